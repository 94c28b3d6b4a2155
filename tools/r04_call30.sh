cd $GRAFT_REPO_ROOT
PYR_ONLY="SAD 7x7, integer" timeout 300 python tools/pyr_throughput.py 2 4 6 8 2>&1 | grep -v amdgpu
PYR_ONLY="LoG 1.4 + NCC" timeout 300 python tools/pyr_throughput.py 4 6 8 2>&1 | grep -v amdgpu
export GPU_MAX_HW_QUEUES=8
PYR_ONLY="SAD 7x7, integer" timeout 300 python tools/pyr_throughput.py 4 6 8 2>&1 | grep -v amdgpu
PYR_ONLY="LoG 1.4 + NCC" timeout 300 python tools/pyr_throughput.py 4 6 8 2>&1 | grep -v amdgpu
