cd $GRAFT_REPO_ROOT
for q in 4 8 16; do
  echo "#### GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/pyr_throughput.py 4 8 16 2>&1 | grep -v amdgpu
done
