cd $GRAFT_REPO_ROOT
PYR_ONLY="SAD 7x7, integer" timeout 300 python tools/pyr_throughput.py 3 4 5 6 8 2>&1 | grep thr
PYR_ONLY="LoG 1.4 + NCC" timeout 300 python tools/pyr_throughput.py 3 4 5 6 8 2>&1 | grep thr
