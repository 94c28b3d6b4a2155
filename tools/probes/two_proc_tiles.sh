# Threaded tile loop (tools/pyr_throughput.py) with the fused / split / automatic pass 2 of the exact-order matchers.  GPU box only.
cd $GRAFT_REPO_ROOT
for sp in 2 1 0; do echo "== OPT_EXACT_SPLIT $sp (2 fused, 1 split, 0 by the longest chain)"; PYR_EXACT_SPLIT=$sp timeout 200 python tools/pyr_throughput.py 1 4 2>&1 | grep -v amdgpu | grep "NCC\|SAD 7x7, int"; done
