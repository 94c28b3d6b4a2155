cd $GRAFT_REPO_ROOT
for o in 23014567 23014756 23104756 01234756 47562301 45672301 23014657; do
echo "order $o: $(VWGPU_SGM_ORDER=$o timeout 120 python tools/time_sgm.py 2048 2048 128 2>&1 | grep -E "calc_disparity|sgm_paths" | tr '\n' ' ')"
done
