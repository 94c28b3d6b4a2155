cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04k} bash tools/final_check.sh
L=gpurun_out/fuzz_campaign_${TAG:-r04k}.txt; : > $L
run() { t0=$(date +%s); echo "\$ $*" >> $L; timeout 900 "$@" 2>&1 | grep -E "cases|MISMATCH|ERROR|Traceback" | tail -5 >> $L; echo "  ($(( $(date +%s) - t0 )) s)" >> $L; }
run python tools/fuzz_pyramid_vs_oracle.py 5000 777 0.6 0,1,2
run python tools/fuzz_borders.py 400 31
cat $L
