cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/fuzz_campaign_r04i.txt; : > $L
run() { t0=$(date +%s); echo "\$ $*" >> $L; timeout 900 "$@" 2>&1 | grep -E "cases|MISMATCH|ERROR|Traceback" | tail -5 >> $L; echo "  ($(( $(date +%s) - t0 )) s)" >> $L; }
run python tools/fuzz_pyramid_vs_oracle.py 8000 424242 0.6 0,1,2
run python tools/fuzz_borders.py 400 99
run python tools/fuzz_borders.py 300 5
cat $L
