SEED=${SEED:-40404}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${TAG:-r04k}
L=gpurun_out/fuzz_campaign_${TAG}_sgm.txt; : > $L
run() { t0=$(date +%s); echo "\$ $*" >> $L; timeout 600 "$@" 2>&1 | grep -E "cases|MISMATCH|ERROR|Traceback" | tail -5 >> $L; echo "  ($(( $(date +%s) - t0 )) s)" >> $L; }
run python tools/fuzz_sgm_vs_oracle.py 4000 $SEED
run python tools/fuzz_sgm_vs_oracle.py 4000 $SEED mgm
run python tools/fuzz_pyramid_sgm_vs_oracle.py 1500 $SEED 1
run python tools/fuzz_pyramid_sgm_vs_oracle.py 1000 $SEED 2
run python tools/fuzz_pyramid_sgm_vs_oracle.py 1000 $SEED 3
run python tools/fuzz_fast_vs_generic.py 3000 $SEED 0
run python tools/fuzz_fast_vs_generic.py 3000 $SEED 1
run python tools/fuzz_fast_vs_generic.py 3000 $SEED 2
run python tools/fuzz_fast_vs_generic.py 1500 $SEED 0 65536
run python tools/fuzz_fast_vs_generic.py 2000 $SEED 1 4096
run python tools/fuzz_fast_vs_generic.py 2000 $SEED 2 4096
cat $L
