# Round-6 randomised parity run on the GPU box (fresh seed per round): the round-5 legs + the SGM / MGM legs, all against the oracle; the packed fast
# paths against the float64 kernel (the packed-u8 SAD matcher now picks the 512-column four-group tile for small grids).
# usage: SEED=6161 bash tools/fuzz_campaign_r06.sh
SEED=${SEED:-6161}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/fuzz_campaign_r06.txt; : > $L
echo "# SEED=$SEED bash tools/fuzz_campaign_r06.sh" >> $L
run() { t0=$(date +%s); echo "\$ ${SGM_PATH_MODE:+SGM_PATH_MODE=$SGM_PATH_MODE }$*" >> $L; timeout 900 "$@" 2>&1 | grep -E "cases|mismatch|MISMATCH|ERROR|Traceback" | tail -5 >> $L; echo "  ($(( $(date +%s) - t0 )) s)" >> $L; }
run python tools/fuzz_pyramid_vs_oracle.py 5000 $SEED 0.6 0,1,2
run python tools/fuzz_pyramid_vs_oracle.py 3000 $SEED corner
run python tools/fuzz_borders.py 400 $SEED
run python tools/fuzz_round5.py 10000 800 $SEED
run python tools/fuzz_fast_vs_generic.py 3000 $SEED 0
run python tools/fuzz_fast_vs_generic.py 3000 $SEED 1
run python tools/fuzz_fast_vs_generic.py 3000 $SEED 2
run python tools/fuzz_fast_vs_generic.py 1500 $SEED 0 65536
run python tools/fuzz_sgm_vs_oracle.py 6000 $SEED
run python tools/fuzz_sgm_vs_oracle.py 4000 $SEED mgm
run python tools/fuzz_pyramid_sgm_vs_oracle.py 1500 $SEED 1
run python tools/fuzz_pyramid_sgm_vs_oracle.py 1000 $SEED 2
run python tools/fuzz_pyramid_sgm_vs_oracle.py 1000 $SEED 3
# ragged boxes with four / two scan lines per wavefront forced (the levels of these small scenes would take one)
SGM_PATH_MODE=16 run python tools/fuzz_sgm_vs_oracle.py 3000 $SEED
SGM_PATH_MODE=128 run python tools/fuzz_sgm_vs_oracle.py 3000 $SEED
SGM_PATH_MODE=16 run python tools/fuzz_pyramid_sgm_vs_oracle.py 800 $SEED 1
SGM_PATH_MODE=128 run python tools/fuzz_pyramid_sgm_vs_oracle.py 800 $SEED 1
cat $L
