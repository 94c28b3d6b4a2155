# Round-5 randomised parity run on the GPU box: the pyramid (single tiles; random textures and the corner scenes), the border cases, the single-level
# float rasters (random and corner cases, fp32 tier on / off) and the tile groups
# against the oracle; the packed fast paths against the float64 kernel.  usage: SEED=5151 bash tools/fuzz_campaign_r05.sh
SEED=${SEED:-5151}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/fuzz_campaign_r05.txt; : > $L
run() { t0=$(date +%s); echo "\$ $*" >> $L; timeout 900 "$@" 2>&1 | grep -E "cases|mismatch|MISMATCH|ERROR|Traceback" | tail -5 >> $L; echo "  ($(( $(date +%s) - t0 )) s)" >> $L; }
run python tools/fuzz_pyramid_vs_oracle.py 5000 $SEED 0.6 0,1,2
run python tools/fuzz_pyramid_vs_oracle.py 3000 $SEED corner
run python tools/fuzz_borders.py 400 $SEED
run python tools/fuzz_round5.py 10000 800 $SEED
run python tools/fuzz_fast_vs_generic.py 2000 $SEED 0
run python tools/fuzz_fast_vs_generic.py 2000 $SEED 1
run python tools/fuzz_fast_vs_generic.py 2000 $SEED 2
cat $L
