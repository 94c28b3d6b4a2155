# rocprofv3 passes of the default bench command (kernel stats, then the PMC passes in runs of their own).  GPU box only.
# usage: TAG=r02a bash tools/prof_bench.sh
TAG=${TAG:-r02a}
mkdir -p gpurun_out/prof && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for tag in main fetch write sq; do
  case $tag in
    main) extra="--stats";;
    fetch) extra="--pmc FETCH_SIZE";;
    write) extra="--pmc WRITE_SIZE";;
    sq) extra="--pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU";;
  esac
  # the stats pass runs long enough for the clocks to settle (its average must agree with bench.py's HIP events); the
  # counter passes serialise every dispatch, a dozen launches are plenty
  if [ $tag = main ]; then n="--steps 200 --warmup 20"; else n="--steps 12 --warmup 3"; fi
  rocprofv3 --kernel-trace $extra -d gpurun_out/prof/${TAG}_$tag -o ${TAG}_$tag -- python bench.py $n --no-cpu-baseline --no-extra --no-traffic > gpurun_out/prof/${TAG}_$tag.log 2>&1
  db=$(find gpurun_out/prof/${TAG}_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_$tag.md >/dev/null 2>&1 || echo "summary $tag failed"
  tail -1 gpurun_out/prof/${TAG}_$tag.log | cut -c1-200
done
# the other kernel families of the path (config 3, SGM block, pyramid tiles): kernel stats only
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/${TAG}_families -o ${TAG}_families -- python tools/profile_families.py > gpurun_out/prof/${TAG}_families.log 2>&1
db=$(find gpurun_out/prof/${TAG}_families -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_families.md >/dev/null 2>&1 || echo "summary families failed"
rm -rf gpurun_out/prof/${TAG}_*/   # the raw databases are large; the summaries are what profiles/ keeps
python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 400 gpurun_out/bench_${TAG}.json
