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
  rocprofv3 --kernel-trace $extra -d gpurun_out/prof/r01g_$tag -o r01g_$tag -- python bench.py $n --no-cpu-baseline > gpurun_out/prof/r01g_$tag.log 2>&1
  db=$(find gpurun_out/prof/r01g_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/r01g_$tag.md >/dev/null 2>&1 || echo "summary $tag failed"
  tail -1 gpurun_out/prof/r01g_$tag.log | cut -c1-200
done
python bench.py > gpurun_out/bench_r01g.json 2> gpurun_out/bench_r01g.err; tail -c 600 gpurun_out/bench_r01g.json
