cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/r04c_main -o r04c_main -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra > gpurun_out/prof/r04c_main.log 2>&1
db=$(find gpurun_out/prof/r04c_main -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/r04c_main.md >/dev/null 2>&1
head -12 gpurun_out/r04c_main.md
tail -1 gpurun_out/prof/r04c_main.log | cut -c1-300
rm -rf gpurun_out/prof/r04c_main
