cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_r04a.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - t0 )) s" >> gpurun_out/pytest_r04a.log
tail -5 gpurun_out/pytest_r04a.log
timeout 300 python tools/sad_timeline.py > gpurun_out/sad_timeline_r04a.txt 2>&1; tail -60 gpurun_out/sad_timeline_r04a.txt
timeout 300 python tools/time_strip_step.py > gpurun_out/strip_step_r04a.txt 2>&1; cat gpurun_out/strip_step_r04a.txt
timeout 600 python bench.py > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; tail -c 1500 gpurun_out/bench_r04a.json
echo "total $(( $(date +%s) - t0 )) s"
