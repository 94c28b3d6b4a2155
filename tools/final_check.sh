# End-of-round verification on the GPU box: the whole gpu suite, smoke(), the profiled bench.  usage: TAG=r02f bash tools/final_check.sh
TAG=${TAG:-r02f}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_${TAG}.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - t0 )) s" >> gpurun_out/pytest_${TAG}.log
grep -E "passed|failed|rc=" gpurun_out/pytest_${TAG}.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
TAG=$TAG timeout 900 bash tools/prof_bench.sh 2>&1 | tail -8
