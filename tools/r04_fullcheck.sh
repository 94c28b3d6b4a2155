cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${TAG:-r04b}
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^certification" > gpurun_out/pytest_${TAG}.log; echo "pytest rc=$? after $(( $(date +%s) - t0 )) s" >> gpurun_out/pytest_${TAG}.log
tail -4 gpurun_out/pytest_${TAG}.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; python - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_us_per_launch"])
for e in d.get("extra", []): print("  ", e.get("name", "")[:90], e.get("kernel_us") or e.get("Mpix_per_s"))
PY
