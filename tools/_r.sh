( time timeout 1700 python bench.py --workload config5 --no-cpu-baseline > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err ) 2>&1 | tail -3; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/c5_full.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["sgm"]["Mpix_per_s"], d["config"]["tiles_vs_oracle"])
PY
tail -3 gpurun_out/c5_full.err
