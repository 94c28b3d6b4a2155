# Per-dispatch FETCH_SIZE / WRITE_SIZE of the SGM path passes of the LAST call of tools/time_sgm.py (one row per direction, in launch
# order) + their durations.  usage: SGM_PATH_MODE=<m> TAG=<t> bash tools/pmc_sgm_dirs.sh   -> gpurun_out/sgm_dirs_<t>.txt.  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcd_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcd_$c -o pmcd_$c -- python tools/time_sgm.py ${SIZE:-2048} ${SIZEH:-2054} 128 > /tmp/pmcd_$c.log 2>&1
done
python - <<PY
import sqlite3, glob, os
out = open("$R/gpurun_out/sgm_dirs_${TAG:-x}.txt", "w")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob("/tmp/pmcd_%s/**/*.db" % c, recursive=True)
    if not dbs:
        out.write(open("/tmp/pmcd_%s.log" % c).read()[-2000:]); continue
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    idc = "dispatch_id" if "dispatch_id" in cols else cols[0]
    rows = list(cur.execute("select %s, kernel_name, sum(value) from counters_collection where counter_name='%s' group by %s, kernel_name order by %s" % (idc, c, idc, idc)))
    rows = [r for r in rows if "path_" in r[1] or "cost_row" in r[1] or "wta" in r[1]]
    n = 8 + 3
    res[c] = rows[-n:] if len(rows) >= n else rows
    try:
        durs = list(cur.execute("select name, start, end from kernels order by start"))
    except Exception:
        durs = []
    if durs and c == "FETCH_SIZE":
        d = [(nm, (e - s) / 1000.0) for nm, s, e in durs if "path_" in nm or "cost_row" in nm]
        res["dur"] = d[-9:]
out.write("mode %s\n" % os.environ.get("SGM_PATH_MODE", "default"))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in res.get(c, []):
        nm = r[1].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        out.write("%-11s %-72s %12.0f KB (x2 for FETCH: %.3f GB)\n" % (c, nm, r[2], r[2] * (2 if c == "FETCH_SIZE" else 1) * 1024 / 1e9))
for nm, us in res.get("dur", []):
    out.write("dur %-72s %9.1f us\n" % (nm.replace("(anonymous namespace)::", "").replace("void ", "")[:70], us))
out.close()
print(open("$R/gpurun_out/sgm_dirs_${TAG:-x}.txt").read())
PY
