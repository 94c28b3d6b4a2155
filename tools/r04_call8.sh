cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in "2,2,11" "0,0,7"; do
  tag=$(echo $c | tr , _)
  PYR_ONLY=$c timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY -d /tmp/pmcz_$tag -o pmcz -- python tools/pyr_profile.py 1024 > /tmp/pmcz_$tag.log 2>&1
  db=$(find /tmp/pmcz_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" gpurun_out/zones_pmc_r04f_$tag.md > /dev/null 2>&1
  grep -E "bm_zones_kernel" gpurun_out/zones_pmc_r04f_$tag.md | head -30
  python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([n for n in names if 'kernel' in n.lower()][:12])
try:
    rows = list(cur.execute("select name, grid_size_x, workgroup_size_x, lds_block_size, vgpr_count, (end - start) from kernels where name like '%bm_zones_kernel%' order by (end - start) desc limit 8"))
    for r in rows: print(r[0][:60], r[1:])
except Exception as e:
    print("query failed:", e)
    try:
        print([d[1] for d in cur.execute("pragma table_info(kernels)")])
    except Exception as e2: print(e2)
PY
done
