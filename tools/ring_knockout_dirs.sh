# needs: make -C visionworkbench_amd/csrc ringdbg
# per-direction durations of the ring kernel under the knock-out build (tools/libexp/libvwgpu_ringdbg.so).  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for d in ${DBGS:-0 7 1 6}; do
  rm -rf /tmp/rk_$d
  VWGPU_LIBRARY=$R/tools/libexp/libvwgpu_ringdbg.so VWGPU_RING_DBG=$d rocprofv3 --kernel-trace --output-format csv -d /tmp/rk_$d -- python tools/time_sgm.py 2048 2054 128 > /tmp/rk_$d.log 2>&1
  python - <<PY
import csv, glob
fs = glob.glob('/tmp/rk_$d/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r['Start_Timestamp']))
p = [r for r in rows if 'path_ring' in r['Kernel_Name']][-8:]
print("dbg $d:", " ".join("%.0f" % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in p), "us (L->R store, R->L, T->B, B->T, 4 diagonals; last = WTA)")
PY
done
