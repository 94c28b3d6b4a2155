#!/bin/bash
# per-launch durations of the SGM kernels (rocprofv3 kernel trace) -> gpurun_out/sgm_trace_<TAG>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sgmtr
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/sgmtr -- python tools/time_sgm.py ${SIZE:-2048} ${SIZEH:-2054} 128 > /tmp/sgmtr.log 2>&1
python - <<PY
import csv,glob,collections
fs=glob.glob('/tmp/sgmtr/**/*kernel_trace.csv',recursive=True)
if not fs: print(open('/tmp/sgmtr.log').read()[-3000:]); raise SystemExit
f=fs[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last call = last occurrence of census kernel onwards
idx=[i for i,r in enumerate(rows) if 'census' in r['Kernel_Name']]
start=idx[-1]
out=open('$R/gpurun_out/sgm_trace_${TAG:-x}.txt','w')
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    n=r['Kernel_Name'].split('(')[0][-60:]
    out.write('%8.1f us  +%7.1f us  %s grid=%s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n,r.get('Grid_Size_X','')))
out.close()
print(open('$R/gpurun_out/sgm_trace_${TAG:-x}.txt').read())
PY
