# Is the threaded tile loop GPU bound?  Sum of the kernel durations (rocprofv3 --kernel-trace) against the wall time of the same run.  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export PYR_ONLY="${PYR_ONLY:-LoG 1.4 + NCC}"
rocprofv3 --kernel-trace --stats -d /tmp/ptl -o ptl -- python tools/pyr_throughput.py ${THREADS:-4} > /tmp/ptl.log 2>&1
grep -v amdgpu /tmp/ptl.log | grep "ms/tile"
db=$(find /tmp/ptl -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/tile_loop_kernels.md > /dev/null 2>&1
python - <<'PY'
import re
tot = 0.0; rows = []
for line in open("gpurun_out/tile_loop_kernels.md"):
    m = re.match(r"\| (.+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m: rows.append((m.group(1), int(m.group(2)), float(m.group(3)))); tot += float(m.group(3))
print("sum of kernel durations: %.1f ms over the whole process" % (tot / 1e3))
for n, c, t in rows[:12]: print("  %-50s %6d calls %9.1f ms" % (n[:50], c, t / 1e3))
PY
