#!/usr/bin/env python3
"""How busy was the GPU during a multi-stream run?  Reads the kernel dispatches of a rocprofv3 --kernel-trace results.db and prints, for the
steady part of the run: wall time, the union of the kernels' intervals (time with at least one kernel running), the sum of their durations,
per-queue busy time, the distribution of the gaps with nothing running, and the kernels by total duration.
usage: python tools/trace_overlap.py results.db [skip_fraction]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
if not rows: sys.exit("no kernels")
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t0 + (t1 - t0) * skip                      # the first part holds warm-up and single-thread reference runs
rows = [r for r in rows if r[1] >= lo]
wall = (max(r[2] for r in rows) - rows[0][1]) / 1e3
union, cur_s, cur_e, gaps = 0.0, rows[0][1], rows[0][2], []
for r in rows[1:]:
    if r[1] > cur_e:
        union += cur_e - cur_s; gaps.append((r[1] - cur_e) / 1e3); cur_s, cur_e = r[1], r[2]
    else:
        cur_e = max(cur_e, r[2])
union += cur_e - cur_s
tot = sum(r[2] - r[1] for r in rows) / 1e3
print("dispatches %d over %.1f ms: at least one kernel running %.1f ms (%.0f %%), sum of durations %.1f ms (%.2f running on average while busy)" %
      (len(rows), wall / 1e3, union / 1e6, 100 * union / 1e3 / wall, tot / 1e3, tot / (union / 1e3)))
if gaps:
    gaps.sort()
    print("idle gaps: %d, total %.1f ms; p50 %.1f us, p90 %.1f us, max %.1f us" % (len(gaps), sum(gaps) / 1e3, gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], gaps[-1]))
if qcol:
    per = collections.Counter()
    for r in rows: per[r[3]] += (r[2] - r[1]) / 1e3
    print("busy per %s: %s" % (qcol, ", ".join("%s: %.1f ms" % (k, v / 1e3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:10])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    agg[n][0] += 1; agg[n][1] += (r[2] - r[1]) / 1e3
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-70s %6d x %8.1f us = %8.1f ms" % (n, c, us / c, us / 1e3))
