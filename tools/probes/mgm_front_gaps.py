#!/usr/bin/env python3
"""Durations of the MGM front launches of a rocprofv3 kernel-trace database and the gaps between consecutive ones.
usage: python tools/probes/mgm_front_gaps.py trace.db"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
fr = [(n, s, e) for n, s, e in rows if "mgm_front" in n]
if not fr:
    sys.exit("no front launches")
names = sorted({n[:40] for n, _, _ in fr})
dur = sorted((e - s) / 1e3 for _, s, e in fr)
gaps = sorted((fr[i + 1][1] - fr[i][2]) / 1e3 for i in range(len(fr) - 1) if fr[i + 1][1] - fr[i][2] < 2e5)
q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
print(names)
print("launches %d  duration us: p10 %.1f p50 %.1f p90 %.1f mean %.1f sum %.1f ms" % (len(dur), q(dur, .1), q(dur, .5), q(dur, .9), sum(dur) / len(dur), sum(dur) / 1e3))
print("gap us: p10 %.1f p50 %.1f p90 %.1f mean %.1f sum %.1f ms" % (q(gaps, .1), q(gaps, .5), q(gaps, .9), sum(gaps) / len(gaps), sum(gaps) / 1e3))
