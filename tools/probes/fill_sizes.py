#!/usr/bin/env python3
"""Which fills does a trace hold: the dispatches of the runtime's fill / copy kernels of a rocprofv3 kernel-trace database, grouped by grid size.
usage: python tools/probes/fill_sizes.py trace.db [name-substring]"""
import collections
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else "fillBuffer"
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
print(cols)
want = [c for c in ("grid_x", "grid_size_x", "grid_size", "workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
rows = cur.execute("select name, start, end, %s from kernels where name like ?" % ", ".join(want), ("%" + sub + "%",))
g = collections.defaultdict(list)
for r in rows:
    g[(r[0][:60],) + tuple(r[3:])].append((r[2] - r[1]) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(k, "n", len(v), "avg us %.1f" % (sum(v) / len(v)), "total ms %.2f" % (sum(v) / 1e3))
