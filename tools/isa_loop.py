#!/usr/bin/env python3
"""Instruction mix of one loop of a kernel (hipcc -S output): every basic block whose label comment says it is in the loop headed by the
given block, plus the header.  usage: python tools/isa_loop.py file.s 'kernel substring' BB15_69"""
import re, sys, collections
path, pat, hdr = sys.argv[1:4]
inside, cur_in, cnt, ops = False, False, collections.Counter(), collections.Counter()
for line in open(path):
    if re.match(r"^_Z\S+:", line):
        inside = pat in line
        cur_in = False
        continue
    if not inside: continue
    m = re.match(r"^(\.LBB\S+):(.*)", line) or re.match(r"^; %bb\.\d+:(.*)", line)
    if m:
        txt = line
        cur_in = ("Header=" + hdr + " " in txt) or (line.startswith(".L" + hdr + ":"))
        continue
    if not cur_in or not line.startswith("\t") or line.startswith("\t;") or line.startswith("\t."): continue
    op = line.split()[0]
    kind = "ds" if op.startswith("ds_") else "global" if op.startswith(("global", "buffer")) else "valu" if op.startswith("v_") else "scalar" if op.startswith("s_") else "other"
    cnt[kind] += 1; ops[op] += 1
print(dict(cnt), "total", sum(cnt.values()))
print(", ".join("%s %d" % kv for kv in ops.most_common(24)))
