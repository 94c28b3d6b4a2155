#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel (hipcc -S, no GPU needed): finds the hot loops of a kernel and what is in them.
usage: python tools/isa_blocks.py file.hip 'kernel substring' [min instructions]"""
import collections
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Iinclude",
                      "-Ivisionworkbench_amd/csrc", "--cuda-device-only", "-S", src, "-o", "-"], stdout=subprocess.PIPE, text=True).stdout
cur_fn, blocks, cur = None, [], None
for line in asm.splitlines():
    m = re.match(r"^(_Z\S+):", line)
    if m:
        cur_fn = m.group(1)
        cur = [cur_fn, "entry", collections.Counter()]
        blocks.append(cur)
        continue
    m = re.match(r"^(\.LBB\S+):", line)
    if m and cur_fn:
        cur = [cur_fn, m.group(1), collections.Counter()]
        blocks.append(cur)
        continue
    if line.startswith("\t.") or not line.startswith("\t") or cur is None:
        continue
    op = line.split()[0]
    key = ("qsad" if "qsad" in op else "sdwa" if "sdwa" in op else "min3" if "min3" in op else "scratch" if op.startswith("scratch") else
           "ds" if op.startswith("ds_") else "global" if op.startswith("global") or op.startswith("buffer") else
           "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
    cur[2][key] += 1
    cur[2]["total"] += 1
for fn, lab, c in blocks:
    if pat in fn and c["total"] >= minn:
        print("%-14s %s" % (lab, " ".join("%s=%d" % kv for kv in sorted(c.items()))))
