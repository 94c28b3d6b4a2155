#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one translation unit, read off hipcc's resource remarks (no GPU needed).
usage: python tools/kernel_resources.py visionworkbench_amd/csrc/bm_sad_u8.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-Iinclude", "-Ivisionworkbench_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
if not rows:
    sys.stderr.write(err)
    sys.exit(1)
try:
    names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines()
except OSError:
    names = [r["name"] for r in rows]
print("%-90s %5s %5s %7s %5s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
    print("%-90s %5d %5d %7d %5d %6d" % (n[:90], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1),
                                         r.get("Occupancy", -1), r.get("LDS Size", -1)))
