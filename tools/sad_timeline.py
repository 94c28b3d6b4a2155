#!/usr/bin/env python3
"""Where the time of one packed-u8 SAD launch goes, workgroup by workgroup (GPU box; needs `make -C visionworkbench_amd/csrc stamps`).

Every workgroup of the instrumented matcher (tools/libexp/libvwgpu_stamps.so, -DVWGPU_TILE_STAMPS) leaves wall-clock stamps (100 MHz) at
its start, after staging, after each byte phase and after its epilogue, the shader clock at start and end, and the XCC / SE / CU it ran
on.  This script prints the launch as a schedule: dispatch ramp, rounds per CU, gaps between a CU's consecutive workgroups, the phase
durations, the average shader clock, and how the wall time of the launch divides into them.
usage: VWGPU_LIBRARY=tools/libexp/libvwgpu_stamps.so python tools/sad_timeline.py [W H SX]..."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VWGPU_LIBRARY", os.path.join(ROOT, "tools", "libexp", "libvwgpu_stamps.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import visionworkbench_amd as vwa  # noqa: E402
from visionworkbench_amd import _lib, core, stereo, synth  # noqa: E402


def run(W, H, SX, ctx, lib, save=None):
    left, right, _ = synth.stereo_pair(W, H, SX, 1)
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    nt = 8192
    buf = torch.zeros(nt * 16, dtype=torch.int64, device="cuda")
    null = ctypes.c_void_p(0)
    assert lib.vwgpu_debug_set_sad_stamps(null) == 0
    box = vwa.bounding_box(left)
    ctx.set_option(core.OPT_DEFER_EXACTNESS, 1)
    if os.environ.get("SAD_GROUPS"): ctx.set_option(core.OPT_SAD_GROUPS, int(os.environ["SAD_GROUPS"]))
    for _ in range(30):
        stereo.calc_disparity(0, lt, rt, box, (SX, 1), (7, 7), ctx=ctx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        stereo.calc_disparity(0, lt, rt, box, (SX, 1), (7, 7), ctx=ctx)
    e1.record()
    torch.cuda.synchronize()
    per_call = e0.elapsed_time(e1) / 20 * 1e3
    assert lib.vwgpu_debug_set_sad_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    stereo.calc_disparity(0, lt, rt, box, (SX, 1), (7, 7), ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_path() == core.PATH_SAD_U8
    assert lib.vwgpu_debug_set_sad_stamps(null) == 0
    st = buf.cpu().numpy().astype(np.uint64).reshape(nt, 16)
    st = st[st[:, 0] != 0]
    n = len(st)
    t0 = st[:, 0].astype(np.float64)
    base = t0.min()
    us = lambda col: (st[:, col].astype(np.float64) - base) / 100.0           # noqa: E731  (100 MHz -> us)
    start, staged, end = us(0), us(3), us(8)
    ph = [us(4 + t) for t in range(4)]
    hw = st[:, 2]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
    cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    key = xcc.astype(np.int64) * 1000 + se * 100 + sh * 50 + cu
    mhz = (st[:, 9].astype(np.float64) - st[:, 1].astype(np.float64)) / np.maximum(end - start, 1e-3)
    print("== %d x %d, sx = %d: %d workgroups, %.1f us per call (events, 20 calls back to back)" % (W, H, SX, n, per_call))
    print("   launch span by stamps: first start 0.0, last start %.1f, last end %.1f us" % (start.max(), end.max()))
    print("   shader clock while a workgroup runs: mean %.0f MHz (min %.0f, max %.0f)" % (mhz.mean(), mhz.min(), mhz.max()))
    cus = np.unique(key)
    print("   distinct CUs: %d, XCCs: %d; workgroups per CU: min %d max %d" % (len(cus), len(np.unique(xcc)), min((key == c).sum() for c in cus), max((key == c).sum() for c in cus)))
    d_stage = staged - start
    d_ph = [ph[0] - staged] + [ph[t] - ph[t - 1] for t in range(1, 4)]
    d_out = end - ph[3]
    q = lambda a: "%.1f / %.1f / %.1f" % (np.percentile(a, 5), np.median(a), np.percentile(a, 95))    # noqa: E731
    print("   per workgroup (p5 / median / p95 us): staging %s | phases %s ; %s ; %s ; %s | vote + output %s | life %s" %
          (q(d_stage), q(d_ph[0]), q(d_ph[1]), q(d_ph[2]), q(d_ph[3]), q(d_out), q(end - start)))
    # rounds: sort each CU's workgroups by start; gap = start of the k-th minus end of the (k - resident)-th
    order = np.argsort(start)
    firsts = np.sort(start)[:min(n, 600)]
    print("   dispatch ramp: workgroup #1 / #64 / #256 / #512 started at %s us" % " / ".join("%.1f" % firsts[min(i, len(firsts) - 1)] for i in (0, 63, 255, 511)))
    gaps, busy, idle_tail = [], [], []
    for c in cus:
        idx = np.where(key == c)[0]
        idx = idx[np.argsort(start[idx])]
        ends = np.sort(end[idx])
        # slot model: a workgroup can start when an earlier one of the CU has ended
        for k in range(2, len(idx)):
            gaps.append(start[idx[k]] - ends[k - 2])
        busy.append((end[idx] - start[idx]).sum())
        idle_tail.append(end.max() - end[idx].max())
    if gaps:
        print("   gap between a workgroup's end and the start of its successor on the CU (2 resident): %s us" % q(np.array(gaps)))
    print("   last end per CU vs last end of the launch (tail imbalance): %s us" % q(np.array(idle_tail)))
    rounds = {}
    for c in cus:
        idx = np.where(key == c)[0]
        for k, i in enumerate(idx[np.argsort(start[idx])]):
            rounds.setdefault(k // 2, []).append((start[i], end[i]))
    for r in sorted(rounds):
        a = np.array(rounds[r])
        print("   round %d: starts %s, ends %s us" % (r, q(a[:, 0]), q(a[:, 1])))
    if save:
        np.save(save, st)
    del order


def main():
    lib = _lib.load()
    lib.vwgpu_debug_set_sad_stamps.argtypes = [ctypes.c_void_p]
    ctx = vwa.Context(0)
    args = [int(v) for v in sys.argv[1:]]
    cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(4096, 4096, 129), (4096, 4096, 33), (4096, 518, 129), (16384, 2054, 129)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for (W, H, SX) in cases:
        run(W, H, SX, ctx, lib, save=os.path.join(ROOT, "gpurun_out", "sad_stamps_%dx%d_%d.npy" % (W, H, SX)))
    ctx.close()


if __name__ == "__main__":
    main()
