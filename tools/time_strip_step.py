"""ms per bench step of rank 0's row strip when config 2 is split N ways (what `bench.py --gpus N` times per rank), on one GPU.
usage: python tools/time_strip_step.py [N...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo, synth, partition
W = H = 4096
KERNEL, SEARCH = (7, 7), (129, 1)
left, right, _ = synth.stereo_pair(W, H, 129, 1)
for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    oh = H - 6
    r0, r1 = partition.row_strip(0, N, oh)
    (la, lb), (ra, rb) = partition.strip_inputs(0, N, H, 7, 1)
    l = torch.from_numpy(left[la:lb]).cuda(); r = torch.from_numpy(right[ra:rb]).cuda()
    ctx = vwa.Context(0); ctx.set_option(core.OPT_DEFER_EXACTNESS, 1)
    if os.environ.get("SAD_GROUPS"): ctx.set_option(core.OPT_SAD_GROUPS, int(os.environ["SAD_GROUPS"]))
    region = vwa.BBox2i(0, 0, W, r1 - r0 + 6)
    step = lambda: stereo.calc_disparity(0, l, r, region, SEARCH, KERNEL, ctx=ctx)
    for _ in range(300): step()
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    F = int(os.environ.get("INFLIGHT", "1"))
    if F > 1:
        # F independent steps in flight: F engine contexts on F streams, steps handed out round-robin (the reference keeps several tiles in
        # flight per device as well, one thread per tile: ImageIO.h:228-251) — the staging burst of one step overlaps the sweep of another
        cs = [vwa.Context(0) for _ in range(F)]
        ss = [torch.cuda.Stream() for _ in range(F)]
        for c_ in cs: c_.set_option(core.OPT_DEFER_EXACTNESS, 1)
        def stepf(i):
            with torch.cuda.stream(ss[i % F]):
                return stereo.calc_disparity(0, l, r, region, SEARCH, KERNEL, ctx=cs[i % F])
        for i in range(100): stepf(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K): o_ = stepf(i)
        torch.cuda.synchronize(); dtf = time.perf_counter() - t0
        same = np.array_equal(o_.cpu().numpy(), step().cpu().numpy())
        print("   %d steps in flight: %.1f us/step (%s the serial result)" % (F, dtf / K * 1e6, "identical to" if same else "DIFFERENT from"))
        for c_ in cs: c_.close()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(20): step()
    torch.cuda.synchronize()
    rec = ctx.profile_read(1 << 10); ctx.profile_enable(False)
    agg = {}
    for n, ms in rec: agg.setdefault(n, []).append(ms)
    if os.environ.get("SAD_GROUPS"):
        got = step().cpu().numpy(); ctx.set_option(core.OPT_SAD_GROUPS, 0); ref = step().cpu().numpy()
        print("   SAD_GROUPS=%s result %s the default flavour's" % (os.environ["SAD_GROUPS"], "identical to" if np.array_equal(got, ref) else "DIFFERENT from"))
    print("N=%d strip rows %d: %.1f us/step; kernels (us, HIP events): %s" % (N, r1 - r0, dt / K * 1e6, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in agg.items()}))
    ctx.close()
