#!/usr/bin/env python3
"""bench.py — disparity Mpix/s of the block-matching hot path on MI355X (BASELINE.json metric).

Workload (BASELINE config 2): 4096x4096 synthetic stereo pair, 7x7 SAD, +-64 px search
(search_volume 129x1), one call of vw::stereo::calc_disparity per step, inputs (float32 PixelGray images)
already resident in HBM, output (PixelMask<Vector2i>, 12 B/px) left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: launched by torch.distributed.run, one rank per GPU.  The pair is split into N row strips
  (output rows [g*oh/N, (g+1)*oh/N) on rank g, input strip + ky-1 halo rows resident on that GPU);
  block matching needs no exchange step, so there is no data-path collective.  Total work is fixed:
  scaling = "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) with three extra objects:
  roofline     HBM roofline of the hot path's kernels: algorithmic bytes / HIP-event time
  cpu_baseline the CPU oracle (restated reference, tile-threaded like the reference) on a bounded sample; its full-image
               pass is also the checker of the timed result (every one of the 4090^2 pixels must be identical)
  extra        (N = 1) the other measured points of the path, each with its algorithmic bytes, kernel time and roofline
               fraction: the same pair at +-16 px (33 disparities: where the HBM bound is arithmetically in reach), BASELINE
               config 3 (11x11 NCC, then parabola_subpixel) and the SGM building block of config 4 (2048^2, census 7x7,
               129 disparities).  They are not part of `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W = H = 4096
KERNEL = (7, 7)
SEARCH = (129, 1)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FCLK_HZ = 2.4e9                # max shader clock
LANES = 256 * 4 * 16           # CUs x SIMDs x lanes/clk of an ordinary VOP3 op (profiles/r01_ubench_valu.txt)


def algorithmic_bytes(lw, lh, kx, ky, sx, sy):
    """read L once (f32), read R once (f32), write the VW-layout disparity once."""
    return 4 * lw * lh + 4 * (lw + sx - 1) * (lh + sy - 1) + 12 * (lw - kx + 1) * (lh - ky + 1)


def cpu_baseline(left, right, budget_s=12.0):
    """The restated reference timed the way the reference runs a big image: the output is split into tiles, T worker
    threads pull tiles from a queue and each tile calls single-threaded calc_disparity on its padded crop
    (src/vw/Image/ImageIO.h:228-251, src/vw/Image/BlockProcessor.h:52-176).  Tile = 256 px, the library default
    (src/vw/Core/Settings.cc:183), so that all T = nproc threads have work; bounded to ~budget_s seconds of wall
    time by processing a prefix of the tile list (at least one tile per thread)."""
    import oracle
    cores = os.cpu_count() or 1
    tile = 256
    ntiles_total = ((W - KERNEL[0] + 1 + tile - 1) // tile) * ((H - KERNEL[1] + 1 + tile - 1) // tile)
    # calibrate on one tile, single thread: the reference's native unit, seconds per (pixel x disparity)
    t0 = time.perf_counter()
    _, done1 = oracle.calc_disparity_tiled(0, left, right, KERNEL, SEARCH, tile=tile, threads=1, max_tiles=1)
    t1 = time.perf_counter() - t0
    ns_per_op = t1 / (done1 * SEARCH[0] * SEARCH[1]) * 1e9
    tiles = int(min(ntiles_total, max(cores, budget_s / t1 * cores * 0.7)))
    # one threaded pass to see how the host really scales (many-core hosts are memory bound here), then repeat
    # it until ~budget_s seconds of wall time have been spent
    t0 = time.perf_counter()
    _, done = oracle.calc_disparity_tiled(0, left, right, KERNEL, SEARCH, tile=tile, threads=cores, max_tiles=tiles)
    first = time.perf_counter() - t0
    reps = 1 + int(max(0, min(40, (budget_s - first) / max(first, 1e-3))))
    for _ in range(reps - 1):
        _, d = oracle.calc_disparity_tiled(0, left, right, KERNEL, SEARCH, tile=tile, threads=cores, max_tiles=tiles)
        done += d
    dt = time.perf_counter() - t0
    return {"value": done / dt / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": "%d x (%d of %d 256x256 output tiles of the same 4096^2 / 7x7 SAD / 129x1 pair) on %d threads, "
                      "%.1f s wall, %.0f core-seconds" % (reps, tiles, ntiles_total, cores, dt, dt * cores),
            "single_thread_ns_per_pixel_disparity": ns_per_op}


def measure(ctx, torch, fn, reps, warm=2):
    """Device-resident timing of fn(): (wall ms per call without profiling, {kernel: avg us per call} from HIP events)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    # fastest of three batches: calls that synchronise with the host (the input-class flag of a non-deferred call) pick up whatever
    # else the host is doing; one slow call of ten moved a 1.07 ms point to 1.63 ms between two runs with identical kernel times
    wall = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / reps * 1e3
        wall = w if wall is None else min(wall, w)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    per = {}
    for name, ms in ctx.profile_read(1 << 14):
        per[name] = per.get(name, 0.0) + ms
    return wall, {k: v / reps * 1e3 for k, v in per.items()}


def extra_points(ctx, torch, stereo, core, vwa, synth, lt, rt, left, right):
    """The other measured points of the path (see the module docstring).  Bounded: a few dozen calls in total."""
    out = []

    def entry(name, model_bytes, model, wall, kern, hot, pixels, **kw):
        t_hot = sum(v for k, v in kern.items() if k in hot)
        e = {"name": name, "algorithmic_bytes": int(model_bytes), "bytes_model": model, "kernels_us": {k: round(v, 2) for k, v in kern.items()},
             "hot_kernels": sorted(k for k in kern if k in hot), "hot_us": round(t_hot, 2), "wall_ms_per_call": round(wall, 4),
             "Mpix_per_s": round(pixels / (wall * 1e-3) / 1e6, 1),
             "roofline_frac": round(model_bytes / (t_hot * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if t_hot > 0 else None}
        e.update(kw)
        out.append(e)

    bb = vwa.bounding_box(left)
    # (1) the headline kernel at +-16 px: 33 disparities
    r33 = rt[:, 48:48 + W + 32].contiguous()
    wall, kern = measure(ctx, torch, lambda: stereo.calc_disparity(0, lt, r33, bb, (33, 1), KERNEL, ctx=ctx), 40)
    entry("4096^2, 7x7 SAD, search 33x1 (+-16 px)", algorithmic_bytes(W, H, 7, 7, 33, 1), "4LW + 4RW + 12 out (SURVEY 8d)", wall, kern,
          {"bm_sad_u8"}, (W - 6) * (H - 6), path=ctx.last_path())
    # (2) BASELINE config 3: 11x11 NCC over 129x1, then parabola_subpixel on the result
    wall, kern = measure(ctx, torch, lambda: stereo.calc_disparity(2, lt, rt, bb, SEARCH, (11, 11), ctx=ctx), 10)
    entry("config 3a: 4096^2, 11x11 NCC, search 129x1", algorithmic_bytes(W, H, 11, 11, 129, 1), "4LW + 4RW + 12 out (SURVEY 8d)", wall, kern,
          {"bm_corr_u8", "ncc_full", "bm_dot_u8"}, (W - 10) * (H - 10), path=ctx.last_path())
    d = stereo.calc_disparity(2, lt, rt, bb, SEARCH, (11, 11), ctx=ctx)
    disp = torch.zeros((H, W, 3), dtype=torch.float32, device=lt.device)
    disp[5:5 + H - 10, 5:5 + W - 10, :2] = d[..., :2].float()
    disp[5:5 + H - 10, 5:5 + W - 10, 2] = (d[..., 2] != 0).float()
    wall, kern = measure(ctx, torch, lambda: stereo.parabola_subpixel(disp, lt, rt, 0, 0.0, (11, 11), ctx=ctx), 10)
    pb = 4 * W * H + 4 * (W + 128) * H + 12 * W * H + 12 * W * H
    entry("config 3b: parabola_subpixel 11x11 on the 4096^2 NCC result", pb, "4LW + 4RW + 12 disparity in + 12 out (SURVEY 8d)", wall, kern,
          {k for k in kern if k.startswith("parabola") or k in ("disparity_range", "float_grain", "edge_extend_sub")}, W * H)
    # (3) SGM building block of config 4: 2048^2, census 7x7, 129 disparities, 8 paths, LC-blend sub-pixel
    n = 2048
    ls, rs_ = lt[:n, :n].contiguous(), rt[:n, :n + 128].contiguous()
    sgm = lambda: stereo.calc_disparity_sgm(3, ls, rs_, vwa.BBox2i(0, 0, n, n), (128, 0), (7, 7), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    wall, kern = measure(ctx, torch, sgm, 5, warm=1)
    npx = (n - 6) * (n - 6)
    entry("config 4 building block: SGM 2048^2, census 7x7, 129 disparities", npx * (20 + 11 * 129),
          "materialised volume (20 + 11 D) B/px (SURVEY 8d; the minimum model is 20 + 4 D = 536 B/px)", wall, kern,
          {k for k in kern if k.startswith("sgm")}, npx, min_model_frac=None)
    e = out[-1]
    if e["hot_us"]:
        e["min_model_frac"] = round(npx * 536 / (e["hot_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    # (3b) the same matcher with use_mgm (accum_mgm_multithread): 1024^2, one launch per front of the 2-D recurrence.  Bytes: 8 directions x
    # (2 reads + 1 write) x 2 B per cell + the sum pass (8 reads + 1 write) x 2 B = 66 B per cell, + costs and images
    n = 1024
    lm_, rm_ = lt[:n, :n].contiguous(), rt[:n, :n + 128].contiguous()
    mgm = lambda: stereo.calc_disparity_sgm(3, lm_, rm_, vwa.BBox2i(0, 0, n, n), (128, 0), (7, 7), use_mgm=True, with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    wall, kern = measure(ctx, torch, mgm, 3, warm=1)
    npx = (n - 6) * (n - 6)
    entry("MGM (use_mgm) 1024^2, census 7x7, 129 disparities", npx * (20 + (66 + 3) * 129), "direction volumes: (20 + 69 D) B/px (DESIGN.md 4.6b)", wall, kern,
          {k for k in kern if k.startswith("sgm")}, npx, fronts=2 * n - 13, note="launch bound: one launch per front")
    # (4) the tile loop of config 5 (tools/correlate.cc:207-266): pyramid_correlate over the 4096^2 pair in 1024^2 tiles, pulled by
    # 4 tile threads, each with its own engine context and stream — the way block_write_image runs the reference's view
    import threading
    rc = rt[:, 64:64 + W].contiguous()
    tiles = [vwa.BBox2i(x, y, 1024, 1024) for y in range(0, H, 1024) for x in range(0, W, 1024)]
    search = vwa.BBox2i.from_corners((-64, -1), (64, 1))
    for label, pf, pw, cost, kk in (("SAD 7x7", 0, 0.0, 0, 7), ("LoG 1.4 + NCC 11x11 (the correlate tool's defaults)", 2, 1.4, 2, 11)):
        T = 4
        ctxs = [vwa.Context(lt.device.index) for _ in range(T)]
        streams = [torch.cuda.Stream(device=lt.device) for _ in range(T)]
        def work(t):
            with torch.cuda.stream(streams[t]):
                for i in range(t, len(tiles), T):
                    stereo.pyramid_correlate(lt, rc, None, None, pf, pw, search, (kk, kk), cost, consistency_threshold=2, filter_half_kernel=5,
                                             max_pyramid_levels=5, bbox=tiles[i], ctx=ctxs[t])
        best = None
        for rep in range(3):
            torch.cuda.synchronize(lt.device); t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            [x.start() for x in th]; [x.join() for x in th]
            torch.cuda.synchronize(lt.device); dt_ = time.perf_counter() - t0
            if rep > 0: best = dt_ if best is None else min(best, dt_)
        for c_ in ctxs: c_.close()
        out.append({"name": "config 5 building block: pyramid_correlate tile loop, 4096^2 in 16 tiles of 1024^2, %s, +-64 x +-1, 5 levels, L/R check, "
                            "4 tile threads" % label, "wall_ms_per_pair": round(best * 1e3, 2), "ms_per_tile": round(best * 1e3 / len(tiles), 3),
                    "Mpix_per_s": round(W * H / best / 1e6, 1), "roofline_frac": None,
                    "note": "throughput of the threaded tile loop; per-kernel times of one tile: tools/pyr_profile.py"})
    return out


def run_config4(args, torch, dist, vwa, core, stereo, synth, partition, rank, world, dev):
    """BASELINE config 4 as the reference can run it (SGM takes census costs only): 16384^2 pair, census 7x7, 129
    disparities, 8 paths, LC-blend sub-pixel, in 8 row strips of 2048 rows with "tile + collar" semantics
    (PyramidCorrelationView::rasterize, CorrelationView.h:123-133: a strip is matched over strip + collar rows and its centre
    kept; parity is against the reference run with the same geometry).  Rank g holds the rows of its 8 / N strips; the
    collar + half-kernel rows above and below come from the neighbouring ranks by RCCL isend / irecv before the clock starts.
    No collective in the timed region."""
    W = H = 16384
    k, D, collar, nstrips = 7, 129, 64, 8
    if nstrips % world:
        sys.exit("config 4 uses 8 row strips: --gpus must divide 8")
    a, b = partition.row_strip(rank, world, H)
    left, right, truth = synth.stereo_pair_rows(W, H, D, a, b)
    halo = collar + k // 2
    lwin = torch.from_numpy(left).to(dev)
    rwin = torch.from_numpy(right).to(dev)
    first = a
    how = "none (single rank holds every row)"
    if world > 1:
        lwin, first = partition.fetch_strip_window(lwin, rank, world, H, halo, halo)
        rwin, _ = partition.fetch_strip_window(rwin, rank, world, H, halo, halo)
        how = "RCCL isend/irecv of %d collar + %d half-kernel rows per neighbour" % (collar, k // 2)
    torch.cuda.synchronize(dev)
    ctx = vwa.Context(dev.index)
    rows_per = H // nstrips
    mine = [s for s in range(nstrips) if a <= s * rows_per < b]

    def step():
        outs = []
        for s_ in mine:
            y0, y1 = s_ * rows_per, (s_ + 1) * rows_per                 # output rows of the strip (image coordinates of the window centre)
            ra, rb = max(0, y0 - halo), min(H, y1 + halo)                # input rows of strip + collar + half kernel
            l = lwin[ra - first:rb - first]
            r = rwin[ra - first:rb - first]
            d = stereo.calc_disparity_sgm(3, l, r, vwa.BBox2i(0, 0, W, rb - ra), (D - 1, 0), (k, k), with_subpixel=True,
                                          memory_limit_mb=200000, ctx=ctx)[0]
            top = max(0, y0 - ra - k // 2)                               # output row j of the call is centred on input row ra + j + k/2
            outs.append(d[top:top + (y1 - y0)])                          # the strip's centre rows (fewer at the image borders)
        return outs

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(1, min(args.warmup, 1))):
        out = step()
    barrier()
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    npx = (W - k + 1) * H
    got = out[0]
    c0 = max(mine[0] * rows_per, k // 2)                                 # centre row of the first kept output row
    tr = truth[c0 - a:c0 - a + got.shape[0], 3:3 + W - 6]
    ok = float((got[:tr.shape[0], :, 0].cpu().numpy() == tr).mean()) if got.shape[0] else 0.0
    if rank == 0:
        per_px = 20 + 11 * D
        print(json.dumps({
            "metric": "disparity Mpix/s, 16384x16384 pair, census 7x7 SGM, 129 disparities, 8 row strips + collar",
            "value": npx * steps / dt / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": steps, "warmup": 1,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8/u16",
            "data": "synthetic (SplitMix64 integer-valued float32 noise pair, 256-px blocks shifted by 64+-48)",
            "config": {"workload": "BASELINE configs[3] (census SGM: the reference's SGM refuses SAD costs, SURVEY F3), tile + collar semantics",
                       "strips": nstrips, "strip_rows": rows_per, "collar_rows": collar, "halo": how,
                       "integer_match_rate_vs_truth_first_strip": ok},
            "roofline": {"bound": "hbm", "achieved": npx * per_px * steps / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": npx * per_px * steps / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "bytes_model": "materialised volume (20 + 11 D) B per pixel, SURVEY 8d", "kernel": "calc_disparity_sgm (wall)"},
            "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_config5(args, torch, dist, vwa, core, stereo, synth, partition, rank, world, dev):
    """BASELINE config 5, the tile loop of tools/correlate (correlate.cc:207-266) at orbital scale: a 32768^2 pair, pyramid_correlate in
    1024^2 tiles (5 levels, +-64 x +-1 search, L/R check, outlier filters), pulled by 4 tile threads per GPU, each with its own engine
    context and stream — once with the tool's block-matching defaults (LoG 1.4 prefilter, NCC 11x11) and once with SGM (census 7x7).
    Rank g owns the tile rows of its row strip; the pyramid halo rows above and below come from the neighbouring ranks by RCCL
    isend / irecv before the clock starts.  No collective in the timed region.  (parabola_subpixel on NCC results: the config-3 point.)"""
    import threading
    W = H = 32768 if not os.environ.get("VWGPU_BENCH_CONFIG5_SIZE") else int(os.environ["VWGPU_BENCH_CONFIG5_SIZE"])
    TILE, LEVELS, D = 1024, 5, 129
    if (H // TILE) % world:
        sys.exit("config 5: --gpus must divide the %d tile rows" % (H // TILE))
    a, b = partition.row_strip(rank, world, H)
    lwin = torch.empty((b - a, W), dtype=torch.float32, device=dev)
    rwin = torch.empty((b - a, W), dtype=torch.float32, device=dev)
    truth0 = None
    for r0 in range(a, b, 2048):                                   # the pair is generated in bands (a whole image would need ~30 GB of host arrays)
        r1 = min(b, r0 + 2048)
        l, r, t = synth.stereo_pair_rows(W, H, D, r0, r1)
        lwin[r0 - a:r1 - a] = torch.from_numpy(l).to(dev)
        rwin[r0 - a:r1 - a] = torch.from_numpy(r[:, 64:64 + W].copy()).to(dev)    # true disparities: 0 +- 48 px inside the +-64 search
        if truth0 is None: truth0 = t[:TILE, :TILE] - 64
    first = a
    how = "none (single rank holds every row)"
    search = vwa.BBox2i.from_corners((-64, -1), (64, 1))
    if world > 1:
        above, below = partition.pyramid_halo_rows(11, LEVELS, -1, 1)
        hal = max(above, below) + 2 * 3                            # SGM R->L runs reach twice the vertical search extent
        lwin, first = partition.fetch_strip_window(lwin, rank, world, H, hal, hal)
        rwin, _ = partition.fetch_strip_window(rwin, rank, world, H, hal, hal)
        how = "RCCL isend/irecv of %d pyramid halo rows per neighbour" % hal
    torch.cuda.synchronize(dev)
    tiles = [(x, y) for y in range(a, b, TILE) for x in range(0, W, TILE)]
    T = 4
    ctxs = [vwa.Context(dev.index) for _ in range(T)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
    keep = {}

    def loop(kw, only=None):
        todo = list(tiles if only is None else tiles[:only])
        lock = threading.Lock()
        def work(t):
            with torch.cuda.stream(streams[t]):
                while True:
                    with lock:
                        if not todo: return
                        x, y = todo.pop()
                    o = stereo.pyramid_correlate(lwin, rwin, None, None, kw["pf"], kw["pfw"], search, kw["kernel"], kw["cost"], consistency_threshold=2,
                                                 filter_half_kernel=5, max_pyramid_levels=LEVELS, algorithm=kw["alg"],
                                                 bbox=vwa.BBox2i(x, y - first, TILE, TILE), ctx=ctxs[t])
                    if (x, y) == (0, a): keep[kw["name"]] = o
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    modes = [dict(name="bm", pf=2, pfw=1.4, kernel=(11, 11), cost=2, alg=0), dict(name="sgm", pf=0, pfw=0.0, kernel=(7, 7), cost=3, alg=1)]
    res = {}
    for kw in modes:
        loop(kw, only=2 * T)                                       # warm-up: arenas of every context sized
        barrier()
        t0 = time.perf_counter()
        loop(kw)
        barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        res[kw["name"]] = float(tmax.item())
    for c_ in ctxs: c_.close()
    if rank == 0:
        npx = W * H
        ok = {}
        for name, o in keep.items():
            g = o.cpu().numpy()
            ok[name] = float(((np.rint(g[..., 0]) == truth0) & (g[..., 2] != 0)).mean())
        print(json.dumps({
            "metric": "disparity Mpix/s, %dx%d pair, pyramid_correlate tile loop (LoG 1.4 + NCC 11x11, 5 levels, +-64 x +-1, L/R check)" % (W, H),
            "value": npx / res["bm"] / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": res["bm"] * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic (SplitMix64 integer-valued float32 noise pair, 256-px blocks shifted by 0+-48)",
            "config": {"workload": "BASELINE configs[4]: orbital-scale pair through the reference's tile loop; block matching with the correlate "
                                   "tool's defaults and, separately, SGM (census 7x7)", "tile": TILE, "tiles": (W // TILE) * (H // TILE),
                       "tile_threads_per_gpu": T, "halo": how,
                       "sgm": {"Mpix_per_s": npx / res["sgm"] / 1e6, "s_per_pair": res["sgm"]},
                       "truth_match_rate_first_tile": ok},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)     # 0.36 ms each: long enough to amortise the ~1.3 ms of barrier + first-launch latency
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=250.0, dest="settle_ms")   # untimed load before the warm-up steps (clock ramp)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measured points (config 3, SGM block, +-16 px)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config4", "config5"],
                    help="config2 (default, the headline metric): 4096^2 7x7 SAD; config4: 16384^2 census SGM in 8 strips + collar")
    args = ap.parse_args()

    # the host driver supports dmabuf IPC only: without this RCCL's peer mappings fail (hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import visionworkbench_amd as vwa
    from visionworkbench_amd import core, stereo, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "config5":
        from visionworkbench_amd import partition
        return run_config5(args, torch, dist, vwa, core, stereo, synth, partition, rank, world, dev)
    if args.workload == "config4":
        from visionworkbench_amd import partition
        return run_config4(args, torch, dist, vwa, core, stereo, synth, partition, rank, world, dev)

    kx, ky = KERNEL
    sx, sy = SEARCH
    left, right, _ = synth.stereo_pair(W, H, sx, sy)
    ow, oh = W - kx + 1, H - ky + 1
    # row strip of this rank (output rows), plus the halo rows its windows read
    from visionworkbench_amd import partition
    r0, r1 = partition.row_strip(rank, world, oh)
    (la, lb), (ra, rb) = partition.strip_inputs(rank, world, H, ky, sy)
    halo = "none (single strip)"
    l_strip = r_strip = None
    if world > 1:
        # The source pair is row-sharded across the GPUs (disjoint rows per HBM); the ky-1 (+sy-1) halo rows a strip's
        # windows read come from the neighbour over RCCL point-to-point (one xGMI link per neighbour pair) — set-up, not
        # part of the timed region (inputs are resident when the clock starts).  Falls back to host-provided halos if
        # the P2P path is unavailable, and says so in the JSON line.
        try:
            lbnd = partition.sharded_bounds(world, left.shape[0], oh, 0, ky - 1)
            rbnd = partition.sharded_bounds(world, right.shape[0], oh, 0, ky - 1 + sy - 1)
            a, b, na, nb = lbnd[rank]
            l_strip = partition.exchange_halo(torch.from_numpy(left[a:b].copy()).to(dev), a, b, na, nb, rank, world, lbnd)
            a, b, na, nb = rbnd[rank]
            r_strip = partition.exchange_halo(torch.from_numpy(right[a:b].copy()).to(dev), a, b, na, nb, rank, world, rbnd)
            torch.cuda.synchronize(dev)
            l_strip = l_strip[:lb - la].contiguous()
            r_strip = r_strip[:rb - ra].contiguous()
            ok = bool(torch.equal(l_strip.cpu(), torch.from_numpy(left[la:lb])) and torch.equal(r_strip.cpu(), torch.from_numpy(right[ra:rb])))
            if not ok:
                raise RuntimeError("halo exchange returned wrong rows")
            halo = "RCCL isend/irecv of %d (+%d) rows between neighbouring strips" % (ky - 1, sy - 1)
        except Exception as e:  # noqa: BLE001
            halo = "host-provided halo rows (RCCL P2P unavailable: %s)" % (str(e)[:80],)
            l_strip = r_strip = None
    if l_strip is None:
        l_strip = torch.from_numpy(left[la:lb]).to(dev)
        r_strip = torch.from_numpy(right[ra:rb]).to(dev)
    region = vwa.BBox2i(0, 0, W, r1 - r0 + ky - 1)
    ctx = vwa.Context(local)
    # K steps are queued back to back: the engine must not wait for the input-class flags of each call (a host round trip
    # per step).  It runs the packed kernels, keeps the float64 kernel behind the device flag, and vwgpu_last_path() says
    # afterwards which family produced the result — asserted below, together with the result itself.
    ctx.set_option(core.OPT_DEFER_EXACTNESS, 1)

    def step():
        return stereo.calc_disparity(core.ABSOLUTE_DIFFERENCE, l_strip, r_strip, region, SEARCH, KERNEL, ctx=ctx)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Clock settle: a fresh box idles at ~600 MHz and takes some tens of milliseconds of load to reach its sustained clock; a short
    # run (the driver's 20 steps = 7 ms) would otherwise time the ramp (round 1: 378 us per launch in the driver's run against 348 us
    # in a 400-step run).  Untimed, before the W warm-up steps, reported in config.clock_settle_ms.
    out = None
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(32):
            out = step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        out = step()
    barrier()
    path = ctx.last_path()

    # Kernel durations come from HIP events the engine records on its own stream, live inside the timed region — on every
    # 4th step only: an event pair around each of a step's three launches costs ~10 us of dispatch serialisation per
    # launch, which would otherwise be charged to `value`.
    ctx.profile_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ctx.profile_enable(i % 4 == 0)
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    recs = ctx.profile_read(16 * args.steps + 16)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # sanity: the result of the timed work is a real disparity map, produced by the kernel family the line reports
    got = out[:8, :64].cpu().numpy()
    assert got.shape == (8, 64, 3) and (got[..., 0] >= 0).all() and (got[..., 0] < sx).all()
    assert ctx.last_path() == path, "the kernel family changed during the timed region"
    ctx.set_option(core.OPT_DEFER_EXACTNESS, 0)

    if rank == 0:
        per_kernel = {}
        for name, ms in recs:
            per_kernel.setdefault(name, []).append(ms)
        kavg_us = {k: 1e3 * float(np.mean(v)) for k, v in per_kernel.items()}
        # dominant kernel of the path: the single matcher launch reads both float images and writes the disparity
        # image, i.e. it moves exactly the algorithmic bytes of SURVEY.md §8(d)
        hot = ["bm_sad_u8"] if path == core.PATH_SAD_U8 else ["bm_generic"]
        t_hot_us = sum(kavg_us.get(k, 0.0) for k in hot)
        strip_bytes = algorithmic_bytes(W, r1 - r0 + ky - 1, kx, ky, sx, sy)
        achieved = strip_bytes / (t_hot_us * 1e-6) / 1e9 if t_hot_us > 0 else 0.0
        evals = (r1 - r0) * ow * sx * sy
        traffic = traffic_src = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run" % tj.get("round", "r01")
            except Exception:
                traffic = None
        res = {
            "metric": "disparity Mpix/s, 4096x4096 pair, 7x7 SAD, +-64-px search",
            "value": ow * oh * args.steps / dt / 1e6,
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8" if path == core.PATH_SAD_U8 else "f64",
            "data": "synthetic (SplitMix64 integer-valued float32 noise pair, 256-px blocks shifted by 64+-48)",
            "config": {"workload": "BASELINE configs[1]: 4096x4096 pair, 7x7 SAD, search_volume 129x1, calc_disparity",
                       "kernel": list(KERNEL), "search_volume": list(SEARCH),
                       "partition": "%d row strip(s), no collective in the timed region" % world, "halo": halo,
                       "clock_settle_ms": args.settle_ms,
                       "path": {core.PATH_SAD_U8: "packed-u8 qsad", core.PATH_GENERIC_F64: "generic f64"}.get(path, "?")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "+".join(hot), "algorithmic_bytes_per_launch": strip_bytes,
                         "avg_us_per_launch": {k: kavg_us.get(k) for k in kavg_us},
                         "issue_bound_frac": (evals / (t_hot_us * 1e-6)) / (LANES * FCLK_HZ) if t_hot_us > 0 else None,
                         "note": "rank-0 strip; issue_bound_frac = (pixel x disparity evaluations per second) / "
                                 "(256 CU x 64 lane-ops/clk x 2.4 GHz): evaluations per VOP3 issue slot — the kernel "
                                 "is VALU-issue bound (~4.25 slots per evaluation at best), see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(left, right)
            # the checker: one full-image pass of the oracle against the result of the timed work, every pixel
            import oracle
            want, _ = oracle.calc_disparity_tiled(0, left, right, KERNEL, SEARCH, tile=256, threads=os.cpu_count() or 1)
            same = bool(np.array_equal(out.cpu().numpy(), want))
            res["cpu_baseline"]["result_identical_to_oracle"] = same
            assert same, "the timed result differs from the CPU oracle"
        else:
            res["cpu_baseline"] = None
        if world == 1 and not args.no_extra:
            try:                                  # the headline line must not depend on the side measurements
                res["extra"] = extra_points(ctx, torch, stereo, core, vwa, synth, l_strip, r_strip, left, right)
            except Exception as e:  # noqa: BLE001
                res["extra"] = [{"name": "extra points failed", "error": "%s: %s" % (type(e).__name__, str(e)[:300])}]
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
