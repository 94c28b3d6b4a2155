#!/usr/bin/env python3
"""bench.py — disparity Mpix/s of the block-matching hot path on MI355X (BASELINE.json metric).

Workload (BASELINE config 2): 4096x4096 synthetic stereo pair, 7x7 SAD, +-64 px search
(search_volume 129x1), one call of vw::stereo::calc_disparity per step, inputs (float32 PixelGray images)
already resident in HBM, output (PixelMask<Vector2i>, 12 B/px) left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: one rank per GPU.  Either launched by torch.distributed.run (RANK / WORLD_SIZE in the environment), or typed as
  above: bench.py then re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
  --master-addr 127.0.0.1 --master-port <free port>`.  The pair is split into N row strips (output rows
  [g*oh/N, (g+1)*oh/N) on rank g); the SOURCE images are row-sharded over the GPUs and the ky-1 (+sy-1) halo rows of a
  strip are fetched from the neighbouring rank before the clock starts — through the engine's C ABI
  (vwgpu_fetch_strip_window_dev: RCCL send / recv, csrc/halo.hip), with the torch.distributed mirror as the stated
  fallback; the choice is made collectively (an all-reduced flag), never per rank.  Block matching needs no exchange
  step inside the timed region: no data-path collective.  Total work is fixed: scaling = "strong".

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


def maybe_self_launch(gpus):
    """`python bench.py --gpus N` typed as is (no RANK in the environment): become the launcher of N ranks on this node."""
    if gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver supports dmabuf IPC only (RCCL peer mappings)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


class HaloFetcher:
    """Halo rows of a row-sharded source image.  First choice: the engine's own RCCL exchange behind the C ABI
    (partition.EngineComm -> vwgpu_fetch_strip_window_dev; the 128-byte unique id is made on rank 0 and broadcast over the
    process group).  Fallback: the torch.distributed mirror (partition.fetch_strip_window).  Every decision is COLLECTIVE —
    a flag all-reduced with MIN over the ranks — so that no rank takes one path while its neighbour waits in the other."""

    def __init__(self, torch, dist, vwa, partition, rank, world, dev):
        self.torch, self.dist, self.partition, self.rank, self.world, self.dev = torch, dist, partition, rank, world, dev
        self.comm = self.ctx = None
        self.how = "none (single rank holds every row)"
        if world == 1:
            return
        err = ""
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        puid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                uid = torch.tensor(list(partition.EngineComm.unique_id()), dtype=torch.uint8, device=dev)
                puid = torch.tensor(list(partition.EngineComm.unique_id()), dtype=torch.uint8, device=dev)
            except Exception as e:  # noqa: BLE001
                err = "unique id: %s" % (str(e)[:80],)
        # first contact in child processes (see halo_probe): a crash or a hang of the engine path ends a child, not this rank
        if self.agree(not err) and os.environ.get("VWGPU_BENCH_HALO_PROBE", "1") != "0":
            dist.broadcast(puid, 0)
            try:
                import subprocess
                spec = "%s:%d:%d:%d" % (bytes(puid.cpu().numpy().tobytes()).hex(), rank, world, dev.index)
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--halo-probe", spec], capture_output=True, text=True, timeout=120)
                if pr.returncode != 0:
                    err = "probe child exited %d: %s" % (pr.returncode, (pr.stderr or "")[-80:].replace("\n", " "))
            except Exception as e:  # noqa: BLE001  (incl. the timeout: a hung exchange)
                err = "probe: %s" % (str(e)[:80],)
            if not self.agree(not err) and not err:
                err = "probe failed on another rank"
        if self.agree(not err):
            dist.broadcast(uid, 0)
            try:
                self.ctx = vwa.Context(dev.index)
                self.comm = partition.EngineComm(self.ctx, bytes(uid.cpu().numpy().tobytes()), rank, world)
            except Exception as e:  # noqa: BLE001
                err = "communicator: %s" % (str(e)[:80],)
                self.comm = None
            if not self.agree(self.comm is not None):
                if self.comm is not None:
                    self.comm.close()
                self.comm = None
        self.how = ("engine C ABI: vwgpu_fetch_strip_window_dev (RCCL send/recv, csrc/halo.hip)" if self.comm is not None else
                    "torch.distributed isend/irecv (engine RCCL communicator unavailable%s)" % ((": " + err) if err else " on some rank"))

    def agree(self, ok):
        if self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def fetch(self, owned, rows_total, above, below):
        """owned = rows row_strip(rank, world, rows_total) of the image (contiguous, on the GPU) -> (window, first row)."""
        if self.world == 1:
            return owned, 0
        # The request is agreed on BEFORE anyone enters an exchange: if the ranks were handed different images or halos,
        # every rank learns it from the same all-reduced pair and raises — none is left waiting for a neighbour that took another way.
        hdr = self.torch.tensor([int(rows_total), int(above), int(below), int(owned.shape[1]) * owned.element_size()],
                                dtype=self.torch.int64, device=self.dev)
        lo, hi = hdr.clone(), hdr.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        if not bool((lo == hi).all().item()):
            raise ValueError("HaloFetcher.fetch: the ranks disagree about (rows, halo above, halo below, bytes per row): min %s max %s"
                             % (lo.tolist(), hi.tolist()))
        if self.comm is not None:
            win = first = None
            try:
                win, first = self.comm.fetch_strip_window(owned.contiguous(), rows_total, above, below)
                self.torch.cuda.synchronize(self.dev)
            except Exception as e:  # noqa: BLE001
                win = None
                self.how = "torch.distributed isend/irecv (engine exchange failed: %s)" % (str(e)[:80],)
            if self.agree(win is not None):
                return win, first
            self.comm.close()
            self.comm = None
            if not self.how.startswith("torch"):
                self.how = "torch.distributed isend/irecv (engine exchange failed on another rank)"
        return self.partition.fetch_strip_window(owned, self.rank, self.world, rows_total, above, below)

    def close(self):
        if self.comm is not None:
            self.comm.close()
        if self.ctx is not None:
            self.ctx.close()
        self.comm = self.ctx = None


def threaded_tiles(fn, jobs, threads, budget_s):
    """Runs fn(job) over `jobs` on a pool of `threads` host threads the way the reference's block rasteriser runs tiles
    (src/vw/Image/ImageIO.h:228-251: tasks pulled from a queue, one tile each), and stops handing out tiles once budget_s
    seconds of wall time have passed.  Returns (jobs finished, wall seconds).  (The oracle is C behind ctypes: no GIL.)"""
    import threading
    lock = threading.Lock()
    state = {"next": 0, "done": 0}
    t0 = time.perf_counter()

    def work():
        while True:
            with lock:
                i = state["next"]
                if i >= len(jobs) or "error" in state or (i >= threads and time.perf_counter() - t0 > budget_s):
                    return
                state["next"] = i + 1
            try:
                fn(jobs[i])
            except BaseException as e:                             # a worker's exception must not vanish with its thread
                with lock:
                    state.setdefault("error", e)
                return
            with lock:
                state["done"] += 1

    th = [threading.Thread(target=work) for _ in range(threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    if "error" in state:
        raise state["error"]
    return state["done"], time.perf_counter() - t0


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
    return {"value": done / dt / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port", "tile": tile,
            "tile_note": "256-px tiles = the library default (src/vw/Core/Settings.cc:183), SURVEY 8d's stated alternative to the 1024-px tiles of "
                         "tools/correlate.cc:266: a 4096^2 image is only 16 tiles of 1024^2, which would leave most of the host's cores idle",
            "sample": "%d x (%d of %d 256x256 output tiles of the same 4096^2 / 7x7 SAD / 129x1 pair) on %d threads, "
                      "%.1f s wall, %.0f core-seconds" % (reps, tiles, ntiles_total, cores, dt, dt * cores),
            "single_thread_ns_per_pixel_disparity": ns_per_op}


def cpu_baseline_sgm(synth, cost_type, k, D, budget_s=12.0):
    """The SGM oracle the way the reference runs a big pair: 1024^2 output tiles (tools/correlate.cc:266), one tile per task on T
    host threads, each tile single-threaded inside.  Bounded: every thread takes at least one tile, no new tile after budget_s."""
    import oracle
    cores = os.cpu_count() or 1
    T = max(1, min(cores, 32))                                            # ~0.45 GB of cost + path sums per tile in flight
    n = 1024 + k - 1
    left, right, _ = synth.stereo_pair(n, n, D, 1)
    oracle.set_sgm_host_threads(1)
    jobs = list(range(4 * T))
    if cost_type == 0:
        fn = lambda j: oracle.calc_disparity_sgm(0, left, right, (D - 1, 0), k, allow_block_cost=True)
    else:
        fn = lambda j: oracle.calc_disparity_sgm(cost_type, left, right, (D - 1, 0), k)
    done, dt = threaded_tiles(fn, jobs, T, budget_s)
    return {"value": done * 1024 * 1024 / dt / 1e6, "unit": "Mpix/s", "cores": T, "kind": "port", "tile": 1024,
            "sample": "%d tiles of 1024^2 (+ kernel rim) of the same kind of pair, %d disparities, one tile per task on %d host threads (of %d), %.1f s wall"
                      % (done, D, T, cores, dt)}


def cpu_baseline_pyramid(left, right, modes, search, levels, budget_s=8.0, keep=None):
    """oracle pyramid_correlate over 1024^2 tiles of the given pair on T host threads (tools/correlate.cc:207-266 + ImageIO.h:228-251).
    keep (a dict): receives {(mode name, tile bbox): the oracle's tile} for the tiles that were computed — the checker of the GPU tile loop."""
    import oracle
    cores = os.cpu_count() or 1
    H, W = left.shape
    tiles = [(x, y, 1024, 1024) for y in range(0, H - 1023, 1024) for x in range(0, W - 1023, 1024)]
    T = max(1, min(cores, 64))
    out = {}
    for m in modes:
        if m["alg"] == 0:
            run = lambda bb, m=m: oracle.pyramid_correlate(left, right, None, None, m["pf"], m["pfw"], search, m["kernel"], m["cost"], 0, 0.0, 2.0, 5, levels, bbox=bb)
        else:
            run = lambda bb, m=m: oracle.pyramid_correlate_sgm(left, right, None, None, search, m["kernel"][0], m["cost"], 2.0, 0, 5, levels, bbox=bb, algorithm=m["alg"])

        def fn(bb, m=m, run=run):
            o = run(bb)
            if keep is not None:
                keep.setdefault((m["name"], bb), o)
        jobs = [tiles[i % len(tiles)] for i in range(8 * T)]
        done, dt = threaded_tiles(fn, jobs, T, budget_s)
        out[m["name"]] = {"value": done * 1024 * 1024 / dt / 1e6, "unit": "Mpix/s", "cores": T, "kind": "port", "tile": 1024,
                          "sample": "%d tiles of 1024^2 of a %dx%d pair through the oracle's pyramid_correlate, one tile per task on %d host threads (of %d), "
                                    "%.1f s wall" % (done, W, H, T, cores, dt)}
    return out


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


def extra_points(ctx, torch, stereo, core, vwa, synth, lt, rt, left, right, with_cpu=True):
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

    cores = os.cpu_count() or 1

    def cpu(fn, jobs, threads, px_per_job, what, budget_s=4.0):
        """A bounded CPU leg beside a GPU point: the oracle on `threads` host threads, one tile per task (see threaded_tiles)."""
        if not with_cpu:
            return None
        done, dt = threaded_tiles(fn, jobs, threads, budget_s)
        return {"value": done * px_per_job / dt / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
                "sample": "%d x %s on %d host threads (of %d), %.1f s wall" % (done, what, threads, cores, dt)}

    import oracle
    bb = vwa.bounding_box(left)
    # (1) the headline kernel at +-16 px: 33 disparities
    r33 = rt[:, 48:48 + W + 32].contiguous()
    wall, kern = measure(ctx, torch, lambda: stereo.calc_disparity(0, lt, r33, bb, (33, 1), KERNEL, ctx=ctx), 40)
    entry("4096^2, 7x7 SAD, search 33x1 (+-16 px)", algorithmic_bytes(W, H, 7, 7, 33, 1), "4LW + 4RW + 12 out (SURVEY 8d)", wall, kern,
          {"bm_sad_u8"}, (W - 6) * (H - 6), path=ctx.last_path(),
          cpu_baseline=cpu(lambda j: oracle.calc_disparity(0, left[j:j + 262, :1030], right[j:j + 262, 48:48 + 1030 + 32], KERNEL, (33, 1)),
                           [256 * (i % 15) for i in range(16 * cores)], cores, 256 * 1024, "256 x 1024-px output tile, 7x7 SAD, 33 disparities"))
    # (2) BASELINE config 3: 11x11 NCC over 129x1, then parabola_subpixel on the result
    wall, kern = measure(ctx, torch, lambda: stereo.calc_disparity(2, lt, rt, bb, SEARCH, (11, 11), ctx=ctx), 10)
    entry("config 3a: 4096^2, 11x11 NCC, search 129x1", algorithmic_bytes(W, H, 11, 11, 129, 1), "4LW + 4RW + 12 out (SURVEY 8d)", wall, kern,
          {"bm_corr_u8", "ncc_full", "bm_dot_u8"}, (W - 10) * (H - 10), path=ctx.last_path(),
          cpu_baseline=cpu(lambda j: oracle.calc_disparity(2, left[j:j + 266, :266], right[j:j + 266, :266 + 128], (11, 11), SEARCH),
                           [256 * (i % 15) for i in range(8 * cores)], cores, 256 * 256, "256^2 output tile (the library's default tile), 11x11 NCC, 129 disparities"))
    # (2b) the same call on 12-bit imagery (pixels * 16 + 7: integers in [7, 4087]): v_dot2_u32_u16 pairs instead of v_dot4_u32_u8 quads
    l12, r12 = lt * 16.0 + 7.0, rt * 16.0 + 7.0
    left12, right12 = left * np.float32(16.0) + np.float32(7.0), right * np.float32(16.0) + np.float32(7.0)
    wall, kern = measure(ctx, torch, lambda: stereo.calc_disparity(2, l12, r12, bb, SEARCH, (11, 11), ctx=ctx), 10)
    entry("config 3a on 12-bit imagery: 4096^2, 11x11 NCC, search 129x1", algorithmic_bytes(W, H, 11, 11, 129, 1), "4LW + 4RW + 12 out (SURVEY 8d)", wall, kern,
          {"bm_corr_u16", "ncc_full"}, (W - 10) * (H - 10), path=ctx.last_path(),
          cpu_baseline=cpu(lambda j: oracle.calc_disparity(2, left12[j:j + 266, :266], right12[j:j + 266, :266 + 128], (11, 11), SEARCH),
                           [256 * (i % 15) for i in range(8 * cores)], cores, 256 * 256, "256^2 output tile (the library's default tile), 11x11 NCC, 129 disparities"))
    del l12, r12
    d = stereo.calc_disparity(2, lt, rt, bb, SEARCH, (11, 11), ctx=ctx)
    disp = torch.zeros((H, W, 3), dtype=torch.float32, device=lt.device)
    disp[5:5 + H - 10, 5:5 + W - 10, :2] = d[..., :2].float()
    disp[5:5 + H - 10, 5:5 + W - 10, 2] = (d[..., 2] != 0).float()
    wall, kern = measure(ctx, torch, lambda: stereo.parabola_subpixel(disp, lt, rt, 0, 0.0, (11, 11), ctx=ctx), 10)
    pb = 4 * W * H + 4 * (W + 128) * H + 12 * W * H + 12 * W * H
    entry("config 3b: parabola_subpixel 11x11 on the 4096^2 NCC result", pb, "4LW + 4RW + 12 disparity in + 12 out (SURVEY 8d)", wall, kern,
          {k for k in kern if k.startswith("parabola") or k in ("disparity_range", "float_grain", "edge_extend_sub")}, W * H,
          cpu_baseline=cpu(lambda j, dh=disp[:512, :512].cpu().numpy(): oracle.parabola_subpixel(dh, left[:512, :512], right[:512, :512 + 128], 0, 0.0, (11, 11)),
                           list(range(4 * cores)), cores, 512 * 512, "512^2 crop of the same NCC result through parabola_subpixel 11x11"))
    # (2c) SURVEY 8(d): "one non-integer (float texture) input per config and report mismatch rate".  The float twin of the pair (pixel * 0.37 +
    # uniform noise in [0, 1), independent per image — tests/fuzz_cases.py:47) through the same calc_disparity calls as configs 2 and 3a: which
    # kernel family served it, Mpix/s, roofline fraction, and the whole image against the oracle (single-threaded over the whole raster, as
    # best_of_search_convolution runs it: on such data the running sums' roundings depend on the raster position, Correlation.cc:33-137).
    rng = np.random.default_rng(20260926)
    left_f = (left * np.float32(0.37) + rng.random(left.shape, dtype=np.float32)).astype(np.float32)
    right_f = (right * np.float32(0.37) + rng.random(right.shape, dtype=np.float32)).astype(np.float32)
    lf, rf = torch.from_numpy(left_f).to(lt.device), torch.from_numpy(right_f).to(lt.device)
    want_f = {}
    oth = []
    if with_cpu:
        import threading
        for c_, kk_ in ((0, KERNEL), (2, (11, 11))):
            th_ = threading.Thread(target=lambda c_=c_, kk_=kk_: want_f.__setitem__(c_, oracle.calc_disparity(c_, left_f, right_f, kk_, SEARCH)))
            th_.start(); oth.append(th_)
    fl = []
    for c_, kk_, nm in ((0, KERNEL, "config 2"), (2, (11, 11), "config 3a")):
        wall, kern = measure(ctx, torch, lambda: stereo.calc_disparity(c_, lf, rf, bb, SEARCH, kk_, ctx=ctx), 5, warm=1)
        fl.append((c_, kk_, stereo.calc_disparity(c_, lf, rf, bb, SEARCH, kk_, ctx=ctx)))
        entry("%s on a FLOAT texture (pixel * 0.37 + uniform noise): 4096^2, %dx%d %s, search 129x1" % (nm, kk_[0], kk_[1], "SAD" if c_ == 0 else "NCC"),
              algorithmic_bytes(W, H, kk_[0], kk_[1], 129, 1), "4LW + 4RW + 12 out (SURVEY 8d)", wall, kern, set(kern), (W - kk_[0] + 1) * (H - kk_[1] + 1),
              path=ctx.last_path())
    for th_ in oth:
        th_.join()
    for (c_, kk_, got_), e in zip(fl, out[-2:]):
        if c_ in want_f:
            g = got_.cpu().numpy()
            bad = int((g != want_f[c_]).any(-1).sum())
            e["identical_to_oracle"] = bad == 0
            e["mismatch_rate"] = bad / float(g.shape[0] * g.shape[1])
            e["oracle_check"] = "every pixel of the %d x %d image against oracle.calc_disparity over the whole raster (one host thread)" % (g.shape[1], g.shape[0])
            assert bad == 0, "float-texture %s: %d pixels differ from the oracle" % (e["name"], bad)
    del lf, rf, fl
    # (2d) SURVEY 8(d)'s second column: END TO END — the host-pointer entry of the C ABI (vwgpu_calc_disparity: H2D of both images, the
    # matcher, D2H of the disparity image, one synchronisation), on pageable numpy arrays and on page-locked ones
    def e2e(l_, r_, o_):
        rc_ = ctx._lib.vwgpu_calc_disparity(ctx._h, 0, l_, W, H, W, r_, right.shape[1], right.shape[0], right.shape[1], KERNEL[0], KERNEL[1], SEARCH[0], SEARCH[1], o_, 0)
        ctx.check(rc_)
    ow_, oh_ = W - KERNEL[0] + 1, H - KERNEL[1] + 1
    res_e2e = {}
    out_np = np.empty((oh_, ow_, 3), np.int32)
    pl_, pr_ = torch.from_numpy(left).pin_memory(), torch.from_numpy(right).pin_memory()
    po_ = torch.empty((oh_, ow_, 3), dtype=torch.int32).pin_memory()
    for kind, args_ in (("pageable", (left.ctypes.data, right.ctypes.data, out_np.ctypes.data)), ("pinned", (pl_.data_ptr(), pr_.data_ptr(), po_.data_ptr()))):
        e2e(*args_)
        t0 = time.perf_counter()
        for _ in range(3):
            e2e(*args_)
        dt_ = (time.perf_counter() - t0) / 3
        res_e2e[kind] = {"ms_per_call": round(dt_ * 1e3, 3), "Mpix_per_s": round(ow_ * oh_ / dt_ / 1e6, 1)}
    assert np.array_equal(out_np, po_.numpy())
    moved = 4 * W * H + 4 * right.shape[0] * right.shape[1] + 12 * ow_ * oh_
    out.append({"name": "headline END TO END: vwgpu_calc_disparity on host pointers (H2D + matcher + D2H), 4096^2, 7x7 SAD, search 129x1",
                "bytes_over_pcie": int(moved), "pageable": res_e2e["pageable"], "pinned": res_e2e["pinned"],
                "pcie_GBs_pinned": round(moved / (res_e2e["pinned"]["ms_per_call"] * 1e-3) / 1e9, 1),
                "note": "never `value`: the device-resident column is the headline (inputs in HBM when the clock starts)"})
    del pl_, pr_, po_
    # (2c) the headline with DEFAULT options: no VWGPU_OPT_DEFER_EXACTNESS — every call waits for the input-class flags of its launch (one host
    # round trip) before it returns; inputs device resident
    dctx = vwa.Context(lt.device.index)
    dcall = lambda: stereo.calc_disparity(0, lt, rt, vwa.bounding_box(left), (129, 1), (7, 7), ctx=dctx)
    for _ in range(5): dcall()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): dcall()
    torch.cuda.synchronize()
    dms = (time.perf_counter() - t0) / 20 * 1e3
    dpath = dctx.last_path()
    dcall_ref = dcall()
    torch.cuda.synchronize()
    dctx.close()
    out.append({"name": "headline with DEFAULT options (no VWGPU_OPT_DEFER_EXACTNESS): device-resident calc_disparity, 4096^2, 7x7 SAD, search 129x1",
                "wall_ms_per_call": round(dms, 4), "Mpix_per_s": round((W - 6) * (H - 6) / (dms * 1e-3) / 1e6, 1), "path": int(dpath),
                "note": "matcher + the host round trip for the input-class flags of this launch; the deferred mode of the headline queues calls back to back"})
    # (2d) two headline steps IN FLIGHT: two engine contexts on two streams, steps handed out in turn (the reference keeps several tiles in flight
    # per device, one thread per tile).  The staging burst and the tail of one launch hide behind the other's sweep.  Not `value`: the contract's
    # step is one launch at a time, and the per-launch duration (what `roofline` divides by) grows when two launches share the CUs.
    fctx = [vwa.Context(lt.device.index) for _ in range(2)]
    fstr = [torch.cuda.Stream(device=lt.device) for _ in range(2)]
    for c_ in fctx: c_.set_option(core.OPT_DEFER_EXACTNESS, 1)

    def fstep(i):
        with torch.cuda.stream(fstr[i & 1]):
            return stereo.calc_disparity(0, lt, rt, vwa.bounding_box(left), (129, 1), (7, 7), ctx=fctx[i & 1])
    for i in range(40): fstep(i)                                   # (the allocator's per-stream pools and the clocks settle)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200): fo = fstep(i)
    torch.cuda.synchronize()
    fms = (time.perf_counter() - t0) / 200 * 1e3
    fsame = bool(torch.equal(fo, dcall_ref)) if dcall_ref is not None else None
    for c_ in fctx: c_.close()
    out.append({"name": "headline, TWO steps in flight (two contexts on two streams): 4096^2, 7x7 SAD, search 129x1",
                "wall_ms_per_step": round(fms, 4), "Mpix_per_s": round((W - 6) * (H - 6) / (fms * 1e-3) / 1e6, 1), "identical_to_serial_result": fsame,
                "note": "throughput of a pipelined caller; the headline keeps one launch at a time"})
    # (3) SGM building block of config 4: 2048^2, census 7x7, 129 disparities, 8 paths, LC-blend sub-pixel
    n = 2048
    ls, rs_ = lt[:n, :n].contiguous(), rt[:n, :n + 128].contiguous()
    sgm = lambda: stereo.calc_disparity_sgm(3, ls, rs_, vwa.BBox2i(0, 0, n, n), (128, 0), (7, 7), with_subpixel=True, memory_limit_mb=200000, ctx=ctx)
    wall, kern = measure(ctx, torch, sgm, 5, warm=1)
    npx = (n - 6) * (n - 6)
    entry("config 4 building block: SGM 2048^2, census 7x7, 129 disparities", npx * (20 + 11 * 129),
          "materialised volume (20 + 11 D) B/px (SURVEY 8d; the minimum model is 20 + 4 D = 536 B/px)", wall, kern,
          {k for k in kern if k.startswith("sgm")}, npx, min_model_frac=None,
          cpu_baseline=cpu_baseline_sgm(synth, 3, 7, 129, budget_s=5.0) if with_cpu else None)
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
          {k for k in kern if k.startswith("sgm")}, npx, fronts=2 * n - 13,
          cpu_baseline=cpu(lambda j, a=left[:518, :518].copy(), b=right[:518, :518 + 128].copy(): oracle.calc_disparity_sgm(3, a, b, (128, 0), 7, use_mgm=True),
                           list(range(2 * min(cores, 32))), min(cores, 32), 512 * 512, "512^2 tile, census 7x7, 129 disparities, use_mgm"))
    # (4) the tile loop of config 5 (tools/correlate.cc:207-266): pyramid_correlate over the 4096^2 pair in 1024^2 tiles, pulled by
    # 4 tile threads, each with its own engine context and stream — the way block_write_image runs the reference's view
    import threading
    rc = rt[:, 64:64 + W].contiguous()
    tiles = [vwa.BBox2i(x, y, 1024, 1024) for y in range(0, H, 1024) for x in range(0, W, 1024)]
    search = vwa.BBox2i.from_corners((-64, -1), (64, 1))
    for label, pf, pw, cost, kk in (("SAD 7x7", 0, 0.0, 0, 7), ("LoG 1.4 + NCC 11x11 (the correlate tool's defaults)", 2, 1.4, 2, 11)):
        # the tiles are handed over in GROUPS (vwgpu_pyramid_correlate_batch_dev, round 5): the 4 tiles of a tile row go through the level loop
        # together — every launch serves the group, one host round trip per level — one group per tile thread
        T, G = 4, 4
        ctxs = [vwa.Context(lt.device.index) for _ in range(T)]
        streams = [torch.cuda.Stream(device=lt.device) for _ in range(T)]
        outs = [None] * len(tiles)
        groups = [list(range(i, min(i + G, len(tiles)))) for i in range(0, len(tiles), G)]
        def work(t):
            with torch.cuda.stream(streams[t]):
                for gi in range(t, len(groups), T):
                    got_ = stereo.pyramid_correlate_batch(lt, rc, None, None, pf, pw, search, (kk, kk), cost, [tiles[i] for i in groups[gi]], consistency_threshold=2,
                                                          filter_half_kernel=5, max_pyramid_levels=5, ctx=ctxs[t])
                    for i, o_ in zip(groups[gi], got_): outs[i] = o_
        best = None
        for rep in range(7):                                   # one warm-up pass, then the fastest of six (a pass is 10 - 40 ms: host noise shows)
            torch.cuda.synchronize(lt.device); t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            [x.start() for x in th]; [x.join() for x in th]
            torch.cuda.synchronize(lt.device); dt_ = time.perf_counter() - t0
            if rep > 0: best = dt_ if best is None else min(best, dt_)
        # launches of one more pass (the kernels the engine brackets with events; fills and copies are not counted)
        for c_ in ctxs: c_.profile_reset(); c_.profile_enable(True)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize(lt.device)
        launches = 0
        for c_ in ctxs:
            c_.profile_enable(False)
            launches += len(c_.profile_read(1 << 16))
        for c_ in ctxs: c_.close()
        nl = [1024 * 1024 / 4 ** l for l in range(6)]
        tile_bytes = 2 * (4.0 / 3.0) * 5 * nl[0] + sum((20 + 24) * n for n in nl) + 20 * nl[0]       # the config-5 byte model (run_config5)
        cb = ident = None
        if with_cpu:
            # the oracle leg is also the checker: every tile it computes is compared with the tile of the LAST timed pass, pixel for pixel
            kept = {}
            cb = cpu_baseline_pyramid(left, np.ascontiguousarray(right[:, 64:64 + W]), [dict(name="bm", pf=pf, pfw=pw, kernel=(kk, kk), cost=cost, alg=0)],
                                      (-64, -1, 64, 1), 5, budget_s=4.0, keep=kept)["bm"]
            same = 0
            for (_, bbt), want in kept.items():
                i = next(k for k, tb in enumerate(tiles) if (tb.min[0], tb.min[1]) == (bbt[0], bbt[1]))
                same += int(np.array_equal(outs[i].cpu().numpy(), want))
            ident = "%d / %d" % (same, len(kept))
            assert same == len(kept), "tile loop (%s): %s tiles identical to the oracle" % (label, ident)
        del outs
        out.append({"name": "config 5 building block: pyramid_correlate tile loop, 4096^2 in 16 tiles of 1024^2, %s, +-64 x +-1, 5 levels, L/R check, "
                            "4 tile threads x groups of 4 tiles (vwgpu_pyramid_correlate_batch_dev)" % label,
                    "kernel_launches_per_tile": round(launches / len(tiles), 1),
                    "wall_ms_per_pair": round(best * 1e3, 2), "ms_per_tile": round(best * 1e3 / len(tiles), 3),
                    "Mpix_per_s": round(W * H / best / 1e6, 1), "algorithmic_bytes": int(tile_bytes * len(tiles)),
                    "bytes_model": "SURVEY 8d summed over the levels of a tile (pyramid build + BM bytes per level, level 0 twice + clean-up chain)",
                    "roofline_frac": round(tile_bytes * len(tiles) / best / 1e9 / HBM_PEAK_GBS, 5), "cpu_baseline": cb,
                    "tiles_identical_to_oracle": ident,
                    "note": "throughput of the threaded tile loop (launch / latency bound, not byte bound); per-kernel times of one tile: tools/pyr_profile.py"})
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
    fetcher = HaloFetcher(torch, dist, vwa, partition, rank, world, dev)
    if world > 1:
        lwin, first = fetcher.fetch(lwin, H, halo, halo)
        rwin, _ = fetcher.fetch(rwin, H, halo, halo)
    how = fetcher.how if world == 1 else "%s: %d collar + %d half-kernel rows per neighbour" % (fetcher.how, collar, k // 2)
    fetcher.close()
    torch.cuda.synchronize(dev)
    ctx = vwa.Context(dev.index)
    rows_per = H // nstrips
    mine = [s for s in range(nstrips) if a <= s * rows_per < b]
    hk = k // 2
    cost_type = 3 if args.cost == "census" else 0                        # 0 = ABSOLUTE_DIFFERENCE: the MAD block cost (opt-in, see --cost)

    def strip_rows(s_):
        """(first / last+1 centre row the strip keeps, first / last+1 input row of strip + collar + half kernel): a strip keeps the
        centre rows of [y0, y1) that have a full window, [hk, H - hk) — the kept rows of all strips partition the valid rows."""
        y0, y1 = s_ * rows_per, (s_ + 1) * rows_per
        ka, kb = max(y0, hk), min(y1, H - hk)
        return ka, kb, max(0, ka - hk - collar), min(H, kb + hk + collar)

    def step():
        outs = []
        for s_ in mine:
            ka, kb, ra, rb = strip_rows(s_)
            l = lwin[ra - first:rb - first]
            r = rwin[ra - first:rb - first]
            d = stereo.calc_disparity_sgm(cost_type, l, r, vwa.BBox2i(0, 0, W, rb - ra), (D - 1, 0), (k, k), with_subpixel=True,
                                          memory_limit_mb=200000, allow_block_cost=(cost_type == 0), ctx=ctx)[0]
            top = ka - (ra + hk)                                         # output row j of the call is centred on input row ra + hk + j
            outs.append(d[top:top + (kb - ka)])
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
    npx = (W - k + 1) * (H - k + 1)                                      # the kept rows of all strips: every pixel with a full window, once
    got = out[0]
    ka, kb, _, _ = strip_rows(mine[0])
    assert got.shape[0] == kb - ka
    tr = truth[ka - a:kb - a, hk:hk + W - k + 1]
    ok = float((got[..., 0].cpu().numpy() == tr).mean())
    if rank == 0:
        per_px = 20 + 11 * D
        cpu = None if args.no_cpu_baseline else cpu_baseline_sgm(synth, cost_type, k, D)
        print(json.dumps({
            "metric": "disparity Mpix/s, 16384x16384 pair, %s SGM, 129 disparities, 8 row strips + collar" % ("census 7x7" if cost_type == 3 else "7x7 MAD block cost"),
            "value": npx * steps / dt / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": steps, "warmup": 1,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8/u16",
            "data": "synthetic (SplitMix64 integer-valued float32 noise pair, 256-px blocks shifted by 64+-48)",
            "config": {"workload": "BASELINE configs[3], tile + collar semantics; cost = %s" % (
                           "census 7x7 (what the reference's SGM accepts, SURVEY F3)" if cost_type == 3 else
                           "7x7 mean-abs-difference block cost = the config as written; reference code path present but unreachable upstream "
                           "(SGM.cc:1651-1738 behind the throw at :1887-1892), enabled here by an explicit opt-in"),
                       "strips": nstrips, "strip_rows": rows_per, "collar_rows": collar, "halo": how,
                       "integer_match_rate_vs_truth_first_strip": ok},
            "roofline": {"bound": "hbm", "achieved": npx * per_px * steps / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": npx * per_px * steps / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "min_model_frac": npx * (20 + 4 * D) * steps / dt / 1e9 / HBM_PEAK_GBS,
                         "bytes_model": "materialised volume (20 + 11 D) B per pixel, SURVEY 8d (min_model_frac: its minimum model, 20 + 4 D)",
                         "kernel": "calc_disparity_sgm (wall)"},
            "cpu_baseline": cpu}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def check_config5_tiles(synth, keep, bm_xy, sgm_xy, modes, W, H, D, TILE, LEVELS):
    """The sampled tiles of the config-5 loop against the oracle (checker leg, after the clock): the oracle runs pyramid_correlate on the
    window of the pair a tile can touch — tile grown by half_kernel * 2^levels + twice the search extent + the search + 8 on every side, cut at
    the image borders, exactly what the library's host entry stages (csrc/pyramid.hip, vwgpu_pyramid_correlate) — so its tile is the tile
    of the whole pair."""
    import oracle
    search = (-64, -1, 64, 1)                                     # the corners of run_config5's BBox2i.from_corners((-64, -1), (64, 1))
    out = {"bm_checked": 0, "bm_identical": 0, "sgm_checked": 0, "sgm_identical": 0, "tiles": []}
    jobs = [("bm", x, y) for (x, y) in bm_xy] + [("sgm", x, y) for (x, y) in sgm_xy]
    rows = {}

    def window(x, y, kernel):
        pad_x = (kernel // 2) * (1 << LEVELS) + 2 * 128 + 8 + 64
        pad_y = (kernel // 2) * (1 << LEVELS) + 2 * 2 + 8 + 1
        x0, y0 = max(0, x - pad_x), max(0, y - pad_y)
        x1, y1 = min(W, x + TILE + pad_x), min(H, y + TILE + pad_y)
        key = (y0, y1)
        if key not in rows:
            l, r, _ = synth.stereo_pair_rows(W, H, D, y0, y1)
            rows[key] = (l, np.ascontiguousarray(r[:, 64:64 + W]))
            if len(rows) > 2:
                rows.pop(next(iter(rows)))
        l, r = rows[key]
        return np.ascontiguousarray(l[:, x0:x1]), np.ascontiguousarray(r[:, x0:x1]), (x - x0, y - y0, TILE, TILE)

    work = []
    for name, x, y in sorted(jobs, key=lambda j: (j[2], j[1])):      # by tile row: the row band is generated once
        m = next(mm for mm in modes if mm["name"] == name)
        work.append((name, x, y, m) + window(x, y, m["kernel"][0]))

    def fn(w):
        name, x, y, m, l, r, bb = w
        if m["alg"] == 0:
            o = oracle.pyramid_correlate(l, r, None, None, m["pf"], m["pfw"], search, m["kernel"], m["cost"], 0, 0.0, 2.0, 5, LEVELS, bbox=bb)
        else:
            o = oracle.pyramid_correlate_sgm(l, r, None, None, search, m["kernel"][0], m["cost"], 2.0, 0, 5, LEVELS, bbox=bb, algorithm=m["alg"])
        g = keep[(name, x, y)].cpu().numpy()
        same = bool(np.array_equal(g, o))
        out[name + "_checked"] += 1
        out[name + "_identical"] += int(same)
        out["tiles"].append([name, x, y, same])

    threaded_tiles(fn, work, max(1, min(os.cpu_count() or 1, len(work))), 1e9)
    out["tiles"].sort()
    assert out["bm_checked"] == len(bm_xy) and out["sgm_checked"] == len(sgm_xy), out      # every sampled tile was compared
    return out


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
    fetcher = HaloFetcher(torch, dist, vwa, partition, rank, world, dev)
    how = fetcher.how
    if world > 1:
        above, below = partition.pyramid_halo_rows(11, LEVELS, -1, 1)
        hal = max(above, below) + 2 * 3                            # SGM R->L runs reach twice the vertical search extent
        lwin, first = fetcher.fetch(lwin, H, hal, hal)
        rwin, _ = fetcher.fetch(rwin, H, hal, hal)
        how = "%s: %d pyramid halo rows per neighbour" % (fetcher.how, hal)
    fetcher.close()
    torch.cuda.synchronize(dev)
    tiles = [(x, y) for y in range(a, b, TILE) for x in range(0, W, TILE)]
    T = 4
    ctxs = [vwa.Context(dev.index) for _ in range(T)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
    keep = {}
    # tiles compared with the oracle after the clock has stopped: corners, the four borders, interior (rank 0's strip; VWGPU_BENCH_CONFIG5_CHECK
    # = how many, default 16 at full size, 6 below; 0 = none); one of them also for SGM
    ntx, nty = W // TILE, (b - a) // TILE
    ncheck = int(os.environ.get("VWGPU_BENCH_CONFIG5_CHECK", "16" if W >= 32768 else "6"))
    cand = [(0, 0), (ntx - 1, nty - 1), (ntx - 1, 0), (0, nty - 1), (ntx // 2, 0), (0, nty // 2), (ntx - 1, nty // 2), (ntx // 2, nty - 1),
            (ntx // 2, nty // 2), (1, 1), (ntx // 3, nty // 4), (ntx - 2, nty // 3), (ntx // 4, nty - 2), (2 * ntx // 3, 2 * nty // 3), (ntx // 5, nty // 2), (3 * ntx // 4, nty // 5)]
    sample = []
    for c_ in cand:
        if c_ not in sample and len(sample) < ncheck: sample.append(c_)
    sample_xy = {(tx * TILE, a + ty * TILE) for tx, ty in sample} if rank == 0 else set()      # (independent of --no-cpu-baseline: a checker, not a baseline)
    sgm_xy = {(0, a)} if sample_xy else set()

    G = 8                                                          # tiles per group (vwgpu_pyramid_correlate_batch_dev; SGM tiles run one by one inside the call)

    def loop(kw, only=None):
        todo = list(tiles if only is None else tiles[-only:])      # (the timed loop pops from the end: warm up with the tiles it starts with)
        lock = threading.Lock()
        def work(t):
            with torch.cuda.stream(streams[t]):
                while True:
                    with lock:
                        if not todo: return
                        grp = [todo.pop() for _ in range(min(G if kw["alg"] == 0 else 1, len(todo)))]
                    got_ = stereo.pyramid_correlate_batch(lwin, rwin, None, None, kw["pf"], kw["pfw"], search, kw["kernel"], kw["cost"],
                                                          [vwa.BBox2i(x, y - first, TILE, TILE) for x, y in grp], consistency_threshold=2,
                                                          filter_half_kernel=5, max_pyramid_levels=LEVELS, algorithm=kw["alg"], ctx=ctxs[t])
                    for (x, y), o in zip(grp, got_):
                        if (x, y) == (0, a): keep[kw["name"]] = o
                        if (x, y) in (sample_xy if kw["alg"] == 0 else sgm_xy): keep[(kw["name"], x, y)] = o
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    modes = [dict(name="bm", pf=2, pfw=1.4, kernel=(11, 11), cost=2, alg=0), dict(name="sgm", pf=0, pfw=0.0, kernel=(7, 7), cost=3, alg=1)]
    res, passes = {}, {}
    for kw in modes:
        loop(kw, only=(G if kw["alg"] == 0 else 2) * T)            # warm-up: arenas of every context sized
        barrier()
        t0 = time.perf_counter()
        loop(kw)
        barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        res[kw["name"]] = float(tmax.item())
        passes.setdefault(kw["name"], []).append(res[kw["name"]])
        if kw["alg"] == 0 and W <= 16384:
            # A second pass of the block-matching loop, reported beside the first: four Python threads feed the groups, and which thread
            # gets the last groups of a 64-tile loop decides 10-20 % of its wall time (two runs of the same child in round 5: 1424 and 1712
            # Mpix/s).  `value` stays the FIRST pass (the contract times one step); `passes_s` shows the spread.
            barrier()
            t0 = time.perf_counter()
            loop(kw)
            barrier()
            passes[kw["name"]].append(time.perf_counter() - t0)
    for c_ in ctxs: c_.close()
    if rank == 0:
        npx = W * H
        ok = {}
        for name, o in keep.items():
            if not isinstance(name, str): continue
            g = o.cpu().numpy()
            ok[name] = float(((np.rint(g[..., 0]) == truth0) & (g[..., 2] != 0)).mean())
        # Byte model of one tile (summed over the levels): the pyramid build reads 4 N_src and writes N_src per image and
        # level ( ~ (4/3) 5 N per image), every level's matcher moves the BM bytes of its size (4 L + 4 R + 12 out; the top level over the
        # whole search, level 0 twice: L->R and R->L for the consistency check), every level's disparity is read and written once more by
        # the clean-up chain (24 B/px).  SGM adds the materialised volume of its per-pixel boxes: (3 x 25 + 2 x 8 x 25) B/px at 5 x 5 boxes.
        nl = [TILE * TILE / 4 ** l for l in range(LEVELS + 1)]
        bm_tile = 2 * (4.0 / 3.0) * 5 * nl[0] + sum((20 + 24) * n for n in nl) + 20 * nl[0]
        sgm_tile = bm_tile + sum(11 * 25 * n for n in nl) + 11 * 25 * nl[0]
        ntiles = (W // TILE) * (H // TILE)
        cpu = None
        checked = None
        if sample_xy:
            checked = check_config5_tiles(synth, keep, sorted(sample_xy), sorted(sgm_xy), modes, W, H, D, TILE, LEVELS)
            assert checked["bm_identical"] == checked["bm_checked"] and checked["sgm_identical"] == checked["sgm_checked"], checked
        if not args.no_cpu_baseline:
            l0, r0, _ = synth.stereo_pair_rows(W, H, D, 0, 2 * TILE)
            cpu = cpu_baseline_pyramid(np.ascontiguousarray(l0[:, :8 * TILE]), np.ascontiguousarray(r0[:, 64:64 + 8 * TILE]), modes, (-64, -1, 64, 1), LEVELS)
        print(json.dumps({
            "metric": "disparity Mpix/s, %dx%d pair, pyramid_correlate tile loop (LoG 1.4 + NCC 11x11, 5 levels, +-64 x +-1, L/R check)" % (W, H),
            "value": npx / res["bm"] / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": res["bm"] * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic (SplitMix64 integer-valued float32 noise pair, 256-px blocks shifted by 0+-48)",
            "config": {"workload": "BASELINE configs[4]: orbital-scale pair through the reference's tile loop; block matching with the correlate "
                                   "tool's defaults and, separately, SGM (census 7x7)", "tile": TILE, "tiles": ntiles,
                       "tile_threads_per_gpu": T, "tiles_per_group": G, "halo": how,
                       "sgm": {"Mpix_per_s": npx / res["sgm"] / 1e6, "s_per_pair": res["sgm"],
                               "roofline_frac": ntiles * sgm_tile / res["sgm"] / 1e9 / HBM_PEAK_GBS,
                               "cpu_baseline": None if cpu is None else cpu["sgm"]},
                       "truth_match_rate_first_tile": ok, "tiles_vs_oracle": checked,
                       "passes_s": {k: [round(v, 4) for v in vs] for k, vs in passes.items()}},
            "roofline": {"bound": "hbm", "achieved": ntiles * bm_tile / res["bm"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ntiles * bm_tile / res["bm"] / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "pyramid_correlate tile loop (wall)",
                         "algorithmic_bytes_per_tile": int(bm_tile),
                         "bytes_model": "SURVEY 8d summed over the levels of a tile: pyramid build + BM bytes per level (level 0 twice) + clean-up chain; "
                                        "the loop is launch / latency bound (about 100 dependent launches per tile), not byte bound"},
            "cpu_baseline": None if cpu is None else cpu["bm"]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_traffic():
    """HBM bytes per launch of the headline kernel from the PMC counters, measured NOW: two rocprofv3 passes of this very command (short:
    8 steps, no CPU leg, no extras), one per counter — FETCH_SIZE and WRITE_SIZE do not fit one pass — read back from the rocpd database.
    gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts the 128-byte requests of wide
    coalesced reads at 64 bytes -> x2; WRITE_SIZE as is; both in KiB.  Returns (bytes per launch or None, how it was obtained)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="vwgpu_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "8", "--warmup", "2", "--settle-ms", "30", "--no-cpu-baseline", "--no-extra", "--no-traffic"]
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if p.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, p.returncode)
            cur = sqlite3.connect(dbs[0]).cursor()
            row = cur.execute("select count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like '%bm_sad_u8_kernel%'",
                              (counter,)).fetchone()
            if not row or not row[0]:
                return None, "no %s samples of bm_sad_u8_kernel in the profile" % counter
            got[counter] = (int(row[0]), float(row[1]))
        except Exception as e:  # noqa: BLE001  (the headline line must not depend on the profiler)
            return None, "PMC pass failed: %s" % (str(e)[:80],)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    total = int(got["FETCH_SIZE"][1] * 1024 * 2 + got["WRITE_SIZE"][1] * 1024)
    return total, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --steps 8` "
                   "(%d / %d launches): FETCH_SIZE %.0f KiB x 2 (gfx950 correction) + WRITE_SIZE %.0f KiB"
                   % (got["FETCH_SIZE"][0], got["WRITE_SIZE"][0], got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]))


def sub_workload(name, extra_args, env_extra, keys, timeout=900):
    """One of the other workloads of this file (config 4, config 5) as a child process on the same GPU, its JSON line condensed into an
    `extra` entry — so that the driver's run times them too, at sizes that keep the default command within minutes."""
    import subprocess
    env = dict(os.environ)
    env.update(env_extra)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
    if p.returncode != 0 or line is None:
        return {"name": name, "error": "child exited %d: %s" % (p.returncode, p.stderr[-200:])}
    d = json.loads(line)
    out = {"name": name, "command": "python bench.py " + " ".join(extra_args) + ("  (" + " ".join("%s=%s" % kv for kv in env_extra.items()) + ")" if env_extra else ""),
           "child_wall_s": round(time.perf_counter() - t0, 1)}
    for k in keys:
        if k in d:
            out[k] = d[k]
    return out


def halo_probe(spec):
    """`bench.py --halo-probe UIDHEX:RANK:WORLD:DEVICE` — one rank of a throw-away communicator: the engine's own RCCL exchange (csrc/halo.hip, RCCL
    behind dlopen) has met more than one rank on no machine yet, and a crash or a hang inside it must not cost the run its JSON line.  So the
    ranks of `bench.py --gpus N` first let CHILD processes do one small halo fetch through the engine path (each child = this function, under a
    timeout); only if every child comes back clean does the parent open its own engine communicator, otherwise the torch.distributed mirror
    serves the halo.  Exit code 0 = the fetched window is the rows the plan promises."""
    uid_hex, rank, world, device = spec.split(":")
    rank, world, device = int(rank), int(world), int(device)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import visionworkbench_amd as vwa
    from visionworkbench_amd import partition
    torch.cuda.set_device(device)
    ctx = vwa.Context(device)
    comm = partition.EngineComm(ctx, bytes.fromhex(uid_hex), rank, world)
    rows, above, below = 16 * world, 3, 5
    full = torch.arange(rows * 24, dtype=torch.float32, device="cuda").reshape(rows, 24)
    a, b = partition.row_strip(rank, world, rows)
    win, first = comm.fetch_strip_window(full[a:b].contiguous(), rows, above, below)
    torch.cuda.synchronize()
    _, _, na, nb = partition.halo_plan(rank, world, rows, above, below)
    ok = first == na and torch.equal(win, full[na:nb])
    comm.close()
    ctx.close()
    sys.exit(0 if ok else 4)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--halo-probe":
        return halo_probe(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)     # 0.36 ms each: long enough to amortise the ~1.3 ms of barrier + first-launch latency
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=250.0, dest="settle_ms")   # untimed load before the warm-up steps (clock ramp)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measured points (config 3, SGM block, +-16 px, config 4 / 5 runs)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--cost", default="census", choices=["census", "mad"],
                    help="config4 only: census (what the reference's SGM accepts) or mad = the 7x7 mean-abs-difference block cost of the config as "
                         "written (SGM.cc:1651-1738; unreachable upstream, explicit opt-in here)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config4", "config5"],
                    help="config2 (default, the headline metric): 4096^2 7x7 SAD; config4: 16384^2 census SGM in 8 strips + collar")
    args = ap.parse_args()

    maybe_self_launch(args.gpus)              # `python bench.py --gpus N` as typed: re-executes under torch.distributed.run (never returns)
    # the host driver supports dmabuf IPC only: without this RCCL's peer mappings fail (hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: rank %d: --gpus %d but WORLD_SIZE is %d (launch with --nproc-per-node %d, or type `python bench.py --gpus %d` "
                 "and let bench.py launch its ranks)" % (rank, args.gpus, world, args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    import visionworkbench_amd as vwa
    from visionworkbench_amd import core, stereo, synth

    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        # (the product has no CPU path; tests/test_bench_contract.py drives the launcher up to this line on a GPU-less host)
        print("bench.py: rank %d of %d (local rank %d): no GPU visible to this rank (%d device(s)) - nothing to measure"
              % (rank, world, local, torch.cuda.device_count() if torch.cuda.is_available() else 0), file=sys.stderr)
        sys.exit(3)
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
    rank_info = None
    if world > 1:
        # The source pair is row-sharded across the GPUs (rank g holds rows row_strip(g, N, rows) of each image, disjoint per HBM);
        # the ky-1 (+sy-1) halo rows a strip's windows read come from the neighbouring rank over RCCL point-to-point (one xGMI
        # link per neighbour pair) through the engine's C ABI — set-up, not part of the timed region (inputs are resident when
        # the clock starts).  Whether the fetched rows are used is decided by ALL ranks together (all-reduced flags): engine
        # exchange -> torch.distributed exchange -> host-provided halo rows, and the JSON line says which.
        fetcher = HaloFetcher(torch, dist, vwa, partition, rank, world, dev)
        hal = ky - 1 + sy - 1
        ok, why = True, ""
        try:
            a, b = partition.row_strip(rank, world, left.shape[0])
            lwin, lfirst = fetcher.fetch(torch.from_numpy(left[a:b].copy()).to(dev), left.shape[0], hal, hal)
            a, b = partition.row_strip(rank, world, right.shape[0])
            rwin, rfirst = fetcher.fetch(torch.from_numpy(right[a:b].copy()).to(dev), right.shape[0], hal, hal)
            torch.cuda.synchronize(dev)
            l_strip = lwin[la - lfirst:lb - lfirst].contiguous()
            r_strip = rwin[ra - rfirst:rb - rfirst].contiguous()
            if not (l_strip.shape[0] == lb - la and r_strip.shape[0] == rb - ra and
                    torch.equal(l_strip.cpu(), torch.from_numpy(left[la:lb])) and torch.equal(r_strip.cpu(), torch.from_numpy(right[ra:rb]))):
                ok, why = False, "the exchange returned wrong rows"
        except Exception as e:  # noqa: BLE001  (collective calls inside fetch() are matched on every rank: see HaloFetcher)
            ok, why = False, str(e)[:80]
        if fetcher.agree(ok):
            halo = "%s: %d (+%d) rows between neighbouring strips" % (fetcher.how, ky - 1, sy - 1)
        else:
            halo = "host-provided halo rows (halo exchange unusable on some rank%s)" % ((": " + why) if why else "")
            l_strip = r_strip = None
        # a SCALE record must check itself: every rank says which device it drove, how many ranks its RCCL communicator
        # held, which halo path it took and whether the rows it received equal the rows of the full image — gathered into config.ranks
        mine = {"rank": rank, "device": torch.cuda.get_device_name(dev), "pci_bus": torch.cuda.get_device_properties(dev).pci_bus_id
                if hasattr(torch.cuda.get_device_properties(dev), "pci_bus_id") else None,
                "rccl_ranks_seen": (fetcher.comm.world if (fetcher.comm is not None and hasattr(fetcher.comm, "world")) else (world if fetcher.comm is not None else 0)),
                "torch_distributed_world": dist.get_world_size(), "halo_path": fetcher.how, "halo_rows_verified": bool(ok),
                "output_rows": [int(r0), int(r1)]}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)
        fetcher.close()
    if l_strip is None:
        l_strip = torch.from_numpy(left[la:lb]).to(dev)
        r_strip = torch.from_numpy(right[ra:rb]).to(dev)
    region = vwa.BBox2i(0, 0, W, r1 - r0 + ky - 1)
    ctx = vwa.Context(local)
    # K steps are queued back to back: the engine must not wait for the input-class flags of each call (a host round trip
    # per step).  With VWGPU_OPT_DEFER_EXACTNESS a step is ONE launch — the packed matcher, validity sweep included — and
    # vwgpu_last_path() says afterwards whether the kernel accepted the data (PATH_REFUSED: no result) — asserted below,
    # together with the result itself.
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
    assert path != core.PATH_REFUSED, "the packed kernel refused the synthetic pair (not byte imagery?)"

    # Kernel durations come from HIP events the engine records on its own stream, live inside the timed region — on every
    # 4th step only: an event pair around a launch costs ~10 us of dispatch serialisation, which would otherwise be
    # charged to `value`.
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
        if world == 1 and not args.no_traffic:
            traffic, traffic_src = measure_traffic()
        if traffic is None:
            why = traffic_src
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, NOT measured in this run%s" % (
                        tj.get("round", "r01"), (" (" + why + ")") if why else "")
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
                       "partition": "%d row strip(s), no collective in the timed region" % world, "halo": halo, "ranks": rank_info,
                       "clock_settle_ms": args.settle_ms,
                       "path": {core.PATH_SAD_U8: "packed-u8 qsad", core.PATH_GENERIC_F64: "generic f64"}.get(path, "?")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "+".join(hot), "algorithmic_bytes_per_launch": strip_bytes,
                         "avg_us_per_launch": {k: kavg_us.get(k) for k in kavg_us},
                         "issue_bound_frac": (evals / (t_hot_us * 1e-6)) / (LANES * FCLK_HZ) if t_hot_us > 0 else None,
                         "note": "rank-0 strip; issue_bound_frac = (pixel x disparity evaluations per second) / "
                                 "(256 CU x 64 lane-ops/clk x 2.4 GHz): evaluations per VOP3 issue slot — the kernel "
                                 "is VALU-issue bound (~4.25 slots per evaluation at best), see DESIGN.md 4.1.  north_star's 0.70 of the HBM "
                                 "roofline is out of reach for +-64-px SAD on this ISA: the cost volume has W H D unique abs-diffs, but a 7-wide "
                                 "window is 4 + 3 bytes — SAD4-class instructions (4 abs-diffs per issue slot, summed) cannot share row partials "
                                 "between neighbouring pixels of an odd window — which puts the floor of the formulation near 234 us per 4096^2 launch "
                                 "(frac 0.18); the +-16-px point in `extra` shows the same kernel where the arithmetic shrinks 4x.  One launch per "
                                 "step since round 4 (validity sweep folded into the matcher); avg_us_per_launch comes from HIP events around the "
                                 "kernel on every 4th step, whose two event records cost ~1.5 us that the other steps do not pay"},
        }
        if not args.no_cpu_baseline:
            # rank 0, for every N: the same leg (the other ranks wait at the final barrier)
            res["cpu_baseline"] = cpu_baseline(left, right)
            # the checker: one full-image pass of the oracle against the result of the timed work — every pixel of this rank's strip
            import oracle
            want, _ = oracle.calc_disparity_tiled(0, left, right, KERNEL, SEARCH, tile=256, threads=os.cpu_count() or 1)
            same = bool(np.array_equal(out.cpu().numpy(), want[r0:r1]))
            res["cpu_baseline"]["result_identical_to_oracle"] = same
            res["cpu_baseline"]["rows_checked"] = [int(r0), int(r1)]
            assert same, "the timed result differs from the CPU oracle"
        else:
            res["cpu_baseline"] = None
        if world == 1 and not args.no_extra:
            try:                                  # the headline line must not depend on the side measurements
                res["extra"] = extra_points(ctx, torch, stereo, core, vwa, synth, l_strip, r_strip, left, right, with_cpu=not args.no_cpu_baseline)
                # BASELINE configs[3] and configs[4] themselves on this GPU: config 4 at full size (one pair = 0.45 s), config 5 on an
                # 32768^2 pair — the same code as `bench.py --workload config4 / config5`
                ckeys = ["metric", "value", "unit", "ms_per_step", "roofline", "config"]
                res["extra"].append(sub_workload("BASELINE configs[3] on one GPU: 16384^2 pair, census 7x7 SGM, 8 strips + collar (full size)",
                                                 ["--workload", "config4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], {}, ckeys))
                # config 5 at its OWN size since round 6 (1024 tiles through the batch entry: 0.56 s of block matching + 7.7 s of SGM per pair, 36 s
                # with the host-side generation of the pair and the oracle on 16 + 1 sampled tiles); VWGPU_BENCH_CONFIG5_SIZE=8192 is the reduced pair
                c5 = {"VWGPU_BENCH_CONFIG5_SIZE": os.environ["VWGPU_BENCH_CONFIG5_SIZE"]} if os.environ.get("VWGPU_BENCH_CONFIG5_SIZE") else {}
                res["extra"].append(sub_workload("BASELINE configs[4] on one GPU at full size: 32768^2 pair (1024 tiles of 1024^2), pyramid_correlate tile loop, "
                                                 "LoG 1.4 + NCC 11x11 and census SGM; 16 + 1 tiles compared with the oracle",
                                                 ["--workload", "config5", "--no-cpu-baseline"], c5, ckeys, timeout=1500))
            except Exception as e:  # noqa: BLE001
                res["extra"] = [{"name": "extra points failed", "error": "%s: %s" % (type(e).__name__, str(e)[:300])}]
            # Every point once more in a dozen short entries: inside `config` (the part of the line the driver's record keeps whole) and as the
            # LAST key of the line (the part its 2000-character tail keeps).  [Mpix/s, fraction of the HBM roofline, checked against the oracle]
            pts = {}
            short = [("search 33x1", "sad_pm16"), ("config 3a: ", "c3a_ncc11"), ("config 3a on 12-bit", "c3a_ncc11_12bit"), ("config 3b", "c3b_parabola"),
                     ("config 2 on a FLOAT", "c2_float"), ("config 3a on a FLOAT", "c3a_float"), ("END TO END", "c2_end_to_end"), ("DEFAULT options", "c2_default_opts"), ("TWO steps in flight", "c2_two_in_flight"),
                     ("SGM 2048^2", "sgm_2048_block"), ("MGM (use_mgm)", "mgm_1024"), ("tile loop, 4096^2 in 16 tiles of 1024^2, SAD", "loop_sad"),
                     ("tile loop, 4096^2 in 16 tiles of 1024^2, LoG", "loop_log_ncc"), ("configs[3] on one GPU", "config4_full"), ("configs[4] on one GPU", "config5_full")]
            for e in res["extra"]:
                key = next((k for pat, k in short if pat in e.get("name", "")), None)
                if key is None or "error" in e: continue
                v = e.get("Mpix_per_s", e.get("value"))
                fr = e.get("roofline_frac", (e.get("roofline") or {}).get("frac"))
                chk = e.get("tiles_identical_to_oracle", e.get("identical_to_oracle"))
                if key == "config5_full":
                    tv = (e.get("config") or {}).get("tiles_vs_oracle") or {}
                    chk = "%s / %s bm, %s / %s sgm" % (tv.get("bm_identical"), tv.get("bm_checked"), tv.get("sgm_identical"), tv.get("sgm_checked"))
                    pts["config5_full_sgm"] = [round(((e.get("config") or {}).get("sgm") or {}).get("Mpix_per_s", 0.0), 1), None, None]
                pts[key] = [None if v is None else round(float(v), 1), None if fr is None else round(float(fr), 4), chk]
                if key == "c2_end_to_end": pts[key] = [(e.get("pinned") or {}).get("Mpix_per_s"), None, None]
            res["config"]["points"] = pts
            res["points"] = pts
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
