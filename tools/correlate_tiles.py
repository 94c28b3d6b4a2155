"""Tile loop of tools/correlate (src/vw/tools/correlate.cc:207-266): pyramid_correlate over a big pair in 1024^2 tiles, T host
threads each with its own engine context (the reference runs one tile per thread).  Device-resident inputs.  GPU box only."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core
from visionworkbench_amd.core import BBox2i
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TILE = 1024
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
tiles = [BBox2i(x, y, min(TILE, W - x), min(TILE, W - y)) for y in range(0, W, TILE) for x in range(0, W, TILE)]
search = BBox2i.from_corners((-64, -1), (64, 1))
for name, kw in (("BM SAD 7x7", dict(kernel=(7, 7), cost=0, algorithm=0)), ("BM NCC 11x11", dict(kernel=(11, 11), cost=2, algorithm=0)),
                 ("SGM census 7x7", dict(kernel=(7, 7), cost=3, algorithm=1))):
    for T in (1, 4, 8):
        ctxs = [core.Context(0) for _ in range(T)]
        streams = [torch.cuda.Stream() for _ in range(T)]
        out = [None] * len(tiles)
        def work(t):
            with torch.cuda.stream(streams[t]):
                for i in range(t, len(tiles), T):
                    out[i] = stereo.pyramid_correlate(Lg, Rg, None, None, 0, 0.0, search, kw["kernel"], kw["cost"], consistency_threshold=2,
                                                      filter_half_kernel=5, max_pyramid_levels=5, algorithm=kw["algorithm"], bbox=tiles[i], ctx=ctxs[t])
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            [x.start() for x in th]; [x.join() for x in th]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        valid = float(np.mean([float((o[..., 2] != 0).float().mean()) for o in out]))
        print("%-15s %d tiles, %d thread(s): %.1f ms  = %.1f Mpix/s (valid %.3f)" % (name, len(tiles), T, dt * 1e3, W * W / dt / 1e6, valid))
        for c in ctxs: c.close()
