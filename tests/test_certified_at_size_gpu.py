"""The certified fast path pinned at the size it is benchmarked on ("What's weak" 1).

bench.py's tile-loop points run pyramid_correlate over 1024^2 tiles of the 4096^2 synthetic pair with the `correlate` tool's defaults
(LoG 1.4 + NCC 11x11, search (-64,-1)...(64,1), 5 levels, consistency threshold 2, filter half kernel 5: tools/correlate.cc:85,207-223).
There the per-pixel certificate of csrc/bm_zones.hip carries W + H ~ 1000 in its error bound, zones are up to 512 x 512 with 100+
disparities, and border tiles search far beyond the other image (the "cannot matter" certificate).  Every case here is one such tile
against oracle.pyramid_correlate (src/vw/Stereo/CorrelationView.cc:596-760 restated), with VWGPU_OPT_CERTIFY on (default) and off (every
rounding level in the reference's own summation order), and asserts through VWGPU_OPT_CERT_PERMILLE that the certificate really engaged.
The oracle tiles run on the host cores of the GPU box, one tile per thread (the oracle is C behind ctypes: no GIL)."""
import concurrent.futures
import os

import numpy as np
import pytest

import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo, synth
from visionworkbench_amd.core import BBox2i

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8
SEARCH = (-64, -1, 64, 1)                 # BBox2i(Vector2i(-64,-1), Vector2i(64,1)): half open, 128 x 2 disparities
TILES = {"interior": (1024, 1024, 1024, 1024), "left border": (0, 2048, 1024, 1024), "corner": (3072, 3072, 1024, 1024),
         "right border": (3072, 1024, 1024, 1024)}


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    c = vwa.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def pair():
    left, right, _ = synth.stereo_pair(4096, 4096, 129, 1)
    return left, np.ascontiguousarray(right[:, 64:64 + 4096])         # same size as left; true disparity 0 +- 48 inside the +-64 search


@pytest.fixture(scope="module")
def float_pair(pair):
    """The float-texture twin of the pair (as tests/fuzz_cases.py:47 makes its float scenes): pixel * 0.37 + uniform noise, independent
    noise per image — non-integer imagery whose box sums round position-dependently."""
    left, right = pair
    rng = np.random.default_rng(20260926)
    lf = (left * np.float32(0.37) + rng.random(left.shape, dtype=np.float32)).astype(np.float32)
    rf = (right * np.float32(0.37) + rng.random(right.shape, dtype=np.float32)).astype(np.float32)
    return lf, rf


def _gpu_tile(ctx, lt, rt, pf, pfw, kernel, cost, bbox, certify):
    ctx.set_option(core.OPT_CERTIFY, certify)
    ctx.set_option(core.OPT_TRACE, 4)                 # certification statistics (resets the running total)
    try:
        got = stereo.pyramid_correlate(lt, rt, None, None, pf, pfw, BBox2i.from_corners(SEARCH[:2], SEARCH[2:]), kernel, cost, 0, 0.0, 2.0, 0, 5, 5,
                                       bbox=BBox2i(*bbox), ctx=ctx)
        import torch
        torch.cuda.synchronize()
        share = ctx.get_option(core.OPT_CERT_PERMILLE)
    finally:
        ctx.set_option(core.OPT_TRACE, 0)
        ctx.set_option(core.OPT_CERTIFY, 1)
    return got.cpu().numpy(), share


def _run(ctx, oracle, left, right, pf, pfw, kernel, cost, names, min_share):
    import torch
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(NCPU, len(names))) as pool:
        futs = {n: pool.submit(oracle.pyramid_correlate, left, right, None, None, pf, pfw, SEARCH, kernel, cost, 0, 0.0, 2.0, 5, 5, bbox=TILES[n])
                for n in names}
        got = {n: {c: _gpu_tile(ctx, lt, rt, pf, pfw, kernel, cost, TILES[n], c) for c in (1, 0)} for n in names}
        want = {n: f.result() for n, f in futs.items()}
    for n in names:
        w = want[n]
        assert w.shape == (1024, 1024, 3) and (w[..., 2] != 0).mean() > 0.5, (n, "the oracle tile is mostly invalid: not a test")
        g0, _ = got[n][0]
        g1, share = got[n][1]
        assert np.array_equal(g0, w), (n, "exact-order schedule", int((g0 != w).any(-1).sum()))
        assert np.array_equal(g1, w), (n, "certified schedule", int((g1 != w).any(-1).sum()))
        # the certificate engaged (share == -1: no level of the tile was certified at all) and proved most of the tile
        assert (min_share is None and share == -1) or share >= (900 if min_share is None else min_share), (n, "certified share (per mille)", share)


@pytest.mark.parametrize("name", list(TILES))
def test_log_ncc_tiles_of_the_bench_pair(ctx, oracle, pair, name):
    """BASELINE configs[4]'s real form and bench.py's "LoG 1.4 + NCC 11x11" tile-loop point: 1024^2 tiles of the 4096^2 pair."""
    _run(ctx, oracle, pair[0], pair[1], 2, float(np.float32(1.4)), (11, 11), 2, [name], 900)


@pytest.mark.parametrize("cost,kernel", [(0, (7, 7)), (1, (7, 7))])
def test_float_texture_tiles(ctx, oracle, float_pair, cost, kernel):
    """The same three kinds of tile on a float texture (non-integer left / right, no prefilter) with SAD and SSD.  (SAD: 31 significant bits
    per pixel + 6 for the 49 addends fit float64 — the level is order free, nothing to certify, the share reads -1; SSD squares them.)"""
    _run(ctx, oracle, float_pair[0], float_pair[1], 0, 0.0, kernel, cost, ["interior", "left border", "corner"], None if cost == 0 else 900)


def test_float_texture_ncc_meansub_tile(ctx, oracle, float_pair):
    """One SubtractedMean case (PREFILTER_MEANSUB, Gaussian sigma 3: src/vw/Stereo/PreFilter.h:52-74), NCC 11x11, corner tile."""
    _run(ctx, oracle, float_pair[0], float_pair[1], 1, float(np.float32(3.0)), (11, 11), 2, ["corner"], 900)
