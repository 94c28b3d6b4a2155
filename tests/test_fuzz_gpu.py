"""Seeded, bounded randomised cross-checks under pytest -m gpu (the long-running versions are tools/fuzz_*.py):
  * packed SAD / SSD / NCC matchers vs the generic float64 kernel (itself pinned to the oracle in test_bm_gpu.py),
  * pyramid_correlate vs the oracle (tiles, masks, thresholds, filters, level counts; integer scenes, float scenes and
    LoG / mean-subtracted prefilters — the inputs on which the reference's running box sums are order dependent),
  * calc_disparity_sgm vs the oracle (masks, previous-level bounds, memory levels, every sub-pixel mode).
Every case must be IDENTICAL; a failure prints (generator, seed, index) for tools/replay_pyramid_case.py."""
import os

import numpy as np
import pytest

import fuzz_cases
import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo
from visionworkbench_amd.core import BBox2i

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    c = vwa.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _default_options(ctx):
    """Variant-pinning options (same results, other kernels / schedules) never leak from one test into the next."""
    yield
    ctx.set_option(core.OPT_SAD_GROUPS, 0)
    ctx.set_option(core.OPT_EXACT_SCRATCH_MB, 4096)


@pytest.mark.parametrize("cost,n,seed,scale", [(0, 120, 101, 256), (1, 60, 102, 256), (2, 60, 103, 256),
                                               (0, 40, 104, 65536), (1, 60, 105, 4096), (2, 60, 106, 4096)])
def test_fuzz_fast_paths_equal_generic(ctx, monkeypatch, cost, n, seed, scale):
    """Packed kernels (u8; 16-bit SAD; 12-bit SSD / NCC) against the float64 kernel on seeded random cases."""
    import torch
    bad = []
    for c in fuzz_cases.bm_cases(n, seed, cost, scale):
        ctx.set_option(core.OPT_SAD_GROUPS, 1 + int(c["split"]))
        lt, rt = torch.from_numpy(c["left"]).cuda(), torch.from_numpy(c["right"]).cuda()
        ctx.force_path(core.PATH_NONE)
        a = stereo.calc_disparity(cost, lt, rt, vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx).cpu().numpy()
        ctx.force_path(core.PATH_GENERIC_F64)
        b = stereo.calc_disparity(cost, lt, rt, vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx).cpu().numpy()
        ctx.force_path(core.PATH_NONE)
        if not np.array_equal(a, b):
            bad.append((c["it"], c["kernel"], c["search"], c["left"].shape, int((a != b).any(-1).sum())))
    assert not bad, "bm_cases(seed=%d, cost=%d): %s" % (seed, cost, bad)


def _pyramid_pair(oracle, c):
    s = c["search"]
    g = stereo.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], BBox2i.from_corners(s[:2], s[2:]), c["kernel"],
                                 c["cost"], 0, 0.0, c["thr"], 0, c["filt"], c["levels"], bbox=None if c["bbox"] is None else BBox2i(*c["bbox"]))
    o = oracle.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], s, c["kernel"], c["cost"], 0, 0.0, c["thr"],
                                 c["filt"], c["levels"], bbox=c["bbox"])
    return g, o


@pytest.mark.parametrize("n,seed,float_p,prefilters,costs,certify", [
    (70, 201, 0.0, (0,), (0, 0, 1), 1),            # the round-1 generator: integer scenes, SAD / SSD
    (40, 202, 0.0, (0,), (2,), 1),                 # NCC
    (50, 203, 1.0, (0,), (0, 1, 2), 1),            # float textures: order-dependent running sums (certified pixels + redone zones)
    (50, 204, 0.0, (1, 2), (0, 1, 2), 1),          # mean-subtracted / LoG prefilters (tools/correlate.cc default)
    (25, 203, 1.0, (0,), (0, 1, 2), 0),            # the same classes with VWGPU_OPT_CERTIFY = 0: every zone through the exact-order kernels
    (25, 204, 0.0, (1, 2), (0, 1, 2), 0),
])
def test_fuzz_pyramid_identical_to_oracle(oracle, n, seed, float_p, prefilters, costs, certify):
    from visionworkbench_amd import core
    ctx = core.default_context(0)
    ctx.set_option(core.OPT_CERTIFY, certify)
    bad = []
    try:
        for c in fuzz_cases.pyramid_cases(n, seed, prefilters=prefilters, costs=costs, float_scene=float_p):
            g, o = _pyramid_pair(oracle, c)
            if not np.array_equal(g, o):
                bad.append((c["it"], int((g != o).any(-1).sum())))
    finally:
        ctx.set_option(core.OPT_CERTIFY, 1)
    assert not bad, "pyramid_cases(seed=%d, float=%g, prefilters=%s, costs=%s, certify=%d): (index, differing pixels) %s" % (seed, float_p, prefilters, costs, certify, bad)


def test_round1_fuzz_finding_ssd_ties_in_mean_filled_border(oracle):
    """Case 153 of seed 10 (round 1: 10 differing pixels — exact SSD ties inside the mean-filled area next to a masked
    border, broken by the rounding order of the reference's running box sums, Algorithms.h:81-110)."""
    c = None
    for c in fuzz_cases.pyramid_cases(154, 10):
        pass
    for thr in (c["thr"], -1.0):
        c2 = dict(c, thr=thr)
        g, o = _pyramid_pair(oracle, c2)
        assert np.array_equal(g, o), int((g != o).any(-1).sum())


@pytest.mark.parametrize("n,seed,mgm", [(120, 301, False), (120, 302, True)])
def test_fuzz_sgm_identical_to_oracle(oracle, n, seed, mgm):
    bad = []
    for c in fuzz_cases.sgm_cases(n, seed):
        h, w = c["left"].shape
        gi, gs = stereo.calc_disparity_sgm(c["cost"], c["left"], c["right"], BBox2i(0, 0, w, h), c["search"], (c["k"], c["k"]),
                                           subpixel_mode=c["sub"], search_buffer=(2, 2), memory_limit_mb=c["mem"], left_mask=c["lm"],
                                           right_mask=c["rm"], prev_disparity=c["prev"], with_subpixel=True, use_mgm=mgm)
        oi, os_ = oracle.calc_disparity_sgm(c["cost"], c["left"], c["right"], c["search"], c["k"], subpixel=c["sub"], search_buffer=(2, 2),
                                            memory_limit_mb=c["mem"], left_mask=c["lm"], right_mask=c["rm"], prev_disparity=c["prev"], use_mgm=mgm)
        if not (np.array_equal(gi, oi) and (gs is None or np.abs(gs - os_).max() < 1e-5)):
            bad.append((c["it"], int((gi != oi).any(-1).sum())))
    assert not bad, "sgm_cases(seed=%d, mgm=%s): %s" % (seed, mgm, bad)


@pytest.mark.parametrize("n,seed,algorithm", [(24, 301, 1), (24, 302, 2), (24, 303, 3)])
def test_fuzz_pyramid_sgm_identical_to_oracle(oracle, n, seed, algorithm):
    """The SGM branch of the pyramid: ragged boxes from the previous level, R->L runs, consistency levels, masks, sub-tiles;
    algorithm 2 / 3 = VW_CORRELATION_MGM / _FINAL_MGM."""
    bad = []
    for c in fuzz_cases.pyramid_sgm_cases(n, seed):
        s = c["search"]
        g = stereo.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], 0, 0.0, BBox2i.from_corners(s[:2], s[2:]), (c["k"], c["k"]), c["cost"],
                                     consistency_threshold=c["thr"], min_consistency_level=c["mcl"], filter_half_kernel=c["filt"],
                                     max_pyramid_levels=c["levels"], algorithm=algorithm, bbox=None if c["bbox"] is None else BBox2i(*c["bbox"]))
        o = oracle.pyramid_correlate_sgm(c["left"], c["right"], c["lm"], c["rm"], s, c["k"], c["cost"], c["thr"], c["mcl"], c["filt"], c["levels"],
                                         bbox=c["bbox"], algorithm=algorithm)
        if not (np.array_equal(g[..., 2], o[..., 2]) and np.abs(g[..., :2] - o[..., :2]).max() < 1e-5):
            bad.append(c["it"])
    assert not bad, "pyramid_sgm_cases(seed=%d, algorithm=%d) mismatching indices %s" % (seed, algorithm, bad)


@pytest.mark.parametrize("n,seed", [(90, 501), (90, 502)])
def test_fuzz_single_level_float_identical_to_oracle(ctx, oracle, n, seed):
    """calc_disparity on float rasters through the default dispatch (round 5): the zone matcher with the whole raster as one zone — order-free
    data without, rounding data with the per-pixel certificate (fp32 tier first) and the exact-order fallback — against the oracle's
    whole-raster running sums."""
    bad, paths = [], {}
    for c in fuzz_cases.bm_float_cases(n, seed):
        got = stereo.calc_disparity(c["cost"], c["left"], c["right"], vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx)
        p = ctx.last_path()
        paths[p] = paths.get(p, 0) + 1
        want = oracle.calc_disparity(c["cost"], c["left"], c["right"], c["kernel"], c["search"])
        if not np.array_equal(got, want):
            bad.append((c["it"], c["cost"], c["kernel"], c["search"], c["left"].shape, c["kind"], p, int((got != want).any(-1).sum())))
    assert not bad, "bm_float_cases(seed=%d): %s" % (seed, bad)
    assert paths.get(core.PATH_CERTIFIED, 0) >= n // 5 and paths.get(core.PATH_EXACT_ORDER, 0) >= 3, paths      # both outcomes of the certificate occur


@pytest.mark.parametrize("n,seed", [(70, 521)])
def test_fuzz_pyramid_corner_scenes_identical_to_oracle(ctx, oracle, n, seed):
    """pyramid_correlate on nodata / saturation blocks, ramps, decades of range, negative values, four-level images
    (fuzz_cases.pyramid_corner_cases): all costs and prefilters."""
    bad = []
    for c in fuzz_cases.pyramid_corner_cases(n, seed):
        s = c["search"]
        g = stereo.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], BBox2i.from_corners(s[:2], s[2:]), c["kernel"], c["cost"],
                                     0, 0.0, c["thr"], 0, c["filt"], c["levels"], bbox=None if c["bbox"] is None else BBox2i(*c["bbox"]),
                                     blob_filter_area=c["blob"], ctx=ctx)
        oracle.set_blob_filter_area(c["blob"])
        try:
            o = oracle.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], s, c["kernel"], c["cost"], 0, 0.0, c["thr"], c["filt"],
                                         c["levels"], bbox=c["bbox"])
        finally:
            oracle.set_blob_filter_area(0)
        if not np.array_equal(g, o):
            bad.append((c["it"], c["kind"], c["cost"], c["pf"], int((g != o).any(-1).sum())))
    assert not bad, "pyramid_corner_cases(seed=%d): %s" % (seed, bad)


@pytest.mark.parametrize("n,seed", [(250, 511)])
def test_fuzz_single_level_float_corner_cases_identical_to_oracle(ctx, oracle, n, seed):
    """The thinly covered corners (fuzz_cases.bm_float_corner_cases): one to a few disparities, windows with a side of 1, box sums that cancel,
    squares that underflow, blocks of zeros (NaN costs), constant images — through the default dispatch, fp32 tier on and off."""
    bad = []
    for f32 in (1, 0):
        ctx.set_option(core.OPT_CERT_F32, f32)
        try:
            for c in fuzz_cases.bm_float_corner_cases(n, seed + f32):
                got = stereo.calc_disparity(c["cost"], c["left"], c["right"], vwa.bounding_box(c["left"]), c["search"], c["kernel"], ctx=ctx)
                want = oracle.calc_disparity(c["cost"], c["left"], c["right"], c["kernel"], c["search"])
                if not np.array_equal(got, want):
                    bad.append((f32, c["it"], c["cost"], c["kernel"], c["search"], c["left"].shape, c["kind"], ctx.last_path(), int((got != want).any(-1).sum())))
        finally:
            ctx.set_option(core.OPT_CERT_F32, 1)
    assert not bad, "bm_float_corner_cases(seed=%d): %s" % (seed, bad)


@pytest.mark.parametrize("n,seed", [(30, 601)])
def test_fuzz_batch_identical_to_oracle(ctx, oracle, n, seed):
    """pyramid_correlate_batch on random scenes cut into random tile grids: every tile against the oracle's tile."""
    bad = []
    for c in fuzz_cases.batch_cases(n, seed):
        s = c["search"]
        got = stereo.pyramid_correlate_batch(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], BBox2i.from_corners(s[:2], s[2:]), c["kernel"], c["cost"],
                                             [BBox2i(*b) for b in c["boxes"]], consistency_threshold=c["thr"], filter_half_kernel=c["filt"],
                                             max_pyramid_levels=c["levels"], ctx=ctx)
        for b, g in zip(c["boxes"], got):
            o = oracle.pyramid_correlate(c["left"], c["right"], c["lm"], c["rm"], c["pf"], c["pfw"], s, c["kernel"], c["cost"], 0, 0.0, c["thr"], c["filt"], c["levels"], bbox=b)
            if not np.array_equal(g, o):
                bad.append((c["it"], b, int((g != o).any(-1).sum())))
    assert not bad, "batch_cases(seed=%d): (index, tile, differing pixels) %s" % (seed, bad)
