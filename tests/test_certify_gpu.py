"""Certified tile-parallel matching of pyramid levels whose box sums round (csrc/bm_zones.hip, VWGPU_OPT_CERTIFY).

The reference's box sums are serial running sums (src/vw/Stereo/Algorithms.h:43-129); on prefiltered or float imagery their roundings depend
on the raster position.  The engine matches such levels with the tile-parallel float64 kernels, proves per pixel that the winner leads the
runner-up by more than twice a bound on the difference between the two summation orders, and redoes only the zones that hold an unproven
pixel in the reference's own order.  Everything here must be IDENTICAL to the oracle with the option on (default) and off (round-3 schedule),
and the option must actually certify most of a textured scene."""
import numpy as np
import pytest

import scenes
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


def _both(ctx, oracle, left, right, lm, rm, pf, pfw, search, kernel, cost, thr, filt, levels):
    box = BBox2i.from_corners(search[:2], search[2:])
    out = {}
    for certify in (1, 0):
        ctx.set_option(core.OPT_CERTIFY, certify)
        ctx.set_option(core.OPT_TRACE, 4)             # certification statistics (resets the running total)
        try:
            out[certify] = stereo.pyramid_correlate(left, right, lm, rm, pf, pfw, box, kernel, cost, 0, 0.0, thr, 0, filt, levels, ctx=ctx)
            permille = ctx.get_option(core.OPT_CERT_PERMILLE)
        finally:
            ctx.set_option(core.OPT_TRACE, 0)
            ctx.set_option(core.OPT_CERTIFY, 1)
        if certify:
            share = permille
    want = oracle.pyramid_correlate(left, right, lm, rm, pf, pfw, search, kernel, cost, 0, 0.0, thr, filt, levels)
    assert np.array_equal(out[0], want), ("exact-order schedule", int((out[0] != want).any(-1).sum()))
    assert np.array_equal(out[1], want), ("certified schedule", int((out[1] != want).any(-1).sum()))
    return share


@pytest.mark.parametrize("cost,kernel", [(2, (11, 11)), (1, (7, 7)), (0, (7, 7))])
def test_log_filtered_scene_is_mostly_certified(ctx, oracle, cost, kernel):
    """The `correlate` tool's defaults (LoG 1.4, tools/correlate.cc:85,210) on the reference's test scene: identical tiles, and most pixels
    proven by the tile-parallel pass (SAD on this scene is order free: nothing to certify, the share reads -1)."""
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    share = _both(ctx, oracle, left, right, None, None, 2, float(np.float32(1.4)), search, kernel, cost, 2, 5, 5)
    assert share == -1 or share >= 700, share


@pytest.mark.parametrize("cost", [0, 1, 2])
def test_float_textures_with_masks(ctx, oracle, cost):
    """Float textures, mean-filled nodata areas (exact cost ties there: the rounding order of the running sums decides — those zones must
    go to the exact-order kernels) and a consistency check."""
    rng = np.random.default_rng(808 + cost)
    H, W = 220, 300
    left = (rng.random((H, W)) * 200.0).astype(np.float32)
    right = np.roll(left, 6, axis=1)
    right[:, :6] = (rng.random((H, 6)) * 200.0).astype(np.float32)
    lm = np.full((H, W), 255, np.uint8); lm[50:120, 40:130] = 0
    rm = np.full((H, W), 255, np.uint8); rm[:, -30:] = 0
    share = _both(ctx, oracle, left, right, lm, rm, 0, 0.0, (-10, -2, 11, 3), (7, 7), cost, 2, 3, 3)
    assert share == -1 or share >= 300, share


def test_repeated_texture_near_ties_go_to_the_exact_kernels(ctx, oracle):
    """A right image that repeats the left texture with a period inside the search range: two disparities whose NCC costs are equal in
    exact arithmetic and differ only by the rounding history of the running sums.  No certificate exists for those pixels."""
    rng = np.random.default_rng(99)
    H, W, P = 160, 240, 7
    base = (rng.random((H, W + 64)) * 90.0 + 5.0).astype(np.float32)
    left = base[:, :W].copy()
    right = base[:, :W].copy()
    right[:, P:] = np.where(rng.random((H, W - P)) < 0.5, right[:, P:], left[:, :W - P])       # half the pixels: the texture shifted by P as well
    _both(ctx, oracle, left, right, None, None, 0, 0.0, (-9, -1, 10, 2), (5, 5), 2, -1, 0, 2)


@pytest.mark.parametrize("cost", [1, 2])
def test_wide_range_floats(ctx, oracle, cost):
    """Magnitudes over 9 decades: the error bound is as large as the largest pixel allows, few pixels certify — still the oracle's tile."""
    rng = np.random.default_rng(5 + cost)
    H, W = 160, 200
    v = (rng.random((H, W)) * 10.0 ** (rng.random((H, W)) * 9 - 4.5)).astype(np.float32)
    right = np.roll(v, 4, axis=1)
    _both(ctx, oracle, v, right, None, None, 0, 0.0, (-8, -2, 9, 3), (7, 7), cost, 2, 3, 3)


@pytest.mark.parametrize("case", range(12))
def test_search_across_the_image_borders(ctx, oracle, case):
    """A search range that reaches beyond both sides of the right image.  Out there the crop is mean-filled nodata: under a LoG or mean
    prefilter its windows are all-zero (NCC: 0 * inf = NaN costs — a NaN first candidate is the reference's winner whatever follows) or
    bit-identical copies (exact ties).  No certificate can order those candidates; the "cannot matter" certificate (bm_zones.hip, ZEdge)
    proves instead that whichever of them the reference picks, the pixel is erased by the mask pass (L->R) or fails the consistency check
    (R->L), and that no neighbour's clean-up count can see it.  Costs, prefilters, clean-up kernels (0 = no filter and NO mask pass: the
    certificate must stay off), consistency thresholds, levels and nodata masks vary; the tile must equal the oracle's in every case."""
    rng = np.random.default_rng(4100 + case)
    H, W = int(rng.integers(150, 230)), int(rng.integers(200, 320))
    shift = int(rng.integers(-9, 10))
    base = (rng.random((H, W + 64)) * 180.0 + 20.0).astype(np.float32)
    base = (base + np.roll(base, 1, axis=1) + np.roll(base, 1, axis=0)) / np.float32(3.0)           # some correlation between neighbours
    left = base[:, 32:32 + W].copy()
    right = base[:, 32 - shift:32 - shift + W].copy()
    cost = case % 3
    kernel = [(7, 7), (5, 5), (11, 11)][cost]
    pf, pfw = [(2, float(np.float32(1.4))), (1, float(np.float32(3.0))), (2, float(np.float32(1.4))), (0, 0.0)][case % 4]
    filt = [5, 3, 1, 0, 5, 2][case % 6]
    thr = [2, 0, -1, 7.5, 2, 1][(case // 2) % 6]
    levels = [3, 2, 0, 4][case % 4]
    reach = int(rng.integers(12, 30))
    search = (-reach, -2, reach + 1, 3)
    lm = rm = None
    if case % 5 == 3:
        lm = np.full((H, W), 255, np.uint8); lm[20:60, :25] = 0
        rm = np.full((H, W), 255, np.uint8); rm[:, -18:] = 0; rm[100:130, 5:40] = 0
    _both(ctx, oracle, left, right, lm, rm, pf, pfw, search, kernel, cost, thr, filt, levels)


# ---- single-level calc_disparity (round 5): the whole raster as the one zone of the certified pass (csrc/vwgpu_abi.hip) -----------------

def _float_scene(rng, h, w, sx, sy, decades=0.0):
    """Non-integer left / right rasters with a shifted copy + independent noise, magnitudes over `decades` powers of ten."""
    base = rng.random((h + sy - 1, w + sx - 1)) * 200.0
    if decades:
        base = base * 10.0 ** (rng.random(base.shape) * decades - decades / 2)
    right = base.astype(np.float32)
    d = (int(rng.integers(0, sx)), int(rng.integers(0, sy)))
    left = (right[d[1]:d[1] + h, d[0]:d[0] + w] + rng.random((h, w)).astype(np.float32) * np.float32(0.5)).astype(np.float32)
    return left, right


@pytest.mark.parametrize("cost,kernel,search", [(0, (7, 7), (33, 1)), (1, (7, 7), (40, 3)), (2, (11, 11), (65, 1)), (1, (5, 9), (9, 9)), (2, (3, 3), (20, 2))])
def test_single_level_float_raster_is_certified(ctx, oracle, cost, kernel, search):
    """calc_disparity on a float raster whose box sums round: served by the certified tile-parallel pass (VWGPU_PATH_CERTIFIED), identical
    to the oracle's whole-raster running sums; the same call with VWGPU_OPT_CERTIFY = 0 goes to the exact-order kernels: same image."""
    import torch
    rng = np.random.default_rng(7000 + cost * 10 + kernel[0])
    left, right = _float_scene(rng, 150, 260, search[0], search[1], decades=3.0)
    want = oracle.calc_disparity(cost, left, right, kernel, search)
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    path = ctx.last_path()
    assert np.array_equal(got, want), (path, int((got != want).any(-1).sum()))
    assert path in (core.PATH_CERTIFIED, core.PATH_GENERIC_F64), path          # (GENERIC_F64: the data happened to be order free)
    ctx.set_option(core.OPT_CERTIFY, 0)
    try:
        got0 = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
        assert ctx.last_path() in (core.PATH_EXACT_ORDER, core.PATH_GENERIC_F64)
    finally:
        ctx.set_option(core.OPT_CERTIFY, 1)
    assert np.array_equal(got0, want)
    # a region of a larger device image (row strides != widths), as vw::stereo::calc_disparity crops its inputs (Correlation.cc:353-359)
    big_l = torch.zeros((left.shape[0] + 9, left.shape[1] + 13), dtype=torch.float32, device="cuda")
    big_r = torch.zeros((right.shape[0] + 9, right.shape[1] + 13 + 5), dtype=torch.float32, device="cuda")
    big_l[4:4 + left.shape[0], 6:6 + left.shape[1]] = torch.from_numpy(left).cuda()
    big_r[4:4 + right.shape[0], 6:6 + right.shape[1]] = torch.from_numpy(right).cuda()
    got_d = stereo.calc_disparity(cost, big_l, big_r, BBox2i(6, 4, left.shape[1], left.shape[0]), search, kernel, ctx=ctx)
    torch.cuda.synchronize()
    assert np.array_equal(got_d.cpu().numpy(), want)


@pytest.mark.parametrize("cost", [0, 1, 2])
def test_single_level_exact_ties_fall_back_to_the_reference_order(ctx, oracle, cost):
    """A flat patch inside a float texture: inside it every disparity has the same cost in exact arithmetic and the running sums' rounding
    residue decides — no certificate exists, the call must be redone in the reference's order (VWGPU_PATH_EXACT_ORDER) and match."""
    rng = np.random.default_rng(7100 + cost)
    left, right = _float_scene(rng, 120, 200, 17, 1, decades=6.0)
    left[40:80, 60:140] = np.float32(3.3)
    right[40:80, 50:170] = np.float32(3.3)
    want = oracle.calc_disparity(cost, left, right, (7, 7), (17, 1))
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), (17, 1), (7, 7), ctx=ctx)
    assert ctx.last_path() == core.PATH_EXACT_ORDER
    assert np.array_equal(got, want), int((got != want).any(-1).sum())


def test_single_level_one_disparity_and_a_nan_cost_in_the_reference(ctx, oracle):
    """Case 3889 of `tools/fuzz_round5.py 10000 800 7373` (round 5, found after the certificate had passed 26 000 others): NCC, 1 x 7 window, ONE
    disparity, data of 12 decades.  One pixel's running box sum of squares cancels to <= 0 in the reference — its precision is infinite or
    negative, the one cost a NaN, `best == worst` false: the pixel stays VALID (Correlation.cc:121-133) — while the tile sums of the certified
    pass are fine and "one disparity => invalid" was returned without asking whether the reference's cost is a number."""
    import fuzz_cases
    case = next(c for c in fuzz_cases.bm_float_cases(3890, 7374) if c["it"] == 3889)
    assert case["cost"] == 2 and case["kernel"] == (1, 7) and case["search"] == (1, 1)
    want = oracle.calc_disparity(2, case["left"], case["right"], (1, 7), (1, 1))
    assert (want[..., 2] != 0).sum() == 1, "the reference keeps exactly one pixel valid here"
    for f32 in (1, 0):
        ctx.set_option(core.OPT_CERT_F32, f32)
        try:
            got = stereo.calc_disparity(2, case["left"], case["right"], vwa.bounding_box(case["left"]), (1, 1), (1, 7), ctx=ctx)
        finally:
            ctx.set_option(core.OPT_CERT_F32, 1)
        assert np.array_equal(got, want), (f32, int((got != want).any(-1).sum()))


def test_single_level_order_free_float_raster(ctx, oracle):
    """Order-free float data (here: quarter-integers) takes the tile-parallel kernels without a certificate — any order returns the bits."""
    rng = np.random.default_rng(7200)
    right = (np.floor(rng.random((100, 300)) * 1024) / 4).astype(np.float32)
    left = right[:, 11:11 + 240].copy()
    left[30:50, 100:130] += np.float32(0.25)
    for cost, kernel in ((0, (7, 7)), (1, (9, 5)), (2, (11, 11))):
        want = oracle.calc_disparity(cost, left, right, kernel, (61, 1))
        got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), (61, 1), kernel, ctx=ctx)
        assert ctx.last_path() == core.PATH_GENERIC_F64
        assert np.array_equal(got, want), (cost, int((got != want).any(-1).sum()))


@pytest.mark.parametrize("f32,tile16", [(0, 2), (1, 1), (0, 1)])
def test_tier_and_tile_options_return_the_same_image(ctx, oracle, f32, tile16):
    """VWGPU_OPT_CERT_F32 = 0 (the certified pass in float64 only) and VWGPU_OPT_ZONE_TILE16 = 1 (zones that fit 16 x 16 on the one-wavefront
    tile kernels) select other kernels for the same arithmetic: pyramid tiles (LoG + NCC, float SSD) and a single-level float raster."""
    ctx.set_option(core.OPT_CERT_F32, f32)
    ctx.set_option(core.OPT_ZONE_TILE16, tile16)
    try:
        left, right, scale, trans, search = scenes.pyramid_scene("u8")
        for cost, kernel, pf in ((2, (11, 11), 2), (1, (7, 7), 0), (0, (5, 5), 2)):
            l, r = left, right
            if pf == 0:
                l = (left * np.float32(0.37) + np.float32(0.11)).astype(np.float32); r = (right * np.float32(0.37) + np.float32(0.07)).astype(np.float32)
            pfw = float(np.float32(1.4)) if pf else 0.0
            got = stereo.pyramid_correlate(l, r, None, None, pf, pfw, BBox2i.from_corners(search[:2], search[2:]), kernel, cost, 0, 0.0, 2, 0, 5, 5, ctx=ctx)
            want = oracle.pyramid_correlate(l, r, None, None, pf, pfw, search, kernel, cost, 0, 0.0, 2, 5, 5)
            assert np.array_equal(got, want), (cost, int((got != want).any(-1).sum()))
        rng = np.random.default_rng(4242)
        fl, fr = _float_scene(rng, 130, 250, 41, 2, decades=2.0)
        for cost, kernel in ((1, (7, 7)), (2, (9, 9))):
            got = stereo.calc_disparity(cost, fl, fr, vwa.bounding_box(fl), (41, 2), kernel, ctx=ctx)
            assert np.array_equal(got, oracle.calc_disparity(cost, fl, fr, kernel, (41, 2)))
    finally:
        ctx.set_option(core.OPT_CERT_F32, 1)
        ctx.set_option(core.OPT_ZONE_TILE16, 2)


@pytest.mark.parametrize("sxc", [1, 5, 64, 4096])
def test_zone_sxc_option_returns_the_same_image(ctx, oracle, sxc):
    """VWGPU_OPT_ZONE_SXC (horizontal disparities per staged right patch): 1 and 5 cut every search row into several patches, 64 and
    4096 go beyond the default 16 (the LDS budget caps them) — the lanes of a tile without a pixel then read the precision image's first row
    at offsets up to the zone's search width (nd <= z.sx <= pb.w).  LoG + NCC pyramid tile with its 16 x 16 leaf zones (partly filled tiles on
    every level), a float SSD pyramid and a single-level float NCC raster with a 70-wide search."""
    ctx.set_option(core.OPT_ZONE_SXC, sxc)
    try:
        left, right, scale, trans, search = scenes.pyramid_scene("u8")
        for cost, kernel, pf in ((2, (11, 11), 2), (1, (7, 7), 0)):
            l, r = left, right
            if pf == 0:
                l = (left * np.float32(0.37) + np.float32(0.11)).astype(np.float32); r = (right * np.float32(0.37) + np.float32(0.07)).astype(np.float32)
            pfw = float(np.float32(1.4)) if pf else 0.0
            got = stereo.pyramid_correlate(l, r, None, None, pf, pfw, BBox2i.from_corners(search[:2], search[2:]), kernel, cost, 0, 0.0, 2, 0, 5, 5, ctx=ctx)
            want = oracle.pyramid_correlate(l, r, None, None, pf, pfw, search, kernel, cost, 0, 0.0, 2, 5, 5)
            assert np.array_equal(got, want), (cost, int((got != want).any(-1).sum()))
        rng = np.random.default_rng(4343)
        fl, fr = _float_scene(rng, 90, 150, 70, 2, decades=2.0)
        got = stereo.calc_disparity(2, fl, fr, vwa.bounding_box(fl), (70, 2), (9, 9), ctx=ctx)
        assert np.array_equal(got, oracle.calc_disparity(2, fl, fr, (9, 9), (70, 2)))
    finally:
        ctx.set_option(core.OPT_ZONE_SXC, 0)


@pytest.mark.parametrize("cost,kernel,search", [(0, (7, 7), (21, 1)), (1, (5, 5), (9, 3)), (2, (11, 11), (33, 2))])
def test_single_level_partial_redo_of_flagged_tile_rows(ctx, oracle, cost, kernel, search):
    """A few unprovable pixels in a large float raster: only the tile rows that hold them are redone in the reference's order — the column
    chains run from the raster's first row (bm_exact.hip, row ranges) — in the middle, at the top, at the bottom of the raster, in two
    separate bands; and a raster flagged nearly everywhere (the whole call in the reference's order).  Always the oracle's image."""
    rng = np.random.default_rng(7300 + cost)
    H, W = 420, 520
    left, right = _float_scene(rng, H, W, search[0], search[1], decades=4.0)
    def flat(y0, y1, x0, x1):
        left[y0:y1, x0:x1] = np.float32(2.7)
        right[y0:y1 + search[1] - 1, x0:x1 + search[0] + 30] = np.float32(2.7)
    # (round 6: the redo also stops at the last flagged tile COLUMN — patches at the left edge, in the middle, at the right edge, and two bands
    # whose patches end at different columns)
    for bands in ([(200, 230)], [(0, 25)], [(H - 30, H)], [(40, 70), (300, 340)], [(0, H)], [(120, 150, 0, 40)], [(120, 150, W - 90, W)],
                  [(60, 80, 10, 50), (260, 290, 300, 420)]):
        l0, r0 = left.copy(), right.copy()
        for bd in bands:
            a, b = bd[0], bd[1]
            x0, x1 = (bd[2], bd[3]) if len(bd) == 4 else (100, 260)
            flat(a, b, x0, x1)
        want = oracle.calc_disparity(cost, left, right, kernel, search)
        got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
        assert np.array_equal(got, want), (bands, ctx.last_path(), int((got != want).any(-1).sum()))
        # (SAD on this scene may be order free — every sum representable: then no certificate is needed at all)
        assert ctx.last_path() == core.PATH_EXACT_ORDER or (cost == 0 and ctx.last_path() == core.PATH_GENERIC_F64), (bands, ctx.last_path())
        left[:], right[:] = l0, r0
