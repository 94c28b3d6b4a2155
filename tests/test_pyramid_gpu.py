"""pyramid_correlate (block matching), the disparity clean-up filters and the zone scheduler: libvwgpu.so vs the oracle.

Parity bar: every stage is integer / index work on top of kernels that are bit-exact on their own, so whole-tile results
must be IDENTICAL to the oracle's (dx, dy, valid) — checked here on 8-bit, 16-bit and float scenes, all three costs,
with and without the L/R check, masks and prefilters, plus the reference's own pass thresholds.
"""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vw():
    from visionworkbench_amd import stereo
    return stereo


def _box(search):
    from visionworkbench_amd.core import BBox2i
    return BBox2i.from_corners(search[:2], search[2:])


def _run_both(vw, oracle, left, right, lm, rm, pf, pfw, search, kernel, cost, thr, filt, levels, bbox=None):
    from visionworkbench_amd.core import BBox2i
    g = vw.pyramid_correlate(left, right, lm, rm, pf, pfw, _box(search), kernel, cost, 0, 0.0, thr, 0, filt, levels,
                             bbox=None if bbox is None else BBox2i(*bbox))
    o = oracle.pyramid_correlate(left, right, lm, rm, pf, pfw, search, kernel, cost, 0, 0.0, thr, filt, levels, bbox=bbox)
    return g, o


@pytest.mark.parametrize("channel", ["u8", "i16", "f32"])
@pytest.mark.parametrize("cost", [0, 1, 2])
@pytest.mark.parametrize("thr", [-1, 2])
def test_reference_scene_identical_to_oracle(vw, oracle, channel, cost, thr):
    left, right, scale, trans, search = scenes.pyramid_scene(channel)
    g, o = _run_both(vw, oracle, left, right, None, None, 0, 0.0, search, (7, 7), cost, thr, 5, 5)
    assert g.shape == o.shape == left.shape + (3,)
    assert np.array_equal(g, o), int((g != o).any(axis=2).sum())      # (NCC included: levels whose sums round are certified or matched in the reference's order)
    c, a = scenes.pyramid_score(g, scale, trans)
    assert c > (.87 if cost == 2 else .90) and a > .99           # TestPyramidCorrelationView.cxx thresholds


@pytest.mark.parametrize("pf,pfw", [(1, 1.4), (2, 3.0)])
def test_prefiltered_identical(vw, oracle, pf, pfw):
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    g, o = _run_both(vw, oracle, left, right, None, None, pf, float(np.float32(pfw)), search, (7, 7), 0, 2, 5, 5)
    assert np.array_equal(g, o)
    if pf == 1:        # (mean subtraction with a wide window does poorly on white noise in the oracle too: identity only)
        c, a = scenes.pyramid_score(g, scale, trans)
        assert c > .85 and a > .95


def test_masks_and_subtile(vw, oracle):
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    lm = np.full(left.shape, 255, np.uint8)
    rm = np.full(right.shape, 255, np.uint8)
    lm[40:90, 100:160] = 0
    rm[120:, 200:] = 0
    g, o = _run_both(vw, oracle, left, right, lm, rm, 0, 0.0, search, (7, 7), 0, 2, 3, 5)
    assert np.array_equal(g, o)
    assert (g[45:85, 105:155, 2] == 0).all()                     # masked source pixels never come out valid
    # one interior tile, as the block rasteriser would request it
    g, o = _run_both(vw, oracle, left, right, lm, rm, 0, 0.0, search, (9, 5), 1, -1, 2, 2, bbox=(64, 32, 128, 96))
    assert g.shape == (96, 128, 3) and np.array_equal(g, o)
    # everything masked -> an all-invalid tile, no error
    g = vw.pyramid_correlate(left, right, np.zeros(left.shape, np.uint8), rm, 0, 0.0, _box(search), (7, 7), 0,
                             filter_half_kernel=3)
    assert not g.any()


@pytest.mark.parametrize("algorithm", [0, 1])
def test_interior_tile_of_a_larger_pair_stages_a_window(vw, oracle, algorithm):
    """The host entry ships only the window of the sources a tile can touch (tile + pyramid padding + search); for a tile
    well inside a larger pair that window is a strict sub-rectangle on every side, and the result must not change —
    BM with masks, and the SGM branch (R->L runs read further out)."""
    from visionworkbench_amd.core import BBox2i
    rng = np.random.default_rng(5)
    H, W = 420, 640
    left = np.floor(rng.random((H, W)) * 256).astype(np.float32)
    right = np.empty_like(left)
    for y0 in range(0, H, 60):
        sh = int(rng.integers(-5, 6))
        right[y0:y0 + 60] = np.roll(left[y0:y0 + 60], sh, axis=1)
    lm = np.full(left.shape, 255, np.uint8)
    rm = np.full(right.shape, 255, np.uint8)
    lm[230:250, 300:330] = 0
    search = (-6, -1, 6, 1)
    bbox = (260, 190, 96, 64)
    if algorithm == 0:
        g, o = _run_both(vw, oracle, left, right, lm, rm, 0, 0.0, search, (7, 7), 0, 2, 3, 2, bbox=bbox)
    else:
        g = vw.pyramid_correlate(left, right, lm, rm, 0, 0.0, _box(search), (5, 5), 3, consistency_threshold=2, filter_half_kernel=3,
                                 max_pyramid_levels=2, algorithm=1, bbox=BBox2i(*bbox))
        o = oracle.pyramid_correlate_sgm(left, right, lm, rm, search, 5, 3, 2, 0, 3, 2, bbox=bbox)
    assert g.shape == (64, 96, 3)
    if algorithm == 0:
        assert np.array_equal(g, o)
    else:
        assert np.array_equal(g[..., 2], o[..., 2]) and np.abs(g - o).max() < 1e-5
    assert (g[..., 2] != 0).mean() > 0.5


def test_level_count_edge_cases(vw, oracle):
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    for levels, srch in [(0, (-3, -2, 4, 3)), (1, (-6, -2, 7, 3)), (5, (0, 0, 1, 1)), (5, (-30, 0, 31, 1))]:
        g, o = _run_both(vw, oracle, left, right, None, None, 0, 0.0, srch, (5, 5), 0, -1, 0, levels)
        assert np.array_equal(g, o), (levels, srch)


def test_torch_device_entry(vw, oracle):
    import torch
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    g = vw.pyramid_correlate(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), None, None, 0, 0.0,
                             _box(search), (7, 7), 0, consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5)
    o = oracle.pyramid_correlate(left, right, None, None, 0, 0.0, search, (7, 7), 0, 0, 0.0, 2, 5, 5)
    assert g.is_cuda and np.array_equal(g.cpu().numpy(), o)


def test_argument_errors(vw):
    from visionworkbench_amd.core import ArgumentErr, NoImplErr, BBox2i
    left, right, *_ = scenes.pyramid_scene("u8")
    with pytest.raises(ArgumentErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, BBox2i(0, 0, 0, 0), (7, 7), 0)
    with pytest.raises(ArgumentErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, BBox2i(0, 0, 4, 4), (6, 7), 0)
    with pytest.raises(NoImplErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, BBox2i(0, 0, 4, 4), (7, 7), 0, algorithm=1)


def _random_disparity(rng, h, w, p_invalid=0.15):
    V = np.iinfo(np.int32).max
    d = np.zeros((h, w, 3), np.int32)
    d[..., 0] = rng.integers(-4, 5, (h, w)) + (np.arange(w)[None, :] // 16)
    d[..., 1] = rng.integers(-2, 3, (h, w))
    d[..., 2] = np.where(rng.random((h, w)) < p_invalid, 0, V)
    return d


@pytest.mark.parametrize("cleanup", [0, 1])
@pytest.mark.parametrize("hk", [(1, 1), (3, 2), (5, 5)])
def test_disparity_filters_identical(vw, oracle, cleanup, hk):
    rng = np.random.default_rng(7)
    d = _random_disparity(rng, 67, 93)
    fn = vw.disparity_cleanup_using_thresh if cleanup else vw.rm_outliers_using_thresh
    for pthr, rthr in [(3.0, 0.5), (1.0, 0.3), (0.0, 0.05)]:
        g = fn(d, hk[0], hk[1], pthr, rthr)
        assert np.array_equal(g, oracle.disparity_filter(d, hk[0], hk[1], pthr, rthr, cleanup))


@pytest.mark.parametrize("cleanup", [0, 1])
def test_disparity_filter_thresholds_and_shapes(vw, oracle, cleanup):
    """The LDS-tiled filter compares |int32 difference| <= floor(pthr): fractional, zero, huge, negative and NaN pixel thresholds,
    wide / tall neighbourhoods, images smaller than a 32 x 8 tile and not a multiple of it."""
    rng = np.random.default_rng(70)
    fn = vw.disparity_cleanup_using_thresh if cleanup else vw.rm_outliers_using_thresh
    for (h, w), hk in [((5, 7), (2, 1)), ((40, 33), (7, 3)), ((9, 130), (1, 6)), ((64, 64), (4, 4)), ((30, 40), (45, 40))]:   # the last: no LDS tile
        d = _random_disparity(rng, h, w)
        for pthr, rthr in [(0.5, 0.4), (2.75, 0.6), (1e12, 0.9), (-1.0, 0.1), (float("nan"), 0.1), (3.0, 0.0), (3.0, 1.5)]:
            g = fn(d, hk[0], hk[1], pthr, rthr)
            assert np.array_equal(g, oracle.disparity_filter(d, hk[0], hk[1], pthr, rthr, cleanup)), (h, w, hk, pthr, rthr)


@pytest.mark.parametrize("cleanup", [0, 1])
def test_disparity_filter_packed_form_limits(vw, oracle, cleanup):
    """The filter's packed 16-bit form serves tiles whose valid disparities are below 2^13 with a threshold below 2^13; everything else
    takes the int32 form in the same launch: values at and beyond the limit (in a corner of the image only, so that both forms run),
    int32 extremes (the difference wraps around like the reference's), garbage under invalid pixels, thresholds around 2^13."""
    rng = np.random.default_rng(71)
    fn = vw.disparity_cleanup_using_thresh if cleanup else vw.rm_outliers_using_thresh
    V = np.iinfo(np.int32).max
    for case in range(4):
        d = _random_disparity(rng, 70, 150)
        if case == 0:
            d[:12, :40, 0] += rng.choice([8180, 8192, -8180, -8192, 20000], (12, 40))
        elif case == 1:
            d[30:, 100:, 1] = rng.choice([V, -V - 1, V - 3, 0, 5], d[30:, 100:, 1].shape)
        elif case == 2:
            inval = d[..., 2] == 0
            d[..., 0][inval] = rng.integers(-2**31, 2**31 - 1, inval.sum())
            d[..., 1][inval] = rng.integers(-2**31, 2**31 - 1, inval.sum())
        else:
            d[..., 0] = rng.integers(-8191, 8192, d.shape[:2]); d[..., 1] = rng.integers(-8191, 8192, d.shape[:2])
            d[40:, :, 1] = rng.integers(-9000, 9000, d[40:, :, 1].shape)
        for pthr in (3.0, 8191.0, 8192.0, 40000.0):
            for hk in ((5, 5), (2, 3)):
                g = fn(d, hk[0], hk[1], pthr, 0.3)
                assert np.array_equal(g, oracle.disparity_filter(d, hk[0], hk[1], pthr, 0.3, cleanup)), (case, pthr, hk)


def test_disparity_mask_identical(vw, oracle):
    rng = np.random.default_rng(8)
    d = _random_disparity(rng, 50, 70)
    lm = (rng.random((50, 70)) > 0.1).astype(np.uint8) * 255
    rm = (rng.random((55, 80)) > 0.1).astype(np.uint8) * 255
    assert np.array_equal(vw.disparity_mask(d, lm, rm), oracle.disparity_mask(d, lm, rm))


def test_subdivide_regions_identical(vw, oracle):
    rng = np.random.default_rng(9)
    for h, w, k in [(64, 96, (7, 7)), (33, 47, (5, 9)), (128, 200, (3, 3))]:
        d = _random_disparity(rng, h, w, 0.3)
        d[: h // 2, : w // 3, 0] += 25
        d[h // 3:, w // 2:, 2] = 0
        z = vw.subdivide_regions(d, k)
        zo = oracle.subdivide_regions(d, k)
        got = [r.min + r.max + s.min + s.max for r, s in z]
        assert got == [list(map(int, row)) for row in zo]


def test_blob_filter_identical(vw, oracle):
    rng = np.random.default_rng(21)
    for h, w, p in [(6, 6, 0.6), (64, 96, 0.55), (200, 300, 0.45), (128, 128, 0.62)]:
        d = _random_disparity(rng, h, w, p)
        for area in (0, 1, 2, 5, 40, 1000, h * w):
            assert np.array_equal(vw.disparity_blob_filter(d, area), oracle.disparity_blob_filter(d, area)), (h, w, area)
    # one long snake: a single component whose union-find tree is deep
    d = np.zeros((40, 400, 3), np.int32)
    d[::4, :, 2] = np.iinfo(np.int32).max
    d[2::8, -1, 2] = d[6::8, 0, 2] = np.iinfo(np.int32).max
    d[1::8, -1, 2] = d[3::8, -1, 2] = d[5::8, 0, 2] = d[7::8, 0, 2] = np.iinfo(np.int32).max
    for area in (399, 3999, 4100):
        assert np.array_equal(vw.disparity_blob_filter(d, area), oracle.disparity_blob_filter(d, area))


@pytest.mark.parametrize("algorithm", [0, 1])
def test_pyramid_with_blob_filter(vw, oracle, algorithm):
    from visionworkbench_amd.core import BBox2i
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    box = BBox2i.from_corners(search[:2], search[2:])
    oracle.set_blob_filter_area(40)
    try:
        if algorithm == 0:
            g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (7, 7), 0, consistency_threshold=2, filter_half_kernel=3,
                                     max_pyramid_levels=3, blob_filter_area=40)
            o = oracle.pyramid_correlate(left, right, None, None, 0, 0.0, search, (7, 7), 0, 0, 0.0, 2, 3, 3)
            assert np.array_equal(g, o)
        else:
            g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (5, 5), 3, consistency_threshold=2, filter_half_kernel=3,
                                     max_pyramid_levels=3, algorithm=1, blob_filter_area=40)
            o = oracle.pyramid_correlate_sgm(left, right, None, None, search, 5, 3, 2, 0, 3, 3)
            assert np.array_equal(g[..., 2], o[..., 2]) and np.abs(g[..., :2] - o[..., :2]).max() < 1e-5
    finally:
        oracle.set_blob_filter_area(0)
    # the filter must have had something to do in this scene
    g0 = vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (7, 7) if algorithm == 0 else (5, 5), 0 if algorithm == 0 else 3,
                              consistency_threshold=2, filter_half_kernel=3, max_pyramid_levels=3, algorithm=algorithm)
    assert (g0[..., 2] != 0).sum() >= (g[..., 2] != 0).sum()


def test_cross_corr_consistency_check_with_diff(vw, oracle):
    """cross_corr_consistency_check's optional lr_disp_diff output (Correlate.cc:1441-1502)."""
    rng = np.random.default_rng(31)
    V = np.iinfo(np.int32).max
    l2r = np.zeros((40, 60, 3), np.int32)
    l2r[..., 0] = rng.integers(0, 6, (40, 60)); l2r[..., 1] = rng.integers(0, 3, (40, 60))
    l2r[..., 2] = np.where(rng.random((40, 60)) < 0.2, 0, V)
    r2l = np.zeros((44, 68, 3), np.int32)
    r2l[..., 0] = -rng.integers(0, 6, (44, 68)); r2l[..., 1] = -rng.integers(0, 3, (44, 68))
    r2l[..., 2] = np.where(rng.random((44, 68)) < 0.2, 0, V)
    for thr in (0, 1, 2):
        a, b = l2r.copy(), l2r.copy()
        da = np.full((50, 70, 2), -7.0, np.float32); db = da.copy()
        vw.cross_corr_consistency_check(a, r2l, thr, lr_disp_diff=da, ul_corner_offset=(5, 3))
        oracle.cross_corr_consistency_check_diff(b, r2l, thr, db, (5, 3))
        assert np.array_equal(a, b) and np.array_equal(da, db)
        kept = a[..., 2] != 0
        assert (da[3:43, 5:65, 1][kept] == 1.0).all() and (da[3:43, 5:65, 0][kept] <= thr).all()
        assert (da[3:43, 5:65][~kept] == -7.0).all()                      # untouched where the pixel was rejected
    from visionworkbench_amd.core import ArgumentErr
    with pytest.raises(ArgumentErr):
        vw.cross_corr_consistency_check(l2r.copy(), r2l, 1, lr_disp_diff=np.zeros((30, 70, 2), np.float32))


@pytest.mark.parametrize("algorithm", [0, 1])
def test_pyramid_lr_disp_diff(vw, oracle, algorithm):
    from visionworkbench_amd.core import BBox2i, ArgumentErr
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    box = BBox2i.from_corners(search[:2], search[2:])
    bb = (40, 24, 200, 150)
    dg = np.zeros((170, 230, 2), np.float32)          # covers image pixels [30, 260) x [20, 190)
    do = dg.copy()
    oracle.set_lr_disp_diff(do, (30, 20))
    try:
        if algorithm == 0:
            g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (7, 7), 0, consistency_threshold=2, filter_half_kernel=3,
                                     max_pyramid_levels=3, bbox=BBox2i(*bb), lr_disp_diff=dg, region_ul=(30, 20))
            o = oracle.pyramid_correlate(left, right, None, None, 0, 0.0, search, (7, 7), 0, 0, 0.0, 2, 3, 3, bbox=bb)
            assert np.array_equal(g, o)
        else:
            g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (5, 5), 3, consistency_threshold=2, filter_half_kernel=3,
                                     max_pyramid_levels=3, algorithm=1, bbox=BBox2i(*bb), lr_disp_diff=dg, region_ul=(30, 20))
            o = oracle.pyramid_correlate_sgm(left, right, None, None, search, 5, 3, 2, 0, 3, 3, bbox=bb)
            assert np.array_equal(g[..., 2], o[..., 2])
    finally:
        oracle.set_lr_disp_diff(None)
    assert np.array_equal(dg, do)
    inside = dg[4:154, 10:210]
    assert (inside[..., 1] == (g[..., 2] != 0)).all() and (inside[..., 0][g[..., 2] != 0] <= 2).all()
    assert not dg[:4].any() and not dg[:, :10].any()                      # nothing outside the tile was touched
    with pytest.raises(ArgumentErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (7, 7), 0, consistency_threshold=2, bbox=BBox2i(0, 0, 100, 100),
                             lr_disp_diff=dg, region_ul=(30, 20))


def test_corr_timeout_estimate(vw, oracle):
    """corr_timeout (CorrelationView.cc:620-637): zones are dropped when seconds_per_op * search volume would pass the budget.
    While the estimate stays within 2 s of the last wall-clock measurement it is the estimate alone that decides — identical to
    the oracle; a seconds_per_op calibrated for a slow CPU is re-measured against the wall clock and cannot stop the GPU early."""
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    # per-zone estimates of a few ms: the budget runs out in the middle of level 0 on both sides
    g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, _box(search), (7, 7), 0, 1, 2e-7, -1, 0, 5, 5)
    o = oracle.pyramid_correlate(left, right, None, None, 0, 0.0, search, (7, 7), 0, 1, 2e-7, -1, 5, 5)
    assert np.array_equal(g, o)
    full = vw.pyramid_correlate(left, right, None, None, 0, 0.0, _box(search), (7, 7), 0, 0, 0.0, -1, 0, 5, 5)
    assert (g[..., 2] != 0).sum() < 0.8 * (full[..., 2] != 0).sum()          # the budget really ran out
    # a CPU-calibrated seconds_per_op (85 s of estimate for this tile, budget 5 s): the reference would re-measure after 2 s of ESTIMATE and carry on with the
    # wall clock; so does the engine: the tile completes
    g2 = vw.pyramid_correlate(left, right, None, None, 0, 0.0, _box(search), (7, 7), 0, 5, 1e-5, -1, 0, 5, 5)
    assert (g2[..., 2] != 0).mean() > 0.5


@pytest.mark.parametrize("k", [(15, 15), (25, 25), (5, 23)])
def test_large_and_oblong_kernels(vw, oracle, k):
    """Kernels beyond the instantiated sizes (ASP users run up to 25 x 25): run-time window loops of the zone kernel, the exact-order
    kernels under LoG, levels 0 and 3."""
    from visionworkbench_amd.core import BBox2i
    rng = np.random.default_rng(3)
    H, W = 220, 300
    left = np.floor(rng.random((H, W)) * 256).astype(np.float32)
    right = np.roll(left, 4, axis=1)
    right[100:] = np.roll(left[100:], -3, axis=1)
    s = (-8, -2, 9, 3)
    for cost in (0, 2):
        for pf, pfw in ((0, 0.0), (2, 1.4)):
            for levels in (0, 3):
                g = vw.pyramid_correlate(left, right, None, None, pf, pfw, BBox2i.from_corners(s[:2], s[2:]), k, cost, 0, 0.0, 2, 0, 3, levels)
                o = oracle.pyramid_correlate(left, right, None, None, pf, pfw, s, k, cost, 0, 0.0, 2, 3, levels)
                assert np.array_equal(g, o), (k, cost, pf, levels)


def test_concurrent_tile_threads_return_the_single_thread_result():
    """The reference pulls tiles with a thread pool (ImageIO.h:228-251): four host threads, a context and a stream each, the correlate tool's
    defaults (LoG + NCC: the exact-order kernels, their tables, flags and scratch arenas) and integer SAD — every tile, every time, the bits of
    the single-threaded run."""
    import threading
    import torch
    from visionworkbench_amd import core, stereo, synth
    from visionworkbench_amd.core import BBox2i
    W, T = 1536, 512
    L, R, _ = synth.stereo_pair(W, W, 129, 1)
    Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, 64:64 + W].copy()).cuda()
    tiles = [(x, y) for y in range(0, W, T) for x in range(0, W, T)]
    for pf, cost, k in [(2, 2, 11), (0, 0, 7)]:
        run = lambda c, x, y: stereo.pyramid_correlate(Lg, Rg, None, None, pf, 1.4 if pf else 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (k, k), cost,
                                                       consistency_threshold=2, filter_half_kernel=5, max_pyramid_levels=5, bbox=BBox2i(x, y, T, T), ctx=c)
        ref_ctx = core.Context(0)
        want = {t: run(ref_ctx, *t).cpu().numpy() for t in tiles}
        ref_ctx.close()
        todo = list(tiles) * 3
        lock = threading.Lock()
        bad = []
        ctxs = [core.Context(0) for _ in range(4)]
        streams = [torch.cuda.Stream() for _ in range(4)]
        def work(c, st):
            with torch.cuda.stream(st):
                while True:
                    with lock:
                        if not todo:
                            return
                        t = todo.pop()
                    got = run(c, *t).cpu().numpy()
                    if not np.array_equal(got, want[t]):
                        with lock:
                            bad.append(t)
        th = [threading.Thread(target=work, args=(c, st)) for c, st in zip(ctxs, streams)]
        for t in th: t.start()
        for t in th: t.join()
        for c in ctxs: c.close()
        assert not bad, (pf, cost, k, bad)
