"""Semi-global matching on the GPU (csrc/sgm.hip) vs the oracle's restatement of SemiGlobalMatcher.

Integer work end to end (u8 images, census words, u8 costs, saturating u16 accumulation, WTA): the integer disparity and
validity must be IDENTICAL to the oracle's; the sub-pixel disparity (float64 cos / erf) must agree within 1e-5."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CENSUS, TERNARY = 3, 4


@pytest.fixture(scope="module")
def vw():
    from visionworkbench_amd import stereo
    return stereo


def _box(w, h):
    from visionworkbench_amd.core import BBox2i
    return BBox2i(0, 0, w, h)


def _pair(rng, h, w, sx, sy, shift=(3, 2), smooth=False, scale=255.0):
    base = rng.random((h + sy + 8, w + sx + 8))
    if smooth:
        k = np.ones(5) / 5
        base = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, base)
        base = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, base)
    base = (base * scale).astype(np.float32)
    left = base[4:4 + h, 4:4 + w]
    right = base[4 - shift[1]:4 - shift[1] + h + sy, 4 - shift[0]:4 - shift[0] + w + sx]     # left(x,y) = right(x+sx_, y+sy_)
    return np.ascontiguousarray(left), np.ascontiguousarray(right)


def _both(vw, oracle, cost, left, right, search, k, sub, sb=(2, 2), mem=6000, lm=None, rm=None, prev=None, mgm=False, p1=0, p2=0, block=False):
    h, w = left.shape
    gi, gs = vw.calc_disparity_sgm(cost, left, right, _box(w, h), search, (k, k), use_mgm=mgm, subpixel_mode=sub, search_buffer=sb,
                                   memory_limit_mb=mem, left_mask=lm, right_mask=rm, prev_disparity=prev, with_subpixel=True, p1=p1, p2=p2,
                                   allow_block_cost=block)
    oi, os_ = oracle.calc_disparity_sgm(cost, left, right, search, k, subpixel=sub, search_buffer=sb, memory_limit_mb=mem,
                                        left_mask=lm, right_mask=rm, prev_disparity=prev, use_mgm=mgm, p1=p1, p2=p2, allow_block_cost=block)
    return gi, gs, oi, os_


@pytest.mark.parametrize("mgm", [False, True])
def test_reference_fixture_constant_offset(vw, oracle, mgm):
    """TestSGM.cxx:28-75 on the reference's own images (the reference's test runs use_mgm = false only)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgm_fixture.npz"))
    left, right = d["left"].astype(np.float32), d["right"].astype(np.float32)
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (9, 9), 3, 5, sb=(4, 4), mem=1024, mgm=mgm)
    assert gi.shape == (398, 398, 3)
    assert ((gi[..., 0] - 4 == 2) & (gi[..., 1] - 4 == 1)).mean() > 0.99
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("cost", [CENSUS, TERNARY])
@pytest.mark.parametrize("k", [3, 5, 7, 9])
def test_cost_types_and_kernels_identical(vw, oracle, cost, k):
    rng = np.random.default_rng(10 * cost + k)
    left, right = _pair(rng, 60, 83, 9, 5, smooth=(k % 4 == 1))
    gi, gs, oi, os_ = _both(vw, oracle, cost, left, right, (9, 5), k, 5)
    assert gi.shape == oi.shape == (60 - k + 1, 83 - k + 1, 3)
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5
    inner = gi[8:-8, 8:-8]
    assert ((inner[..., 0] == 3) & (inner[..., 1] == 2)).mean() > 0.9


@pytest.mark.parametrize("sub", [0, 1, 2, 3, 4, 5])
def test_subpixel_modes(vw, oracle, sub):
    rng = np.random.default_rng(sub)
    left, right = _pair(rng, 48, 64, 6, 4, smooth=True)
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (6, 4), 5, sub)
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5
    assert np.array_equal(gs[..., 2] != 0, gi[..., 2] != 0)


def test_flat_image_tie_smoothing(vw, oracle):
    """Large flat areas give many tied minima: the smoothing iterations of select_best_disparity must match."""
    rng = np.random.default_rng(5)
    left, right = _pair(rng, 50, 70, 8, 6)
    left[10:40, 15:55] = 100.0
    right[12:42, 18:58] = 100.0
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (8, 6), 5, 5)
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5


def test_masks_prev_disparity_and_memory_levels(vw, oracle):
    rng = np.random.default_rng(11)
    left, right = _pair(rng, 64, 96, 12, 12)
    k = 5
    oh, ow = 64 - k + 1, 96 - k + 1
    lm = np.full((oh, ow), 255, np.uint8)
    lm[10:20, 10:30] = 0
    rm = np.full((oh + 12, ow + 12), 255, np.uint8)
    rm[:, -9:] = 0
    rm[:3] = 0
    prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
    prev[..., 0], prev[..., 1], prev[..., 2] = 2, 1, np.iinfo(np.int32).max
    prev[5:12, 8:20, 2] = 0                                  # untrusted -> full search -> constrained by neighbours
    prev[20:25, 30:40, 0] = 0                                # on the edge of a >= 10 wide range: untrusted
    for mem in (6000, 1):                                    # 1 MB forces the conservation levels (SGM.cc:468-491)
        gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (12, 12), k, 5, lm=lm, rm=rm, prev=prev, mem=mem)
        assert np.array_equal(gi, oi), mem
        assert np.abs(gs - os_).max() < 1e-5
    assert (gi[10:20, 10:30, 2] == 0).all()


@pytest.mark.parametrize("search", [(30, 20), (40, 30)])
def test_large_two_d_search_with_masks(vw, oracle, search):
    """Ragged vectors with many disparities: 31 x 21 = 651 (in-place kernel, 16 slots per lane) and 41 x 31 = 1271 (the
    three-phase kernel that takes any size); the left mask makes the boxes non-uniform."""
    rng = np.random.default_rng(21)
    sx, sy = search
    left, right = _pair(rng, 40, 56, sx, sy)
    k = 5
    oh, ow = 40 - k + 1, 56 - k + 1
    lm = np.full((oh, ow), 255, np.uint8)
    lm[5:12, 8:20] = 0
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, search, k, 5, lm=lm)
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5


def test_float_range_is_stretched(vw, oracle):
    """u8_convert (ImageThresh.h:275-286): any float range, including negative values and a constant image."""
    rng = np.random.default_rng(12)
    left, right = _pair(rng, 40, 50, 5, 3, scale=1.0)
    left, right = left * 3000.0 - 1000.0, right * 3000.0 - 1000.0
    gi, gs, oi, os_ = _both(vw, oracle, TERNARY, left, right, (5, 3), 7, 5)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5
    flat = np.full((30, 40), 7.5, np.float32)
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, flat, np.full((33, 45), 7.5, np.float32), (5, 3), 3, 5)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5


def test_torch_device_entry_and_errors(vw, oracle):
    import torch
    from visionworkbench_amd.core import ArgumentErr, NoImplErr
    rng = np.random.default_rng(13)
    left, right = _pair(rng, 40, 50, 5, 3)
    g = vw.calc_disparity_sgm(CENSUS, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), _box(50, 40), (5, 3), (5, 5))
    oi, _ = oracle.calc_disparity_sgm(CENSUS, left, right, (5, 3), 5)
    assert g.is_cuda and np.array_equal(g.cpu().numpy(), oi)
    with pytest.raises(NoImplErr):
        vw.calc_disparity_sgm(0, left, right, _box(50, 40), (5, 3), (5, 5))          # block cost with SGM
    with pytest.raises(NoImplErr):
        vw.calc_disparity_sgm(CENSUS, left, right, _box(50, 40), (5, 3), (11, 11))    # census sizes 3..9 only
    with pytest.raises(NoImplErr):
        vw.calc_disparity_sgm(0, left, right, _box(50, 40), (5, 3), (5, 5), use_mgm=True)   # block cost with MGM
    with pytest.raises(ArgumentErr):
        vw.calc_disparity_sgm(CENSUS, left, right, _box(51, 40), (5, 3), (5, 5))


# ---- the fused raster sweeps (VWGPU_OPT_SGM_SWEEP; csrc/sgm.hip sweep_uniform_kernel): another schedule, the same sums -------------

@pytest.mark.parametrize("sweep", [1, 2, 5])
@pytest.mark.parametrize("k,sx,w,h,cost", [(7, 128, 300, 40, CENSUS), (5, 16, 64, 20, CENSUS), (3, 60, 1000, 90, TERNARY), (9, 130, 257, 70, CENSUS),
                                           (5, 255, 300, 33, CENSUS), (7, 95, 100, 300, TERNARY), (7, 8, 41, 25, CENSUS)])
def test_fused_sweeps_identical_to_oracle(vw, oracle, sweep, k, sx, w, h, cost):
    """Two concurrent sweeps of four directions each, one wavefront per row, vectors handed from row to row through LDS rings and
    from workgroup to workgroup through HBM (1 = as many rows per workgroup as fit, 2 / 5 = pinned: more hand-offs): identical
    integer disparities, sub-pixel values within 1e-5, for 1 .. 8 disparity pairs per lane and rows that end inside a workgroup."""
    from visionworkbench_amd import core
    ctx = core.default_context(0)
    rng = np.random.default_rng(7 * k + sx + sweep)
    left = np.floor(rng.random((h, w)) * 256).astype(np.float32)
    right = np.floor(rng.random((h, w + sx)) * 256).astype(np.float32)
    right[:, sx // 3:sx // 3 + w] = left
    ctx.set_option(core.OPT_SGM_SWEEP, sweep)
    try:
        gi, gs = vw.calc_disparity_sgm(cost, left, right, _box(w, h), (sx, 0), (k, k), with_subpixel=True, ctx=ctx)
    finally:
        ctx.set_option(core.OPT_SGM_SWEEP, 0)
    oi, os_ = oracle.calc_disparity_sgm(cost, left, right, (sx, 0), k)
    assert np.array_equal(gi, oi), int((gi != oi).any(-1).sum())
    assert np.abs(gs - os_).max() < 1e-5


# ---- the one-direction-per-launch schedule in all its forms (VWGPU_OPT_SGM_PATH_MODE): path_ring_kernel (LDS ring, the default),
# path_uniform_reg_kernel (round 2), lines per workgroup, XCD clusters, census costs formed in the ring kernel -------------------------

@pytest.mark.parametrize("mode", [0, 8, 512 + 8, 1024, 2048, 2048 + 512, 2048 + 8, 32, 32 + 1, 32 + 8])
@pytest.mark.parametrize("k,sx,w,h,cost", [(7, 128, 300, 40, CENSUS), (5, 16, 64, 20, CENSUS), (3, 60, 500, 90, TERNARY), (9, 130, 257, 70, CENSUS),
                                           (7, 127, 100, 300, TERNARY), (7, 8, 41, 25, CENSUS), (5, 159, 64, 47, CENSUS), (5, 200, 90, 33, CENSUS),
                                           (7, 128, 12, 9, CENSUS), (7, 40, 3000, 9, CENSUS)])
def test_path_modes_identical_to_oracle(vw, oracle, mode, k, sx, w, h, cost):
    """Lines shorter than a chunk of the ring, widths that leave workgroups partly empty, 129 disparities (the 33rd lane), more than 160
    bytes of stride (falls back to the register kernel), the winner-take-all inside the last direction: the same disparities every time."""
    from visionworkbench_amd import core
    ctx = core.default_context(0)
    rng = np.random.default_rng(11 * k + sx + mode)
    left = np.floor(rng.random((h, w)) * 256).astype(np.float32)
    right = np.floor(rng.random((h, w + sx)) * 256).astype(np.float32)
    right[:, sx // 3:sx // 3 + w] = left
    left[h // 2:, : w // 2] = 90.0                                  # a flat patch: ties for the smoothing kernel
    ctx.set_option(core.OPT_SGM_PATH_MODE, mode)
    try:
        gi, gs = vw.calc_disparity_sgm(cost, left, right, _box(w, h), (sx, 0), (k, k), with_subpixel=True, ctx=ctx)
    finally:
        ctx.set_option(core.OPT_SGM_PATH_MODE, 0)
    oi, os_ = oracle.calc_disparity_sgm(cost, left, right, (sx, 0), k)
    assert np.array_equal(gi, oi), int((gi != oi).any(-1).sum())
    assert np.abs(gs - os_).max() < 1e-5


def test_fused_sweeps_flat_image_and_user_penalties(vw, oracle):
    """Ties everywhere (the packed winner-take-all hands the pixel to the smoothing kernel, which must find the SUM of the two sweeps'
    volumes), and penalties large enough for the u16 sums to wrap."""
    from visionworkbench_amd import core
    ctx = core.default_context(0)
    flat = np.full((30, 90), 77.0, np.float32); flat[0, 0] = 0; flat[-1, -1] = 255
    rflat = np.full((30, 130), 77.0, np.float32); rflat[0, 0] = 0; rflat[-1, -1] = 255
    rng = np.random.default_rng(3)
    left = np.floor(rng.random((40, 120)) * 256).astype(np.float32)
    right = np.floor(rng.random((40, 160)) * 256).astype(np.float32)
    right[:, 20:140] = left
    ctx.set_option(core.OPT_SGM_SWEEP, 1)
    try:
        gi, gs = vw.calc_disparity_sgm(CENSUS, flat, rflat, _box(90, 30), (40, 0), (5, 5), with_subpixel=True, ctx=ctx)
        g2 = vw.calc_disparity_sgm(CENSUS, left, right, _box(120, 40), (40, 0), (7, 7), p1=2000, p2=60000, ctx=ctx)
    finally:
        ctx.set_option(core.OPT_SGM_SWEEP, 0)
    oi, os_ = oracle.calc_disparity_sgm(CENSUS, flat, rflat, (40, 0), 5)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5
    o2, _ = oracle.calc_disparity_sgm(CENSUS, left, right, (40, 0), 7, p1=2000, p2=60000)
    assert np.array_equal(g2, o2)


# ---- the mean-abs-difference block cost (fill_costs_block, SGM.cc:1651-1738): behind the reference's throw, opt-in here ------

@pytest.mark.parametrize("cost", [0, 1])
@pytest.mark.parametrize("k,sx,w,h", [(7, 128, 300, 40), (3, 33, 530, 20), (5, 64, 257, 23), (9, 40, 100, 30), (11, 17, 70, 30),
                                      (13, 20, 80, 36), (1, 12, 60, 20), (15, 9, 64, 40)])
def test_block_cost_one_search_row(vw, oracle, cost, k, sx, w, h):
    """BASELINE configs[3] as written (a SAD-class cost into SGM): uniform one-row searches take the packed qsad kernel for kernels
    3 .. 11 and the general kernel otherwise; ABSOLUTE_DIFFERENCE and SQUARED_DIFFERENCE both mean "mean of abs differences"."""
    rng = np.random.default_rng(100 * k + sx + cost)
    left = np.floor(rng.random((h, w)) * 256).astype(np.float32)
    right = np.floor(rng.random((h, w + sx)) * 256).astype(np.float32)
    right[:, sx // 3:sx // 3 + w] = left                                     # left(x, y) = right(x + sx / 3, y)
    left[0, 0], left[-1, -1] = 0.0, 255.0                                    # u8_convert stretches [min, max] to [0, 255]: keep the values
    gi, gs, oi, os_ = _both(vw, oracle, cost, left, right, (sx, 0), k, 5, block=True)
    assert gi.shape == oi.shape == (h - k + 1, w - k + 1, 3)
    assert np.array_equal(gi, oi), int((gi != oi).any(-1).sum())
    assert np.abs(gs - os_).max() < 1e-5
    assert (gi[..., 0] == sx // 3).mean() > 0.8


def test_block_cost_extreme_values_and_division(vw, oracle):
    """Costs of 0 and 255 (black against white), pad bytes of the packed kernel next to 255-valued pixels, a flat pair."""
    rng = np.random.default_rng(4)
    left = (rng.random((30, 140)) > 0.5).astype(np.float32) * 255.0
    right = np.concatenate([255.0 - left[:, :20], left, (rng.random((30, 60)) > 0.5).astype(np.float32) * 255.0], axis=1).astype(np.float32)
    for k in (3, 5, 7, 9, 11):
        gi, gs, oi, os_ = _both(vw, oracle, 0, left, right, (80, 0), k, 5, block=True)
        assert np.array_equal(gi, oi), k
    flat = np.full((20, 90), 77.0, np.float32); flat[0, 0] = 0; flat[-1, -1] = 255
    rflat = np.full((20, 120), 77.0, np.float32); rflat[0, 0] = 0; rflat[-1, -1] = 255
    gi, gs, oi, os_ = _both(vw, oracle, 0, flat, rflat, (30, 0), 7, 5, block=True)
    assert np.array_equal(gi, oi)


def test_block_cost_ragged_boxes_two_d_and_mgm(vw, oracle):
    rng = np.random.default_rng(12)
    left, right = _pair(rng, 64, 96, 12, 12)
    k = 5
    oh, ow = 64 - k + 1, 96 - k + 1
    lm = np.full((oh, ow), 255, np.uint8)
    lm[10:20, 10:30] = 0
    rm = np.full((oh + 12, ow + 12), 255, np.uint8)
    rm[:, -9:] = 0
    prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
    prev[..., 0], prev[..., 1], prev[..., 2] = 2, 1, np.iinfo(np.int32).max
    prev[5:12, 8:20, 2] = 0
    gi, gs, oi, os_ = _both(vw, oracle, 0, left, right, (12, 12), k, 5, lm=lm, rm=rm, prev=prev, block=True)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5
    gi, gs, oi, os_ = _both(vw, oracle, 0, left, right, (12, 12), k, 5, block=True)              # uniform 2-D search
    assert np.array_equal(gi, oi)
    gi, gs, oi, os_ = _both(vw, oracle, 0, left, right, (12, 12), k, 5, lm=lm, mgm=True, block=True, p1=5, p2=90)
    assert np.array_equal(gi, oi)


def test_block_cost_needs_the_opt_in(vw, oracle):
    from visionworkbench_amd.core import ArgumentErr, NoImplErr
    rng = np.random.default_rng(13)
    left, right = _pair(rng, 40, 50, 5, 3)
    for cost in (0, 1, 2):
        with pytest.raises(NoImplErr):
            vw.calc_disparity_sgm(cost, left, right, _box(50, 40), (5, 3), (5, 5))                 # SGM.cc:1887-1892
    with pytest.raises(NoImplErr):
        vw.calc_disparity_sgm(2, left, right, _box(50, 40), (5, 3), (5, 5), allow_block_cost=True)  # the NCC flavour is not restated
    with pytest.raises(ArgumentErr):
        vw.calc_disparity_sgm(0, left, right, _box(50, 40), (5, 3), (17, 17), allow_block_cost=True)
    with pytest.raises(ValueError):
        oracle.calc_disparity_sgm(0, left, right, (5, 3), 5)


# ---- pyramid_correlate with VW_CORRELATION_SGM ------------------------------------------------------------------------

@pytest.mark.parametrize("thr", [-1, 2])
@pytest.mark.parametrize("k,cost", [(5, CENSUS), (7, TERNARY)])
def test_pyramid_sgm_identical_to_oracle(vw, oracle, thr, k, cost):
    """The SGM branch of PyramidCorrelationView::prerasterize (CorrelationView.cc:391-595, 862-875) on the scene of
    TestPyramidCorrelationView.cxx: integer part and validity identical, sub-pixel part within 1e-5."""
    import scenes
    from visionworkbench_amd.core import BBox2i
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, BBox2i.from_corners(search[:2], search[2:]), (k, k), cost,
                             consistency_threshold=thr, filter_half_kernel=5, max_pyramid_levels=5, algorithm=1)
    o = oracle.pyramid_correlate_sgm(left, right, None, None, search, k, cost, thr, 0, 5, 5)
    assert g.shape == o.shape == left.shape + (3,)
    assert np.array_equal(g[..., 2], o[..., 2])
    assert np.abs(g[..., :2] - o[..., :2]).max() < 1e-5
    c, a = scenes.pyramid_score(np.concatenate([np.rint(g[..., :2]), g[..., 2:]], axis=2), scale, trans)
    assert c > 0.8 and a > 0.75


def test_pyramid_sgm_masks_and_subtile(vw, oracle):
    import scenes
    from visionworkbench_amd.core import BBox2i, NoImplErr
    left, right, scale, trans, search = scenes.pyramid_scene("u8")
    lm = np.full(left.shape, 255, np.uint8)
    rm = np.full(right.shape, 255, np.uint8)
    lm[40:90, 100:160] = 0
    rm[120:, 200:] = 0
    box = BBox2i.from_corners(search[:2], search[2:])
    g = vw.pyramid_correlate(left, right, lm, rm, 0, 0.0, box, (5, 5), CENSUS, consistency_threshold=2, min_consistency_level=1,
                             filter_half_kernel=3, max_pyramid_levels=3, algorithm=1, bbox=BBox2i(32, 16, 200, 150))
    o = oracle.pyramid_correlate_sgm(left, right, lm, rm, search, 5, CENSUS, 2, 1, 3, 3, bbox=(32, 16, 200, 150))
    assert np.array_equal(g[..., 2], o[..., 2]) and np.abs(g[..., :2] - o[..., :2]).max() < 1e-5
    with pytest.raises(NoImplErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (5, 5), 0, algorithm=1)        # block cost with SGM
    with pytest.raises(NoImplErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (5, 5), 0, algorithm=2)        # block cost with MGM
    with pytest.raises(NoImplErr):
        vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (5, 5), CENSUS, algorithm=4)   # VW_CORRELATION_OTHER


@pytest.mark.parametrize("seed", range(6))
def test_randomized_masks_and_previous_level(vw, oracle, seed):
    """Differential test of populate_disp_bound_image + constrain_disp_bound_image + the ragged path recurrence: random
    kernel / search sizes, masks with empty borders, a previous-level disparity with untrusted holes and odd sizes."""
    rng = np.random.default_rng(100 + seed)
    V = np.iinfo(np.int32).max
    k = int(rng.choice([3, 5, 7]))
    sx, sy = int(rng.integers(2, 14)), int(rng.integers(1, 6))
    h, w = int(rng.integers(40, 70)), int(rng.integers(50, 90))
    base = rng.integers(0, 256, (h + sy + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[2:2 + h + sy, 1:1 + w + sx])
    oh, ow = h - k + 1, w - k + 1
    lm = np.full((oh, ow), 255, np.uint8)
    lm[:int(rng.integers(0, 5))] = 0
    lm[:, :int(rng.integers(0, 5))] = 0
    rm = np.full((oh + sy + int(rng.integers(0, 4)), ow + sx + int(rng.integers(0, 4))), 255, np.uint8)
    rm[:int(rng.integers(0, sy + 2))] = 0
    rm[:, :int(rng.integers(0, sx + 2))] = 0
    rm[-int(rng.integers(1, 4)):] = 0
    ph, pw = (oh + 1) // 2 + int(rng.integers(-3, 4)), (ow + 1) // 2 + int(rng.integers(-3, 4))
    prev = np.zeros((ph, pw, 3), np.int32)
    prev[..., 0] = rng.integers(0, sx // 2 + 1, (ph, pw))
    prev[..., 1] = rng.integers(0, sy // 2 + 1, (ph, pw))
    prev[..., 2] = np.where(rng.random((ph, pw)) < 0.3, 0, V)
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (sx, sy), k, 5, lm=lm, rm=rm, prev=prev)
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("sx,sy,k,w,h", [(70, 0, 5, 150, 40), (128, 0, 7, 200, 60), (64, 2, 5, 150, 50), (128, 2, 7, 220, 60), (200, 1, 5, 260, 40)])
def test_uniform_path_many_disparities(vw, oracle, sx, sy, k, w, h):
    """Full-range boxes everywhere -> the in-place one-wavefront path kernel, 2..7 disparities per lane, 1-D and 2-D."""
    rng = np.random.default_rng(sx + sy)
    base = rng.integers(0, 256, (h + sy + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[3:3 + h + sy, 1:1 + w + sx])
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (sx, sy), k, 5)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("sx,k,w,h", [(6, 3, 11, 9), (63, 5, 120, 21), (126, 5, 180, 24), (127, 5, 180, 24), (129, 5, 190, 22), (254, 5, 300, 20),
                                      (255, 5, 300, 20), (300, 5, 340, 18), (383, 3, 400, 12), (511, 3, 540, 11)])
def test_register_path_kernel_lane_layouts(vw, oracle, sx, k, w, h):
    """path_uniform_reg_kernel: 1, 2 and 4 disparity pairs per lane (3 is served as 4), odd and even counts, vectors that fill the
    stride exactly, lines shorter than one prefetch chunk (the 9 x 7 output), the store / read-modify-write direction order."""
    rng = np.random.default_rng(1000 + sx)
    base = rng.integers(0, 256, (h + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[4:4 + h, 2:2 + w + sx])
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (sx, 0), k, 5)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("p1,p2", [(5, 60), (200, 9000), (1, 65000)])
def test_user_penalties_incl_u16_wrap(vw, oracle, p1, p2):
    """User-provided P1 / P2 (SGM.cc:106-160).  With P2 > 7937 the eight path costs of a pixel can exceed 65535: the reference's u16
    accumulator wraps, and so does the v_pk_add_u16 accumulation of the one-direction-per-launch path kernel."""
    rng = np.random.default_rng(p2)
    sx, k, h, w = 40, 5, 30, 90
    base = rng.integers(0, 256, (h + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[4:4 + h, 1:1 + w + sx])
    box = _box(w, h)
    noise = rng.integers(0, 256, right.shape).astype(np.float32)          # an unrelated right image: large path costs everywhere
    for rr in (right, noise):
        gi, gs = vw.calc_disparity_sgm(CENSUS, left, rr, box, (sx, 0), (k, k), subpixel_mode=5, with_subpixel=True, p1=p1, p2=p2)
        oi, os_ = oracle.calc_disparity_sgm(CENSUS, left, rr, (sx, 0), k, subpixel=5, p1=p1, p2=p2)
        assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5
    # the other path kernels: a 2-D search (uniform, LDS-resident vectors) and ragged boxes (masks + a previous level)
    sy = 2
    base2 = rng.integers(0, 256, (h + sy + 8, w + 12 + 8)).astype(np.float32)
    l2, r2 = np.ascontiguousarray(base2[4:4 + h, 4:4 + w]), np.ascontiguousarray(rng.integers(0, 256, (h + sy, w + 12)).astype(np.float32))
    gi, gs = vw.calc_disparity_sgm(CENSUS, l2, r2, box, (12, sy), (k, k), subpixel_mode=5, with_subpixel=True, p1=p1, p2=p2)
    oi, os_ = oracle.calc_disparity_sgm(CENSUS, l2, r2, (12, sy), k, subpixel=5, p1=p1, p2=p2)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5
    oh, ow = h - k + 1, w - k + 1
    lm = np.full((oh, ow), 255, np.uint8); lm[:3] = 0; lm[:, -4:] = 0
    rm = np.full((oh + sy, ow + 12), 255, np.uint8); rm[:, :5] = 0
    prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
    prev[..., 0] = rng.integers(0, 7, prev.shape[:2]); prev[..., 1] = rng.integers(0, 2, prev.shape[:2])
    prev[..., 2] = np.where(rng.random(prev.shape[:2]) < 0.2, 0, np.iinfo(np.int32).max)
    gi, gs, oi, os_ = None, None, None, None
    gi, gs = vw.calc_disparity_sgm(CENSUS, l2, r2, box, (12, sy), (k, k), subpixel_mode=5, with_subpixel=True, p1=p1, p2=p2,
                                   left_mask=lm, right_mask=rm, prev_disparity=prev)
    oi, os_ = oracle.calc_disparity_sgm(CENSUS, l2, r2, (12, sy), k, subpixel=5, p1=p1, p2=p2, left_mask=lm, right_mask=rm, prev_disparity=prev)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("mode", [16, 128, 64, 0])
@pytest.mark.parametrize("p1,p2,sx,sy,w,h,bad", [(0, 0, 40, 2, 150, 97, 0.03), (0, 0, 127, 2, 90, 140, 0.2), (1, 65000, 24, 1, 77, 50, 0.05),
                                                  (0, 0, 30, 0, 200, 33, 0.0), (0, 0, 70, 3, 64, 64, 0.5)])
def test_several_lines_per_wavefront_on_ragged_boxes(vw, oracle, mode, p1, p2, sx, sy, w, h, bad):
    """Ragged boxes from a previous level through path_multi_kernel (SGM_PATH_MODE bit 4: four lines per wavefront, bit 7: two, bit 6: the
    one-line kernel, 0: chosen by the level): boxes of up to 16 / 32 cells in a group's own lanes, pixels without a trusted coarser disparity
    (whole range: more cells than a wavefront has lanes) on all lanes, patches of them, lines of unequal length in one wavefront, a number of
    lines that does not fill the last wavefront, penalties that separate the directions (plain u16 sums instead of atomics)."""
    from visionworkbench_amd import core
    ctx = core.default_context(0)
    k = 5
    rng = np.random.default_rng(sx * 7 + sy + w)
    shift = sx // 2
    base = rng.integers(0, 256, (h + sy + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4 + shift:4 + shift + w])
    right = np.ascontiguousarray(base[4:4 + h + sy, 4:4 + w + sx])
    oh, ow = h - k + 1, w - k + 1
    prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
    prev[..., 0] = shift // 2 + rng.integers(-1, 2, prev.shape[:2]); prev[..., 1] = rng.integers(0, sy // 2 + 1, prev.shape[:2])
    prev[..., 2] = np.where(rng.random(prev.shape[:2]) < bad, 0, np.iinfo(np.int32).max)
    prev[prev.shape[0] // 3:prev.shape[0] // 3 + 9, 5:25, 2] = 0                 # a patch of untrusted pixels
    ctx.set_option(core.OPT_SGM_PATH_MODE, mode)
    try:
        gi, gs = vw.calc_disparity_sgm(CENSUS, left, right, _box(w, h), (sx, sy), (k, k), subpixel_mode=5, with_subpixel=True, p1=p1, p2=p2,
                                       prev_disparity=prev, ctx=ctx)
    finally:
        ctx.set_option(core.OPT_SGM_PATH_MODE, 0)
    oi, os_ = oracle.calc_disparity_sgm(CENSUS, left, right, (sx, sy), k, subpixel=5, p1=p1, p2=p2, prev_disparity=prev)
    assert np.array_equal(gi, oi), int((gi != oi).any(-1).sum())
    assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("bad,patch", [(0.001, False), (0.01, True), (0.3, False)])
def test_ragged_level_large_enough_for_the_chosen_form(vw, oracle, bad, patch):
    """A level of 6 000 scan lines and more, default options: the library picks four, two or one scan line per wavefront from the level's
    own boxes (clean coarser disparities: four; patches of untrusted pixels: two; many untrusted pixels: one) — the same image each time."""
    k, sx, sy, w, h = 5, 64, 2, 724, 704
    rng = np.random.default_rng(int(bad * 1000) + patch)
    shift = 30
    base = rng.integers(0, 256, (h + sy + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4 + shift:4 + shift + w])
    right = np.ascontiguousarray(base[4:4 + h + sy, 4:4 + w + sx])
    oh, ow = h - k + 1, w - k + 1
    prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
    prev[..., 0] = shift // 2 + rng.integers(0, 2, prev.shape[:2]); prev[..., 1] = rng.integers(0, 2, prev.shape[:2])
    prev[..., 2] = np.where(rng.random(prev.shape[:2]) < bad, 0, np.iinfo(np.int32).max)
    if patch:
        prev[40:90, 100:140, 2] = 0; prev[200:215, 10:300, 2] = 0
    gi, gs = vw.calc_disparity_sgm(CENSUS, left, right, _box(w, h), (sx, sy), (k, k), subpixel_mode=5, with_subpixel=True, prev_disparity=prev)
    oi, os_ = oracle.calc_disparity_sgm(CENSUS, left, right, (sx, sy), k, subpixel=5, prev_disparity=prev)
    assert np.array_equal(gi, oi), int((gi != oi).any(-1).sum())
    assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("sx,sy,k,w,h", [(4, 0, 7, 40, 30), (8, 0, 7, 60, 48), (3, 1, 5, 20, 16), (16, 0, 7, 100, 80)])
def test_uniform_path_detected_from_the_boxes(vw, oracle, sx, sy, k, w, h):
    """All-valid masks (the top level of every pyramid): the boxes come out full everywhere and the uniform kernel runs;
    fewer than 16 disparities exercises the one-quantum-per-pixel stride."""
    rng = np.random.default_rng(7 * sx + sy)
    base = rng.integers(0, 256, (h + sy + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[3:3 + h + sy, 1:1 + w + sx])
    oh, ow = h - k + 1, w - k + 1
    lm = np.full((oh, ow), 255, np.uint8)
    rm = np.full((oh + sy, ow + sx), 255, np.uint8)
    gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (sx, sy), k, 5, lm=lm, rm=rm)
    assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5


def test_pyramid_sgm_large_tile(vw, oracle):
    """A 640x512 tile of a 1400^2 pair, +-64 x +-1 search, 5 levels: 387 disparities at level 0 on ragged boxes."""
    from visionworkbench_amd import synth
    from visionworkbench_amd.core import BBox2i
    L, R, _ = synth.stereo_pair(1400, 1400, 129, 1)
    R = R[:, 64:64 + 1400].copy()
    bb = (256, 192, 640, 512)
    g = vw.pyramid_correlate(L, R, None, None, 0, 0.0, BBox2i.from_corners((-64, -1), (64, 1)), (7, 7), CENSUS, consistency_threshold=2,
                             filter_half_kernel=5, max_pyramid_levels=5, algorithm=1, bbox=BBox2i(*bb))
    o = oracle.pyramid_correlate_sgm(L, R, None, None, (-64, -1, 64, 1), 7, CENSUS, 2, 0, 5, 5, bbox=bb)
    assert np.array_equal(g[..., 2], o[..., 2]) and np.abs(g[..., :2] - o[..., :2]).max() < 1e-5


# ---- MGM (use_mgm; accum_mgm_multithread, SGM.cc:2619-2700) ----------------------------------------------------------------------
# The reference's tests hold no MGM vector (TestSGM.cxx:47 runs use_mgm = false), so the oracle's accumulate_mgm is a restatement
# that nothing pins: these tests show that the GPU fronts and the oracle's raster loops compute the same function.

@pytest.mark.parametrize("cost,k,sx,sy,w,h", [(CENSUS, 5, 9, 0, 83, 60), (CENSUS, 3, 6, 4, 64, 48), (TERNARY, 7, 12, 2, 90, 41), (CENSUS, 9, 40, 0, 120, 33),
                                              (CENSUS, 5, 129, 0, 200, 24), (CENSUS, 5, 3, 3, 12, 70), (CENSUS, 3, 2, 0, 3, 3), (CENSUS, 3, 2, 1, 40, 3),
                                              (CENSUS, 5, 62, 60, 44, 30), (CENSUS, 3, 300, 0, 330, 12)])
def test_mgm_identical_to_oracle(vw, oracle, cost, k, sx, sy, w, h):
    """Every direction's front order (anti-diagonals, rows, columns), wide / tall / one-pixel outputs, 1-D and 2-D searches; 63 x 61 = 3843
    disparities (one wavefront per workgroup, vectors longer than the prefetched 128 elements) and 301 on one row (4 pairs per lane)."""
    rng = np.random.default_rng(17 * sx + sy + k)
    left, right = _pair(rng, h, w, sx, sy, shift=(min(3, sx), min(2, sy)), smooth=(k == 5))
    gi, gs, oi, os_ = _both(vw, oracle, cost, left, right, (sx, sy), k, 5, mgm=True)
    assert gi.shape == oi.shape == (h - k + 1, w - k + 1, 3)
    assert np.array_equal(gi, oi)
    assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("opt", [0, 1, 2, 5])
@pytest.mark.parametrize("k,sx,w,h,cost", [(7, 128, 150, 40, CENSUS), (5, 16, 40, 130, CENSUS), (3, 60, 300, 25, TERNARY), (9, 200, 64, 64, CENSUS),
                                           (5, 250, 90, 31, CENSUS), (7, 9, 21, 200, TERNARY), (5, 30, 10, 10, CENSUS)])
def test_mgm_sweeps_and_fronts_identical_to_oracle(vw, oracle, opt, k, sx, w, h, cost):
    """use_mgm on full-range one-row searches: the eight passes as four concurrent sweeps in the frames (x, y), (-x, -y), (y, -x), (-y, x)
    (VWGPU_OPT_MGM_SWEEP 0; 2 / 5 = pinned lines per workgroup: hand-offs through HBM every few lines, and the ticket deal when the row
    and column sweeps have different block counts) and as one launch per front (1): wide, tall and tiny outputs, 1 .. 8 pairs per lane."""
    from visionworkbench_amd import core
    ctx = core.default_context(0)
    rng = np.random.default_rng(11 * k + sx + opt)
    left = np.floor(rng.random((h, w)) * 256).astype(np.float32)
    right = np.floor(rng.random((h, w + sx)) * 256).astype(np.float32)
    right[:, sx // 3:sx // 3 + w] = left
    ctx.set_option(core.OPT_MGM_SWEEP, opt)
    try:
        gi, gs = vw.calc_disparity_sgm(cost, left, right, _box(w, h), (sx, 0), (k, k), use_mgm=True, with_subpixel=True, ctx=ctx)
    finally:
        ctx.set_option(core.OPT_MGM_SWEEP, 0)
    oi, os_ = oracle.calc_disparity_sgm(cost, left, right, (sx, 0), k, use_mgm=True)
    assert np.array_equal(gi, oi), int((gi != oi).any(-1).sum())
    assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("seed", range(4))
def test_mgm_ragged_boxes_masks_and_previous_level(vw, oracle, seed):
    """Masks with empty borders and a previous level with untrusted holes: pixels without disparities are skipped but still serve as
    (empty) predecessors; ragged vectors on every front.  mem = 1 forces the MGM small-buffer term through the conservation levels."""
    rng = np.random.default_rng(300 + seed)
    V = np.iinfo(np.int32).max
    k = int(rng.choice([3, 5, 7]))
    sx, sy = int(rng.integers(2, 14)), int(rng.integers(0, 5))
    h, w = int(rng.integers(36, 60)), int(rng.integers(40, 80))
    base = rng.integers(0, 256, (h + sy + 8, w + sx + 8)).astype(np.float32)
    left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
    right = np.ascontiguousarray(base[4 - min(2, sy):4 - min(2, sy) + h + sy, 1:1 + w + sx])
    oh, ow = h - k + 1, w - k + 1
    lm = np.full((oh, ow), 255, np.uint8)
    lm[:int(rng.integers(0, 4))] = 0
    lm[:, :int(rng.integers(0, 4))] = 0
    lm[10:14, 12:30] = 0
    rm = np.full((oh + sy + 2, ow + sx + 1), 255, np.uint8)
    rm[:, :int(rng.integers(0, sx + 2))] = 0
    rm[-int(rng.integers(1, 4)):] = 0
    ph, pw = (oh + 1) // 2 + int(rng.integers(-2, 3)), (ow + 1) // 2 + int(rng.integers(-2, 3))
    prev = np.zeros((ph, pw, 3), np.int32)
    prev[..., 0] = rng.integers(0, sx // 2 + 1, (ph, pw))
    prev[..., 1] = rng.integers(0, sy // 2 + 1, (ph, pw))
    prev[..., 2] = np.where(rng.random((ph, pw)) < 0.3, 0, V)
    for mem in (6000, 1):
        gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, right, (sx, sy), k, 5, lm=lm, rm=rm, prev=prev, mgm=True, mem=mem)
        assert np.array_equal(gi, oi), mem
        assert np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("p1,p2", [(5, 60), (200, 9000), (1, 65000)])
def test_mgm_user_penalties(vw, oracle, p1, p2):
    """P2 > 7937: the eight directions can no longer share packed 16-bit atomics and run one at a time, the u16 sums wrap like the reference's."""
    rng = np.random.default_rng(p2 + 1)
    sx, sy, k, h, w = 20, 1, 5, 30, 70
    left, right = _pair(rng, h, w, sx, sy)
    noise = rng.integers(0, 256, right.shape).astype(np.float32)
    for rr in (right, noise):
        gi, gs, oi, os_ = _both(vw, oracle, CENSUS, left, rr, (sx, sy), k, 5, mgm=True, p1=p1, p2=p2)
        assert np.array_equal(gi, oi) and np.abs(gs - os_).max() < 1e-5


@pytest.mark.parametrize("algorithm", [2, 3])
def test_pyramid_mgm_and_final_mgm(vw, oracle, algorithm):
    """VW_CORRELATION_MGM: every level with use_mgm; VW_CORRELATION_FINAL_MGM: level 0 only (CorrelationView.cc:365-366)."""
    from visionworkbench_amd.core import BBox2i
    rng = np.random.default_rng(40 + algorithm)
    base = rng.integers(0, 256, (260, 340)).astype(np.float32)
    k = np.ones(3) / 3
    base = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, base).astype(np.float32)
    left = np.ascontiguousarray(base[10:210, 20:300])
    right = np.ascontiguousarray(base[8:208, 14:294])
    search = (-8, -4, 8, 4)
    box = BBox2i.from_corners(search[:2], search[2:])
    g = vw.pyramid_correlate(left, right, None, None, 0, 0.0, box, (5, 5), CENSUS, consistency_threshold=2, min_consistency_level=0,
                             filter_half_kernel=3, max_pyramid_levels=2, algorithm=algorithm, bbox=BBox2i(16, 8, 220, 160))
    o = oracle.pyramid_correlate_sgm(left, right, None, None, search, 5, CENSUS, 2, 0, 3, 2, bbox=(16, 8, 220, 160), algorithm=algorithm)
    assert np.array_equal(g[..., 2], o[..., 2]) and np.abs(g[..., :2] - o[..., :2]).max() < 1e-5
