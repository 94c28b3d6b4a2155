"""The reference's summation order on the GPU (csrc/bm_exact.hip): fast_box_sum is two families of serial float64
recurrences (src/vw/Stereo/Algorithms.h:62-110); when partial sums are not exactly representable the roundings depend on
the raster position and decide exact cost ties.  These tests feed inputs of that class — wide dynamic range floats, values
next to zero, NaNs — and ask for results IDENTICAL to the oracle's literal restatement."""
import numpy as np
import pytest

import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo

pytestmark = pytest.mark.gpu

ABS, SQ, NCC = 0, 1, 2


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    c = vwa.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["split", "fused", "tiled", "certified"])
def _default_options(request, ctx):
    """Every test of this module runs with all three forms of pass 2 (OPT_EXACT_SPLIT: the recurrence alone + a parallel selection, the
    fused kernel, and the tiled kernel that transposes a row's sums through LDS) with VWGPU_OPT_CERTIFY off — every call on rounding data
    goes to the exact-order kernels — and once more with the default dispatch: the certified tile-parallel pass first
    (VWGPU_PATH_CERTIFIED), the exact-order kernels only for calls with an unproven pixel.  Variant-pinning options (same results,
    other kernels / schedules) never leak from one test into the next."""
    ctx.set_option(core.OPT_EXACT_SPLIT, {"split": 1, "fused": 2, "tiled": 3, "certified": 0}[request.param])
    ctx.set_option(core.OPT_CERTIFY, 1 if request.param == "certified" else 0)
    yield
    ctx.set_option(core.OPT_CERTIFY, 1)
    ctx.set_option(core.OPT_EXACT_SPLIT, 0)
    ctx.set_option(core.OPT_SAD_GROUPS, 0)
    ctx.set_option(core.OPT_EXACT_SCRATCH_MB, 4096)


def wide_range(rng, h, w, decades=14, signed=True):
    """float32 texture whose magnitudes span `decades` powers of ten: box sums of such data round all the time."""
    v = rng.random((h, w)) * 10.0 ** (rng.random((h, w)) * decades - decades / 2)
    if signed:
        v *= rng.choice([-1.0, 1.0], (h, w))
    return v.astype(np.float32)


def test_fast_box_sum_golden_and_order(ctx, oracle):
    """TestAlgorithms.cxx:46-174 known answers, then float data on which the order of the running sums matters."""
    ramp = np.arange(1, 36, dtype=np.float32).reshape(5, 7)
    got = stereo.fast_box_sum(ramp, (5, 3), ctx=ctx)
    assert got.shape == (3, 3)
    assert np.array_equal(got[0], [150, 165, 180]) and np.array_equal(got[1], [255, 270, 285]) and np.array_equal(got[2], [360, 375, 390])
    assert np.array_equal(stereo.fast_box_sum(ramp, (1, 1), ctx=ctx), ramp.astype(np.float64))
    assert stereo.fast_box_sum(ramp, (7, 5), ctx=ctx).item() == 630.0
    rng = np.random.default_rng(5)
    for (h, w, k) in [(40, 53, (7, 7)), (64, 300, (3, 9)), (9, 9, (9, 9)), (130, 70, (11, 5)), (33, 1000, (15, 15))]:
        img = wide_range(rng, h, w)
        want = oracle.fast_box_sum(img, k)
        got = stereo.fast_box_sum(img, k, ctx=ctx)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (h, w, k)
        # direct window sums differ on this data: the test would not notice an order-free implementation otherwise
    import torch
    img = wide_range(rng, 90, 200)
    got = stereo.fast_box_sum(torch.from_numpy(img).cuda(), (5, 5), ctx=ctx).cpu().numpy()
    assert np.array_equal(got.view(np.uint64), oracle.fast_box_sum(img, (5, 5)).view(np.uint64))
    direct = np.lib.stride_tricks.sliding_window_view(img.astype(np.float64), (5, 5)).sum((-1, -2))
    assert not np.array_equal(direct, got)
    with pytest.raises(vwa.ArgumentErr):
        stereo.fast_box_sum(ramp, (4, 3), ctx=ctx)          # Algorithms.h:45-46
    with pytest.raises(vwa.ArgumentErr):
        stereo.fast_box_sum(ramp, (9, 3), ctx=ctx)


def test_fast_box_sum_row_bands(ctx, oracle, monkeypatch):
    ctx.set_option(core.OPT_EXACT_SCRATCH_MB, 16)      # 16 MB of column sums = bands of ~1000 rows at 2048 columns
    rng = np.random.default_rng(6)
    img = wide_range(rng, 2300, 2048, decades=10)
    got = stereo.fast_box_sum(img, (7, 7), ctx=ctx)
    assert np.array_equal(got.view(np.uint64), oracle.fast_box_sum(img, (7, 7)).view(np.uint64))


def _pair(rng, h, w, sx, sy, shift, decades):
    left = wide_range(rng, h, w, decades)
    right = wide_range(rng, h + sy - 1, w + sx - 1, decades)
    right[shift[1]:shift[1] + h, shift[0]:shift[0] + w] = np.where(rng.random((h, w)) < 0.8, left, right[shift[1]:shift[1] + h, shift[0]:shift[0] + w])
    return left, right


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
@pytest.mark.parametrize("h,w,kernel,search,shift", [
    (40, 90, (7, 7), (17, 1), (8, 0)),        # one chunk, two rows per wave
    (33, 70, (5, 5), (5, 5), (2, 3)),         # 2-D search
    (50, 300, (11, 11), (129, 1), (64, 0)),   # three 64-disparity chunks per lane
    (24, 64, (3, 9), (1, 1), (0, 0)),         # a single disparity: all invalid unless NaN
    (60, 130, (13, 5), (70, 3), (30, 1)),     # 210 disparities
    (20, 40, (7, 7), (40, 12), (11, 5)),      # 480 disparities (the largest chunk count)
])
def test_calc_disparity_in_reference_order(ctx, oracle, cost, h, w, kernel, search, shift):
    import torch
    rng = np.random.default_rng(h * 1000 + w + cost)
    left, right = _pair(rng, h, w, search[0], search[1], shift, decades=14)
    want = oracle.calc_disparity(cost, left, right, kernel, search)
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    assert ctx.last_path() in (core.PATH_EXACT_ORDER, core.PATH_CERTIFIED)
    assert np.array_equal(got, want), int((got != want).any(-1).sum())
    got_d = stereo.calc_disparity(cost, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), vwa.bounding_box(left), search, kernel, ctx=ctx)
    assert np.array_equal(got_d.cpu().numpy(), want)
    # the tile-parallel float64 kernel sums in another order: on this data it must NOT be what served the call
    ctx.force_path(core.PATH_GENERIC_F64)
    try:
        other = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    finally:
        ctx.force_path(core.PATH_NONE)
    assert other.shape == want.shape


def test_order_free_floats_keep_the_fast_kernels(ctx, oracle):
    """Float textures whose partial sums are all representable (most imagery): the float64 tile kernel, identical result."""
    from visionworkbench_amd import synth
    left = synth.noise_f32(41, 48, 120, 0.0, 1.0)
    right = np.concatenate([synth.noise_f32(42, 48, 8), left, synth.noise_f32(43, 48, 8)], axis=1)
    for cost in (ABS, SQ, NCC):
        got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), (17, 1), (7, 7), ctx=ctx)
        assert ctx.last_path() in (core.PATH_GENERIC_F64, core.PATH_EXACT_ORDER, core.PATH_CERTIFIED)
        assert np.array_equal(got, oracle.calc_disparity(cost, left, right, (7, 7), (17, 1)))
    i16 = np.floor(synth.noise_f32(44, 40, 100, 0.0, 32767.0)).astype(np.float32)
    r16 = np.concatenate([i16[:, 5:], np.floor(synth.noise_f32(45, 40, 21, 0.0, 32767.0))], axis=1).astype(np.float32)
    got = stereo.calc_disparity(ABS, i16, r16, vwa.bounding_box(i16), (17, 1), (7, 7), ctx=ctx)
    assert ctx.last_path() == core.PATH_SAD_U16                   # integers below 2^16: the packed-u16 kernel
    assert np.array_equal(got, oracle.calc_disparity(ABS, i16, r16, (7, 7), (17, 1)))
    got = stereo.calc_disparity(SQ, i16, r16, vwa.bounding_box(i16), (17, 1), (7, 7), ctx=ctx)
    assert ctx.last_path() == core.PATH_GENERIC_F64               # SSD of 16-bit integers: order free, float64 tile kernel
    assert np.array_equal(got, oracle.calc_disparity(SQ, i16, r16, (7, 7), (17, 1)))


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
def test_nan_and_inf_costs_follow_the_compare_chain(ctx, oracle, cost):
    """NaN never wins, makes `worst` NaN and the pixel stays valid (Correlation.cc:91-133); all-zero NCC windows give 1/0."""
    rng = np.random.default_rng(77 + cost)
    left = np.floor(rng.random((40, 80)) * 256).astype(np.float32)
    right = np.floor(rng.random((42, 100)) * 256).astype(np.float32)
    right[1:41, 9:89] = left
    left[10, 20] = np.nan
    right[25, 60] = np.nan
    left[30:40, 50:70] = 0.0
    right[28:42, 45:100] = 0.0
    left[5, 70] = np.inf
    want = oracle.calc_disparity(cost, left, right, (7, 7), (21, 3))
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), (21, 3), (7, 7), ctx=ctx)
    assert np.array_equal(got, want), int((got != want).any(-1).sum())


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
@pytest.mark.parametrize("h,w,kernel,search,shift,log", [
    (70, 90, (7, 7), (129, 5), (64, 2), False),     # 645 disparities: two groups (512 + 133); 27 MB of column sums per group
    (20, 48, (7, 7), (300, 5), (150, 3), False),    # 1500 disparities: three groups
    (16, 40, (5, 5), (513, 1), (200, 0), False),    # a last group of ONE disparity
    (26, 64, (7, 7), (129, 5), (64, 2), True),      # LoG-filtered imagery (values next to zero), the verdict's example
    (22, 50, (9, 9), (300, 5), (100, 1), True),
])
def test_search_volumes_beyond_512_disparities(ctx, oracle, cost, h, w, kernel, search, shift, log):
    """best_of_search_convolution loops over ANY search volume (Correlation.cc:64-66): zones of more than 512 disparities are swept
    in disparity groups with the compare-chain state carried between them — still the reference's order, never the tile-local sums."""
    from visionworkbench_amd import filters
    rng = np.random.default_rng(h * 31 + w + cost)
    if log:
        left = np.floor(rng.random((h, w)) * 256).astype(np.float32)
        right = np.floor(rng.random((h + search[1] - 1, w + search[0] - 1)) * 256).astype(np.float32)
        right[shift[1]:shift[1] + h, shift[0]:shift[0] + w] = left
        left = filters.prefilter_image(left, 1, 1.4, ctx=ctx)           # LaplacianOfGaussian(1.4), PreFilter.h:76-95
        right = filters.prefilter_image(right, 1, 1.4, ctx=ctx)
    else:
        left, right = _pair(rng, h, w, search[0], search[1], shift, decades=14)
    want = oracle.calc_disparity(cost, left, right, kernel, search)
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    # (LoG + SAD of 8-bit imagery is order free — every partial sum fits 53 bits — and may take the tile kernel: same bits)
    assert ctx.last_path() in (core.PATH_EXACT_ORDER, core.PATH_CERTIFIED) or (log and cost == ABS and ctx.last_path() == core.PATH_GENERIC_F64)
    assert np.array_equal(got, want), int((got != want).any(-1).sum())
    ctx.force_path(core.PATH_EXACT_ORDER)
    try:
        got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    finally:
        ctx.force_path(core.PATH_NONE)
    assert np.array_equal(got, want)
    ctx.set_option(core.OPT_EXACT_SCRATCH_MB, 16)                       # row bands inside every disparity group
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
def test_nan_costs_across_disparity_groups(ctx, oracle, cost):
    """The chain state handed from group to group can hold a NaN `worst` (or a NaN `best` when disparity 0 is NaN): the
    next group replays the chain from that state (Correlation.cc:91-117)."""
    rng = np.random.default_rng(500 + cost)
    h, w, search = 18, 44, (260, 3)
    left = (rng.random((h, w)) * 10.0 ** (rng.random((h, w)) * 8 - 4)).astype(np.float32)
    right = (rng.random((h + 2, w + 259)) * 10.0 ** (rng.random((h + 2, w + 259)) * 8 - 4)).astype(np.float32)
    right[1:1 + h, 100:100 + w] = left
    right[0, 3] = np.nan            # NaN cost at the very first disparities of a few pixels (best = NaN forever)
    right[8, 130] = np.nan          # NaN costs in the first group only
    right[12, 290] = np.nan         # NaN costs that straddle the group boundary (index 512 = (dx 252, dy 1))
    right[17, 200] = np.inf
    left[5, 30] = np.nan            # a pixel window with NaN at every disparity
    want = oracle.calc_disparity(cost, left, right, (7, 7), search)
    got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, (7, 7), ctx=ctx)
    assert ctx.last_path() == core.PATH_EXACT_ORDER
    assert np.array_equal(got, want), int((got != want).any(-1).sum())


def test_search_volume_beyond_65535(ctx, oracle):
    """The 16-bit index fields belong to the packed kernels only; the float64 and exact-order paths index with ints."""
    rng = np.random.default_rng(65536)
    h, w, search = 9, 12, (300, 220)                # 66 000 disparities
    left = np.floor(rng.random((h, w)) * 256).astype(np.float32)
    right = np.floor(rng.random((h + 219, w + 299)) * 256).astype(np.float32)
    right[201:201 + h, 280:280 + w] = left
    want = oracle.calc_disparity(ABS, left, right, (3, 3), search)
    got = stereo.calc_disparity(ABS, left, right, vwa.bounding_box(left), search, (3, 3), ctx=ctx)
    assert np.array_equal(got, want)
    assert np.all(got[..., 0] == 280) and np.all(got[..., 1] == 201)
    left2, right2 = left + np.float32(0.1), right + np.float32(0.1)      # not order free: disparity groups
    want = oracle.calc_disparity(SQ, left2, right2, (3, 3), search)
    got = stereo.calc_disparity(SQ, left2, right2, vwa.bounding_box(left2), search, (3, 3), ctx=ctx)
    assert ctx.last_path() in (core.PATH_EXACT_ORDER, core.PATH_CERTIFIED)
    assert np.array_equal(got, want)


def test_table_ring_wraps_while_the_device_is_behind(oracle):
    """The disparity-group loop queues one table upload per 512 disparities and never synchronises: with a 64 KiB pinned ring
    the 40 groups of this call change ring halves a dozen times while earlier uploads are still pending — the ring must wait for
    the copies that read a half before it rewrites it.  Same result as with the default ring, which never wraps here."""
    rng = np.random.default_rng(4242)
    h, w, search = 40, 70, (255, 80)                # 20 400 disparities = 40 groups; NCC: two table sets per group
    left, right = _pair(rng, h, w, search[0], search[1], (101, 33), decades=10)
    want = oracle.calc_disparity(NCC, left, right, (5, 5), search)
    c = vwa.Context(0)
    try:
        c.set_option(core.OPT_HOST_RING_KB, 64)
        c.set_option(core.OPT_CERTIFY, 0)                                # (the exact-order kernels are what queues the tables)
        got = stereo.calc_disparity(NCC, left, right, vwa.bounding_box(left), search, (5, 5), ctx=c)
        assert c.last_path() == core.PATH_EXACT_ORDER
        wraps = c.get_option(core.OPT_HOST_RING_WRAPS)
        assert np.array_equal(got, want), int((got != want).any(-1).sum())
        assert wraps >= 4, wraps
        c.set_option(core.OPT_HOST_RING_KB, 16384)                       # (re-allocates the ring)
        assert np.array_equal(stereo.calc_disparity(NCC, left, right, vwa.bounding_box(left), search, (5, 5), ctx=c), want)
    finally:
        c.close()


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
@pytest.mark.parametrize("h,w,kernel,search,shift", [
    (38, 38, (7, 7), (5, 5), (2, 3)),          # the typical level-0 zone of a pyramid tile: 32 x 32 outputs, 5 x 5 disparities
    (20, 58, (7, 7), (1, 1), (0, 0)),          # 52 output columns: cw = 58 chains in pass 1, one disparity
    (64, 30, (11, 11), (3, 4), (1, 2)),        # 54 output rows
    (25, 33, (3, 9), (27, 8), (13, 4)),        # 216 disparities
    (12, 12, (11, 11), (9, 1), (4, 0)),        # 2 x 2 outputs
    (70, 40, (5, 5), (4, 3), (1, 1)),          # 66 output rows
])
def test_small_zones_in_reference_order(ctx, oracle, cost, h, w, kernel, search, shift):
    """Zone-sized rasters (what a pyramid level hands to calc_disparity) through the exact-order kernels: the reference's summation
    order and compare chain, wide-range data with NaN / Inf pixels and all-zero NCC windows."""
    rng = np.random.default_rng(h * 100 + w + cost)
    left, right = _pair(rng, h, w, search[0], search[1], shift, decades=14)
    left[h // 2, w // 3] = np.nan
    right[h // 3, w // 2] = np.inf
    left[:kernel[1], :kernel[0]] = 0.0
    want = oracle.calc_disparity(cost, left, right, kernel, search)
    ctx.force_path(core.PATH_EXACT_ORDER)
    try:
        got = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
    finally:
        ctx.force_path(core.PATH_NONE)
    assert np.array_equal(got, want), int((got != want).any(-1).sum())


def test_whole_raster_in_row_bands(ctx, oracle, monkeypatch):
    ctx.set_option(core.OPT_EXACT_SCRATCH_MB, 16)
    rng = np.random.default_rng(9)
    left, right = _pair(rng, 300, 500, 33, 1, (16, 0), decades=12)
    want = oracle.calc_disparity(SQ, left, right, (7, 7), (33, 1))
    got = stereo.calc_disparity(SQ, left, right, vwa.bounding_box(left), (33, 1), (7, 7), ctx=ctx)
    assert ctx.last_path() in (core.PATH_EXACT_ORDER, core.PATH_CERTIFIED)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
@pytest.mark.parametrize("thr", [-1, 2])
def test_pyramid_on_wide_range_floats(oracle, cost, thr):
    """pyramid_correlate with masks (mean fill), deep levels and data whose sums round at every level."""
    from visionworkbench_amd.core import BBox2i
    rng = np.random.default_rng(31 + cost)
    H, W = 200, 260
    left = wide_range(rng, H, W, decades=9, signed=False)
    right = np.roll(left, 4, axis=1)
    right[:, :4] = wide_range(rng, H, 4, decades=9, signed=False)
    lm = np.full((H, W), 255, np.uint8)
    lm[60:100, 30:90] = 0
    rm = np.full((H, W), 255, np.uint8)
    rm[:, -20:] = 0
    search = (-8, -2, 9, 3)
    g = stereo.pyramid_correlate(left, right, lm, rm, 0, 0.0, BBox2i.from_corners(search[:2], search[2:]), (7, 7), cost, 0, 0.0, thr, 0, 3, 3)
    o = oracle.pyramid_correlate(left, right, lm, rm, 0, 0.0, search, (7, 7), cost, 0, 0.0, thr, 3, 3)
    assert np.array_equal(g, o), int((g != o).any(-1).sum())
