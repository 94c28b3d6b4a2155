"""GPU parity tests of the block-matching hot path, through the C ABI (libvwgpu.so), against the CPU oracle.

Integer disparities and validity must be BIT-EXACT on integer-valued inputs; the golden cases
are the reference's own (src/vw/Stereo/tests/TestCorrelation.cxx, TestCorrelate.cxx)."""
import numpy as np
import pytest

import visionworkbench_amd as vwa
from visionworkbench_amd import core, synth

pytestmark = pytest.mark.gpu

ABS, SQ, NCC = 0, 1, 2


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def ctx(torch_cuda):
    c = vwa.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _default_options(ctx):
    """Variant-pinning options (same results, other kernels / schedules) never leak from one test into the next."""
    yield
    ctx.set_option(core.OPT_SAD_GROUPS, 0)
    ctx.set_option(core.OPT_EXACT_SCRATCH_MB, 4096)


@pytest.fixture(params=["auto", "one-group", "two-groups", "narrow-four-groups"])
def sad_variant(request, ctx):
    """Small images run the packed-u8 matcher with several wave groups per tile (the launcher's choice for grids that do not
    fill the chip: two groups on 1024-column tiles or, round 6, four groups on 512-column tiles of 16 rows); OPT_SAD_GROUPS pins the
    one-group kernel the full-size case uses and either split flavour, so all of them see every case."""
    ctx.set_option(core.OPT_SAD_GROUPS, {"auto": 0, "one-group": 1, "two-groups": 2, "narrow-four-groups": 3}[request.param])
    yield request.param
    ctx.set_option(core.OPT_SAD_GROUPS, 0)


def _gpu(ctx, cost, left, right, kernel, search, path=core.PATH_NONE, device=True):
    from visionworkbench_amd import stereo
    import torch
    ctx.force_path(path)
    try:
        if device:
            l = torch.from_numpy(left).cuda()
            r = torch.from_numpy(right).cuda()
            out = stereo.calc_disparity(cost, l, r, vwa.bounding_box(left), search, kernel, ctx=ctx)
            torch.cuda.synchronize()
            ctx.synchronize()
            return out.cpu().numpy(), ctx.last_path()
        out = stereo.calc_disparity(cost, left, right, vwa.bounding_box(left), search, kernel, ctx=ctx)
        return out, ctx.last_path()
    finally:
        ctx.force_path(core.PATH_NONE)


def _correlation_fixture(scale):
    """SetUp of TestCorrelation.cxx:45-53 (see tests/test_oracle_golden.py)."""
    u = (synth.splitmix64(10, 25 * 25) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    left = u.reshape(25, 25) * scale
    if scale > 1:
        left = np.floor(left)
    left = left.astype(np.float32)
    ys = np.clip(np.arange(46) - 8, 0, 24)
    xs = np.clip(np.arange(31) - 3, 0, 24)
    return left, np.ascontiguousarray(left[np.ix_(ys, xs)])


@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
@pytest.mark.parametrize("scale", [255.0, 32767.0, 1.0])
@pytest.mark.parametrize("device", [True, False])
def test_golden_test_correlation(ctx, oracle, cost, scale, device):
    """TestCorrelation.cxx:73-214: 19x21 output, all valid, all == (3,8)."""
    left, right = _correlation_fixture(scale)
    d, _ = _gpu(ctx, cost, left, right, (7, 5), (7, 12), device=device)
    assert d.shape == (21, 19, 3)
    assert (d[..., 2] == core.VALID_I32).all()
    assert (d[..., 0] == 3).all() and (d[..., 1] == 8).all()
    if scale > 1:   # integer-valued: bit-exact vs the oracle as well
        assert np.array_equal(d, oracle.calc_disparity(cost, left, right, (7, 5), (7, 12)))


CASES = [
    # (w, h, kernel, search)
    (96, 40, (7, 7), (33, 1)),
    (300, 70, (5, 5), (33, 1)),
    (257, 33, (7, 7), (129, 1)),
    (130, 50, (3, 3), (7, 5)),
    (200, 45, (9, 9), (18, 3)),
    (190, 37, (11, 11), (21, 2)),
    (64, 30, (7, 5), (1, 4)),
    (33, 21, (13, 15), (6, 2)),      # no packed path -> generic
    (1100, 20, (7, 7), (10, 1)),     # wider than one workgroup tile
]


@pytest.mark.parametrize("w,h,kernel,search", CASES)
@pytest.mark.parametrize("cost", [ABS, SQ, NCC])
def test_parity_vs_oracle_integer_inputs(ctx, oracle, w, h, kernel, search, cost):
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32, seeds=(21, 22, 23))
    want = oracle.calc_disparity(cost, left, right, kernel, search)
    got, _ = _gpu(ctx, cost, left, right, kernel, search)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "mismatching pixels: %d" % int((got != want).any(-1).sum())


@pytest.mark.parametrize("w,h,kernel,search", [c for c in CASES if c[2] != (13, 15)])
def test_packed_u8_path_is_used_and_equals_generic(ctx, oracle, sad_variant, w, h, kernel, search):
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32, seeds=(31, 32, 33), smooth=True)
    fast, p_fast = _gpu(ctx, ABS, left, right, kernel, search, path=core.PATH_SAD_U8)
    gen, p_gen = _gpu(ctx, ABS, left, right, kernel, search, path=core.PATH_GENERIC_F64)
    assert p_fast == core.PATH_SAD_U8 and p_gen == core.PATH_GENERIC_F64
    want = oracle.calc_disparity(ABS, left, right, kernel, search)
    assert np.array_equal(gen, want)
    assert np.array_equal(fast, want), "mismatching pixels: %d" % int((fast != want).any(-1).sum())


def test_ties_and_flat_regions(ctx, oracle, sad_variant):
    """Many exact ties and constant areas: first-wins tie-breaking and best==worst invalidation."""
    rng = np.random.RandomState(5)
    left = (rng.randint(0, 3, (60, 140)) * 100).astype(np.float32)
    right = (rng.randint(0, 3, (62, 172)) * 100).astype(np.float32)
    left[10:40, 20:90] = 50.0
    right[5:50, 10:150] = 50.0
    left[45:, :] = 255.0
    right[45:, :] = 0.0
    for kernel, search in [((7, 7), (33, 3)), ((5, 5), (16, 1)), ((3, 3), (33, 2))]:
        want = oracle.calc_disparity(ABS, left, right, kernel, search)
        assert (want[..., 2] == 0).any() and (want[..., 2] != 0).any()
        got, path = _gpu(ctx, ABS, left, right, kernel, search)
        assert path == core.PATH_SAD_U8
        assert np.array_equal(got, want)
        for cost in (SQ, NCC):
            assert np.array_equal(_gpu(ctx, cost, left, right, kernel, search)[0],
                                  oracle.calc_disparity(cost, left, right, kernel, search))


@pytest.mark.parametrize("sx", [2, 4, 8, 13])
def test_flat_images_with_a_narrow_search(ctx, oracle, sad_variant, sx):
    """Round-6 campaign (seed 6161): a constant image with a search of 2 .. 8 disparities.  The validity sweep reads the RIGHT base tile after
    the wave-group merge and the epilogue have borrowed the entry array; on the 512-column four-group tile a narrow search made that array
    smaller than what they borrow, and the pixels came out valid.  Wide enough for several tiles, a textured stripe for contrast."""
    rng = np.random.RandomState(17 + sx)
    left = np.full((70, 1300), 90.0, np.float32)
    right = np.full((70, 1300 + sx - 1), 90.0, np.float32)
    left[30:45, 500:900] = rng.randint(0, 256, (15, 400)).astype(np.float32)
    right[30:45, 500 + sx // 2:900 + sx // 2] = left[30:45, 500:900]
    want = oracle.calc_disparity(ABS, left, right, (7, 7), (sx, 1))
    assert (want[..., 2] == 0).mean() > 0.5 and (want[..., 2] != 0).any()
    got, path = _gpu(ctx, ABS, left, right, (7, 7), (sx, 1))
    assert path == core.PATH_SAD_U8
    assert np.array_equal(got, want), int((got != want).any(-1).sum())


def test_search_volume_one_all_invalid(ctx, oracle):
    left, right, _ = synth.stereo_pair(80, 30, 1, 1)
    for cost in (ABS, SQ, NCC):
        got, _ = _gpu(ctx, cost, left, right, (7, 7), (1, 1))
        assert (got[..., 2] == 0).all() and (got[..., :2] == 0).all()
        assert np.array_equal(got, oracle.calc_disparity(cost, left, right, (7, 7), (1, 1)))


def test_minimum_size_image(ctx, oracle):
    """Kernel as large as the region: a single output pixel."""
    left, right, _ = synth.stereo_pair(7, 7, 9, 2)
    for cost in (ABS, SQ, NCC):
        got, _ = _gpu(ctx, cost, left, right, (7, 7), (9, 2))
        assert got.shape == (1, 1, 3)
        assert np.array_equal(got, oracle.calc_disparity(cost, left, right, (7, 7), (9, 2)))


def test_non_integer_input_falls_back_to_generic(ctx, oracle):
    """Float textures are outside the packed kernels' domain: they must refuse them (device flag) and a float64 kernel
    must take over — the tile kernel when every partial box sum is representable (any order gives the reference's bits),
    the exact-order kernels otherwise (tests/test_exact_order_gpu.py).  Either way the result is the oracle's."""
    left = synth.noise_f32(41, 48, 120, 0.0, 1.0)
    right = np.concatenate([synth.noise_f32(42, 48, 8), left, synth.noise_f32(43, 48, 8)], axis=1)
    got, path = _gpu(ctx, ABS, left, right, (7, 7), (17, 1))
    assert path in (core.PATH_GENERIC_F64, core.PATH_EXACT_ORDER, core.PATH_CERTIFIED)
    want = oracle.calc_disparity(ABS, left, right, (7, 7), (17, 1))
    assert (got[..., 0] == 8).mean() > 0.99
    assert np.array_equal(got, want)
    # 255.5 / negative / NaN also force the fallback
    for bad in (255.5, -1.0, float("nan"), 256.0):
        l2, r2, _ = synth.stereo_pair(64, 24, 9)
        l2[3, 5] = bad
        g2, p = _gpu(ctx, ABS, l2, r2, (5, 5), (9, 1))
        assert p in (core.PATH_GENERIC_F64, core.PATH_EXACT_ORDER, core.PATH_CERTIFIED) or (bad == 256.0 and p == core.PATH_SAD_U16)
        assert np.array_equal(g2, oracle.calc_disparity(ABS, l2, r2, (5, 5), (9, 1)))


def test_deferred_mode_returns_the_reference_bits_or_says_it_returned_none(ctx, oracle):
    """VWGPU_OPT_DEFER_EXACTNESS = 1 (pipelined callers, ONE launch per call, no host round trip): byte imagery is served by the
    packed kernel; anything else — a stray non-integer, float textures, LoG-filtered imagery, pixels over 14 decades (classes whose
    box sums round, src/vw/Stereo/Algorithms.h:43-129) — must never be answered by tile-local sums: vwgpu_last_path() says
    PATH_REFUSED and the same call with the option off returns the oracle's bits."""
    from visionworkbench_amd import filters
    l8, r8, _ = synth.stereo_pair(200, 60, 17)
    rng = np.random.default_rng(77)
    decades = (rng.random((60, 216)).astype(np.float32) * np.float32(10.0) ** rng.integers(-7, 8, (60, 216)).astype(np.float32)).astype(np.float32)
    cases = [("bytes", l8, r8, core.PATH_SAD_U8)]
    l2 = l8.copy(); l2[3, 5] = 0.5
    cases.append(("one half-integer pixel", l2, r8, core.PATH_REFUSED))
    cases.append(("LoG filtered", filters.prefilter_image(l8, 2, 1.4, ctx=ctx), filters.prefilter_image(r8, 2, 1.4, ctx=ctx), core.PATH_REFUSED))
    cases.append(("14 decades", decades[:, :200].copy(), decades, core.PATH_REFUSED))
    cases.append(("16-bit integers", l8 * 200.0, r8 * 200.0, core.PATH_REFUSED))
    for name, left, right, want_path in cases:
        want = oracle.calc_disparity(ABS, left, right, (7, 7), (17, 1))
        ctx.set_option(core.OPT_DEFER_EXACTNESS, 1)
        try:
            got, p = _gpu(ctx, ABS, left, right, (7, 7), (17, 1))
        finally:
            ctx.set_option(core.OPT_DEFER_EXACTNESS, 0)
        assert p == want_path, (name, p)
        if p != core.PATH_REFUSED:
            assert np.array_equal(got, want), name
        got, p = _gpu(ctx, ABS, left, right, (7, 7), (17, 1))          # the repeat the header prescribes
        assert p != core.PATH_REFUSED and np.array_equal(got, want), (name, p)
    # a forced packed path on data outside its domain reports the same way
    ctx.force_path(core.PATH_SAD_U8)
    try:
        _, p = _gpu(ctx, ABS, l2, r8, (7, 7), (17, 1), path=core.PATH_SAD_U8)
    finally:
        ctx.force_path(core.PATH_NONE)
    assert p == core.PATH_REFUSED


def test_trim_gives_the_arenas_back(oracle):
    """vwgpu_trim (memory policy in include/vwgpu.h): the scratch arenas go back to the device and the next call simply allocates again."""
    from visionworkbench_amd import stereo
    left, right, _ = synth.stereo_pair(300, 90, 17)
    c = vwa.Context(0)
    try:
        want = oracle.calc_disparity(NCC, left, right, (7, 7), (17, 1))
        assert np.array_equal(stereo.calc_disparity(NCC, left, right, vwa.bounding_box(left), (17, 1), (7, 7), ctx=c), want)
        freed = c.trim()
        assert freed >= left.nbytes + right.nbytes          # at least the staging copies of the host-pointer entry
        assert c.trim() == 0
        assert np.array_equal(stereo.calc_disparity(NCC, left, right, vwa.bounding_box(left), (17, 1), (7, 7), ctx=c), want)
        assert np.array_equal(stereo.calc_disparity(ABS, left, right, vwa.bounding_box(left), (17, 1), (7, 7), ctx=c),
                              oracle.calc_disparity(ABS, left, right, (7, 7), (17, 1)))
    finally:
        c.close()


def test_strided_region_crop(ctx, oracle, sad_variant):
    """calc_disparity with a left_region inside a larger image (Correlation.cc:356-359 crops)."""
    import torch
    from visionworkbench_amd import stereo
    left, right, _ = synth.stereo_pair(200, 80, 20, 2, block=32)
    region = vwa.BBox2i(13, 9, 150, 50)
    want = oracle.calc_disparity(ABS, left[9:59, 13:163], right[9:59 + 1, 13:163 + 19], (7, 7), (20, 2))
    got = stereo.calc_disparity(ABS, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), region,
                                (20, 2), (7, 7), ctx=ctx)
    assert np.array_equal(got.cpu().numpy(), want)
    got_h = stereo.calc_disparity(ABS, left, right, region, (20, 2), (7, 7), ctx=ctx)
    assert np.array_equal(got_h, want)


def test_argument_errors(ctx):
    """Reference checks (Correlation.cc:341-351, Algorithms.h:45-46) surface as ArgumentErr."""
    from visionworkbench_amd import stereo
    left, right, _ = synth.stereo_pair(40, 30, 9)
    bb = vwa.bounding_box(left)
    with pytest.raises(vwa.ArgumentErr):
        stereo.calc_disparity(ABS, left, right, bb, (9, 1), (4, 5), ctx=ctx)       # even kernel
    with pytest.raises(vwa.ArgumentErr):
        stereo.calc_disparity(ABS, left, right, bb, (0, 1), (5, 5), ctx=ctx)       # empty search
    with pytest.raises(vwa.ArgumentErr):
        stereo.calc_disparity(ABS, left, right, bb, (9, 1), (41, 5), ctx=ctx)      # kernel > region
    with pytest.raises(vwa.ArgumentErr):
        stereo.calc_disparity(ABS, left, right, vwa.BBox2i(0, 0, 41, 30), (9, 1), (5, 5), ctx=ctx)
    with pytest.raises(vwa.NoImplErr):
        stereo.calc_disparity(3, left, right, bb, (9, 1), (5, 5), ctx=ctx)         # census is not a BM cost


def test_lr_check_golden_and_random(ctx, oracle):
    """TestCorrelate.cxx:29-55 + random parity, host and device entry."""
    import torch
    from visionworkbench_amd import stereo
    V = core.VALID_I32
    l2r = np.zeros((3, 3, 3), np.int32)
    r2l = np.zeros((3, 3, 3), np.int32)
    l2r[..., 2] = V
    r2l[..., 2] = V
    l2r[:, 2, 0:2] = 2
    l2r[0, 0, 0:2] = 1
    r2l[1, 1, 0:2] = -1
    l2r[0, 1, 0:2] = 1
    for thr in (0, 2):
        want = oracle.cross_corr_consistency_check(l2r, r2l, thr)
        got = stereo.cross_corr_consistency_check(l2r.copy(), r2l, thr, ctx=ctx)
        assert np.array_equal(got, want)
    rng = np.random.RandomState(9)
    a = rng.randint(-6, 7, (70, 90, 3)).astype(np.int32)
    b = rng.randint(-6, 7, (64, 95, 3)).astype(np.int32)
    a[..., 2] = np.where(rng.rand(70, 90) < 0.8, V, 0)
    b[..., 2] = np.where(rng.rand(64, 95) < 0.8, V, 0)
    for thr in (0.0, 1.0, 2.5):
        want = oracle.cross_corr_consistency_check(a, b, thr)
        got = stereo.cross_corr_consistency_check(torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b).cuda(), thr, ctx=ctx)
        assert np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(vwa.ArgumentErr):
        stereo.cross_corr_consistency_check(a.copy(), b, -1.0, ctx=ctx)


def test_full_size_config2_sampled_parity(ctx, oracle):
    """BASELINE config 2 (4096^2, 7x7 SAD, 129x1) at full size: output tiles are independent units
    (SURVEY §8e), so the oracle is run on sampled padded crops and compared bit-for-bit; plus the
    size-independent property that pixels whose true-shift window survived must reach zero SAD there."""
    import torch
    from visionworkbench_amd import stereo
    W = H = 4096
    left, right, truth = synth.stereo_pair(W, H, 129, 1)
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    got = stereo.calc_disparity(ABS, lt, rt, vwa.bounding_box(left), (129, 1), (7, 7), ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_path() == core.PATH_SAD_U8
    got = got.cpu().numpy()
    assert got.shape == (4090, 4090, 3)
    rng = np.random.RandomState(3)
    spots = [(0, 0), (4090 - 96, 4090 - 48), (1000, 4090 - 48), (4090 - 96, 7)] + \
            [(int(rng.randint(0, 4090 - 96)), int(rng.randint(0, 4090 - 48))) for _ in range(6)]
    for (x, y) in spots:
        tw, th = 96, 48
        want = oracle.calc_disparity(ABS, left[y:y + th + 6, x:x + tw + 6], right[y:y + th + 6, x:x + tw + 6 + 128],
                                     (7, 7), (129, 1))
        assert np.array_equal(got[y:y + th, x:x + tw], want), (x, y)
    # property: where the whole 7x7 window (and its copy in the right image) lies inside one 256-block that was
    # not overwritten, cost(truth) == 0 so the winner is at or before the true shift and valid.
    t = truth[:4090, :4090]
    assert (got[..., 2] == core.VALID_I32).mean() > 0.999
    assert (got[..., 0] == t).mean() > 0.9


@pytest.mark.parametrize("rows,variant", [(1028, None), (517, None), (1028, "0"), (2051, "1"), (517, "2"), (1028, "2"), (261, "2")])
def test_row_strip_sizes_sampled_parity(ctx, oracle, monkeypatch, rows, variant):
    """The strips a 4096^2 pair is cut into on 4 and 8 GPUs (and the tile-height / wave-group variants the launcher picks
    for them: 16-row two-group tiles for 1/4, 8-row two-group tiles for 1/8; OPT_SAD_GROUPS pins the other flavour): full-width
    strip on the GPU, the oracle on sampled padded crops, bit for bit."""
    import torch
    from visionworkbench_amd import stereo
    ctx.set_option(core.OPT_SAD_GROUPS, {None: 0, "0": 1, "1": 2, "2": 3}[variant])
    W = 4096
    left, right, _ = synth.stereo_pair(W, rows, 129, 1)
    lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    got = stereo.calc_disparity(ABS, lt, rt, vwa.bounding_box(left), (129, 1), (7, 7), ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.last_path() == core.PATH_SAD_U8
    got = got.cpu().numpy()
    oh = rows - 6
    assert got.shape == (oh, 4090, 3)
    rng = np.random.RandomState(rows)
    spots = [(0, 0), (4090 - 96, oh - 48), (1000, oh - 48), (4090 - 96, 3), (1020, 5)] + \
            [(int(rng.randint(0, 4090 - 96)), int(rng.randint(0, oh - 48))) for _ in range(5)]
    for (x, y) in spots:
        tw, th = 96, 48
        want = oracle.calc_disparity(ABS, left[y:y + th + 6, x:x + tw + 6], right[y:y + th + 6, x:x + tw + 6 + 128],
                                     (7, 7), (129, 1))
        assert np.array_equal(got[y:y + th, x:x + tw], want), (x, y)
    assert (got[..., 2] == core.VALID_I32).mean() > 0.999


# ---- packed dot-product path (SSD / NCC on integer-valued data) ----------------------------------------------------------

@pytest.mark.parametrize("cost", [1, 2])
@pytest.mark.parametrize("kernel,search", [((3, 3), (5, 1)), ((7, 7), (33, 1)), ((11, 11), (129, 1)), ((5, 9), (64, 1)),
                                           ((13, 5), (17, 1)), ((15, 15), (100, 1)), ((9, 31), (7, 1))])
def test_dot_path_bit_exact(oracle, cost, kernel, search):
    from visionworkbench_amd import core, stereo, synth
    w, h = 203, 77
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32)
    ctx = core.Context(0)
    got = stereo.calc_disparity(cost, left, right, core.BBox2i(0, 0, w, h), search, kernel, ctx=ctx)
    assert ctx.last_path() == core.PATH_DOT_U8
    want = oracle.calc_disparity(cost, left, right, kernel, search)
    assert np.array_equal(got, want)
    ctx.close()


@pytest.mark.parametrize("cost", [1, 2])
def test_dot_path_ties_and_flat_areas(oracle, cost):
    """Constant regions: every disparity ties (all costs equal -> invalid), partial ties pick the first disparity."""
    from visionworkbench_amd import core, stereo
    rng = np.random.default_rng(4)
    left = rng.integers(1, 256, (60, 150)).astype(np.float32)
    right = rng.integers(1, 256, (60, 150 + 32)).astype(np.float32)
    left[10:40, 20:90] = 77.0
    right[5:45, 10:140] = 77.0
    right[:, 100:] = np.tile(right[:, 96:100], (1, 21))[:, :82]           # periodic texture: repeated exact matches
    ctx = core.Context(0)
    got = stereo.calc_disparity(cost, left, right, core.BBox2i(0, 0, 150, 60), (33, 1), (7, 7), ctx=ctx)
    want = oracle.calc_disparity(cost, left, right, (7, 7), (33, 1))
    # NCC: flat / periodic pixels are near ties in fp32 and are evaluated over every disparity in float64; when they are most of
    # the image (as here) the packed kernel hands the whole image to the float64 kernel instead
    assert ctx.last_path() == core.PATH_DOT_U8 or (cost == 2 and ctx.last_path() == core.PATH_GENERIC_F64)
    assert np.array_equal(got, want)
    assert (got[15:30, 25:50, 2] == 0).all()
    if cost == 2:       # a few flat pixels in a textured image stay on the packed path (queued for the float64 evaluation)
        left2 = rng.integers(1, 256, (200, 400)).astype(np.float32)
        right2 = rng.integers(1, 256, (200, 400 + 32)).astype(np.float32)
        right2[:, 9:409] = left2
        left2[50:70, 100:130] = 9.0                 # (a flat RIGHT patch scores higher than any texture: keep it small, every
        right2[45:75, 100:150] = 9.0                #  pixel that can see it becomes a near tie)
        got2 = stereo.calc_disparity(cost, left2, right2, core.BBox2i(0, 0, 400, 200), (33, 1), (7, 7), ctx=ctx)
        assert ctx.last_path() == core.PATH_DOT_U8
        assert np.array_equal(got2, oracle.calc_disparity(cost, left2, right2, (7, 7), (33, 1)))
        assert (got2[55:60, 105:111, 2] == 0).all()        # windows and all 33 right windows inside the flat patches
    ctx.close()


def test_dot_path_falls_back(oracle):
    """Non-integer pixels or an all-zero window (NCC: 1/0) raise the flag and the float64 kernel recomputes the image."""
    from visionworkbench_amd import core, stereo, synth
    left, right, _ = synth.stereo_pair(120, 50, 17, 1, block=32)
    ctx = core.Context(0)
    l2 = left.copy(); l2[20, 30] += 0.5
    got = stereo.calc_disparity(1, l2, right, core.BBox2i(0, 0, 120, 50), (17, 1), (5, 5), ctx=ctx)
    assert ctx.last_path() == core.PATH_GENERIC_F64
    assert np.array_equal(got, oracle.calc_disparity(1, l2, right, (5, 5), (17, 1)))
    r0 = right.copy(); r0[10:30, 40:70] = 0.0
    got = stereo.calc_disparity(2, left, r0, core.BBox2i(0, 0, 120, 50), (17, 1), (5, 5), ctx=ctx)
    assert ctx.last_path() == core.PATH_GENERIC_F64
    assert np.array_equal(got, oracle.calc_disparity(2, left, r0, (5, 5), (17, 1)))
    # SSD has no division: zero windows stay on the fast path
    got = stereo.calc_disparity(1, left, r0, core.BBox2i(0, 0, 120, 50), (17, 1), (5, 5), ctx=ctx)
    assert ctx.last_path() == core.PATH_DOT_U8
    assert np.array_equal(got, oracle.calc_disparity(1, left, r0, (5, 5), (17, 1)))
    # two search rows: not a dot-path shape
    left2, right2, _ = synth.stereo_pair(120, 50, 9, 3, block=32)
    got = stereo.calc_disparity(1, left2, right2, core.BBox2i(0, 0, 120, 50), (9, 3), (5, 5), ctx=ctx)
    assert ctx.last_path() == core.PATH_GENERIC_F64
    assert np.array_equal(got, oracle.calc_disparity(1, left2, right2, (5, 5), (9, 3)))
    ctx.close()


# ---- packed-u16 SAD path (16-bit imagery) ---------------------------------------------------------------------------------

@pytest.mark.parametrize("w,h,kernel,sx", [(300, 70, (7, 7), 129), (97, 40, (5, 5), 9), (640, 33, (11, 11), 64), (258, 50, (3, 3), 1),
                                           (513, 21, (7, 5), 200), (1100, 60, (9, 9), 33), (64, 16, (7, 7), 2)])
def test_u16_path_bit_exact(ctx, oracle, w, h, kernel, sx):
    """Integer-valued pixels above 255 (16-bit sensors; the scale-32767 cases of TestCorrelation.cxx:45-214): v_sad_u16 kernel,
    identical to the oracle — shifted copies, flat patches (validity), values at both ends of the range."""
    rng = np.random.default_rng(w * 7 + sx)
    left = np.floor(rng.random((h, w)) * 65536).astype(np.float32)
    right = np.floor(rng.random((h, w + sx - 1)) * 65536).astype(np.float32)
    d = int(rng.integers(0, sx))
    right[:, d:d + w] = np.where(rng.random((h, w)) < 0.7, left, right[:, d:d + w])
    left[3:12, 10:40] = 65535.0
    right[2:14, 5:60 + sx] = 65535.0
    left[0, 0] = 0.0
    got, path = _gpu(ctx, ABS, left, right, kernel, (sx, 1))
    assert path == core.PATH_SAD_U16
    want = oracle.calc_disparity(ABS, left, right, kernel, (sx, 1))
    assert np.array_equal(got, want), int((got != want).any(-1).sum())
    ctx.force_path(core.PATH_NONE)
    # negative or fractional pixels leave the domain: the float64 kernel serves them, same result as the oracle
    left[5, 7] = -3.0
    got, path = _gpu(ctx, ABS, left, right, kernel, (sx, 1))
    assert path == core.PATH_GENERIC_F64
    assert np.array_equal(got, oracle.calc_disparity(ABS, left, right, kernel, (sx, 1)))


def test_u16_path_random_cross_check(ctx):
    """Seeded random sizes / searches: packed-u16 kernel vs the float64 kernel (pinned to the oracle above)."""
    import torch
    from visionworkbench_amd import stereo
    rng = np.random.default_rng(55)
    kernels = [(3, 3), (5, 5), (7, 7), (7, 5), (9, 9), (11, 11)]
    for it in range(60):
        kx, ky = kernels[rng.integers(len(kernels))]
        sx = int(rng.integers(1, 200))
        w, h = int(rng.integers(kx, 1500)), int(rng.integers(ky, 120))
        scale = float(rng.choice([65536, 4096, 300]))
        left = np.floor(rng.random((h, w)) * scale).astype(np.float32)
        right = np.floor(rng.random((h, w + sx - 1)) * scale).astype(np.float32)
        d = int(rng.integers(0, sx))
        right[:, d:d + w] = np.where(rng.random((h, w)) < 0.6, left, right[:, d:d + w])
        lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        ctx.force_path(core.PATH_SAD_U16)
        try:
            a = stereo.calc_disparity(ABS, lt, rt, vwa.bounding_box(left), (sx, 1), (kx, ky), ctx=ctx).cpu().numpy()
        except vwa.NoImplErr:                                     # search too wide for the kernel's LDS tile
            ctx.force_path(core.PATH_NONE)
            continue
        ctx.force_path(core.PATH_GENERIC_F64)
        b = stereo.calc_disparity(ABS, lt, rt, vwa.bounding_box(left), (sx, 1), (kx, ky), ctx=ctx).cpu().numpy()
        ctx.force_path(core.PATH_NONE)
        assert np.array_equal(a, b), (it, kx, ky, sx, w, h, int((a != b).any(-1).sum()))


# ---- packed SSD / NCC for 11- / 12-bit imagery (v_dot2_u32_u16) --------------------------------------------------------------

@pytest.mark.parametrize("cost", [SQ, NCC])
@pytest.mark.parametrize("w,h,kernel,sx", [(300, 70, (7, 7), 129), (97, 40, (5, 5), 9), (640, 33, (11, 11), 64), (258, 50, (3, 3), 1),
                                           (1100, 60, (9, 9), 33), (64, 16, (7, 7), 2), (520, 45, (11, 11), 129)])
def test_u12_corr_path_bit_exact(ctx, oracle, cost, w, h, kernel, sx):
    """Integer-valued pixels in [0,4095]: float32 products / squared differences are exact, the v_dot2_u32_u16 kernel returns
    the oracle's bits — shifted copies, saturated patches (ties, validity), both ends of the range."""
    rng = np.random.default_rng(w * 11 + sx + cost)
    left = np.floor(rng.random((h, w)) * 4096).astype(np.float32)
    right = np.floor(rng.random((h, w + sx - 1)) * 4096).astype(np.float32)
    d = int(rng.integers(0, sx))
    right[:, d:d + w] = np.where(rng.random((h, w)) < 0.7, left, right[:, d:d + w])
    left[3:12, 10:40] = 4095.0
    right[2:14, 5:60 + sx] = 4095.0
    left[0, 0] = 0.0
    got, path = _gpu(ctx, cost, left, right, kernel, (sx, 1))
    # (one disparity under NCC: best == worst everywhere, every pixel would be queued for the float64 sequence — the float64 kernel takes the image)
    assert path == (core.PATH_GENERIC_F64 if (cost == NCC and sx == 1) else core.PATH_DOT_U16)
    want = oracle.calc_disparity(cost, left, right, kernel, (sx, 1))
    assert np.array_equal(got, want), int((got != want).any(-1).sum())
    # one pixel at 4096: a product may round in float32, the float64 kernel serves the image
    left[5, 7] = 4096.0
    got, path = _gpu(ctx, cost, left, right, kernel, (sx, 1))
    assert path == core.PATH_GENERIC_F64
    assert np.array_equal(got, oracle.calc_disparity(cost, left, right, kernel, (sx, 1)))
    # fractional pixels: not this path either
    left[5, 7] = 100.5
    got, path = _gpu(ctx, cost, left, right, kernel, (sx, 1))
    assert path in (core.PATH_GENERIC_F64, core.PATH_EXACT_ORDER, core.PATH_CERTIFIED)
    assert np.array_equal(got, oracle.calc_disparity(cost, left, right, kernel, (sx, 1)))


def test_u12_corr_zero_windows_and_flat_images(ctx, oracle):
    """An all-zero window under NCC (1/0 in the reference) and images that are mostly flat (every score ties): the packed kernel
    hands the image over, the result is the oracle's either way."""
    rng = np.random.default_rng(4)
    left = np.floor(rng.random((40, 200)) * 4096).astype(np.float32)
    right = np.floor(rng.random((40, 232)) * 4096).astype(np.float32)
    right[10:30, 50:120] = 0.0
    for cost in (SQ, NCC):
        got, path = _gpu(ctx, cost, left, right, (7, 7), (33, 1))
        assert path == (core.PATH_DOT_U16 if cost == SQ else core.PATH_GENERIC_F64)
        assert np.array_equal(got, oracle.calc_disparity(cost, left, right, (7, 7), (33, 1)))
    flat_l = np.full((40, 200), 1000.0, np.float32)
    flat_r = np.full((40, 232), 1000.0, np.float32)
    flat_l[20, 100] = 1001.0
    for cost in (SQ, NCC):
        got, _ = _gpu(ctx, cost, flat_l, flat_r, (7, 7), (33, 1))
        assert np.array_equal(got, oracle.calc_disparity(cost, flat_l, flat_r, (7, 7), (33, 1)))


def test_u12_corr_path_random_cross_check(ctx):
    """Seeded random sizes / searches: packed-u16 SSD / NCC vs the float64 kernel (pinned to the oracle above)."""
    import torch
    from visionworkbench_amd import stereo
    rng = np.random.default_rng(77)
    kernels = [(3, 3), (5, 5), (7, 7), (9, 9), (11, 11)]
    ran = 0
    for it in range(80):
        kx, ky = kernels[rng.integers(len(kernels))]
        cost = int(rng.choice([SQ, NCC]))
        sx = int(rng.integers(1, 200))
        w, h = int(rng.integers(kx, 1500)), int(rng.integers(ky, 120))
        scale = float(rng.choice([4096, 2048, 300, 40]))
        left = np.floor(rng.random((h, w)) * scale).astype(np.float32) + (1.0 if cost == NCC else 0.0)
        right = np.floor(rng.random((h, w + sx - 1)) * scale).astype(np.float32) + (1.0 if cost == NCC else 0.0)
        left, right = np.minimum(left, 4095.0), np.minimum(right, 4095.0)
        d = int(rng.integers(0, sx))
        right[:, d:d + w] = np.where(rng.random((h, w)) < 0.6, left, right[:, d:d + w])
        lt, rt = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        ctx.force_path(core.PATH_DOT_U16)
        try:
            a = stereo.calc_disparity(cost, lt, rt, vwa.bounding_box(left), (sx, 1), (kx, ky), ctx=ctx).cpu().numpy()
            assert ctx.last_path() == core.PATH_DOT_U16
        except vwa.NoImplErr:                                     # search too wide for the kernel's LDS tile
            ctx.force_path(core.PATH_NONE)
            continue
        ctx.force_path(core.PATH_GENERIC_F64)
        b = stereo.calc_disparity(cost, lt, rt, vwa.bounding_box(left), (sx, 1), (kx, ky), ctx=ctx).cpu().numpy()
        ctx.force_path(core.PATH_NONE)
        assert np.array_equal(a, b), (it, cost, kx, ky, sx, w, h, int((a != b).any(-1).sum()))
        ran += 1
    assert ran >= 50
