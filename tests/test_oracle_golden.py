"""Pins the CPU oracle against the reference's own known-answer tests.

Each test is the re-typed body of a reference gtest; the citation is in the docstring.
CPU only (no GPU needed).
"""
import numpy as np
import pytest

from visionworkbench_amd import synth


def test_fast_box_float_ramp(oracle):
    """src/vw/Stereo/tests/TestAlgorithms.cxx:46-72 (FastBoxFloat): 7x5 ramp 1..35, kernels (5,3) and (3,3)."""
    img = np.arange(1, 36, dtype=np.float32).reshape(5, 7)
    o53 = oracle.fast_box_sum(img, (5, 3))
    assert o53.shape == (3, 3)
    assert oracle.fast_box_sum(img, (3, 3)).shape == (3, 5)
    # output(col,row) in VW == o[row, col] here
    assert o53[0, 0] == 150 and o53[0, 1] == 165 and o53[0, 2] == 180
    assert o53[2, 0] == 360 and o53[2, 1] == 375 and o53[2, 2] == 390


def test_fast_box_double(oracle):
    """TestAlgorithms.cxx:100-126 (FastBoxDouble): ones, first row = 2, input(6,0) = 3; kernel (3,3)."""
    img = np.ones((5, 7), np.float64)
    img[0, :] = 2
    img[0, 6] = 3
    o = oracle.fast_box_sum(img, (3, 3))
    assert o.shape == (3, 5)
    assert o[0, 0] == 12 and o[1, 0] == 9 and o[2, 0] == 9
    assert o[0, 1] == 12 and o[0, 2] == 12 and o[0, 3] == 12 and o[0, 4] == 13


def test_fast_box_char_values(oracle):
    """TestAlgorithms.cxx:128-152 (FastBoxChar): 30-filled, first row 40, (6,0)=50; (5,3) kernel; 255 fill (5,5)."""
    img = np.full((5, 7), 30, np.float32)
    img[0, :] = 40
    img[0, 6] = 50
    o = oracle.fast_box_sum(img, (5, 3))
    assert o[2, 0] == 450 and o[2, 2] == 450 and o[0, 0] == 500 and o[0, 2] == 510 and o[1, 1] == 450
    assert oracle.fast_box_sum(np.full((5, 7), 255, np.float32), (5, 5))[0, 0] == 6375


def test_fast_box_pixel_u8_values(oracle):
    """TestAlgorithms.cxx:154-174 (FastBoxPixelU8): 27-filled, first row 40, kernel (5,3)."""
    img = np.full((5, 7), 27, np.float32)
    img[0, :] = 40
    o = oracle.fast_box_sum(img, (5, 3))
    assert (o[0, :] == 470).all() and (o[1, :] == 405).all() and o[2, 0] == 405 and o[2, 2] == 405


def test_fast_box_even_kernel_rejected(oracle):
    """Always-on VW_ASSERT, src/vw/Stereo/Algorithms.h:45-46."""
    with pytest.raises(ValueError):
        oracle.fast_box_sum(np.zeros((5, 7), np.float32), (4, 3))


def _cost_fixture():
    """SetUp of TestCostFunctions.cxx:40-50 for PixelGray<float> (ChannelRange max = 1.0)."""
    a = np.zeros((2, 2), np.float32)
    b = np.zeros((2, 2), np.float32)
    a[0, 0] = b[0, 0] = 128          # input(0,0)
    a[0, 1] = b[1, 0] = 10           # input1(1,0) = input2(0,1) = 10   [row, col] = (y, x)
    b[0, 1] = a[1, 0] = 100          # input2(1,0) = input1(0,1) = 100
    a[1, 1] = 0
    b[1, 1] = 1.0
    return a, b


def test_cost_functions_per_pixel(oracle):
    """TestCostFunctions.cxx:54-84: AbsDiff 0,90,90,max; SquaredDiff 0,8100,8100,max^2; CrossCorr 16384,1000,1000,0."""
    a, b = _cost_fixture()
    r = oracle.cost_image(oracle.ABSOLUTE_DIFFERENCE, a, b)
    assert r[0, 0] == 0 and r[1, 0] == 90 and r[0, 1] == 90 and r[1, 1] == 1.0
    r = oracle.cost_image(oracle.SQUARED_DIFFERENCE, a, b)
    assert r[0, 0] == 0 and r[1, 0] == 8100 and r[0, 1] == 8100 and r[1, 1] == 1.0
    r = oracle.cost_image(oracle.CROSS_CORRELATION, a, b)
    assert r[0, 0] == 16384 and r[1, 0] == 1000 and r[0, 1] == 1000 and r[1, 1] == 0


def _correlation_fixture(dtype_scale):
    """SetUp of TestCorrelation.cxx:45-53: 25x25 noise; right = crop(edge_extend(left, Constant), -3, -8, 31, 46),
    i.e. R(x, y) = L_clamped(x-3, y-8); kernel (7,5); search (7,12); solution (3,8).  Any PRNG will do."""
    u = (synth.splitmix64(10, 25 * 25) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    left = (u.reshape(25, 25) * dtype_scale)
    if dtype_scale > 1:
        left = np.floor(left)
    left = left.astype(np.float32)
    ys = np.clip(np.arange(46) - 8, 0, 24)
    xs = np.clip(np.arange(31) - 3, 0, 24)
    right = left[np.ix_(ys, xs)]
    return left, right


@pytest.mark.parametrize("cost", [0, 1, 2])
@pytest.mark.parametrize("scale", [255.0, 32767.0, 1.0])   # u8-, i16- and float-like fixtures
def test_calc_disparity_recovers_shift(oracle, cost, scale):
    """TestCorrelation.cxx:73-214: output 19x21, every pixel valid and == (3,8), for ABS/SQ/NCC."""
    left, right = _correlation_fixture(scale)
    d = oracle.calc_disparity(cost, left, right, (7, 5), (7, 12))
    assert d.shape == (21, 19, 3)
    assert (d[..., 2] == oracle.VALID).all()
    assert (d[..., 0] == 3).all() and (d[..., 1] == 8).all()


def test_search_volume_one_is_always_invalid(oracle):
    """best == worst for a single disparity (Correlation.cc:110-117,121-133)."""
    left, right = _correlation_fixture(255.0)
    d = oracle.calc_disparity(0, left, right, (7, 5), (1, 1))
    assert (d[..., 2] == 0).all() and (d[..., :2] == 0).all()


def test_cross_corr_consistency(oracle):
    """TestCorrelate.cxx:29-55 (CrossCorrConsistency), thresholds 0 and 2."""
    V = oracle.VALID
    l2r = np.zeros((3, 3, 3), np.int32)
    r2l = np.zeros((3, 3, 3), np.int32)
    l2r[..., 2] = V
    r2l[..., 2] = V
    l2r[:, 2, 0:2] = 2            # crop(l2r,2,0,1,3) = (2,2)
    l2r[0, 0, 0:2] = 1            # l2r(0,0) = (1,1)
    r2l[1, 1, 0:2] = -1           # r2l(1,1) = (-1,-1)
    l2r[0, 1, 0:2] = 1            # l2r(1,0) = (1,1)

    o = oracle.cross_corr_consistency_check(l2r, r2l, 0)
    assert o[0, 2, 2] == 0 and o[1, 2, 2] == 0 and o[2, 2, 2] == 0
    assert o[0, 0, 2] == V and o[0, 1, 2] == 0

    o = oracle.cross_corr_consistency_check(l2r, r2l, 2)
    assert o[0, 2, 2] == 0 and o[1, 2, 2] == 0 and o[2, 2, 2] == 0
    assert o[0, 0, 2] == V and o[0, 1, 2] == V


def test_tiled_equals_whole(oracle):
    """Tile-threaded execution (ImageIO.h:228-251) must equal the single call — the same property
    src/vw/Image/tests/TestBlockRasterize.cxx:26-47 checks for block rasterisation."""
    left, right, _ = synth.stereo_pair(150, 90, 9, 2, block=32)
    whole = oracle.calc_disparity(0, left, right, (5, 5), (9, 2))
    tiled, done = oracle.calc_disparity_tiled(0, left, right, (5, 5), (9, 2), tile=64, threads=3)
    assert done == whole.shape[0] * whole.shape[1]
    assert np.array_equal(whole, tiled)


def test_synthetic_pair_truth(oracle):
    """The benchmark generator: interior pixels of each pasted block must recover centre + s_b."""
    left, right, truth = synth.stereo_pair(128, 64, 33, 1, block=32)
    d = oracle.calc_disparity(0, left, right, (5, 5), (33, 1))
    oh, ow = d.shape[:2]
    t = truth[:oh, :ow]
    # pixels whose 5x5 window at the true disparity survived later pastes (zero SAD there)
    yy, xx = np.mgrid[0:oh, 0:ow]
    zero = np.ones((oh, ow), bool)
    for j in range(5):
        for i in range(5):
            zero &= left[yy + j, xx + i] == right[yy + j, xx + i + t]
    assert zero.mean() > 0.5
    assert (d[..., 0][zero] <= t[zero]).all()          # first zero-cost disparity wins
    assert (d[..., 2][zero] == oracle.VALID).all()
    assert (d[..., 0][zero] == t[zero]).mean() > 0.99


# ---- image filters on the path (pyramid, prefilters) ------------------------------------------------------------

def test_gaussian_kernel_golden(oracle):
    """src/vw/Image/tests/TestFilter.cxx:45-74 (GaussianKernel), each tap at the tolerance the reference's own EXPECT_NEAR asks."""
    k = oracle.generate_gaussian_kernel(1.0, 5, np.float64)
    assert len(k) == 5
    for got, want, tol in zip(k, [0.06135958087, 0.2447702197, 0.3877403988, 0.2447702197, 0.06135958087], [1e-8, 1e-7, 1e-7, 1e-7, 1e-8]):
        assert abs(got - want) <= tol
    k = oracle.generate_gaussian_kernel(1.0, 4, np.float64)
    assert len(k) == 4
    for got, want in zip(k, [0.1423836140, 0.3576163860, 0.3576163860, 0.1423836140]):
        assert abs(got - want) <= 1e-7
    k = oracle.generate_gaussian_kernel(1.5, 0, np.float64)
    assert len(k) == 9
    for got, want, tol in zip(k, [0.008488347404, 0.03807782601, 0.1111650246, 0.2113567063, 0.2618241916, 0.2113567063, 0.1111650246,
                                  0.03807782601, 0.008488347404], [1e-9, 1e-8, 1e-7, 1e-7, 1e-7, 1e-7, 1e-7, 1e-8, 1e-9]):
        assert abs(got - want) <= tol
    assert len(oracle.generate_gaussian_kernel(0, 0, np.float64)) == 0


# ---- independent evidence for rows the reference's tests do not pin ("unpinned oracle rows") ------------------------------
# A second formulation written from the mathematics, not from the reference's text: scipy box filters in float64 (exact on small integer
# imagery), argmin / argmax with first-wins ties, exact rational comparison for NCC.  The oracle must agree with it.

def _window_sums(img, ky, kx):
    """Sum over every full ky x kx window, float64 (exact for integer data of this size)."""
    from scipy.ndimage import uniform_filter
    s = uniform_filter(img.astype(np.float64), size=(ky, kx), mode="constant") * (ky * kx)
    return np.rint(s[ky // 2: img.shape[0] - ky // 2, kx // 2: img.shape[1] - kx // 2])


def _independent_winners(cost, left, right, kernel, search):
    kx, ky = kernel
    sx, sy = search
    h, w = left.shape
    l = left.astype(np.int64)
    vols = []
    for dy in range(sy):
        for dx in range(sx):
            r = right[dy:dy + h, dx:dx + w].astype(np.int64)
            e = np.abs(l - r) if cost == 0 else (l - r) ** 2 if cost == 1 else l * r
            vols.append(_window_sums(e, ky, kx))
    vol = np.stack(vols)                                   # [disparity index (dy outer, dx inner), y, x]
    return vol


@pytest.mark.parametrize("cost", [0, 1])
@pytest.mark.parametrize("kernel,search", [((5, 5), (9, 3)), ((7, 3), (16, 1)), ((3, 9), (5, 4))])
def test_sad_ssd_winners_against_an_independent_formulation(oracle, cost, kernel, search):
    """AbsoluteCost / SquaredCost through calc_disparity: the winner is the FIRST minimum in (dy outer, dx inner) order, a pixel is invalid
    exactly when all its costs are equal (Correlation.cc:91-133 reduces to that for costs that are numbers)."""
    rng = np.random.default_rng(100 * cost + kernel[0] + search[0])
    h, w = 40, 52
    left = rng.integers(0, 256, (h, w)).astype(np.float32)
    right = rng.integers(0, 256, (h + search[1] - 1, w + search[0] - 1)).astype(np.float32)
    dy0, dx0 = min(2, search[1] - 1), min(3, search[0] - 1)
    right[dy0:dy0 + h, dx0:dx0 + w][8:, 10:] = left[8:, 10:]       # a true shift over most of the image
    left[:6, :20] = 17.0; right[:12, :40] = 17.0           # a flat corner: every cost equal -> invalid
    vol = _independent_winners(cost, left, right, kernel, search)
    idx = vol.argmin(axis=0)                               # first minimum
    want_dx, want_dy = idx % search[0], idx // search[0]
    want_valid = ~(vol == vol[0]).all(axis=0)
    got = oracle.calc_disparity(cost, left, right, kernel, search)
    assert np.array_equal(got[..., 0], want_dx) and np.array_equal(got[..., 1], want_dy)
    assert np.array_equal(got[..., 2] != 0, want_valid)
    assert (~want_valid).sum() > 0


@pytest.mark.parametrize("kernel,search", [((5, 5), (9, 3)), ((11, 11), (17, 1))])
def test_ncc_winners_against_exact_rational_arithmetic(oracle, kernel, search):
    """NCCCost (CostFunctions.h:207-236): the winner maximises S_lr / sqrt(S_ll S_rr).  With integer imagery the three sums are exact integers,
    so two candidates compare exactly as S_lr_a^2 S_rr_b vs S_lr_b^2 S_rr_a in Python integers; wherever the exact best leads the exact
    runner-up by more than a relative 1e-12 the oracle's float64 pipeline (1 / S, *, sqrt, *) must name the same first maximum."""
    rng = np.random.default_rng(7 + kernel[0])
    h, w = 36, 44
    kx, ky = kernel
    sx, sy = search
    left = rng.integers(1, 256, (h, w)).astype(np.float32)
    right = rng.integers(1, 256, (h + sy - 1, w + sx - 1)).astype(np.float32)
    dy0 = 1 if sy > 1 else 0
    right[dy0:dy0 + h, 4:4 + w][6:, 5:] = left[6:, 5:]
    slr = _independent_winners(2, left, right, kernel, search)
    srr_full = _window_sums(right.astype(np.int64) ** 2, ky, kx)
    oh, ow = slr.shape[1:]
    got = oracle.calc_disparity(2, left, right, kernel, search)
    checked = 0
    for y in range(oh):
        for x in range(ow):
            best, bi, second = None, -1, None
            for i in range(sx * sy):
                dy, dx = divmod(i, sx)
                a, b = int(slr[i, y, x]), int(srr_full[y + dy, x + dx])          # value = a / sqrt(b) (S_ll is common to all candidates), a > 0
                if best is None or a * a * best[1] > best[0] * best[0] * b:       # strictly greater: first maximum wins
                    if best is not None: second = best if second is None or best[0] ** 2 * second[1] > second[0] ** 2 * best[1] else second
                    best, bi = (a, b), i
                elif second is None or a * a * second[1] > second[0] ** 2 * b:
                    second = (a, b)
            va, vb = best[0] / np.sqrt(best[1]), second[0] / np.sqrt(second[1])
            if va - vb > 1e-12 * va:
                checked += 1
                assert (int(got[y, x, 0]), int(got[y, x, 1])) == (bi % sx, bi // sx), (y, x)
                assert got[y, x, 2] != 0
    assert checked > 0.99 * oh * ow


@pytest.mark.parametrize("k,sx", [(5, 12), (7, 20), (3, 6)])
def test_sgm_integer_disparities_against_a_numpy_formulation(oracle, k, sx):
    """SemiGlobalMatcher on full boxes (SGM.cc:2462-2612, :1013-1150, CensusTransform.h), written again from the textbook recurrence
    L_r(p, d) = C(p, d) + min(L_r(p - r, d), L_r(p - r, d +- 1) + P1, min_k L_r(p - r, k) + P2') - min_k L_r(p - r, k), P2' = max(P1, P2 / |dI|),
    the first pixel of a line taking its plain costs, the eight directions summed in u16, census bits = (neighbour > centre): wherever the sum
    has a UNIQUE minimum (ties go through the reference's smoothing loop, which this formulation leaves out) the oracle must name it."""
    rng = np.random.default_rng(50 + k)
    H, W = 26, 34
    left = rng.integers(0, 256, (H, W)).astype(np.float32)
    right = rng.integers(0, 256, (H, W + sx)).astype(np.float32)
    right[:, sx // 2:sx // 2 + W][4:, 6:] = left[4:, 6:]
    left[0, 0], left[0, 1], right[0, 0], right[0, 1] = 0, 255, 0, 255          # full range: the u8 conversion is the identity
    hk = k // 2
    oh, ow = H - 2 * hk, W - 2 * hk
    Li, Ri = left.astype(np.int64), right.astype(np.int64)

    def census(img):
        h, w = img.shape
        c = img[hk:h - hk, hk:w - hk]
        bits = [img[hk + j:h - hk + j, hk + i:w - hk + i] > c for j in range(-hk, hk + 1) for i in range(-hk, hk + 1) if (i, j) != (0, 0)]
        return np.stack(bits, -1)
    cl, cr = census(Li), census(Ri)                           # (oh, ow, n) and (oh, ow + sx, n)
    D = sx + 1
    C = np.stack([(cl != cr[:, d:d + ow]).sum(-1) for d in range(D)], -1).astype(np.int64)      # (oh, ow, D)
    p1 = {3: 3, 5: 15, 7: 30}[k]
    p2 = {3: 70, 5: 750, 7: 1500}[k]
    grey = Li[hk:hk + oh, hk:hk + ow]
    S = np.zeros((oh, ow, D), np.int64)
    for dc, dr in [(0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, 1), (1, -1), (-1, -1)]:
        starts = set()
        for c in range(ow):
            for r in range(oh):
                if not (0 <= c - dc < ow and 0 <= r - dr < oh):
                    starts.add((c, r))
        for (c, r) in starts:
            prev = None
            while 0 <= c < ow and 0 <= r < oh:
                if prev is None:
                    cur = C[r, c].copy()
                else:
                    g = abs(int(grey[r, c]) - int(grey[r - dr, c - dc]))
                    pen = max(p1, p2 // g if g > 0 else p2)
                    mp = prev.min()
                    lo = np.concatenate(([prev[0]], prev[:-1])); hi = np.concatenate((prev[1:], [prev[-1]]))
                    cur = C[r, c] + np.minimum(np.minimum(prev, np.minimum(lo, hi) + p1), mp + pen) - mp
                S[r, c] += cur
                prev = cur
                c += dc; r += dr
    assert S.max() < 65536
    best = S.argmin(-1)
    unique = (S == S.min(-1, keepdims=True)).sum(-1) == 1
    oi, _ = oracle.calc_disparity_sgm(3, left, right, (sx, 0), k)
    assert oi.shape == (oh, ow, 3)
    assert unique.mean() > 0.9
    assert np.array_equal(oi[..., 0][unique], best[unique])
    assert (oi[..., 1][unique] == 0).all() and (oi[..., 2][unique] != 0).all()


def test_parabola_offsets_against_a_least_squares_fit(oracle):
    """ParabolaSubpixelView (ParabolaSubpixelView.cc:31-274, .h:83-88) from the mathematics: nine windowed SADs around the integer disparity, a
    least-squares quadric z = a x^2 + b y^2 + c xy + d x + e y + f through them (numpy.linalg.lstsq instead of the reference's tabulated
    pseudo-inverse), its stationary point as the offset, kept when shorter than 5 pixels.  Interior pixels (no edge extension), no prefilter."""
    rng = np.random.default_rng(77)
    h, w, kx, ky = 28, 36, 5, 7
    hx, hy = kx // 2, ky // 2
    left = (rng.random((h, w)) * 200).astype(np.float32)
    right = (rng.random((h + 6, w + 8)) * 200).astype(np.float32)
    disp = np.zeros((h, w, 3), np.float32)
    disp[..., 0] = rng.integers(1, 5, (h, w)); disp[..., 1] = rng.integers(1, 3, (h, w)); disp[..., 2] = 1.0
    for y in range(h):                                      # a smooth right image around the true match: the quadric has a minimum nearby
        for x in range(w):
            right[y + int(disp[y, x, 1]), x + int(disp[y, x, 0])] = left[y, x] * 0.9 + 5.0
    disp[3, 4, 2] = 0.0
    got = oracle.parabola_subpixel(disp, left, right, 0, 0.0, (kx, ky))
    gx, gy = np.meshgrid([-1.0, 0.0, 1.0], [-1.0, 0.0, 1.0])
    Amat = np.stack([gx.ravel() ** 2, gy.ravel() ** 2, (gx * gy).ravel(), gx.ravel(), gy.ravel(), np.ones(9)], 1)
    checked = 0
    for y in range(hy + 1, h - hy - 1):
        for x in range(hx + 1, w - hx - 1):
            if disp[y, x, 2] == 0:
                assert (got[y, x] == 0).all()
                continue
            Dx, Dy = int(disp[y, x, 0]), int(disp[y, x, 1])
            lwin = left[y - hy:y + hy + 1, x - hx:x + hx + 1].astype(np.float64)
            z = np.empty(9)
            for ddy in (-1, 0, 1):
                for ddx in (-1, 0, 1):
                    ry, rx = y + Dy + ddy, x + Dx + ddx
                    z[(ddy + 1) * 3 + ddx + 1] = np.abs(lwin - right[ry - hy:ry + hy + 1, rx - hx:rx + hx + 1]).sum()
            a, b, c, d, e, _ = np.linalg.lstsq(Amat, z, rcond=None)[0]
            den = 4 * a * b - c * c
            ox, oy = (c * e - 2 * b * d) / den, (c * d - 2 * a * e) / den
            want = (Dx + ox, Dy + oy) if np.hypot(ox, oy) < 5.0 else (Dx, Dy)
            if abs(np.hypot(ox, oy) - 5.0) < 1e-3: continue      # on the acceptance threshold: float32 may fall either way
            assert abs(got[y, x, 0] - want[0]) < 2e-3 * max(1.0, abs(ox)) and abs(got[y, x, 1] - want[1]) < 2e-3 * max(1.0, abs(oy)), (y, x, got[y, x], want)
            assert got[y, x, 2] != 0
            checked += 1
    assert checked > 400


def test_pyramid_level_against_scipy(oracle):
    """subsample(separable_convolution_filter(img, k, k), 2) (CorrelationView.cc:38-63): scipy's correlate1d in float64 with the nearest-pixel
    edge, every second pixel — the oracle's float accumulation agrees to float32 rounding."""
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(5)
    img = (rng.random((37, 50)) * 200).astype(np.float32)
    k = np.asarray(oracle.pyramid_smoothing_kernel(), np.float64)
    assert len(k) == 5 and abs(k.sum() - 1.0) < 1e-6 and np.allclose(k, k[::-1])
    ref = correlate1d(correlate1d(img.astype(np.float64), k, axis=1, mode="nearest"), k, axis=0, mode="nearest")[::2, ::2]
    lvl = oracle.separable_convolution(img, k.astype(np.float32), k.astype(np.float32), subsample=2)
    assert lvl.shape == ref.shape == (19, 25)
    np.testing.assert_allclose(lvl, ref, rtol=2e-6, atol=1e-4)


def _src22():
    # src(0,0)=1; src(1,0)=2; src(0,1)=3; src(1,1)=4   ->  [row, col]
    return np.array([[1.0, 2.0], [3.0, 4.0]])


def test_separable_convolution_golden(oracle):
    """src/vw/Image/tests/TestConvolution.cxx:110-215: SeparableView, _0x2, _2x0 with ZeroEdgeExtension."""
    krn = np.array([1.0, -1.0])
    d = oracle.separable_convolution(_src22(), krn, krn, edge=oracle.EDGE_ZERO)
    assert d[0, 0] == 1 and d[1, 0] == 2 and d[0, 1] == 1 and d[1, 1] == 0          # dst(col,row) == d[row,col]
    src = np.array([[1.0, 2.0], [4.0, 6.0]])
    d = oracle.separable_convolution(src, np.array([]), krn, edge=oracle.EDGE_ZERO)  # SeparableView_0x2
    assert d[0, 0] == 1 and d[1, 0] == 3 and d[0, 1] == 2 and d[1, 1] == 4
    d = oracle.separable_convolution(_src22(), krn, np.array([]), edge=oracle.EDGE_ZERO)  # SeparableView_2x0
    assert d[0, 0] == 1 and d[0, 1] == 1 and d[1, 0] == 3 and d[1, 1] == 1


def test_convolution_2d_golden(oracle):
    """TestConvolution.cxx:80-108 (View) and TestFilter.cxx:141-150 (Laplacian), ZeroEdgeExtension."""
    krn = np.array([[2.0, -1.0], [0.0, 3.0]])      # krn(0,0)=2; krn(1,0)=-1; krn(0,1)=0; krn(1,1)=3
    d = oracle.convolution_2d(_src22(), krn, edge=oracle.EDGE_ZERO)
    assert d[0, 0] == 2 and d[0, 1] == 3 and d[1, 0] == 6 and d[1, 1] == 8
    lap = np.array([[0.0, 1, 0], [1, -4, 1], [0, 1, 0]])
    d = oracle.convolution_2d(_src22(), lap, 1, 1, edge=oracle.EDGE_ZERO)
    assert d[0, 0] == 1 and d[0, 1] == -3 and d[1, 0] == -7 and d[1, 1] == -11


def test_gaussian_filter_golden(oracle):
    """TestFilter.cxx:132-139 (Gaussian): gaussian_filter(src, 1.0, 0, 5, 0, ZeroEdgeExtension) — x only."""
    k = oracle.generate_gaussian_kernel(1.0, 5, np.float64)
    d = oracle.separable_convolution(_src22(), k, np.array([]), edge=oracle.EDGE_ZERO)
    np.testing.assert_allclose(d, [[0.3877403988 * 1 + 0.2447702197 * 2, 0.3877403988 * 2 + 0.2447702197 * 1],
                                   [0.3877403988 * 3 + 0.2447702197 * 4, 0.3877403988 * 4 + 0.2447702197 * 3]], atol=1e-7)


def test_prerasterize_fixture(oracle):
    """TestConvolution.cxx:203-215 (Prerasterize): gaussian(1.5) of a constant-1 image stays 1 inside (float taps
    sum to ~1), zero-extended outside."""
    img = np.ones((40, 60), np.float32)
    k = oracle.generate_gaussian_kernel(1.5)
    d = oracle.separable_convolution(img, k, k)
    assert abs(d[20, 30] - 1.0) < 1e-6 and abs(d[0, 0] - 1.0) < 1e-6


def test_prefilters_are_the_compositions(oracle):
    """src/vw/Stereo/tests/TestPreFilter.cxx:47-112: the prefilter structs equal the explicit compositions."""
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (37, 53)).astype(np.float32)
    # the prefilter structs hold the width as float (PreFilter.h:53,68): sigma = (double)1.4f
    k = oracle.generate_gaussian_kernel(float(np.float32(1.4)))
    assert len(k) == 9
    g = oracle.separable_convolution(img, k, k)
    lap = np.array([[0.0, 1, 0], [1, -4, 1], [0, 1, 0]], np.float32)
    assert np.array_equal(oracle.prefilter_image(img, oracle.PREFILTER_LOG, 1.4), oracle.convolution_2d(g, lap, 1, 1))
    k5 = oracle.generate_gaussian_kernel(5.0)
    assert len(k5) == 35
    assert np.array_equal(oracle.prefilter_image(img, oracle.PREFILTER_MEANSUB, 5.0),
                          img - oracle.separable_convolution(img, k5, k5))
    assert np.array_equal(oracle.prefilter_image(img, oracle.PREFILTER_NONE, 0.0), img)


def test_pyramid_level_and_mask(oracle):
    """subsample(separable_convolution_filter(level, k, k), 2) and subsample_mask_by_two
    (src/vw/Stereo/CorrelationView.cc:38-63,210-216): sizes 1+(N-1)/2; constant images stay constant."""
    k = oracle.pyramid_smoothing_kernel()
    img = np.full((9, 14), 100.0, np.float32)
    lvl = oracle.separable_convolution(img, k, k, subsample=2)
    assert lvl.shape == (5, 7) and (lvl == 100.0).all()
    mask = np.zeros((5, 7), np.uint8)
    mask[0, 0] = 255                # one of four -> off
    mask[2, 2] = mask[2, 3] = 9     # two of four -> on
    mask[4, 6] = 1                  # corner block partly outside (zero extension): one -> off
    m2 = oracle.subsample_mask_by_two(mask)
    assert m2.shape == (3, 4)
    assert m2[0, 0] == 0 and m2[1, 1] == 255 and m2[2, 3] == 0


# ---- zone subdivision and parabola sub-pixel ----------------------------------------------------------------------

def _pm2f(dx, dy, h, w, valid=1.0):
    d = np.zeros((h, w, 3), np.float32)
    d[..., 0], d[..., 1], d[..., 2] = dx, dy, valid
    return d


@pytest.mark.parametrize("mode", [0, 2])
def test_parabola_null_test(oracle, mode):
    """src/vw/Stereo/tests/TestSubPixel.cxx:93-124 (NullTest): constant images 0.5 / 0.6, disparity (1,1), 3x3 kernel,
    PREFILTER_NONE and PREFILTER_LOG(1.4): every pixel valid and within 0.1 of (1,1)."""
    left = np.full((5, 5), 0.5, np.float32)
    right = np.full((5, 5), 0.6, np.float32)
    out = oracle.parabola_subpixel(_pm2f(1, 1, 5, 5), left, right, mode, 1.4, (3, 3))
    assert out.shape == (5, 5, 3)
    assert (out[..., 2] == 1.0).all()
    assert np.abs(out[..., :2] - 1.0).max() < 0.1


def test_parabola_recovers_fractional_shift(oracle):
    """In the spirit of TestSubPixel.cxx:130-140 (stretched copy, mean error < 0.6, nothing invalid): a smooth texture
    shifted by a fraction of a pixel; starting from the truncated disparity the refined one must be much closer."""
    yy, xx = np.mgrid[0:80, 0:120].astype(np.float64)

    def tex(x, y):
        return 120 + 50 * np.sin(x / 3.1) * np.cos(y / 4.3) + 40 * np.sin((x + 2 * y) / 5.7) + 20 * np.cos(x / 1.9 + y / 2.3)
    left = tex(xx, yy).astype(np.float32)
    right = tex(xx - 2.35, yy - 0.6).astype(np.float32)              # right(x + 2.35, y + 0.6) == left(x, y)
    out = oracle.parabola_subpixel(_pm2f(2, 1, 80, 120), left, right, 0, 0.0, (7, 7))   # nearest integer start
    core = out[10:-10, 10:-10]
    assert (core[..., 2] == 1.0).all()
    err = np.abs(core[..., 0] - 2.35) + np.abs(core[..., 1] - 0.6)
    assert err.mean() < 0.4                                          # the integer start has error 0.35 + 0.4 = 0.75


def test_subdivide_regions_basic(oracle):
    """subdivide_regions (Correlation.cc:139-328): zones tile the image, each zone's range contains its pixels'
    disparities, a uniform image stays one zone, a two-valued image splits."""
    V = oracle.VALID
    d = np.zeros((64, 96, 3), np.int32)
    d[..., 0], d[..., 1], d[..., 2] = 3, 1, V
    z = oracle.subdivide_regions(d, (7, 7))
    assert len(z) == 1 and list(z[0]) == [0, 0, 96, 64, 3, 1, 4, 2]
    d[:, 48:, 0] = 40                                                  # right half far away
    z = oracle.subdivide_regions(d, (7, 7))
    assert len(z) >= 2
    cover = np.zeros((64, 96), np.int32)
    for x0, y0, x1, y1, rx0, ry0, rx1, ry1 in z:
        cover[y0:y1, x0:x1] += 1
        blk = d[y0:y1, x0:x1]
        assert (blk[..., 0] >= rx0).all() and (blk[..., 0] < rx1).all()
        assert (blk[..., 1] >= ry0).all() and (blk[..., 1] < ry1).all()
    assert (cover == 1).all()
    # invalid areas produce no zone
    d[..., 2] = 0
    assert len(oracle.subdivide_regions(d, (7, 7))) == 0


# ---- pyramid_correlate (BM) and the disparity clean-up chain ---------------------------------------------------------

# thresholds of src/vw/Stereo/tests/TestPyramidCorrelationView.cxx:92-410: (cost, consistency threshold) -> (correct, attempted)
_PYR_U8 = {(0, -1): (.909, .9985), (0, 2): (.91, .990), (1, -1): (.90, .9985), (1, 2): (.90, .990),
           (2, -1): (.90, .998), (2, 2): (.90, .990)}
_PYR_I16 = {**_PYR_U8, (2, -1): (.87, .99), (2, 2): (.87, .99)}


@pytest.mark.parametrize("channel,table", [("u8", _PYR_U8), ("i16", _PYR_I16), ("f32", _PYR_U8)])
def test_pyramid_correlate_reference_thresholds(oracle, channel, table):
    """TestPyramidCorrelationView.cxx NullPreprocess (GRAYU8 :92, GRAYI16 :172, GRAYF32 :253): the same scene
    (rand48 noise 300x200, affine 0.9/0.95 + (15, 5), bicubic), arguments and pass thresholds."""
    import scenes
    left, right, scale, trans, search = scenes.pyramid_scene(channel)
    for (cost, thr), (correct, attempted) in table.items():
        d = oracle.pyramid_correlate(left, right, np.full(left.shape, 255, np.uint8), np.full(right.shape, 255, np.uint8),
                                     0, 0.0, search, (7, 7), cost, 0, 0.0, thr, 5, 5)
        assert d.shape == left.shape + (3,)
        got_c, got_a = scenes.pyramid_score(d, scale, trans)
        assert got_c > correct and got_a > attempted, (channel, cost, thr, got_c, got_a)


def test_disparity_cleanup_known_answer(oracle):
    """After TestDisparity.cxx:222-276 (DisparityFiltering): identity-ramp disparity with a corrupted patch; only the
    corrupted pixels may be rejected.  The reference test drives triple_disparity_cleanup; the path here is
    disparity_cleanup_using_thresh (DisparityMap.h:427-441), so the patch is 2x2 (4/49 < 0.2 matches)."""
    V = oracle.VALID
    n = 100
    d = np.zeros((n, n, 3), np.int32)
    d[..., 0] = np.arange(n)[None, :]
    d[..., 1] = np.arange(n)[:, None]
    d[..., 2] = V
    d[5:7, 5:7, 0], d[5:7, 5:7, 1] = 10000, 5000
    for cleanup in (0, 1):
        f = oracle.disparity_filter(d, 3, 3, 10.0, 0.2, cleanup)
        assert (f[5:7, 5:7, 2] == 0).all() and (f[5:7, 5:7, :2] == 0).all()
        assert int((f[..., 2] == 0).sum()) == 4
        keep = f[..., 2] != 0
        assert (f[keep] == d[keep]).all()
    # a 5x5 patch supports itself (25/49 > 0.2): the threshold filter keeps it, as the functor's definition says
    d[5:10, 5:10, 0], d[5:10, 5:10, 1] = 10000, 5000
    assert int((oracle.disparity_filter(d, 3, 3, 10.0, 0.2, 0)[..., 2] == 0).sum()) == 0


def test_disparity_mask_known_answer(oracle):
    """DisparityMaskView (DisparityMap.h:132-155): masked source, masked / out-of-image target invalidate and zero."""
    V = oracle.VALID
    d = np.zeros((4, 6, 3), np.int32)
    d[..., 0], d[..., 1], d[..., 2] = 2, 1, V
    lm = np.full((4, 6), 255, np.uint8)
    rm = np.full((5, 7), 255, np.uint8)
    lm[0, 0] = 0                         # source masked
    rm[2, 3] = 0                         # target of pixel (1, 1) masked
    d[3, 5, 2] = 0                       # already invalid
    out = oracle.disparity_mask(d, lm, rm)
    expect_valid = np.ones((4, 6), bool)
    expect_valid[0, 0] = expect_valid[1, 1] = expect_valid[3, 5] = False
    expect_valid[:, 5] = False           # x + 2 = 7 is outside the 7-wide right mask
    expect_valid[3, :] &= False          # y + 1 = 4 < 5 stays inside ... row 3 -> 4 is valid, undo below
    expect_valid[3, :5] = True
    expect_valid[3, 5] = False
    assert ((out[..., 2] != 0) == expect_valid).all()
    assert (out[~expect_valid] == 0).all() and (out[expect_valid] == [2, 1, V]).all()


# ---- semi-global matching ----------------------------------------------------------------------------------------------

_CENSUS_SRC = np.array([[1, 2, 7, 2, 2, 8, 5, 2], [1, 4, 2, 9, 8, 8, 2, 6], [5, 2, 7, 2, 2, 2, 4, 6], [1, 2, 2, 2, 1, 4, 5, 2],
                        [6, 6, 3, 7, 2, 2, 5, 5], [1, 2, 9, 2, 2, 2, 2, 2], [7, 9, 2, 8, 5, 2, 3, 2], [1, 2, 2, 2, 2, 2, 2, 1]], np.uint8)


def test_census_transform_point_tests(oracle):
    """src/vw/Image/tests/TestCensusTransform.cxx:25-47 (PointTests): the image is indexed src(col,row) there."""
    c3, c5, c7 = (oracle.census_transform(_CENSUS_SRC, k) for k in (3, 5, 7))
    at = lambda img, col, row, k: int(img[row - k // 2, col - k // 2])
    assert at(c3, 2, 2, 3) == 0x20 and at(c3, 4, 5, 3) == 0x86 and at(c3, 6, 1, 3) == 0xDB
    assert at(c5, 4, 4, 5) == 0x0088F60D and at(c5, 2, 3, 5) == 0x005D03C4
    assert at(c7, 3, 4, 7) == 0x00001C0000041400


def test_hamming_distance_tests(oracle):
    """TestCensusTransform.cxx:49-59 (HammingDist)."""
    h = oracle.hamming_distance
    assert (h(0x01, 0x00), h(0xF0, 0x00), h(0xF0, 0xB1)) == (1, 4, 2)
    assert (h(0x00B06FFF, 0x00B06F11), h(0x0033C8BA, 0x0023C0BA)) == (6, 2)
    assert h(0x00002820A0F038, 0x00002820000030) == 7


def _sgm_fixture():
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgm_fixture.npz"))
    return d["left"], d["right"]


def test_sgm_constant_offset(oracle):
    """src/vw/Stereo/tests/TestSGM.cxx:28-75 on the reference's own images (tests/golden/sgm_fixture.npz, made by
    tests/golden/make_sgm_fixture.py): > 99 % of the pixels must come out as (2, 1) after adding the search minimum."""
    left, right = _sgm_fixture()
    res, sub = oracle.calc_disparity_sgm(oracle.CENSUS_TRANSFORM, left.astype(np.float32), right.astype(np.float32), (9, 9), 3,
                                         subpixel=oracle.SUBPIXEL_LC_BLEND, search_buffer=(4, 4), memory_limit_mb=1024)
    assert res.shape == (398, 398, 3)
    correct = ((res[..., 0] - 4 == 2) & (res[..., 1] - 4 == 1)).mean()
    assert correct > 0.99
    assert (np.abs(sub[..., 0] - 6) < 1).all() and (np.abs(sub[..., 1] - 5) < 1).all()


def test_mgm_constant_offset_on_the_reference_fixture(oracle):
    """The same scene with use_mgm = true.  The reference never runs this branch in its tests (TestSGM.cxx:47 sets use_mgm = false), so
    this is no golden vector — but its own image pair has one true answer, (2, 1) everywhere, and the restated accum_mgm_multithread must
    find it as the plain eight-path accumulation does."""
    left, right = _sgm_fixture()
    res, sub = oracle.calc_disparity_sgm(oracle.CENSUS_TRANSFORM, left.astype(np.float32), right.astype(np.float32), (9, 9), 3,
                                         subpixel=oracle.SUBPIXEL_LC_BLEND, search_buffer=(4, 4), memory_limit_mb=1024, use_mgm=True)
    assert res.shape == (398, 398, 3)
    assert ((res[..., 0] - 4 == 2) & (res[..., 1] - 4 == 1)).mean() > 0.99
    assert (np.abs(sub[..., 0] - 6) < 1).all() and (np.abs(sub[..., 1] - 5) < 1).all()


def test_sgm_parameters_and_errors(oracle):
    """set_parameters defaults (SGM.cc:105-160) and the NoImplErr cases of compute_disparity_costs (:1877-1890)."""
    table = {(3, 3): (3, 70), (3, 5): (15, 750), (3, 7): (30, 1500), (3, 9): (20, 1000),
             (4, 3): (12, 600), (4, 5): (30, 1500), (4, 7): (40, 2000), (4, 9): (40, 2000)}
    for (cost, k), want in table.items():
        assert oracle.SemiGlobalMatcher(cost, 0, 0, 4, 4, k).p1p2() == want
    assert oracle.SemiGlobalMatcher(3, 0, 0, 4, 4, 5, p1=7, p2=99).p1p2() == (7, 99)
    with pytest.raises(ValueError):
        oracle.SemiGlobalMatcher(0, 0, 0, 4, 4, 5)          # block (MAD) cost: "only the census transform ..."
    with pytest.raises(ValueError):
        oracle.SemiGlobalMatcher(3, 0, 0, 4, 4, 11)         # census sizes 3, 5, 7, 9 only


def test_sgm_masks_and_prev_disparity(oracle):
    """populate_disp_bound_image (SGM.cc:241-499): masked left pixels get a zero search area and come out invalid; a
    trusted half-resolution disparity narrows the range to +-search_buffer; buffers are ragged accordingly."""
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (70, 90)).astype(np.uint8)
    left = base[4:60, 4:70]
    right = base[2:2 + 56 + 12, 1:1 + 66 + 12]          # left(x, y) = right(x + 3, y + 2)
    m = oracle.SemiGlobalMatcher(oracle.CENSUS_TRANSFORM, 0, 0, 12, 12, 5)
    d0 = m.semi_global_matching_func(left, right)
    oh, ow = d0.shape[:2]
    assert (oh, ow) == (52, 62)
    inner = d0[6:-6, 6:-6]
    assert ((inner[..., 0] == 3) & (inner[..., 1] == 2)).mean() > 0.98
    lmask = np.full((oh, ow), 255, np.uint8)
    lmask[10:20, 10:30] = 0
    rmask = np.full((oh + 12, ow + 12), 255, np.uint8)
    prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
    prev[..., 0], prev[..., 1], prev[..., 2] = 2, 1, oracle.VALID      # x2 -> (4, 2): within +-2 of the truth
    d1 = m.semi_global_matching_func(left, right, lmask, rmask, prev)
    b, s, c, a = m.buffers()
    assert (d1[10:20, 10:30, 2] == 0).all()
    assert (b[10:20, 10:30] == [0, 0, -1, -1]).all()
    assert (b[30, 30] == [2, 0, 6, 4]).all()                           # 2*2 +- 2 in x, 2*1 +- 2 in y, clipped at 0
    counts = (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1)
    assert int(counts.sum()) == len(c) == len(a)
    assert (s.reshape(-1)[1:] == np.cumsum(counts.reshape(-1))[:-1]).all()
    keep = lmask != 0
    assert ((d1[..., 0] == 3) & (d1[..., 1] == 2))[keep][200:].mean() > 0.95


def test_blob_sizes_known_answer(oracle):
    """src/vw/Image/tests/TestBlobIndex.cxx:97-126 (BlobSizesView): 8-connected blobs of the non-zero pixels, sizes 7 / 2 / 3,
    and the blob filter built on them (CorrelationView.cc:242-271) erases exactly the blobs of at most `area` pixels."""
    img = np.array([[0, 0, 0, 1, 8, 0], [0, 1, 0, 0, 0, 0], [0, 1, 0, 0, 1, 1], [1, 1, 0, 0, 0, 1], [1, 0, 1, 0, 0, 0], [0, 1, 0, 0, 0, 0]], np.float32)
    d = np.zeros((6, 6, 3), np.int32)
    d[..., 0], d[..., 1] = 3, -2
    d[..., 2] = np.where(img != 0, oracle.VALID, 0)
    s = oracle.blob_sizes(d)                       # image(col, row) in the reference = s[row, col] here
    assert (s[3, 3], s[3, 0], s[0, 3], s[3, 5]) == (0, 7, 2, 3)
    assert int((s == 7).sum()) == 7 and int((s == 2).sum()) == 2 and int((s == 3).sum()) == 3
    for area, left_valid in ((0, 12), (1, 12), (2, 10), (3, 7), (6, 7), (7, 0)):
        f = oracle.disparity_blob_filter(d, area)
        assert int((f[..., 2] != 0).sum()) == left_valid
        gone = (d[..., 2] != 0) & (f[..., 2] == 0)
        assert (f[gone] == 0).all() and (f[~gone] == d[~gone]).all()
