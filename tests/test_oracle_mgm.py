"""The oracle's MGM accumulation (oracle/vw_sgm_oracle.cc accumulate_mgm) against an independent pure-Python restatement of
accum_mgm_multithread (src/vw/Stereo/SGM.cc:2619-2700, SmoothPathAccumTask src/vw/Stereo/SGMAssist.h:835-1239) on small cases.

The reference's own tests never run use_mgm = true (TestSGM.cxx:47), so no golden vector exists for this branch: the two restatements
were written separately from the reference text (C++ raster loops over ragged vectors there, numpy over a dense volume here) and must
agree on the accumulated sums, not only on the winning disparity."""
import numpy as np
import pytest

CENSUS = 3


def _evaluate_path(prior, local, p1, p2_mod, bad, ndx, ndy):
    """evaluate_path (SGM.cc:936-984, 1013-1150) for full-range boxes: prior / local are (ndy, ndx) arrays."""
    pr = prior.astype(np.int64)
    min_prior = min(int(pr.min()), bad)
    dj = (min_prior + p2_mod) & 0xffff
    out = np.empty_like(pr)
    for y in range(ndy):
        for x in range(ndx):
            yl, ym = max(y - 1, 0), min(y + 1, ndy - 1)
            xl, xm = max(x - 1, 0), min(x + 1, ndx - 1)
            m = min(pr[yl, x], pr[y, xl], pr[y, xm], pr[ym, x], pr[yl, xl], pr[yl, xm], pr[ym, xl], pr[ym, xm])
            res = min(int(m) + p1, 65535)
            res = min(res, int(pr[y, x]), dj)
            res = min(res + int(local[y, x]), 65535)
            out[y, x] = max(res - min_prior, 0)
    return out


def _mgm_python(left_u8, cost, p1, p2, min_col, min_row):
    """cost: (H, W, ndy, ndx) uint8.  Returns the (H, W, ndy, ndx) uint16 sums of the eight smooth passes."""
    H, W, ndy, ndx = cost.shape
    bad = 255 + p2
    acc = np.zeros(cost.shape, np.int64)
    #        path pred   perp pred   col>0 col<last row>0 row<last   visiting order
    dirs = [((-1, 0), (0, -1), 1, 0, 1, 0, "rows_down"), ((1, 0), (0, 1), 0, 1, 0, 1, "rows_up"),
            ((-1, -1), (1, -1), 1, 1, 1, 0, "rows_down"), ((1, 1), (-1, 1), 1, 1, 0, 1, "rows_up"),
            ((0, -1), (1, 0), 0, 1, 1, 0, "cols_left"), ((0, 1), (-1, 0), 1, 0, 0, 1, "cols_right"),
            ((1, -1), (1, 1), 0, 1, 1, 1, "cols_left"), ((-1, 1), (-1, -1), 1, 0, 1, 1, "cols_right")]
    for (ax, ay), (bx, by), clo, chi, rlo, rhi, order in dirs:
        vol = np.zeros(cost.shape, np.int64)
        if order == "rows_down":
            visit = [(c, r) for r in range(H) for c in range(W)]
        elif order == "rows_up":
            visit = [(c, r) for r in range(H - 1, -1, -1) for c in range(W - 1, -1, -1)]
        elif order == "cols_right":
            visit = [(c, r) for c in range(W) for r in range(H - 1, -1, -1)]
        else:
            visit = [(c, r) for c in range(W - 1, -1, -1) for r in range(H)]
        for c, r in visit:
            ok = (not clo or c > 0) and (not chi or c < W - 1) and (not rlo or r > 0) and (not rhi or r < H - 1)
            if not ok:
                vol[r, c] = cost[r, c]
                continue
            a = int(left_u8[r + min_row, c + min_col])
            b = int(left_u8[r - ay + min_row, c - ax + min_col])          # get_path_pixel_diff: the pixel on the far side (SGM.cc:2715-2721)
            g = abs(a - b)
            p2_mod = p2 // g if g > 0 else p2
            p2_mod = max(p2_mod, p1)
            o1 = _evaluate_path(vol[r + ay, c + ax], cost[r, c], p1, p2_mod, bad, ndx, ndy)
            o2 = _evaluate_path(vol[r + by, c + bx], cost[r, c], p1, p2_mod, bad, ndx, ndy)
            vol[r, c] = (o1 + o2) // 2
        acc += vol
    return (acc & 0xffff).astype(np.uint16)


@pytest.mark.parametrize("sx,sy,k,w,h,p1,p2", [(4, 0, 3, 14, 11, 0, 0), (3, 2, 5, 13, 12, 0, 0), (5, 1, 3, 9, 16, 7, 9000)])
def test_oracle_mgm_sums_equal_python_restatement(oracle, sx, sy, k, w, h, p1, p2):
    rng = np.random.default_rng(sx * 10 + sy)
    left = rng.integers(0, 256, (h, w)).astype(np.uint8)
    right = rng.integers(0, 256, (h + sy, w + sx)).astype(np.uint8)
    right[1:1 + h - 1, 2:2 + w - 2] = left[:h - 1, :w - 2] if sy else right[1:1 + h - 1, 2:2 + w - 2]
    m = oracle.SemiGlobalMatcher(CENSUS, 0, 0, sx, sy, kernel=k, p1=p1, p2=p2, use_mgm=True)
    disp = m.semi_global_matching_func(left, right)
    oh, ow = m.shape
    hk = (k - 1) // 2
    bounds, starts, cost, accum = m.buffers()
    assert (bounds == np.array([0, 0, sx, sy])).all()
    nd = (sx + 1) * (sy + 1)
    cost4 = cost[:oh * ow * nd].reshape(oh, ow, sy + 1, sx + 1)
    q1, q2 = m.p1p2()
    want = _mgm_python(left, cost4, q1, q2, hk, hk)
    got = accum[:oh * ow * nd].reshape(oh, ow, sy + 1, sx + 1)
    # select_best_disparity smooths the vector of a pixel IN PLACE when its minimum is not unique (SGM.cc:1159-1284): those pixels
    # no longer hold the plain sums after the run
    flat = want.reshape(oh, ow, nd)
    unique_min = (flat == flat.min(axis=2, keepdims=True)).sum(axis=2) == 1
    assert unique_min.mean() > 0.9
    assert np.array_equal(got[unique_min], want[unique_min])
    # the winner of those pixels: first minimum, (dy, dx) order
    win = flat.argmin(axis=2)
    assert np.array_equal(disp[..., 0][unique_min], (win % (sx + 1))[unique_min]) and np.array_equal(disp[..., 1][unique_min], (win // (sx + 1))[unique_min])
    # and MGM is not SGM: the sums of the plain eight-path accumulation differ
    s = oracle.SemiGlobalMatcher(CENSUS, 0, 0, sx, sy, kernel=k, p1=p1, p2=p2)
    s.semi_global_matching_func(left, right)
    assert not np.array_equal(s.buffers()[3], accum)
