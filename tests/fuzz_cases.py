"""Seeded random case generators shared by the gpu-marked fuzz tests (tests/test_fuzz_gpu.py) and the command-line
aids in tools/fuzz_*.py / tools/replay_pyramid_case.py.  A case is a plain dict; every generator is a pure function of
(seed, index stream), so a failing case can be replayed from the numbers the test prints."""
import numpy as np

KERNELS_SAD = [(3, 3), (5, 5), (7, 7), (7, 5), (9, 9), (11, 11)]


def bm_cases(n, seed, cost, scale=256):
    """calc_disparity: random sizes / kernels / 1-D and 2-D searches, shifted copies + flat patches (validity).  `scale` = 4096: 12-bit
    imagery (the packed-u16 kernels; SSD / NCC windows are then mostly the instantiated squares)."""
    rng = np.random.default_rng(seed)
    for it in range(n):
        kx, ky = KERNELS_SAD[rng.integers(len(KERNELS_SAD))]
        if cost:
            kx, ky = int(rng.integers(1, 9)) * 2 - 1, int(rng.integers(1, 9)) * 2 - 1
            if scale != 256 and np.random.default_rng([seed, 7000003 + it]).random() < 0.8:
                kx = ky = min(max(kx, 3), 11)
        sx = int(rng.integers(1, 140))
        sy = int(rng.choice([1, 1, 1, 2, 3])) if cost == 0 else 1
        w = int(rng.integers(kx, 2600))
        h = int(rng.integers(ky, 200))
        lo = 1 if cost == 2 else 0                     # NCC: no all-zero windows (1/0 is the float64 kernel's business)
        left = np.floor(rng.random((h, w)) * (scale - lo)).astype(np.float32) + lo
        right = np.floor(rng.random((h + sy - 1, w + sx - 1)) * (scale - lo)).astype(np.float32) + lo
        d = int(rng.integers(0, sx))
        right[:h, d:d + w] = np.where(rng.random((h, w)) < 0.7, left, right[:h, d:d + w])
        if rng.random() < 0.5:
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            left[y0:y0 + 20, x0:x0 + 40] = 7.0
            right[y0:y0 + 24, x0:x0 + 60 + sx] = 7.0
        split = str(int(rng.integers(0, 2)))
        yield dict(it=it, cost=cost, kernel=(kx, ky), search=(sx, sy), left=left, right=right, split=split)


def pyramid_cases(n, seed, prefilters=(0,), costs=(0, 0, 1), float_scene=0.0):
    """pyramid_correlate: random scenes (row bands shifted by different amounts), search boxes, kernels, thresholds, filter
    radii, level counts, masks, interior / border tiles.  `float_scene` = probability of a non-integer (smooth float)
    texture, the input class on which the reference's running box sums round position-dependently."""
    rng = np.random.default_rng(seed)
    for it in range(n):
        H, W = int(rng.integers(90, 360)), int(rng.integers(120, 520))
        left = np.floor(rng.random((H, W)) * 256).astype(np.float32)
        aux = np.random.default_rng([seed, 1000003 + it])   # options added later draw from their own stream, so that
        is_float = aux.random() < float_scene               # (index, seed) of the original generator still replays
        if is_float:
            left = (left * np.float32(0.37) + aux.random((H, W)).astype(np.float32)).astype(np.float32)
        right = np.empty_like(left)
        band = int(rng.integers(20, 90))
        for y0 in range(0, H, band):
            right[y0:y0 + band] = np.roll(left[y0:y0 + band], int(rng.integers(-7, 8)), axis=1)
        if rng.random() < 0.5:
            right = np.roll(right, int(rng.integers(-2, 3)), axis=0)
        mx, my = int(rng.integers(1, 12)), int(rng.integers(0, 4))
        search = (-mx, -my, mx + int(rng.integers(0, 3)), my + 1)
        k = int(rng.choice([3, 5, 7, 9]))
        ky = int(rng.choice([k, k, 5]))
        cost = int(rng.choice(list(costs)))
        thr = float(rng.choice([-1, 1, 2]))
        filt = int(rng.choice([0, 3, 5]))
        levels = int(rng.integers(0, 5))
        lm = rm = None
        if rng.random() < 0.4:
            lm = np.full(left.shape, 255, np.uint8)
            rm = np.full(right.shape, 255, np.uint8)
            y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            lm[y0:y0 + 30, x0:x0 + 50] = 0
            rm[:, -int(rng.integers(1, 40)):] = 0
        bbox = None
        if rng.random() < 0.6:
            bw, bh = int(rng.integers(24, min(200, W))), int(rng.integers(24, min(160, H)))
            bbox = (int(rng.integers(0, W - bw + 1)), int(rng.integers(0, H - bh + 1)), bw, bh)
        pf = int(aux.choice(list(prefilters)))
        pfw = 0.0 if pf == 0 else float(aux.choice([1.4, 2.0, 3.0]))
        yield dict(it=it, left=left, right=right, lm=lm, rm=rm, search=search, kernel=(k, ky), cost=cost, thr=thr,
                   filt=filt, levels=levels, bbox=bbox, pf=pf, pfw=pfw, is_float=is_float)


def pyramid_corner_cases(n, seed):
    """pyramid_correlate on the kinds of imagery the random textures of pyramid_cases do not contain: blocks of exact zeros in both images
    (nodata without a mask: all-zero windows, infinite NCC precisions), saturated blocks (constant 255: every disparity ties), smooth ramps (low
    texture: near-ties everywhere), several decades of dynamic range, negative values — all costs, all prefilters, windows up to 15, the blob
    filter (c["blob"]: the oracle takes it through set_blob_filter_area)."""
    rng = np.random.default_rng([seed, 0xC1])
    for it in range(n):
        H, W = int(rng.integers(70, 260)), int(rng.integers(100, 380))
        kind = int(rng.integers(0, 6))
        tex = rng.random((H, W))
        if kind == 0:
            left = np.floor(tex * 256)
        elif kind == 1:
            left = tex * 0.37 * 256 + rng.random((H, W))
        elif kind == 2:
            yy, xx = np.mgrid[0:H, 0:W]
            left = 40.0 + 0.31 * xx + 0.17 * yy + tex * float(rng.choice([0.0, 0.5, 4.0]))          # a ramp with little or no texture
        elif kind == 3:
            left = tex * 10.0 ** (rng.random((H, W)) * float(rng.choice([3.0, 6.0])))                # decades
        elif kind == 4:
            left = (tex - 0.5) * 500.0
        else:
            left = np.floor(tex * 4.0) * 64.0                                                       # four grey levels
        left = left.astype(np.float32)
        for _ in range(int(rng.integers(0, 4))):                                                     # nodata / saturation blocks
            y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            left[y0:y0 + int(rng.integers(4, 60)), x0:x0 + int(rng.integers(4, 90))] = np.float32(rng.choice([0.0, 0.0, 255.0]))
        right = np.empty_like(left)
        band = int(rng.integers(20, 90))
        for y0 in range(0, H, band):
            right[y0:y0 + band] = np.roll(left[y0:y0 + band], int(rng.integers(-6, 7)), axis=1)
        if rng.random() < 0.3:
            right = right + (rng.random((H, W)).astype(np.float32) - np.float32(0.5)) * np.float32(rng.choice([0.01, 1.0]))
            right = right.astype(np.float32)
        mx, my = int(rng.integers(1, 10)), int(rng.integers(0, 3))
        search = (-mx, -my, mx + int(rng.integers(0, 3)), my + 1)
        k = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        ky = int(rng.choice([k, k, 5]))
        cost = int(rng.integers(0, 3))
        thr = float(rng.choice([-1, 1, 2]))
        filt = int(rng.choice([0, 3, 5]))
        levels = int(rng.integers(0, 5))
        blob = int(rng.choice([0, 0, 12, 80]))                                                      # blob_filter_area (CorrelationView.cc:242-271)
        lm = rm = None
        if rng.random() < 0.3:
            lm = np.full(left.shape, 255, np.uint8); rm = np.full(right.shape, 255, np.uint8)
            y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            lm[y0:y0 + 30, x0:x0 + 50] = 0
            rm[:, -int(rng.integers(1, 40)):] = 0
        bbox = None
        if rng.random() < 0.5:
            bw, bh = int(rng.integers(24, min(200, W))), int(rng.integers(24, min(160, H)))
            bbox = (int(rng.integers(0, W - bw + 1)), int(rng.integers(0, H - bh + 1)), bw, bh)
        pf = int(rng.integers(0, 3))
        pfw = 0.0 if pf == 0 else float(rng.choice([1.4, 2.0, 3.0]))
        yield dict(it=it, left=left, right=right, lm=lm, rm=rm, search=search, kernel=(k, ky), cost=cost, thr=thr,
                   filt=filt, levels=levels, bbox=bbox, pf=pf, pfw=pfw, is_float=kind not in (0, 5), kind=kind, blob=blob)


def sgm_cases(n, seed):
    """calc_disparity_sgm: random sizes, 1-D / 2-D searches, kernels, cost types, masks, previous-level disparities,
    memory limits, sub-pixel modes."""
    rng = np.random.default_rng(seed)
    for it in range(n):
        k = int(rng.choice([3, 5, 7, 9]))
        cost = int(rng.choice([3, 4]))
        sx = int(rng.integers(0, 70))
        sy = int(rng.choice([0, 0, 1, 2, 6]))
        if rng.random() < 0.2:
            sx = int(rng.integers(100, 140))
            sy = 0
        h = int(rng.integers(k + 2, 70))
        w = int(rng.integers(k + 2, 120))
        base = rng.random((h + sy + 8, w + sx + 8))
        scale = float(rng.choice([255.0, 1.0, 4000.0]))
        base = (base * scale).astype(np.float32)
        d0 = (int(rng.integers(0, sx + 1)), int(rng.integers(0, sy + 1)))
        left = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
        right = np.ascontiguousarray(base[4 - min(d0[1], 4):4 - min(d0[1], 4) + h + sy, 4 - min(d0[0], 4):4 - min(d0[0], 4) + w + sx])
        oh, ow = h - k + 1, w - k + 1
        lm = rm = prev = None
        if rng.random() < 0.4:
            lm = np.full((oh, ow), 255, np.uint8)
            y0, x0 = int(rng.integers(0, oh)), int(rng.integers(0, ow))
            lm[y0:y0 + 9, x0:x0 + 14] = 0
        if rng.random() < 0.3:
            rm = np.full((oh + sy, ow + sx), 255, np.uint8)
            rm[:, -int(rng.integers(1, 6)):] = 0
        if rng.random() < 0.4:
            prev = np.zeros(((oh + 1) // 2, (ow + 1) // 2, 3), np.int32)
            prev[..., 0] = rng.integers(0, sx // 2 + 1, prev.shape[:2])
            prev[..., 1] = rng.integers(0, sy // 2 + 1, prev.shape[:2])
            prev[..., 2] = np.where(rng.random(prev.shape[:2]) < 0.85, np.iinfo(np.int32).max, 0)
        sub = int(rng.choice([0, 1, 2, 3, 4, 5])) if (sx > 0 or sy > 0) else 0
        mem = int(rng.choice([6000, 6000, 1]))
        yield dict(it=it, k=k, cost=cost, search=(sx, sy), left=left, right=right, lm=lm, rm=rm, prev=prev, sub=sub, mem=mem)


def pyramid_sgm_cases(n, seed):
    """pyramid_correlate with VW_CORRELATION_SGM: random scenes (row bands shifted by different amounts), 1-D and 2-D search boxes,
    census / ternary kernels, L/R thresholds and consistency levels, filter radii, level counts, masks, interior / border tiles."""
    rng = np.random.default_rng(seed)
    for it in range(n):
        H, W = int(rng.integers(60, 200)), int(rng.integers(80, 260))
        left = np.floor(rng.random((H, W)) * 256).astype(np.float32)
        right = np.empty_like(left)
        band = int(rng.integers(20, 90))
        for y0 in range(0, H, band):
            right[y0:y0 + band] = np.roll(left[y0:y0 + band], int(rng.integers(-6, 7)), axis=1)
        if rng.random() < 0.4:
            right = np.roll(right, int(rng.integers(-1, 2)), axis=0)
        mx, my = int(rng.integers(1, 10)), int(rng.integers(0, 3))
        search = (-mx, -my, mx + int(rng.integers(0, 3)), my + 1)
        k = int(rng.choice([3, 5, 7, 9]))
        cost = int(rng.choice([3, 4]))
        thr = float(rng.choice([-1, 1, 2]))
        mcl = int(rng.choice([0, 0, 1, 2]))
        filt = int(rng.choice([0, 3, 5]))
        levels = int(rng.integers(0, 5))
        lm = rm = None
        if rng.random() < 0.4:
            lm = np.full(left.shape, 255, np.uint8)
            rm = np.full(right.shape, 255, np.uint8)
            y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            lm[y0:y0 + 25, x0:x0 + 40] = 0
            rm[:, -int(rng.integers(1, 30)):] = 0
        bbox = None
        if rng.random() < 0.6:
            bw, bh = int(rng.integers(24, min(160, W))), int(rng.integers(24, min(120, H)))
            bbox = (int(rng.integers(0, W - bw + 1)), int(rng.integers(0, H - bh + 1)), bw, bh)
        yield dict(it=it, left=left, right=right, lm=lm, rm=rm, search=search, k=k, cost=cost, thr=thr, mcl=mcl, filt=filt, levels=levels, bbox=bbox)


def bm_float_cases(n, seed):
    """calc_disparity on NON-integer rasters (round 5: the certified single-level path): smooth float textures, wide dynamic ranges, flat
    patches (exact ties: the call must fall back to the reference's order), LoG-like zero-mean data, negative values; windows from the
    compile-time squares and odd rectangles; 1-D and 2-D searches.  Sizes the oracle finishes in milliseconds."""
    rng = np.random.default_rng(seed)
    for it in range(n):
        cost = int(rng.integers(0, 3))
        if rng.random() < 0.7:
            kx = ky = int(rng.choice([3, 5, 7, 9, 11, 13]))
        else:
            kx, ky = int(rng.integers(1, 8)) * 2 - 1, int(rng.integers(1, 8)) * 2 - 1
        sx = int(rng.integers(1, 70))
        sy = int(rng.choice([1, 1, 2, 3]))
        w = int(rng.integers(kx + 1, 420))
        h = int(rng.integers(ky + 1, 160))
        kind = int(rng.integers(0, 5))
        base = rng.random((h + sy - 1, w + sx - 1))
        if kind == 0:
            right = (base * 200.0)
        elif kind == 1:
            right = base * 10.0 ** (rng.random(base.shape) * rng.choice([2.0, 6.0, 12.0]) - 3.0)
        elif kind == 2:
            right = (base - 0.5) * 60.0                                    # zero-mean (what a LoG / mean-subtraction prefilter leaves)
        elif kind == 3:
            right = np.floor(base * 256) * 0.37 + rng.random(base.shape)   # the bench's float texture
        else:
            right = base * 1e-3
        right = right.astype(np.float32)
        d = (int(rng.integers(0, sx)), int(rng.integers(0, sy)))
        left = right[d[1]:d[1] + h, d[0]:d[0] + w].copy()
        left += (rng.random((h, w)).astype(np.float32) - np.float32(0.5)) * np.float32(rng.choice([0.0, 0.01, 1.0]))
        if rng.random() < 0.3:                                             # a flat patch: ties in exact arithmetic
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            v = np.float32(rng.choice([0.0, 3.3, 100.25]))
            left[y0:y0 + 25, x0:x0 + 45] = v
            right[y0:y0 + 27, x0:x0 + 60 + sx] = v
        yield dict(it=it, cost=cost, kernel=(kx, ky), search=(sx, sy), left=left, right=right, kind=kind)


def bm_float_corner_cases(n, seed):
    """The corners of calc_disparity on float rasters that the campaign of round 5 found thinly covered (its one mismatch in 26 000 cases was a
    1 x 7 window with ONE disparity on 12-decade data): one to a few disparities, windows with a side of 1, data whose running box sums cancel
    (many decades, both signs), squares that underflow to zero in float32, blocks of exact zeros (infinite NCC precisions: NaN costs), constant
    images (every cost ties), tiny rasters."""
    rng = np.random.default_rng([seed, 0xC0])
    sides = [1, 3, 5, 7, 9, 11, 13]
    for it in range(n):
        cost = int(rng.integers(0, 3))
        kx = int(rng.choice(sides))
        ky = kx if rng.random() < 0.5 else int(rng.choice(sides))
        sx = int(rng.choice([1, 1, 2, 3, 4, 8, 17]))
        sy = int(rng.choice([1, 1, 1, 2, 3]))
        w = int(rng.integers(kx + 1, 130))
        h = int(rng.integers(ky + 1, 90))
        kind = int(rng.integers(0, 7))
        shape = (h + sy - 1, w + sx - 1)
        base = rng.random(shape)
        if kind == 0:
            right = base * 10.0 ** (rng.random(shape) * 12.0 - 3.0)                        # 12 decades, positive
        elif kind == 1:
            right = (base - 0.5) * 10.0 ** (rng.random(shape) * 12.0 - 4.0)                # 12 decades, both signs
        elif kind == 2:
            right = base * 1e-21 * 10.0 ** (rng.random(shape) * 4.0)                       # squares underflow (denormal or zero) in float32
        elif kind == 3:
            right = np.full(shape, float(rng.choice([0.0, 1.5, -7.25, 1e-3])))             # a constant image
        elif kind == 4:
            right = base * 100.0
            for _ in range(int(rng.integers(1, 4))):                                       # blocks of exact zeros
                y0, x0 = int(rng.integers(0, shape[0])), int(rng.integers(0, shape[1]))
                right[y0:y0 + int(rng.integers(1, 30)), x0:x0 + int(rng.integers(1, 60))] = 0.0
        elif kind == 5:
            right = np.where(rng.random(shape) < 0.9, 0.0, base * 10.0 ** (rng.random(shape) * 8.0))      # sparse: mostly zero, a few large values
        else:
            right = np.floor(base * 4.0) * 10.0 ** rng.integers(-6, 9)                     # few distinct levels: many exact ties
        right = right.astype(np.float32)
        d = (int(rng.integers(0, sx)), int(rng.integers(0, sy)))
        left = right[d[1]:d[1] + h, d[0]:d[0] + w].copy()
        if rng.random() < 0.5:
            left *= np.float32(1.0 + (rng.random() - 0.5) * rng.choice([0.0, 1e-6, 1e-2]))
        yield dict(it=it, cost=cost, kernel=(kx, ky), search=(sx, sy), left=left, right=right, kind=kind)


def batch_cases(n, seed):
    """pyramid_correlate_batch: the scenes of pyramid_cases cut into grids of tiles (equal tiles + smaller ones at the right / bottom, so
    that groups and lone tiles both occur), every prefilter and cost, integer and float scenes."""
    rng = np.random.default_rng([seed, 99])
    for c in pyramid_cases(n, seed, prefilters=(0, 1, 2), costs=(0, 1, 2), float_scene=0.5):
        H, W = c["left"].shape
        tw, th = int(rng.integers(40, 140)), int(rng.integers(40, 120))
        boxes = [(x, y, min(tw, W - x), min(th, H - y)) for y in range(0, H, th) for x in range(0, W, tw)]
        boxes = [b for b in boxes if b[2] >= 16 and b[3] >= 16]
        if len(boxes) > 12:
            boxes = [boxes[i] for i in sorted(rng.choice(len(boxes), 12, replace=False))]
        c["boxes"] = boxes
        yield c
