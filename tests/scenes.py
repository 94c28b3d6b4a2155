"""Synthetic scenes of the reference's own stereo tests, rebuilt without boost/vw (test infrastructure only).

  rand48_noise      boost::rand48 gen(10) + uniform_noise_view (src/vw/Image/UtilityViews.h:148-157): the LCG
                    x' = (0x5DEECE66D x + 0xB) mod 2^48 seeded (seed << 16) | 0x330E, output x >> 17 scaled by 2^-31,
                    raster order, multiplied by ChannelRange<T>::max() and truncated for integer channels.
  affine_bicubic    transform(img, AffineTransform(diag(sx, sy), (tx, ty)), ConstantEdgeExtension(),
                    BicubicInterpolation()) — src/vw/Image/Interpolation.h:138-186 (weights, 0.25 scale, exact-integer
                    shortcut, round+clamp for integer channels).
  pyramid_scene     the fixture of src/vw/Stereo/tests/TestPyramidCorrelationView.cxx:47-64.
"""
import numpy as np


def rand48_noise(cols, rows, seed=10):
    a, c, m = 0x5DEECE66D, 0xB, (1 << 48) - 1
    x = ((seed << 16) | 0x330E) & m
    out = np.empty(rows * cols, np.float64)
    for i in range(rows * cols):
        x = (a * x + c) & m
        out[i] = (x >> 17) / 2147483648.0
    return out.reshape(rows, cols)


def affine_bicubic(img, sx, sy, tx, ty, integer_max=None):
    """out(i, j) = bicubic(img, ((i - tx)/sx, (j - ty)/sy)) with constant edge extension; img is (rows, cols)."""
    rows, cols = img.shape
    src = img.astype(np.float64)
    ii = (np.arange(cols, dtype=np.float64) - tx) / sx
    jj = (np.arange(rows, dtype=np.float64) - ty) / sy
    x = np.floor(ii).astype(np.int64)
    y = np.floor(jj).astype(np.int64)
    nx, ny = ii - x, jj - y

    def weights(n):
        return [((2 - n) * n - 1) * n, (3 * n - 5) * n * n + 2, ((4 - 3 * n) * n + 1) * n, (n - 1) * n * n]

    s, t = weights(nx), weights(ny)
    res = np.zeros((rows, cols), np.float64)
    for b in range(4):
        yy = np.clip(y - 1 + b, 0, rows - 1)
        row = np.zeros((rows, cols), np.float64)
        for a_ in range(4):
            xx = np.clip(x - 1 + a_, 0, cols - 1)
            term = s[a_][None, :] * src[yy[:, None], xx[None, :]]
            row = term if a_ == 0 else row + term
        res = t[b][:, None] * row if b == 0 else res + t[b][:, None] * row
    res *= 0.25
    exact = (nx == 0)[None, :] & (ny == 0)[:, None]
    plain = src[np.clip(y, 0, rows - 1)[:, None], np.clip(x, 0, cols - 1)[None, :]]
    res = np.where(exact, plain, res)
    if integer_max is not None:
        res = np.clip(np.rint(res), 0, integer_max)
    return res


def pyramid_scene(channel="u8"):
    """(left, right, scale (sx, sy), translation (tx, ty), search box corners) as float32 images."""
    noise = rand48_noise(300, 200)
    if channel == "u8":
        left = np.floor(255.0 * noise)
        right = affine_bicubic(left, 0.9, 0.95, 15.0, 5.0, integer_max=255)
    elif channel == "i16":
        left = np.floor(32767.0 * noise)
        right = affine_bicubic(left, 0.9, 0.95, 15.0, 5.0, integer_max=32767)
    else:
        left = noise.astype(np.float32).astype(np.float64)
        right = affine_bicubic(left, 0.9, 0.95, 15.0, 5.0)
    # BBox2i(BBox2(-1.5 t, 1.5 t)): the corners are truncated towards zero
    return left.astype(np.float32), right.astype(np.float32), (0.9, 0.95), (15.0, 5.0), (-22, -7, 22, 7)


def pyramid_score(disp, scale, translation):
    """check_error of TestPyramidCorrelationView.cxx:66-84: (correct / valid, valid / all)."""
    rows, cols = disp.shape[:2]
    i = np.arange(cols, dtype=np.float64)[None, :]
    j = np.arange(rows, dtype=np.float64)[:, None]
    ox = np.rint(scale[0] * i + translation[0] - i) + 0 * j
    oy = np.rint(scale[1] * j + translation[1] - j) + 0 * i
    valid = disp[..., 2] != 0
    good = valid & (disp[..., 0] == ox) & (disp[..., 1] == oy)
    return good.sum() / max(valid.sum(), 1), valid.sum() / (rows * cols)
