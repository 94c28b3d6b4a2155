"""Synthetic stereo pairs for parity tests and the benchmark.

Integer-valued float32 in [0,255] from a counter-based SplitMix64 (no dependence on boost::rand48).
Left = noise (seed 10); right canvas = independent noise (seed 11) onto which every `block` x `block`
block of the left image is pasted at x + centre + s_b with s_b uniform in [-jitter, jitter] (seed 12),
so the ground-truth disparity index is centre + s_b inside the search box, with occlusion-like seams
at block borders.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, offset=0):
    """n outputs of SplitMix64 started at `seed` (output i uses state seed + (offset+i+1)*golden)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(offset + 1, offset + n + 1, dtype=np.uint64) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def noise_u8(seed, h, w):
    """floor(256*u) with u = top bits of SplitMix64 -> uint8 image (h, w)."""
    return (splitmix64(seed, h * w) >> np.uint64(56)).astype(np.uint8).reshape(h, w)


def stereo_pair(w, h, sx, sy=1, block=256, jitter=None, seeds=(10, 11, 12), smooth=False):
    """Returns (left (h,w) f32, right (h+sy-1, w+sx-1) f32, truth_dx (h,w) int32)."""
    centre = (sx - 1) // 2
    if jitter is None:
        jitter = max(0, min(48, centre - 1, sx - 1 - centre - 1)) if sx > 2 else 0
    left = noise_u8(seeds[0], h, w)
    if smooth:  # "textured-smooth" variant: 3x3 box blur re-quantised to integers (more near-ties)
        p = np.pad(left.astype(np.uint32), 1, mode="edge")
        acc = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
        left = (acc // 9).astype(np.uint8)
    rw, rh = w + sx - 1, h + sy - 1
    right = noise_u8(seeds[1], rh, rw)
    nbx, nby = (w + block - 1) // block, (h + block - 1) // block
    r = splitmix64(seeds[2], nbx * nby)
    shifts = (r % np.uint64(2 * jitter + 1)).astype(np.int64) - jitter
    truth = np.zeros((h, w), np.int32)
    for by in range(nby):
        for bx in range(nbx):
            s = int(shifts[by * nbx + bx])
            y0, y1 = by * block, min(h, (by + 1) * block)
            x0, x1 = bx * block, min(w, (bx + 1) * block)
            right[y0:y1, x0 + centre + s:x1 + centre + s] = left[y0:y1, x0:x1]
            truth[y0:y1, x0:x1] = centre + s
    return left.astype(np.float32), right.astype(np.float32), truth


def noise_f32(seed, h, w, lo=0.0, hi=1.0):
    """Non-integer float texture (mismatch-rate runs; bit-exact parity is not defined on these)."""
    u = (splitmix64(seed, h * w) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(h, w)


def stereo_pair_rows(w, h, sx, r0, r1, block=256, jitter=None, seeds=(10, 11, 12)):
    """Rows [r0, r1) of stereo_pair(w, h, sx, 1, block, jitter, seeds) without generating the rest (one search row, so a
    right row depends on the same left row only): for pairs too large to build whole on every rank."""
    centre = (sx - 1) // 2
    if jitter is None:
        jitter = max(0, min(48, centre - 1, sx - 1 - centre - 1)) if sx > 2 else 0
    n = r1 - r0
    rw = w + sx - 1
    left = (splitmix64(seeds[0], n * w, offset=r0 * w) >> np.uint64(56)).astype(np.uint8).reshape(n, w)
    right = (splitmix64(seeds[1], n * rw, offset=r0 * rw) >> np.uint64(56)).astype(np.uint8).reshape(n, rw)
    nbx, nby = (w + block - 1) // block, (h + block - 1) // block
    r = splitmix64(seeds[2], nbx * nby)
    shifts = (r % np.uint64(2 * jitter + 1)).astype(np.int64) - jitter
    truth = np.zeros((n, w), np.int32)
    for by in range(r0 // block, (r1 + block - 1) // block):
        y0, y1 = max(by * block, r0) - r0, min(h, (by + 1) * block, r1) - r0
        for bx in range(nbx):
            s = int(shifts[by * nbx + bx])
            x0, x1 = bx * block, min(w, (bx + 1) * block)
            right[y0:y1, x0 + centre + s:x1 + centre + s] = left[y0:y1, x0:x1]
            truth[y0:y1, x0:x1] = centre + s
    return left.astype(np.float32), right.astype(np.float32), truth
