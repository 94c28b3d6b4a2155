"""ctypes binding of oracle/libvw_oracle.so (see vw_oracle.h). TEST INFRASTRUCTURE ONLY."""
import ctypes
import threading
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_LOCK = threading.Lock()

ABSOLUTE_DIFFERENCE, SQUARED_DIFFERENCE, CROSS_CORRELATION = 0, 1, 2
VALID = np.iinfo(np.int32).max

EDGE_CONSTANT, EDGE_ZERO = 0, 1
PREFILTER_NONE, PREFILTER_MEANSUB, PREFILTER_LOG = 0, 1, 2

__all__ = ["build", "lib", "fast_box_sum", "cost_image", "calc_disparity", "calc_disparity_tiled",
           "cross_corr_consistency_check", "ABSOLUTE_DIFFERENCE", "SQUARED_DIFFERENCE",
           "CROSS_CORRELATION", "VALID", "generate_gaussian_kernel", "separable_convolution", "convolution_2d",
           "subsample_mask_by_two", "prefilter_image", "pyramid_smoothing_kernel",
           "EDGE_CONSTANT", "EDGE_ZERO", "PREFILTER_NONE", "PREFILTER_MEANSUB", "PREFILTER_LOG",
           "subdivide_regions", "prefilter_region", "parabola_subpixel", "pyramid_correlate", "disparity_filter",
           "disparity_mask", "u8_convert", "census_transform", "hamming_distance", "SemiGlobalMatcher", "calc_disparity_sgm",
           "pyramid_correlate_sgm", "set_sgm_host_threads", "blob_sizes", "disparity_blob_filter", "set_blob_filter_area", "cross_corr_consistency_check_diff", "set_lr_disp_diff", "CENSUS_TRANSFORM", "TERNARY_CENSUS_TRANSFORM", "SUBPIXEL_NONE", "SUBPIXEL_PARABOLA", "SUBPIXEL_LINEAR",
           "SUBPIXEL_POLY4", "SUBPIXEL_COSINE", "SUBPIXEL_LC_BLEND"]


def build(force=False):
    so = os.path.join(_HERE, "libvw_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("vw_oracle.cc", "vw_sgm_oracle.cc", "vw_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return so


def lib():
    """The oracle's shared library with its prototypes.  Thread safe: callers run the oracle on pools of host threads (bench.py's checkers and
    CPU baselines), and a second thread must not see the handle before every prototype is declared (an undeclared double argument raises
    ctypes.ArgumentError — which a worker thread dies of silently)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LIB_LOCK:
        if _LIB is None:
            _LIB = _load()
    return _LIB


def _load():
    so = ctypes.CDLL(build())
    P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    so.vwo_fast_box_sum_f32.argtypes = [P, I, I, I, I, P]
    so.vwo_fast_box_sum_f64.argtypes = [P, I, I, I, I, P]
    so.vwo_cost_image.argtypes = [I, P, P, I, I, P]
    so.vwo_calc_disparity.argtypes = [I, P, I, I, L, P, I, I, L, I, I, I, I, P]
    so.vwo_calc_disparity_tiled.argtypes = [I, P, I, I, P, I, I, I, I, I, I, P, I, I, I, P]
    so.vwo_cross_corr_consistency_check.argtypes = [P, I, I, P, I, I, ctypes.c_float]
    D, F = ctypes.c_double, ctypes.c_float
    so.vwo_generate_gaussian_kernel_f32.argtypes = [D, I, P, I]
    so.vwo_generate_gaussian_kernel_f64.argtypes = [D, I, P, I]
    so.vwo_separable_convolution_f32.argtypes = [P, I, I, P, I, I, P, I, I, I, I, P]
    so.vwo_separable_convolution_f64.argtypes = [P, I, I, P, I, I, P, I, I, I, I, P]
    so.vwo_convolution_2d_f32.argtypes = [P, I, I, P, I, I, I, I, I, P]
    so.vwo_convolution_2d_f64.argtypes = [P, I, I, P, I, I, I, I, I, P]
    so.vwo_subsample_mask_by_two.argtypes = [P, I, I, P]
    so.vwo_prefilter_image.argtypes = [P, I, I, I, F, P]
    so.vwo_subdivide_regions.argtypes = [P, I, I, I, I, P, I]
    so.vwo_prefilter_region.argtypes = [P, I, I, I, F, I, I, I, I, P]
    so.vwo_parabola_subpixel.argtypes = [P, I, I, P, P, I, I, I, F, I, I, P]
    so.vwo_pyramid_correlate.argtypes = [P, I, I, P, I, I, P, P, I, F, I, I, I, I, I, I, I, I, D, F, I, I, I, I, I, I, P]
    so.vwo_disparity_filter.argtypes = [P, I, I, I, I, D, D, I]
    so.vwo_disparity_mask.argtypes = [P, I, I, P, P, I, I]
    Z = ctypes.c_size_t
    so.vwo_blob_sizes.argtypes = [P, I, I, P]
    so.vwo_disparity_blob_filter.argtypes = [P, I, I, I]
    so.vwo_set_blob_filter_area.argtypes = [I]
    so.vwo_set_blob_filter_area.restype = None
    so.vwo_set_sgm_algorithm.argtypes = [I]
    so.vwo_set_sgm_algorithm.restype = None
    so.vwo_cross_corr_consistency_check_diff.argtypes = [P, I, I, P, I, I, F, P, I, I, I, I]
    so.vwo_set_lr_disp_diff.argtypes = [P, I, I, I, I]
    so.vwo_set_lr_disp_diff.restype = None
    so.vwo_u8_convert.argtypes = [P, I, I, P]
    so.vwo_census_transform.argtypes = [P, I, I, I, I, I, P]
    so.vwo_hamming_distance.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    so.vwo_sgm_create.argtypes = [I, I, I, I, I, I, I, I, I, I, Z, I, I, I, I]
    so.vwo_sgm_create.restype = P
    so.vwo_sgm_destroy.argtypes = [P]
    so.vwo_sgm_destroy.restype = None
    so.vwo_sgm_run.argtypes = [P, P, I, I, P, I, I, P, I, I, P, I, I, P, I, I, P, I]
    so.vwo_sgm_output_size.argtypes = [P, P, P]
    so.vwo_sgm_subpixel.argtypes = [P, P, P]
    so.vwo_sgm_buffer_size.argtypes = [P]
    so.vwo_sgm_buffer_size.restype = Z
    so.vwo_sgm_read.argtypes = [P, P, P, P, P]
    so.vwo_sgm_p1p2.argtypes = [P, P, P]
    so.vwo_pyramid_correlate_sgm.argtypes = [P, I, I, P, I, I, P, P, I, I, I, I, I, I, F, I, I, I, I, I, I, Z, I, I, I, I, I, P]
    so.vwo_calc_disparity_sgm.argtypes = [I, P, I, I, P, I, I, I, I, I, I, I, I, Z, I, P, I, I, P, I, I, P, I, I, P, P, P, P]
    so.vwo_calc_disparity_sgm_p.argtypes = [I, P, I, I, P, I, I, I, I, I, I, I, I, Z, I, P, I, I, P, I, I, P, I, I, I, I, P, P, P, P]
    so.vwo_calc_disparity_sgm_x.argtypes = [I, I, P, I, I, P, I, I, I, I, I, I, I, I, Z, I, P, I, I, P, I, I, P, I, I, I, I, P, P, P, P]
    return so


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fast_box_sum(img, kernel):
    """fast_box_sum<double>(img, Vector2i(kx,ky)); img is (rows, cols) float32 or float64."""
    kx, ky = kernel
    h, w = img.shape
    out = np.empty((h - ky + 1, w - kx + 1), np.float64)
    if img.dtype == np.float64:
        a = np.ascontiguousarray(img)
        rc = lib().vwo_fast_box_sum_f64(_p(a), w, h, kx, ky, _p(out))
    else:
        a = np.ascontiguousarray(img, np.float32)
        rc = lib().vwo_fast_box_sum_f32(_p(a), w, h, kx, ky, _p(out))
    if rc:
        raise ValueError("fast_box_sum: Kernel input not sized with odd values." if rc == -1 else "bad size")
    return out


def cost_image(cost_type, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty(a.shape, np.float64)
    rc = lib().vwo_cost_image(cost_type, _p(a), _p(b), a.shape[1], a.shape[0], _p(out))
    assert rc == 0
    return out


def calc_disparity(cost_type, left, right, kernel, search):
    """left (lh,lw) f32 = the cropped left region; right (>=lh+sy-1, >=lw+sx-1) f32.
    Returns int32 (oh, ow, 3) = {dx, dy, valid}."""
    kx, ky = kernel
    sx, sy = search
    left = np.ascontiguousarray(left, np.float32)
    right = np.ascontiguousarray(right, np.float32)
    lh, lw = left.shape
    rh, rw = right.shape
    out = np.empty((lh - ky + 1, lw - kx + 1, 3), np.int32)
    rc = lib().vwo_calc_disparity(cost_type, _p(left), lw, lh, lw, _p(right), rw, rh, rw, kx, ky, sx, sy, _p(out))
    if rc:
        raise ValueError("vwo_calc_disparity rc=%d" % rc)
    return out


def calc_disparity_tiled(cost_type, left, right, kernel, search, tile=1024, threads=1, max_tiles=0):
    kx, ky = kernel
    sx, sy = search
    left = np.ascontiguousarray(left, np.float32)
    right = np.ascontiguousarray(right, np.float32)
    lh, lw = left.shape
    rh, rw = right.shape
    out = np.zeros((lh - ky + 1, lw - kx + 1, 3), np.int32)
    done = ctypes.c_int64(0)
    rc = lib().vwo_calc_disparity_tiled(cost_type, _p(left), lw, lh, _p(right), rw, rh, kx, ky, sx, sy,
                                        _p(out), tile, threads, max_tiles, ctypes.byref(done))
    if rc:
        raise ValueError("vwo_calc_disparity_tiled rc=%d" % rc)
    return out, done.value


def cross_corr_consistency_check(l2r, r2l, thr):
    l2r = np.ascontiguousarray(l2r, np.int32).copy()
    r2l = np.ascontiguousarray(r2l, np.int32)
    rc = lib().vwo_cross_corr_consistency_check(_p(l2r), l2r.shape[1], l2r.shape[0],
                                                _p(r2l), r2l.shape[1], r2l.shape[0], thr)
    assert rc == 0
    return l2r


def generate_gaussian_kernel(sigma, size=0, dtype=np.float32):
    """generate_gaussian_kernel<KernelT>(kernel, sigma, size), src/vw/Image/Filter.tcc:37-78."""
    out = np.zeros(4096, dtype)
    fn = lib().vwo_generate_gaussian_kernel_f32 if dtype == np.float32 else lib().vwo_generate_gaussian_kernel_f64
    n = fn(float(sigma), int(size), _p(out), out.size)
    assert n >= 0
    return out[:n].copy()


def pyramid_smoothing_kernel():
    """generate_pyramid_smoothing_kernel(), src/vw/Image/Filter.h:89-99: float {1,4,6,4,1}/16."""
    return np.array([1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0], np.float32)


def separable_convolution(img, xk, yk, cx=None, cy=None, edge=EDGE_CONSTANT, subsample=1):
    """separable_convolution_filter(img, xk, yk[, cx, cy], edge) rasterised, then subsample(., s)."""
    dt = np.float64 if img.dtype == np.float64 else np.float32
    a = np.ascontiguousarray(img, dt)
    xk = np.ascontiguousarray(xk, dt)
    yk = np.ascontiguousarray(yk, dt)
    if cx is None:
        cx = (len(xk) - 1) // 2 if len(xk) else 0
    if cy is None:
        cy = (len(yk) - 1) // 2 if len(yk) else 0
    h, w = a.shape
    out = np.empty((1 + (h - 1) // subsample, 1 + (w - 1) // subsample), dt)
    fn = lib().vwo_separable_convolution_f64 if dt == np.float64 else lib().vwo_separable_convolution_f32
    rc = fn(_p(a), w, h, _p(xk), len(xk), cx, _p(yk), len(yk), cy, edge, subsample, _p(out))
    assert rc == 0
    return out


def convolution_2d(img, kernel, ci=None, cj=None, edge=EDGE_CONSTANT):
    """convolution_filter(img, kernel[, ci, cj], edge), src/vw/Image/Convolution.h:105-170."""
    dt = np.float64 if img.dtype == np.float64 else np.float32
    a = np.ascontiguousarray(img, dt)
    k = np.ascontiguousarray(kernel, dt)
    kh, kw = k.shape
    if ci is None:
        ci = (kw - 1) // 2
    if cj is None:
        cj = (kh - 1) // 2
    h, w = a.shape
    out = np.empty((h, w), dt)
    fn = lib().vwo_convolution_2d_f64 if dt == np.float64 else lib().vwo_convolution_2d_f32
    rc = fn(_p(a), w, h, _p(k), kw, kh, ci, cj, edge, _p(out))
    assert rc == 0
    return out


def subsample_mask_by_two(mask):
    m = np.ascontiguousarray(mask, np.uint8)
    h, w = m.shape
    out = np.empty((1 + (h - 1) // 2, 1 + (w - 1) // 2), np.uint8)
    assert lib().vwo_subsample_mask_by_two(_p(m), w, h, _p(out)) == 0
    return out


def prefilter_image(img, mode, width):
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(a)
    assert lib().vwo_prefilter_image(_p(a), a.shape[1], a.shape[0], int(mode), float(width), _p(out)) == 0
    return out


def subdivide_regions(disp, kernel):
    """subdivide_regions(disparity, bounding_box(disparity), list, kernel_size); returns an (n, 8) int32 array of
    {region.min.x, region.min.y, region.max.x, region.max.y, range.min.x, range.min.y, range.max.x, range.max.y}."""
    d = np.ascontiguousarray(disp, np.int32)
    h, w = d.shape[:2]
    cap = 1 << 16
    z = np.zeros((cap, 8), np.int32)
    n = lib().vwo_subdivide_regions(_p(d), w, h, kernel[0], kernel[1], _p(z), cap)
    assert 0 <= n <= cap
    return z[:n].copy()


def prefilter_region(img, mode, width, x0, y0, bw, bh):
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty((bh, bw), np.float32)
    assert lib().vwo_prefilter_region(_p(a), a.shape[1], a.shape[0], int(mode), float(width), x0, y0, bw, bh, _p(out)) == 0
    return out


def parabola_subpixel(disparity, left, right, prefilter_mode, prefilter_width, kernel):
    """parabola_subpixel(...) rasterised over the whole image. disparity: (h, w, 3) float32 PixelMask<Vector2f>."""
    d = np.ascontiguousarray(disparity, np.float32)
    l = np.ascontiguousarray(left, np.float32)
    r = np.ascontiguousarray(right, np.float32)
    h, w = l.shape
    assert d.shape == (h, w, 3)
    out = np.empty((h, w, 3), np.float32)
    rc = lib().vwo_parabola_subpixel(_p(d), w, h, _p(l), _p(r), r.shape[1], r.shape[0], int(prefilter_mode),
                                     float(prefilter_width), kernel[0], kernel[1], _p(out))
    assert rc == 0
    return out


def pyramid_correlate(left, right, left_mask, right_mask, prefilter_mode, prefilter_width, search_region, kernel_size,
                      cost_type, corr_timeout, seconds_per_op, consistency_threshold, filter_half_kernel,
                      max_pyramid_levels, bbox=None):
    """One tile of pyramid_correlate(..., VW_CORRELATION_BM) rasterised over bbox = (x, y, w, h) (default: whole left
    image).  search_region = (minx, miny, maxx, maxy), half-open.  Returns (h, w, 3) float32 PixelMask<Vector2f>."""
    l = np.ascontiguousarray(left, np.float32)
    r = np.ascontiguousarray(right, np.float32)
    lm = None if left_mask is None else np.ascontiguousarray(left_mask, np.uint8)
    rm = None if right_mask is None else np.ascontiguousarray(right_mask, np.uint8)
    if bbox is None:
        bbox = (0, 0, l.shape[1], l.shape[0])
    out = np.zeros((bbox[3], bbox[2], 3), np.float32)
    rc = lib().vwo_pyramid_correlate(_p(l), l.shape[1], l.shape[0], _p(r), r.shape[1], r.shape[0],
                                     None if lm is None else _p(lm), None if rm is None else _p(rm),
                                     int(prefilter_mode), float(prefilter_width),
                                     search_region[0], search_region[1], search_region[2], search_region[3],
                                     kernel_size[0], kernel_size[1], int(cost_type), int(corr_timeout), float(seconds_per_op),
                                     float(consistency_threshold), int(filter_half_kernel), int(max_pyramid_levels),
                                     bbox[0], bbox[1], bbox[2], bbox[3], _p(out))
    if rc:
        raise ValueError("vwo_pyramid_correlate rc=%d" % rc)
    return out


def disparity_filter(disp, half_h, half_v, pixel_thr, rej_thr, cleanup):
    d = np.ascontiguousarray(disp, np.int32).copy()
    assert lib().vwo_disparity_filter(_p(d), d.shape[1], d.shape[0], half_h, half_v, pixel_thr, rej_thr, int(cleanup)) == 0
    return d


def disparity_mask(disp, left_mask, right_mask):
    d = np.ascontiguousarray(disp, np.int32).copy()
    lm = np.ascontiguousarray(left_mask, np.uint8)
    rm = np.ascontiguousarray(right_mask, np.uint8)
    assert lib().vwo_disparity_mask(_p(d), d.shape[1], d.shape[0], _p(lm), _p(rm), rm.shape[1], rm.shape[0]) == 0
    return d


# ---- semi-global matching ---------------------------------------------------------------------------------------------

CENSUS_TRANSFORM, TERNARY_CENSUS_TRANSFORM = 3, 4
SUBPIXEL_NONE, SUBPIXEL_PARABOLA, SUBPIXEL_LINEAR, SUBPIXEL_POLY4, SUBPIXEL_COSINE, SUBPIXEL_LC_BLEND = range(6)


def u8_convert(img):
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty(a.shape, np.uint8)
    assert lib().vwo_u8_convert(_p(a), a.shape[1], a.shape[0], _p(out)) == 0
    return out


def census_transform(img, kernel, ternary=False, threshold=5):
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty((a.shape[0] - kernel + 1, a.shape[1] - kernel + 1), np.uint64)
    assert lib().vwo_census_transform(_p(a), a.shape[1], a.shape[0], kernel, int(ternary), threshold, _p(out)) == 0
    return out


def hamming_distance(a, b):
    return lib().vwo_hamming_distance(int(a), int(b))


def _opt(a, dtype):
    if a is None:
        return None, 0, 0
    a = np.ascontiguousarray(a, dtype)
    return a, a.shape[1], a.shape[0]


class SemiGlobalMatcher:
    """SemiGlobalMatcher(cost_type, use_mgm, min_dx, min_dy, max_dx, max_dy, kernel, subpixel, search_buffer,
    memory_limit_mb, p1, p2, ternary_census_threshold) — src/vw/Stereo/SGM.h:108-121."""

    def __init__(self, cost_type, min_dx, min_dy, max_dx, max_dy, kernel=5, subpixel=SUBPIXEL_LC_BLEND, search_buffer=(2, 2),
                 memory_limit_mb=6000, p1=0, p2=0, ternary_thr=5, num_threads=1, use_mgm=False):
        self._h = lib().vwo_sgm_create(int(cost_type), int(bool(use_mgm)), min_dx, min_dy, max_dx, max_dy, kernel, int(subpixel),
                                       search_buffer[0], search_buffer[1], memory_limit_mb, p1, p2, ternary_thr, num_threads)
        if not self._h:
            raise ValueError("vwo_sgm_create: unsupported cost type / kernel size")
        self._geom = (min_dx, min_dy, max_dx, max_dy, kernel)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vwo_sgm_destroy(self._h)
            self._h = None

    def semi_global_matching_func(self, left, right, left_mask=None, right_mask=None, prev_disparity=None):
        l = np.ascontiguousarray(left, np.uint8)
        r = np.ascontiguousarray(right, np.uint8)
        lm, lmw, lmh = _opt(left_mask, np.uint8)
        rm, rmw, rmh = _opt(right_mask, np.uint8)
        pd, pw, ph = _opt(prev_disparity, np.int32)
        out = np.zeros((l.shape[0], l.shape[1], 3), np.int32)
        rc = lib().vwo_sgm_run(self._h, _p(l), l.shape[1], l.shape[0], _p(r), r.shape[1], r.shape[0],
                               None if lm is None else _p(lm), lmw, lmh, None if rm is None else _p(rm), rmw, rmh,
                               None if pd is None else _p(pd), pw, ph, _p(out), l.shape[0] * l.shape[1])
        if rc:
            raise ValueError("vwo_sgm_run rc=%d" % rc)
        ow, oh = ctypes.c_int(), ctypes.c_int()
        lib().vwo_sgm_output_size(self._h, ctypes.byref(ow), ctypes.byref(oh))
        self.shape = (oh.value, ow.value)
        return out.reshape(-1)[:oh.value * ow.value * 3].reshape(oh.value, ow.value, 3).copy()

    def create_disparity_view_subpixel(self, integer_disparity):
        d = np.ascontiguousarray(integer_disparity, np.int32)
        out = np.empty(d.shape, np.float32)
        assert lib().vwo_sgm_subpixel(self._h, _p(d), _p(out)) == 0
        return out

    def buffers(self):
        """(bounds (h,w,4) int32, starts (h,w) uint64, cost uint8[n], accum uint16[n]) after a run."""
        h, w = self.shape
        n = lib().vwo_sgm_buffer_size(self._h)
        b = np.empty((h, w, 4), np.int32)
        s = np.empty((h, w), np.uint64)
        c = np.empty(n, np.uint8)
        a = np.empty(n, np.uint16)
        lib().vwo_sgm_read(self._h, _p(b), _p(s), _p(c), _p(a))
        return b, s, c, a

    def p1p2(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        lib().vwo_sgm_p1p2(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value


def calc_disparity_sgm(cost_type, left, right, search_volume, kernel, subpixel=SUBPIXEL_LC_BLEND, search_buffer=(2, 2),
                       memory_limit_mb=6000, left_mask=None, right_mask=None, prev_disparity=None, num_threads=1, p1=0, p2=0,
                       use_mgm=False, allow_block_cost=False):
    """calc_disparity_sgm on cropped regions: left (lh, lw) float32, right (lh+sy, lw+sx) float32.
    Returns (integer disparity (oh, ow, 3) int32, sub-pixel disparity (oh, ow, 3) float32)."""
    l = np.ascontiguousarray(left, np.float32)
    r = np.ascontiguousarray(right, np.float32)
    lm, lmw, lmh = _opt(left_mask, np.uint8)
    rm, rmw, rmh = _opt(right_mask, np.uint8)
    pd, pw, ph = _opt(prev_disparity, np.int32)
    out = np.zeros((l.shape[0], l.shape[1], 3), np.int32)
    sub = np.zeros((l.shape[0], l.shape[1], 3), np.float32)
    ow, oh = ctypes.c_int(), ctypes.c_int()
    # cost types 0 / 1 = the MAD block cost (SGM.cc:1651-1738), refused like the reference's throw (:1887-1892) unless opted in
    lib().vwo_set_sgm_allow_block_cost(int(bool(allow_block_cost)))
    rc = lib().vwo_calc_disparity_sgm_x(int(cost_type), int(bool(use_mgm)), _p(l), l.shape[1], l.shape[0], _p(r), r.shape[1], r.shape[0],
                                        search_volume[0], search_volume[1], kernel, int(subpixel), search_buffer[0], search_buffer[1],
                                        memory_limit_mb, num_threads, None if lm is None else _p(lm), lmw, lmh,
                                        None if rm is None else _p(rm), rmw, rmh, None if pd is None else _p(pd), pw, ph,
                                        int(p1), int(p2), _p(out), _p(sub), ctypes.byref(ow), ctypes.byref(oh))
    if rc:
        raise ValueError("vwo_calc_disparity_sgm rc=%d" % rc)
    n = ow.value * oh.value * 3
    return (out.reshape(-1)[:n].reshape(oh.value, ow.value, 3).copy(), sub.reshape(-1)[:n].reshape(oh.value, ow.value, 3).copy())


def set_sgm_host_threads(n):
    """Host threads for the SGM oracle's path lines / cost rows (run time only; identical results)."""
    lib().vwo_set_sgm_host_threads(int(n))


def pyramid_correlate_sgm(left, right, left_mask, right_mask, search_region, kernel, cost_type, consistency_threshold=-1.0,
                          min_consistency_level=0, filter_half_kernel=0, max_pyramid_levels=5, subpixel=SUBPIXEL_LC_BLEND,
                          search_buffer=(2, 2), memory_limit_mb=6000, num_threads=1, bbox=None, algorithm=1):
    """One tile of pyramid_correlate(..., VW_CORRELATION_SGM | _MGM (algorithm=2) | _FINAL_MGM (3)): returns (h, w, 3) float32
    sub-pixel PixelMask<Vector2f>."""
    l = np.ascontiguousarray(left, np.float32)
    r = np.ascontiguousarray(right, np.float32)
    lm = None if left_mask is None else np.ascontiguousarray(left_mask, np.uint8)
    rm = None if right_mask is None else np.ascontiguousarray(right_mask, np.uint8)
    if bbox is None:
        bbox = (0, 0, l.shape[1], l.shape[0])
    out = np.zeros((bbox[3], bbox[2], 3), np.float32)
    lib().vwo_set_sgm_algorithm(int(algorithm))
    rc = lib().vwo_pyramid_correlate_sgm(_p(l), l.shape[1], l.shape[0], _p(r), r.shape[1], r.shape[0],
                                         None if lm is None else _p(lm), None if rm is None else _p(rm),
                                         search_region[0], search_region[1], search_region[2], search_region[3], int(kernel), int(cost_type),
                                         float(consistency_threshold), int(min_consistency_level), int(filter_half_kernel),
                                         int(max_pyramid_levels), int(subpixel), search_buffer[0], search_buffer[1], memory_limit_mb,
                                         num_threads, bbox[0], bbox[1], bbox[2], bbox[3], _p(out))
    lib().vwo_set_sgm_algorithm(1)
    if rc:
        raise ValueError("vwo_pyramid_correlate_sgm rc=%d" % rc)
    return out


def blob_sizes(disp):
    d = np.ascontiguousarray(disp, np.int32)
    out = np.zeros(d.shape[:2], np.uint32)
    assert lib().vwo_blob_sizes(_p(d), d.shape[1], d.shape[0], _p(out)) == 0
    return out


def disparity_blob_filter(disp, area):
    d = np.ascontiguousarray(disp, np.int32).copy()
    assert lib().vwo_disparity_blob_filter(_p(d), d.shape[1], d.shape[0], int(area)) == 0
    return d


def set_blob_filter_area(area):
    """blob_filter_area used by the following pyramid_correlate / pyramid_correlate_sgm calls of this thread."""
    lib().vwo_set_blob_filter_area(int(area))


def cross_corr_consistency_check_diff(l2r, r2l, thr, diff, ul=(0, 0)):
    """In place on l2r and diff ((rows, cols, 2) float32 PixelMask<float>)."""
    assert l2r.flags.c_contiguous and l2r.dtype == np.int32 and diff.flags.c_contiguous and diff.dtype == np.float32
    r = np.ascontiguousarray(r2l, np.int32)
    rc = lib().vwo_cross_corr_consistency_check_diff(_p(l2r), l2r.shape[1], l2r.shape[0], _p(r), r.shape[1], r.shape[0], float(thr),
                                                     _p(diff), diff.shape[1], diff.shape[0], int(ul[0]), int(ul[1]))
    if rc:
        raise ValueError("lr_disp_diff does not contain the checked region")
    return l2r


def set_lr_disp_diff(diff, region_ul=(0, 0)):
    """lr_disp_diff buffer ((rows, cols, 2) float32, None = off) of the following pyramid calls of this thread."""
    if diff is None:
        lib().vwo_set_lr_disp_diff(None, 0, 0, 0, 0)
    else:
        assert diff.flags.c_contiguous and diff.dtype == np.float32 and diff.shape[2] == 2
        lib().vwo_set_lr_disp_diff(_p(diff), diff.shape[1], diff.shape[0], int(region_ul[0]), int(region_ul[1]))
