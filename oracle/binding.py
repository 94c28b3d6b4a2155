"""ctypes binding of oracle/libvw_oracle.so (see vw_oracle.h). TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABSOLUTE_DIFFERENCE, SQUARED_DIFFERENCE, CROSS_CORRELATION = 0, 1, 2
VALID = np.iinfo(np.int32).max

__all__ = ["build", "lib", "fast_box_sum", "cost_image", "calc_disparity", "calc_disparity_tiled",
           "cross_corr_consistency_check", "ABSOLUTE_DIFFERENCE", "SQUARED_DIFFERENCE",
           "CROSS_CORRELATION", "VALID"]


def build(force=False):
    so = os.path.join(_HERE, "libvw_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("vw_oracle.cc", "vw_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        _LIB.vwo_fast_box_sum_f32.argtypes = [P, I, I, I, I, P]
        _LIB.vwo_fast_box_sum_f64.argtypes = [P, I, I, I, I, P]
        _LIB.vwo_cost_image.argtypes = [I, P, P, I, I, P]
        _LIB.vwo_calc_disparity.argtypes = [I, P, I, I, L, P, I, I, L, I, I, I, I, P]
        _LIB.vwo_calc_disparity_tiled.argtypes = [I, P, I, I, P, I, I, I, I, I, I, P, I, I, I, P]
        _LIB.vwo_cross_corr_consistency_check.argtypes = [P, I, I, P, I, I, ctypes.c_float]
    return _LIB


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
