"""Host-side mirror of the reference's stereo entry points on the block-matching hot path.

Same names, argument meaning and error behaviour as vw::stereo; every function hands
rasterised images to libvwgpu.so through the C ABI (include/vwgpu.h).  Inputs may be
  * torch CUDA tensors  -> device entry points, asynchronous on the current torch stream, result = CUDA tensor;
  * numpy arrays        -> host entry points (H2D, kernels, D2H), result = numpy array.
There is no CPU implementation here: without the HIP library or a GPU these functions raise.
"""
import ctypes

import numpy as np

from . import core
from .core import ArgumentErr, BBox2i, CostFunctionType

try:  # torch is plumbing (device memory, streams); the host-pointer path works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _ctx_for(x, ctx):
    if ctx is not None:
        return ctx
    dev = x.device.index if _is_tensor(x) and x.is_cuda else 0
    return core.default_context(dev or 0)


def calc_disparity(cost_type, left_in, right_in, left_region, search_volume, kernel_size, ctx=None):
    """vw::stereo::calc_disparity (src/vw/Stereo/Correlation.h:50-57, Correlation.cc:330-375).

    left_in / right_in: (rows, cols) float32 images (PixelGray<float>).  left_region: BBox2i inside the left
    image.  search_volume = (sx, sy) >= 1, kernel_size = (kx, ky) odd.  The right image must cover
    left_region grown by search_volume - 1 on the max side (the reference crops it so, :356-359).
    Returns (rows-ky+1, cols-kx+1, 3) int32 = PixelMask<Vector2i> {dx, dy, valid (INT32_MAX|0)}.
    """
    kx, ky = int(kernel_size[0]), int(kernel_size[1])
    sx, sy = int(search_volume[0]), int(search_volume[1])
    if left_in.ndim != 2 or right_in.ndim != 2:
        raise ArgumentErr("calc_disparity: images must be 2-D (rows, cols)")
    x0, y0 = left_region.min
    x1, y1 = left_region.max
    if x0 < 0 or y0 < 0 or x1 > left_in.shape[1] or y1 > left_in.shape[0]:
        raise ArgumentErr("calc_disparity: Region not inside left image.")
    rx1, ry1 = x1 + sx - 1, y1 + sy - 1
    if rx1 > right_in.shape[1] or ry1 > right_in.shape[0]:
        raise ArgumentErr("calc_disparity: right image does not cover the search region")
    lw, lh = x1 - x0, y1 - y0
    ctx = _ctx_for(left_in, ctx)
    lib = ctx._lib
    ow, oh = lw - kx + 1, lh - ky + 1
    if _is_tensor(left_in):
        if not (left_in.is_cuda and right_in.is_cuda):
            raise ArgumentErr("calc_disparity: torch inputs must be CUDA tensors (no CPU path)")
        if left_in.dtype != torch.float32 or right_in.dtype != torch.float32:
            raise ArgumentErr("calc_disparity: images must be float32")
        if left_in.stride(1) != 1 or right_in.stride(1) != 1:
            left_in, right_in = left_in.contiguous(), right_in.contiguous()
        l = left_in[y0:y1, x0:x1]
        r = right_in[y0:ry1, x0:rx1]
        out = torch.empty((max(oh, 0), max(ow, 0), 3), dtype=torch.int32, device=left_in.device)
        ctx.set_stream(torch.cuda.current_stream(left_in.device).cuda_stream)
        rc = lib.vwgpu_calc_disparity_dev(ctx._h, int(cost_type), l.data_ptr(), lw, lh, l.stride(0),
                                          r.data_ptr(), rx1 - x0, ry1 - y0, r.stride(0),
                                          kx, ky, sx, sy, out.data_ptr(), 0)
        ctx.check(rc)
        return out
    l = np.ascontiguousarray(left_in[y0:y1, x0:x1], np.float32)
    r = np.ascontiguousarray(right_in[y0:ry1, x0:rx1], np.float32)
    out = np.empty((max(oh, 0), max(ow, 0), 3), np.int32)
    rc = lib.vwgpu_calc_disparity(ctx._h, int(cost_type), l.ctypes.data, lw, lh, lw,
                                  r.ctypes.data, r.shape[1], r.shape[0], r.shape[1],
                                  kx, ky, sx, sy, out.ctypes.data, 0)
    ctx.check(rc)
    return out


def cross_corr_consistency_check(l2r, r2l, cross_corr_threshold, ctx=None):
    """vw::stereo::cross_corr_consistency_check (src/vw/Stereo/Correlate.cc:1441-1502), IN PLACE on l2r.

    l2r, r2l: (rows, cols, 3) int32 PixelMask<Vector2i> images."""
    ctx = _ctx_for(l2r, ctx)
    lib = ctx._lib
    if _is_tensor(l2r):
        if not (l2r.is_cuda and r2l.is_cuda and l2r.is_contiguous() and r2l.is_contiguous()):
            raise ArgumentErr("cross_corr_consistency_check: contiguous CUDA tensors required")
        ctx.set_stream(torch.cuda.current_stream(l2r.device).cuda_stream)
        rc = lib.vwgpu_cross_corr_consistency_check_dev(ctx._h, l2r.data_ptr(), l2r.shape[1], l2r.shape[0], 0,
                                                        r2l.data_ptr(), r2l.shape[1], r2l.shape[0], 0,
                                                        float(cross_corr_threshold))
        ctx.check(rc)
        return l2r
    if not (l2r.flags.c_contiguous and l2r.dtype == np.int32):
        raise ArgumentErr("cross_corr_consistency_check: l2r must be a contiguous int32 array (modified in place)")
    r2l = np.ascontiguousarray(r2l, np.int32)
    rc = lib.vwgpu_cross_corr_consistency_check(ctx._h, l2r.ctypes.data, l2r.shape[1], l2r.shape[0], 0,
                                                r2l.ctypes.data, r2l.shape[1], r2l.shape[0], 0,
                                                float(cross_corr_threshold))
    ctx.check(rc)
    return l2r


def parabola_subpixel(disparity, left_image, right_image, prefilter_mode, prefilter_width, kernel_size, ctx=None):
    """vw::stereo::parabola_subpixel (src/vw/Stereo/ParabolaSubpixelView.h:112-117) rasterised over the whole image.

    disparity: (rows, cols, 3) float32 PixelMask<Vector2f> {dx, dy, valid}; same rows/cols as left_image (the
    reference asserts this, ParabolaSubpixelView.h:67-69).  Returns the refined disparity in the same layout."""
    kx, ky = int(kernel_size[0]), int(kernel_size[1])
    if disparity.ndim != 3 or disparity.shape[2] != 3 or tuple(disparity.shape[:2]) != tuple(left_image.shape):
        raise ArgumentErr("SubpixelView: Disparity image must match left image.")
    h, w = left_image.shape
    rh, rw = right_image.shape
    ctx = _ctx_for(left_image, ctx)
    lib = ctx._lib
    if _is_tensor(left_image):
        d, l, r = disparity.contiguous(), left_image.contiguous(), right_image.contiguous()
        if not (d.is_cuda and l.is_cuda and r.is_cuda) or d.dtype != torch.float32:
            raise ArgumentErr("parabola_subpixel: float32 CUDA tensors required")
        out = torch.empty_like(d)
        ctx.set_stream(torch.cuda.current_stream(l.device).cuda_stream)
        ctx.check(lib.vwgpu_parabola_subpixel_dev(ctx._h, d.data_ptr(), w, h, 0, l.data_ptr(), 0, r.data_ptr(), rw, rh, 0,
                                                  int(prefilter_mode), float(prefilter_width), kx, ky, out.data_ptr(), 0))
        return out
    d = np.ascontiguousarray(disparity, np.float32)
    l = np.ascontiguousarray(left_image, np.float32)
    r = np.ascontiguousarray(right_image, np.float32)
    out = np.empty_like(d)
    ctx.check(lib.vwgpu_parabola_subpixel(ctx._h, d.ctypes.data, w, h, 0, l.ctypes.data, 0, r.ctypes.data, rw, rh, 0,
                                          int(prefilter_mode), float(prefilter_width), kx, ky, out.ctypes.data, 0))
    return out


__all__ = ["calc_disparity", "cross_corr_consistency_check", "parabola_subpixel", "BBox2i", "CostFunctionType"]
