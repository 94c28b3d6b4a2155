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

    Device tensors: the call is queued on the current torch stream, but by default it WAITS for the input-class flags of the data
    (one small device-to-host copy: packed integer kernels, the float64 tile kernel or the reference's summation order are chosen
    from the data, so that the result is bit-exact for any input).  Callers that queue many calls set
    ctx.set_option(core.OPT_DEFER_EXACTNESS, 1): no host round trip, ctx.last_path() reports afterwards.
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


def fast_box_sum(image, kernel, ctx=None):
    """vw::stereo::fast_box_sum<double>(image, kernel) (src/vw/Stereo/Algorithms.h:41-129): float64 sums of every
    kx x ky window, (rows-ky+1, cols-kx+1), formed in the reference's running-sum order (bit-identical for any float input)."""
    kx, ky = int(kernel[0]), int(kernel[1])
    if image.ndim != 2:
        raise ArgumentErr("fast_box_sum: the image must be 2-D (rows, cols)")
    h, w = image.shape
    ctx = _ctx_for(image, ctx)
    lib = ctx._lib
    if _is_tensor(image):
        if not image.is_cuda or image.dtype != torch.float32:
            raise ArgumentErr("fast_box_sum: torch input must be a float32 CUDA tensor (no CPU path)")
        if image.stride(1) != 1:
            image = image.contiguous()
        out = torch.empty((max(h - ky + 1, 0), max(w - kx + 1, 0)), dtype=torch.float64, device=image.device)
        ctx.set_stream(torch.cuda.current_stream(image.device).cuda_stream)
        ctx.check(lib.vwgpu_fast_box_sum_dev(ctx._h, image.data_ptr(), w, h, image.stride(0), kx, ky, out.data_ptr(), 0))
        return out
    img = np.ascontiguousarray(image, np.float32)
    out = np.empty((max(h - ky + 1, 0), max(w - kx + 1, 0)), np.float64)
    ctx.check(lib.vwgpu_fast_box_sum(ctx._h, img.ctypes.data, w, h, w, kx, ky, out.ctypes.data, 0))
    return out


def cross_corr_consistency_check(l2r, r2l, cross_corr_threshold, lr_disp_diff=None, ul_corner_offset=(0, 0), ctx=None):
    """vw::stereo::cross_corr_consistency_check (src/vw/Stereo/Correlate.cc:1441-1502), IN PLACE on l2r.

    l2r, r2l: (rows, cols, 3) int32 PixelMask<Vector2i> images.  lr_disp_diff (optional, modified in place): (rows, cols, 2)
    float32 PixelMask<float> {value, valid}; every kept pixel stores its discrepancy at (c, r) + ul_corner_offset."""
    ctx = _ctx_for(l2r, ctx)
    lib = ctx._lib
    if lr_disp_diff is not None:
        ux, uy = int(ul_corner_offset[0]), int(ul_corner_offset[1])
        if lr_disp_diff.ndim != 3 or lr_disp_diff.shape[2] != 2:
            raise ArgumentErr("cross_corr_consistency_check: lr_disp_diff must be (rows, cols, 2) float32")
        dr, dc = lr_disp_diff.shape[:2]
        if _is_tensor(l2r):
            if not (l2r.is_cuda and r2l.is_cuda and lr_disp_diff.is_cuda and l2r.is_contiguous() and r2l.is_contiguous()
                    and lr_disp_diff.is_contiguous() and lr_disp_diff.dtype == torch.float32):
                raise ArgumentErr("cross_corr_consistency_check: contiguous CUDA tensors required")
            ctx.set_stream(torch.cuda.current_stream(l2r.device).cuda_stream)
            ctx.check(lib.vwgpu_cross_corr_consistency_check_diff_dev(ctx._h, l2r.data_ptr(), l2r.shape[1], l2r.shape[0], 0, r2l.data_ptr(),
                                                                      r2l.shape[1], r2l.shape[0], 0, float(cross_corr_threshold),
                                                                      lr_disp_diff.data_ptr(), dc, dr, 0, ux, uy))
            return l2r
        if not (l2r.flags.c_contiguous and l2r.dtype == np.int32 and lr_disp_diff.flags.c_contiguous and lr_disp_diff.dtype == np.float32):
            raise ArgumentErr("cross_corr_consistency_check: contiguous int32 / float32 arrays required (modified in place)")
        r2l = np.ascontiguousarray(r2l, np.int32)
        ctx.check(lib.vwgpu_cross_corr_consistency_check_diff(ctx._h, l2r.ctypes.data, l2r.shape[1], l2r.shape[0], 0, r2l.ctypes.data,
                                                              r2l.shape[1], r2l.shape[0], 0, float(cross_corr_threshold),
                                                              lr_disp_diff.ctypes.data, dc, dr, 0, ux, uy))
        return l2r
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


def _filter_call(name, disparity, hh, hv, pthr, rthr, cleanup, ctx):
    if disparity.ndim != 3 or disparity.shape[2] != 3:
        raise ArgumentErr("%s: disparity must be (rows, cols, 3) int32" % name)
    h, w = disparity.shape[:2]
    if hh <= 0 or hv <= 0:
        raise ArgumentErr("RmOutliersFunc: half kernel sizes must be non-zero.")
    ctx = _ctx_for(disparity, ctx)
    lib = ctx._lib
    if _is_tensor(disparity):
        if not disparity.is_cuda or disparity.dtype != torch.int32:
            raise ArgumentErr("%s: int32 CUDA tensor required" % name)
        d = disparity.contiguous()
        out = torch.empty_like(d)
        ctx.set_stream(torch.cuda.current_stream(d.device).cuda_stream)
        ctx.check(lib.vwgpu_disparity_filter_dev(ctx._h, d.data_ptr(), w, h, int(hh), int(hv), float(pthr), float(rthr),
                                                 int(cleanup), out.data_ptr()))
        return out
    d = np.ascontiguousarray(disparity, np.int32)
    out = np.empty_like(d)
    ctx.check(lib.vwgpu_disparity_filter(ctx._h, d.ctypes.data, w, h, int(hh), int(hv), float(pthr), float(rthr),
                                         int(cleanup), out.ctypes.data))
    return out


def rm_outliers_using_thresh(disparity, half_h_kernel, half_v_kernel, pixel_threshold, rejection_threshold, ctx=None):
    """vw::stereo::rm_outliers_using_thresh (src/vw/Stereo/DisparityMap.h:387-399), rasterised over the whole image
    with the reference's ConstantEdgeExtension.  disparity: (rows, cols, 3) int32 PixelMask<Vector2i>."""
    return _filter_call("rm_outliers_using_thresh", disparity, half_h_kernel, half_v_kernel, pixel_threshold,
                        rejection_threshold, 0, ctx)


def disparity_cleanup_using_thresh(disparity, h_half_kernel, v_half_kernel, threshold, rejection_threshold, ctx=None):
    """vw::stereo::disparity_cleanup_using_thresh (src/vw/Stereo/DisparityMap.h:427-441): the filter above followed
    by a second pass with the reference's fixed (1, 1, 3.0, 0.20)."""
    return _filter_call("disparity_cleanup_using_thresh", disparity, h_half_kernel, v_half_kernel, threshold,
                        rejection_threshold, 1, ctx)


def disparity_mask(disparity, left_mask, right_mask, ctx=None):
    """vw::stereo::disparity_mask (src/vw/Stereo/DisparityMap.h:236-253): invalidate pixels whose source or target
    falls on masked data.  Returns a new image; masks are (rows, cols) uint8 (0 = no data)."""
    if disparity.ndim != 3 or disparity.shape[2] != 3 or tuple(disparity.shape[:2]) != tuple(left_mask.shape):
        raise ArgumentErr("disparity_mask: left mask must match the disparity image")
    h, w = left_mask.shape
    rmh, rmw = right_mask.shape
    ctx = _ctx_for(disparity, ctx)
    lib = ctx._lib
    if _is_tensor(disparity):
        if not (disparity.is_cuda and left_mask.is_cuda and right_mask.is_cuda) or disparity.dtype != torch.int32:
            raise ArgumentErr("disparity_mask: int32 / uint8 CUDA tensors required")
        out = disparity.contiguous().clone()
        m1, m2 = left_mask.contiguous(), right_mask.contiguous()
        ctx.set_stream(torch.cuda.current_stream(out.device).cuda_stream)
        ctx.check(lib.vwgpu_disparity_mask_dev(ctx._h, out.data_ptr(), w, h, m1.data_ptr(), m2.data_ptr(), rmw, rmh))
        return out
    out = np.array(disparity, np.int32, order="C", copy=True)
    m1 = np.ascontiguousarray(left_mask, np.uint8)
    m2 = np.ascontiguousarray(right_mask, np.uint8)
    ctx.check(lib.vwgpu_disparity_mask(ctx._h, out.ctypes.data, w, h, m1.ctypes.data, m2.ctypes.data, rmw, rmh))
    return out


def disparity_blob_filter(disparity, max_blob_area, ctx=None):
    """PyramidCorrelationView::disparity_blob_filter at one level (src/vw/Stereo/CorrelationView.cc:242-271): erase every
    8-connected component of valid pixels with at most max_blob_area pixels.  Returns a new image."""
    if disparity.ndim != 3 or disparity.shape[2] != 3:
        raise ArgumentErr("disparity_blob_filter: disparity must be (rows, cols, 3) int32")
    h, w = disparity.shape[:2]
    ctx = _ctx_for(disparity, ctx)
    lib = ctx._lib
    if _is_tensor(disparity):
        if not disparity.is_cuda or disparity.dtype != torch.int32:
            raise ArgumentErr("disparity_blob_filter: int32 CUDA tensor required")
        out = disparity.contiguous().clone()
        ctx.set_stream(torch.cuda.current_stream(out.device).cuda_stream)
        ctx.check(lib.vwgpu_disparity_blob_filter_dev(ctx._h, out.data_ptr(), w, h, int(max_blob_area)))
        return out
    out = np.array(disparity, np.int32, order="C", copy=True)
    ctx.check(lib.vwgpu_disparity_blob_filter(ctx._h, out.ctypes.data, w, h, int(max_blob_area)))
    return out


def subdivide_regions(disparity, kernel_size):
    """vw::stereo::subdivide_regions(disparity, bounding_box(disparity), list, kernel_size)
    (src/vw/Stereo/Correlation.cc:139-328).  Host logic (the zone scheduler of pyramid_correlate) on a numpy
    PixelMask<Vector2i> image; returns [(region BBox2i, disparity_range BBox2i), ...] in the reference's order."""
    from . import _lib
    lib = _lib.load()
    d = np.ascontiguousarray(disparity, np.int32)
    if d.ndim != 3 or d.shape[2] != 3:
        raise ArgumentErr("subdivide_regions: disparity must be (rows, cols, 3) int32")
    h, w = d.shape[:2]
    cap = 1024
    while True:
        buf = np.empty((cap, 8), np.int32)
        n = lib.vwgpu_subdivide_regions(d.ctypes.data, w, h, int(kernel_size[0]), int(kernel_size[1]), buf.ctypes.data, cap)
        if n < 0:
            raise ArgumentErr("subdivide_regions: bad arguments")
        if n <= cap:
            break
        cap = n
    return [(BBox2i.from_corners(z[0:2], z[2:4]), BBox2i.from_corners(z[4:6], z[6:8])) for z in buf[:n].tolist()]


def pyramid_correlate(left, right, left_mask, right_mask, prefilter_mode, prefilter_width, search_region, kernel_size,
                      cost_type, corr_timeout=0, seconds_per_op=0.0, consistency_threshold=-1.0,
                      min_consistency_level=0, filter_half_kernel=0, max_pyramid_levels=5, algorithm=0,
                      collar_size=0, sgm_subpixel_mode=5, sgm_search_buffer=(2, 2), memory_limit_mb=6000,
                      blob_filter_area=0, bbox=None, sgm_num_threads=1, lr_disp_diff=None, region_ul=(0, 0), ctx=None):
    """vw::stereo::pyramid_correlate (src/vw/Stereo/CorrelationView.h:195-230) rasterised over `bbox`
    (default: the whole left image as ONE tile, i.e. PyramidCorrelationView::prerasterize(bounding_box),
    src/vw/Stereo/CorrelationView.cc:273-886).  The reference rasterises per block-cache tile; pass the same
    bbox to reproduce a tile.

    left / right: (rows, cols) float32; masks: (rows, cols) uint8 or None; search_region: BBox2i (half open).
    Returns (bbox rows, bbox cols, 3) float32 PixelMask<Vector2f> {dx, dy, valid}.
    algorithm 0 = VW_CORRELATION_BM (integer disparities cast to float), 1 = VW_CORRELATION_SGM (census costs only; the
    result is the matcher's sub-pixel view, CorrelationView.cc:862-875), 2 = VW_CORRELATION_MGM, 3 = _FINAL_MGM (MGM at level 0 only).  collar_size is
    the tile rasteriser's business (CorrelationView.h:128-132): pass the collared bbox.
    lr_disp_diff (optional, modified in place): (rows, cols, 2) float32 PixelMask<float> image covering the image pixels from
    region_ul on; the level-0 consistency check stores the L-R / R-L discrepancy of the pixels it keeps there and pixels
    the filters remove are invalidated again (CorrelationView.h:84, .cc:277-283, 683-693, 846-855)."""
    from ._lib import PyramidParams
    if left.ndim != 2 or right.ndim != 2:
        raise ArgumentErr("pyramid_correlate: images must be 2-D (rows, cols)")
    lh, lw = left.shape
    rh, rw = right.shape
    if bbox is None:
        bbox = BBox2i(0, 0, lw, lh)
    (bx, by), (bx1, by1) = bbox.min, bbox.max
    P = PyramidParams(int(prefilter_mode), float(prefilter_width),
                      int(search_region.min[0]), int(search_region.min[1]), int(search_region.max[0]), int(search_region.max[1]),
                      int(kernel_size[0]), int(kernel_size[1]), int(cost_type), int(corr_timeout), float(seconds_per_op),
                      float(consistency_threshold), int(min_consistency_level), int(filter_half_kernel),
                      int(max_pyramid_levels), int(algorithm), int(blob_filter_area), int(sgm_subpixel_mode),
                      int(sgm_search_buffer[0]), int(sgm_search_buffer[1]), int(memory_limit_mb), int(sgm_num_threads),
                      None, 0, 0, 0, int(region_ul[0]), int(region_ul[1]))
    if lr_disp_diff is not None:
        if lr_disp_diff.ndim != 3 or lr_disp_diff.shape[2] != 2:
            raise ArgumentErr("pyramid_correlate: lr_disp_diff must be (rows, cols, 2) float32")
        if _is_tensor(lr_disp_diff) != _is_tensor(left):
            raise ArgumentErr("pyramid_correlate: lr_disp_diff must live where the images live")
        ok = (lr_disp_diff.is_cuda and lr_disp_diff.is_contiguous() and lr_disp_diff.dtype == torch.float32) if _is_tensor(lr_disp_diff) \
            else (lr_disp_diff.flags.c_contiguous and lr_disp_diff.dtype == np.float32)
        if not ok:
            raise ArgumentErr("pyramid_correlate: lr_disp_diff must be contiguous float32")
        P.lr_disp_diff = lr_disp_diff.data_ptr() if _is_tensor(lr_disp_diff) else lr_disp_diff.ctypes.data
        P.lr_disp_diff_rows, P.lr_disp_diff_cols = int(lr_disp_diff.shape[0]), int(lr_disp_diff.shape[1])
    ctx = _ctx_for(left, ctx)
    lib = ctx._lib
    bw, bh = bx1 - bx, by1 - by
    if _is_tensor(left):
        if not (left.is_cuda and right.is_cuda) or left.dtype != torch.float32 or right.dtype != torch.float32:
            raise ArgumentErr("pyramid_correlate: float32 CUDA tensors required (no CPU path)")
        l, r = left.contiguous(), right.contiguous()
        lm = left_mask.contiguous() if left_mask is not None else None
        rm = right_mask.contiguous() if right_mask is not None else None
        for m, shp in ((lm, l.shape), (rm, r.shape)):
            if m is not None and (m.dtype != torch.uint8 or tuple(m.shape) != tuple(shp) or not m.is_cuda):
                raise ArgumentErr("pyramid_correlate: masks must be uint8 CUDA tensors of the image size")
        out = torch.empty((max(bh, 0), max(bw, 0), 3), dtype=torch.float32, device=l.device)
        ctx.set_stream(torch.cuda.current_stream(l.device).cuda_stream)
        ctx.check(lib.vwgpu_pyramid_correlate_dev(ctx._h, l.data_ptr(), lw, lh, 0, r.data_ptr(), rw, rh, 0,
                                                  lm.data_ptr() if lm is not None else None, 0,
                                                  rm.data_ptr() if rm is not None else None, 0,
                                                  ctypes.byref(P), bx, by, bw, bh, out.data_ptr(), 0))
        return out
    l = np.ascontiguousarray(left, np.float32)
    r = np.ascontiguousarray(right, np.float32)
    lm = np.ascontiguousarray(left_mask, np.uint8) if left_mask is not None else None
    rm = np.ascontiguousarray(right_mask, np.uint8) if right_mask is not None else None
    for m, shp in ((lm, l.shape), (rm, r.shape)):
        if m is not None and tuple(m.shape) != tuple(shp):
            raise ArgumentErr("pyramid_correlate: masks must have the image size")
    out = np.empty((max(bh, 0), max(bw, 0), 3), np.float32)
    ctx.check(lib.vwgpu_pyramid_correlate(ctx._h, l.ctypes.data, lw, lh, 0, r.ctypes.data, rw, rh, 0,
                                          lm.ctypes.data if lm is not None else None, 0,
                                          rm.ctypes.data if rm is not None else None, 0,
                                          ctypes.byref(P), bx, by, bw, bh, out.ctypes.data, 0))
    return out


def pyramid_correlate_batch(left, right, left_mask, right_mask, prefilter_mode, prefilter_width, search_region, kernel_size, cost_type,
                            bboxes, corr_timeout=0, seconds_per_op=0.0, consistency_threshold=-1.0, min_consistency_level=0,
                            filter_half_kernel=0, max_pyramid_levels=5, algorithm=0, collar_size=0, sgm_subpixel_mode=5,
                            sgm_search_buffer=(2, 2), memory_limit_mb=6000, blob_filter_area=0, sgm_num_threads=1, ctx=None):
    """Several tiles (`bboxes`: a list of BBox2i) of vw::stereo::pyramid_correlate in ONE call: what the reference's block rasteriser hands
    to its tile threads one at a time (src/vw/Image/ImageIO.h:228-251).  Runs of consecutive tiles of equal size go through the pyramid
    level loop together (vwgpu_pyramid_correlate_batch[_dev], include/vwgpu.h); every tile's result is identical to pyramid_correlate on
    that tile.  Returns a list of (rows, cols, 3) float32 PixelMask<Vector2f> images: CUDA tensors for CUDA inputs, numpy arrays otherwise."""
    from ._lib import PyramidParams
    if left.ndim != 2 or right.ndim != 2:
        raise ArgumentErr("pyramid_correlate: images must be 2-D (rows, cols)")
    lh, lw = left.shape
    rh, rw = right.shape
    P = PyramidParams(int(prefilter_mode), float(prefilter_width),
                      int(search_region.min[0]), int(search_region.min[1]), int(search_region.max[0]), int(search_region.max[1]),
                      int(kernel_size[0]), int(kernel_size[1]), int(cost_type), int(corr_timeout), float(seconds_per_op),
                      float(consistency_threshold), int(min_consistency_level), int(filter_half_kernel),
                      int(max_pyramid_levels), int(algorithm), int(blob_filter_area), int(sgm_subpixel_mode),
                      int(sgm_search_buffer[0]), int(sgm_search_buffer[1]), int(memory_limit_mb), int(sgm_num_threads),
                      None, 0, 0, 0, 0, 0)
    n = len(bboxes)
    IA = ctypes.c_int * max(n, 1)
    bx = IA(*[int(b.min[0]) for b in bboxes]); by = IA(*[int(b.min[1]) for b in bboxes])
    bw = IA(*[int(b.max[0] - b.min[0]) for b in bboxes]); bh = IA(*[int(b.max[1] - b.min[1]) for b in bboxes])
    ctx = _ctx_for(left, ctx)
    lib = ctx._lib
    PA = ctypes.c_void_p * max(n, 1)
    if _is_tensor(left):
        if not (left.is_cuda and right.is_cuda) or left.dtype != torch.float32 or right.dtype != torch.float32:
            raise ArgumentErr("pyramid_correlate: float32 CUDA tensors required (no CPU path)")
        l, r = left.contiguous(), right.contiguous()
        lm = left_mask.contiguous() if left_mask is not None else None
        rm = right_mask.contiguous() if right_mask is not None else None
        for m, shp in ((lm, l.shape), (rm, r.shape)):
            if m is not None and (m.dtype != torch.uint8 or tuple(m.shape) != tuple(shp) or not m.is_cuda):
                raise ArgumentErr("pyramid_correlate: masks must be uint8 CUDA tensors of the image size")
        outs = [torch.empty((max(bh[t], 0), max(bw[t], 0), 3), dtype=torch.float32, device=l.device) for t in range(n)]
        ptrs = PA(*[o.data_ptr() for o in outs])
        ctx.set_stream(torch.cuda.current_stream(l.device).cuda_stream)
        ctx.check(lib.vwgpu_pyramid_correlate_batch_dev(ctx._h, l.data_ptr(), lw, lh, 0, r.data_ptr(), rw, rh, 0,
                                                        lm.data_ptr() if lm is not None else None, 0, rm.data_ptr() if rm is not None else None, 0,
                                                        ctypes.byref(P), n, bx, by, bw, bh, ptrs, None))
        return outs
    l = np.ascontiguousarray(left, np.float32)
    r = np.ascontiguousarray(right, np.float32)
    lm = np.ascontiguousarray(left_mask, np.uint8) if left_mask is not None else None
    rm = np.ascontiguousarray(right_mask, np.uint8) if right_mask is not None else None
    for m, shp in ((lm, l.shape), (rm, r.shape)):
        if m is not None and tuple(m.shape) != tuple(shp):
            raise ArgumentErr("pyramid_correlate: masks must have the image size")
    outs = [np.empty((max(bh[t], 0), max(bw[t], 0), 3), np.float32) for t in range(n)]
    ptrs = PA(*[o.ctypes.data for o in outs])
    ctx.check(lib.vwgpu_pyramid_correlate_batch(ctx._h, l.ctypes.data, lw, lh, 0, r.ctypes.data, rw, rh, 0,
                                                lm.ctypes.data if lm is not None else None, 0, rm.ctypes.data if rm is not None else None, 0,
                                                ctypes.byref(P), n, bx, by, bw, bh, ptrs, None))
    return outs


SUBPIXEL_NONE, SUBPIXEL_PARABOLA, SUBPIXEL_LINEAR, SUBPIXEL_POLY4, SUBPIXEL_COSINE, SUBPIXEL_LC_BLEND = range(6)


def calc_disparity_sgm(cost_type, left_in, right_in, left_region, search_volume, kernel_size, use_mgm=False,
                       subpixel_mode=SUBPIXEL_LC_BLEND, search_buffer=(2, 2), memory_limit_mb=6000,
                       left_mask=None, right_mask=None, prev_disparity=None, p1=0, p2=0, ternary_census_threshold=5,
                       num_threads=1, with_subpixel=False, allow_block_cost=False, ctx=None):
    """vw::stereo::calc_disparity_sgm (src/vw/Stereo/SGM.h:360-375, SGM.cc:167-229).

    allow_block_cost (not an argument of the reference's function): cost types ABSOLUTE_DIFFERENCE / SQUARED_DIFFERENCE raise
    NoImplErr exactly as compute_disparity_costs throws (SGM.cc:1887-1892) unless this is True, which runs the code behind that
    throw — fill_costs_block's mean-abs-difference block cost (SGM.cc:1651-1738; p1 = 3, p2 = 250 by default).

    left_in / right_in: (rows, cols) float32; left_region: BBox2i inside the left image; search_volume = (sx, sy) is
    INCLUSIVE like the reference's (the right crop is left_region grown by search_volume, so (sx+1) x (sy+1) disparities
    are searched); kernel_size = (k, k) with k in {3, 5, 7, 9}; cost_type CENSUS_TRANSFORM / TERNARY_CENSUS_TRANSFORM.
    Masks / prev_disparity as in SemiGlobalMatcher::semi_global_matching_func (SGM.h:149-157).
    Returns the integer disparity (rows-k+1, cols-k+1, 3) int32; with_subpixel=True also returns the matcher's
    create_disparity_view_subpixel result (the reference hands the matcher back through matcher_ptr for that)."""
    from ._lib import SgmParams
    kx, ky = int(kernel_size[0]), int(kernel_size[1])
    sx, sy = int(search_volume[0]), int(search_volume[1])
    if left_in.ndim != 2 or right_in.ndim != 2:
        raise ArgumentErr("calc_disparity_sgm: images must be 2-D (rows, cols)")
    if kx % 2 != 1 or ky % 2 != 1:
        raise ArgumentErr("calc_disparity_sgm: Kernel input not sized with odd values.")
    x0, y0 = left_region.min
    x1, y1 = left_region.max
    if x0 < 0 or y0 < 0 or x1 > left_in.shape[1] or y1 > left_in.shape[0]:
        raise ArgumentErr("calc_disparity_sgm: Region not inside left image.")
    lw, lh = x1 - x0, y1 - y0
    if kx > lw or ky > lh:
        raise ArgumentErr("calc_disparity_sgm: Kernel size too large of active region.")
    rx1, ry1 = min(x1 + sx, right_in.shape[1]), min(y1 + sy, right_in.shape[0])
    P = SgmParams(int(cost_type), int(bool(use_mgm)), kx, int(subpixel_mode), int(search_buffer[0]), int(search_buffer[1]),
                  int(memory_limit_mb), int(p1), int(p2), int(ternary_census_threshold), int(num_threads), int(bool(allow_block_cost)))
    ctx = _ctx_for(left_in, ctx)
    lib = ctx._lib
    ow, oh = ctypes.c_int(), ctypes.c_int()
    cap = lw * lh

    def shape2(a):
        return (0, 0) if a is None else (a.shape[1], a.shape[0])
    if _is_tensor(left_in):
        if not (left_in.is_cuda and right_in.is_cuda) or left_in.dtype != torch.float32 or right_in.dtype != torch.float32:
            raise ArgumentErr("calc_disparity_sgm: float32 CUDA tensors required (no CPU path)")
        l = left_in[y0:y1, x0:x1].contiguous()
        r = right_in[y0:ry1, x0:rx1].contiguous()
        lm = left_mask.contiguous() if left_mask is not None else None
        rm = right_mask.contiguous() if right_mask is not None else None
        pd = prev_disparity.contiguous() if prev_disparity is not None else None
        out = torch.empty((cap, 3), dtype=torch.int32, device=l.device)
        sub = torch.empty((cap, 3), dtype=torch.float32, device=l.device) if with_subpixel else None
        ctx.set_stream(torch.cuda.current_stream(l.device).cuda_stream)
        ptr = lambda t: t.data_ptr() if t is not None else None
        ctx.check(lib.vwgpu_calc_disparity_sgm_dev(ctx._h, ctypes.byref(P), l.data_ptr(), lw, lh, 0, r.data_ptr(), r.shape[1], r.shape[0], 0,
                                                   sx, sy, ptr(lm), *shape2(lm), ptr(rm), *shape2(rm), ptr(pd), *shape2(pd),
                                                   out.data_ptr(), ptr(sub), cap, ctypes.byref(ow), ctypes.byref(oh)))
        n = ow.value * oh.value
        res = out[:n].reshape(oh.value, ow.value, 3)
        return (res, sub[:n].reshape(oh.value, ow.value, 3)) if with_subpixel else res
    l = np.ascontiguousarray(left_in[y0:y1, x0:x1], np.float32)
    r = np.ascontiguousarray(right_in[y0:ry1, x0:rx1], np.float32)
    lm = np.ascontiguousarray(left_mask, np.uint8) if left_mask is not None else None
    rm = np.ascontiguousarray(right_mask, np.uint8) if right_mask is not None else None
    pd = np.ascontiguousarray(prev_disparity, np.int32) if prev_disparity is not None else None
    out = np.empty((cap, 3), np.int32)
    sub = np.empty((cap, 3), np.float32) if with_subpixel else None
    ptr = lambda a: a.ctypes.data if a is not None else None
    ctx.check(lib.vwgpu_calc_disparity_sgm(ctx._h, ctypes.byref(P), l.ctypes.data, lw, lh, 0, r.ctypes.data, r.shape[1], r.shape[0], 0,
                                           sx, sy, ptr(lm), *shape2(lm), ptr(rm), *shape2(rm), ptr(pd), *shape2(pd),
                                           out.ctypes.data, ptr(sub), cap, ctypes.byref(ow), ctypes.byref(oh)))
    n = ow.value * oh.value
    res = out[:n].reshape(oh.value, ow.value, 3).copy()
    return (res, sub[:n].reshape(oh.value, ow.value, 3).copy()) if with_subpixel else res


__all__ = ["calc_disparity", "calc_disparity_sgm", "cross_corr_consistency_check", "parabola_subpixel", "rm_outliers_using_thresh",
           "disparity_cleanup_using_thresh", "disparity_mask", "disparity_blob_filter", "subdivide_regions", "pyramid_correlate", "pyramid_correlate_batch",
           "BBox2i", "CostFunctionType"]
