"""Host-side mirror of the reference's image filters on the stereo path (pyramid smoothing + decimation, Gaussian /
Laplacian prefilters, mask decimation).  Same names and argument meaning as the reference:

  generate_gaussian_kernel           src/vw/Image/Filter.tcc:37-78
  generate_pyramid_smoothing_kernel  src/vw/Image/Filter.h:89-99
  separable_convolution_filter       src/vw/Image/Filter.h:156-191   (rasterised; optional subsample(., s))
  gaussian_filter / laplacian_filter src/vw/Image/Filter.h:205-258, 320-335
  subsample_mask_by_two              src/vw/Stereo/CorrelationView.cc:38-63
  prefilter_image                    src/vw/Stereo/PreFilter.h:76-95

numpy arrays go through the host entry points, torch CUDA tensors through the device entry points (current stream).
"""
import numpy as np

from . import _lib
from .core import ArgumentErr
from .stereo import _ctx_for, _is_tensor

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

ConstantEdgeExtension, ZeroEdgeExtension = 0, 1
PREFILTER_NONE, PREFILTER_MEANSUB, PREFILTER_LOG = 0, 1, 2


def generate_gaussian_kernel(sigma, size=0):
    taps = np.zeros(4096, np.float32)
    n = _lib.load().vwgpu_generate_gaussian_kernel(float(sigma), int(size), taps.ctypes.data, taps.size)
    if n < 0:
        raise ArgumentErr("generate_gaussian_kernel: bad size")
    return taps[:n].copy()


def generate_pyramid_smoothing_kernel():
    return np.array([1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0], np.float32)


def _prep(img, dtype):
    if _is_tensor(img):
        if not img.is_cuda:
            raise ArgumentErr("torch inputs must be CUDA tensors (no CPU path)")
        if img.stride(-1) != 1:
            img = img.contiguous()
        return img
    return np.ascontiguousarray(img, dtype)


def _stream(ctx, img):
    ctx.set_stream(torch.cuda.current_stream(img.device).cuda_stream)


def separable_convolution_filter(src, x_kernel, y_kernel, cx=None, cy=None, edge=ConstantEdgeExtension, subsample=1, ctx=None):
    src = _prep(src, np.float32)
    xk = np.ascontiguousarray(x_kernel, np.float32)
    yk = np.ascontiguousarray(y_kernel, np.float32)
    cx = ((len(xk) - 1) // 2 if len(xk) else 0) if cx is None else cx
    cy = ((len(yk) - 1) // 2 if len(yk) else 0) if cy is None else cy
    h, w = src.shape
    oh, ow = 1 + (h - 1) // subsample, 1 + (w - 1) // subsample
    ctx = _ctx_for(src, ctx)
    lib = ctx._lib
    if _is_tensor(src):
        out = torch.empty((oh, ow), dtype=torch.float32, device=src.device)
        _stream(ctx, src)
        ctx.check(lib.vwgpu_separable_convolution_dev(ctx._h, src.data_ptr(), w, h, src.stride(0), xk.ctypes.data, len(xk), cx,
                                                      yk.ctypes.data, len(yk), cy, edge, subsample, out.data_ptr(), 0))
        return out
    out = np.empty((oh, ow), np.float32)
    ctx.check(lib.vwgpu_separable_convolution(ctx._h, src.ctypes.data, w, h, w, xk.ctypes.data, len(xk), cx,
                                              yk.ctypes.data, len(yk), cy, edge, subsample, out.ctypes.data, 0))
    return out


def gaussian_filter(src, x_sigma, y_sigma=None, x_dim=0, y_dim=0, edge=ConstantEdgeExtension, ctx=None):
    y_sigma = x_sigma if y_sigma is None else y_sigma
    return separable_convolution_filter(src, generate_gaussian_kernel(x_sigma, x_dim), generate_gaussian_kernel(y_sigma, y_dim),
                                        edge=edge, ctx=ctx)


def convolution_filter(src, kernel, ci=None, cj=None, edge=ConstantEdgeExtension, ctx=None):
    src = _prep(src, np.float32)
    k = np.ascontiguousarray(kernel, np.float32)
    kh, kw = k.shape
    ci = (kw - 1) // 2 if ci is None else ci
    cj = (kh - 1) // 2 if cj is None else cj
    h, w = src.shape
    ctx = _ctx_for(src, ctx)
    lib = ctx._lib
    if _is_tensor(src):
        out = torch.empty((h, w), dtype=torch.float32, device=src.device)
        _stream(ctx, src)
        ctx.check(lib.vwgpu_convolution_2d_dev(ctx._h, src.data_ptr(), w, h, src.stride(0), k.ctypes.data, kw, kh, ci, cj, edge,
                                               out.data_ptr(), 0))
        return out
    out = np.empty((h, w), np.float32)
    ctx.check(lib.vwgpu_convolution_2d(ctx._h, src.ctypes.data, w, h, w, k.ctypes.data, kw, kh, ci, cj, edge, out.ctypes.data, 0))
    return out


def laplacian_filter(src, edge=ConstantEdgeExtension, ctx=None):
    return convolution_filter(src, [[0, 1, 0], [1, -4, 1], [0, 1, 0]], 1, 1, edge, ctx=ctx)


def subsample_mask_by_two(mask, ctx=None):
    mask = _prep(mask, np.uint8)
    h, w = mask.shape
    oh, ow = 1 + (h - 1) // 2, 1 + (w - 1) // 2
    ctx = _ctx_for(mask, ctx)
    lib = ctx._lib
    if _is_tensor(mask):
        if mask.dtype != torch.uint8:
            raise ArgumentErr("subsample_mask_by_two: uint8 mask expected")
        out = torch.empty((oh, ow), dtype=torch.uint8, device=mask.device)
        _stream(ctx, mask)
        ctx.check(lib.vwgpu_subsample_mask_by_two_dev(ctx._h, mask.data_ptr(), w, h, mask.stride(0), out.data_ptr(), 0))
        return out
    out = np.empty((oh, ow), np.uint8)
    ctx.check(lib.vwgpu_subsample_mask_by_two(ctx._h, mask.ctypes.data, w, h, w, out.ctypes.data, 0))
    return out


def prefilter_image(image, prefilter_mode, prefilter_width, ctx=None):
    image = _prep(image, np.float32)
    h, w = image.shape
    ctx = _ctx_for(image, ctx)
    lib = ctx._lib
    if _is_tensor(image):
        out = torch.empty((h, w), dtype=torch.float32, device=image.device)
        _stream(ctx, image)
        ctx.check(lib.vwgpu_prefilter_image_dev(ctx._h, image.data_ptr(), w, h, image.stride(0), int(prefilter_mode),
                                                float(prefilter_width), out.data_ptr(), 0))
        return out
    out = np.empty((h, w), np.float32)
    ctx.check(lib.vwgpu_prefilter_image(ctx._h, image.ctypes.data, w, h, w, int(prefilter_mode), float(prefilter_width),
                                        out.ctypes.data, 0))
    return out


def build_gaussian_pyramid(image, levels, ctx=None):
    """The smoothing + decimation chain of build_image_pyramids (src/vw/Stereo/CorrelationView.cc:205-216):
    level i = subsample(separable_convolution_filter(level i-1, k, k), 2) with k = [1 4 6 4 1]/16."""
    k = generate_pyramid_smoothing_kernel()
    out = [image]
    for _ in range(levels):
        out.append(separable_convolution_filter(out[-1], k, k, subsample=2, ctx=ctx))
    return out
