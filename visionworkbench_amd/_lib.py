"""ctypes loader of libvwgpu.so (include/vwgpu.h).  Fails loudly: there is no CPU fallback in the product."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VWGPU_LIBRARY") or os.path.join(_HERE, "lib", "libvwgpu.so")  # override: experiments only
_LIB = None

SYMBOLS = [
    "vwgpu_abi_version", "vwgpu_create", "vwgpu_destroy", "vwgpu_set_stream", "vwgpu_reset_stream", "vwgpu_synchronize", "vwgpu_trim",
    "vwgpu_strerror", "vwgpu_last_error", "vwgpu_force_path", "vwgpu_last_path", "vwgpu_set_option", "vwgpu_get_option",
    "vwgpu_profile_enable", "vwgpu_profile_reset", "vwgpu_profile_read",
    "vwgpu_calc_disparity_dev", "vwgpu_calc_disparity", "vwgpu_fast_box_sum_dev", "vwgpu_fast_box_sum",
    "vwgpu_cross_corr_consistency_check_dev", "vwgpu_cross_corr_consistency_check",
    "vwgpu_cross_corr_consistency_check_diff_dev", "vwgpu_cross_corr_consistency_check_diff",
    "vwgpu_generate_gaussian_kernel",
    "vwgpu_separable_convolution_dev", "vwgpu_separable_convolution",
    "vwgpu_convolution_2d_dev", "vwgpu_convolution_2d",
    "vwgpu_subsample_mask_by_two_dev", "vwgpu_subsample_mask_by_two",
    "vwgpu_prefilter_image_dev", "vwgpu_prefilter_image",
    "vwgpu_parabola_subpixel_dev", "vwgpu_parabola_subpixel",
    "vwgpu_disparity_filter_dev", "vwgpu_disparity_filter",
    "vwgpu_disparity_mask_dev", "vwgpu_disparity_mask",
    "vwgpu_subdivide_regions",
    "vwgpu_disparity_blob_filter_dev", "vwgpu_disparity_blob_filter",
    "vwgpu_pyramid_correlate_dev", "vwgpu_pyramid_correlate", "vwgpu_pyramid_correlate_batch_dev", "vwgpu_pyramid_correlate_batch",
    "vwgpu_calc_disparity_sgm_dev", "vwgpu_calc_disparity_sgm", "vwgpu_mgm_front_count", "vwgpu_mgm_front_pixel",
    "vwgpu_comm_unique_id", "vwgpu_comm_create", "vwgpu_comm_destroy", "vwgpu_halo_plan", "vwgpu_halo_headers_agree", "vwgpu_fetch_strip_window_dev",
]


class SgmParams(ctypes.Structure):
    """struct vwgpu_sgm_params (include/vwgpu.h)."""
    _fields_ = [
        ("cost_type", ctypes.c_int), ("use_mgm", ctypes.c_int), ("kernel_size", ctypes.c_int), ("subpixel_mode", ctypes.c_int),
        ("search_buffer_x", ctypes.c_int), ("search_buffer_y", ctypes.c_int), ("memory_limit_mb", ctypes.c_size_t),
        ("p1", ctypes.c_int), ("p2", ctypes.c_int), ("ternary_census_threshold", ctypes.c_int), ("num_threads", ctypes.c_int),
        ("allow_block_cost", ctypes.c_int),
    ]


class PyramidParams(ctypes.Structure):
    """struct vwgpu_pyramid_params (include/vwgpu.h)."""
    _fields_ = [
        ("prefilter_mode", ctypes.c_int), ("prefilter_width", ctypes.c_float),
        ("search_min_x", ctypes.c_int), ("search_min_y", ctypes.c_int),
        ("search_max_x", ctypes.c_int), ("search_max_y", ctypes.c_int),
        ("kernel_x", ctypes.c_int), ("kernel_y", ctypes.c_int),
        ("cost_type", ctypes.c_int), ("corr_timeout", ctypes.c_int), ("seconds_per_op", ctypes.c_double),
        ("consistency_threshold", ctypes.c_float), ("min_consistency_level", ctypes.c_int),
        ("filter_half_kernel", ctypes.c_int), ("max_pyramid_levels", ctypes.c_int),
        ("algorithm", ctypes.c_int), ("blob_filter_area", ctypes.c_int),
        ("sgm_subpixel_mode", ctypes.c_int), ("sgm_search_buffer_x", ctypes.c_int), ("sgm_search_buffer_y", ctypes.c_int),
        ("memory_limit_mb", ctypes.c_size_t), ("sgm_num_threads", ctypes.c_int),
        ("lr_disp_diff", ctypes.c_void_p), ("lr_disp_diff_cols", ctypes.c_int), ("lr_disp_diff_rows", ctypes.c_int),
        ("lr_disp_diff_stride", ctypes.c_ssize_t), ("region_ul_x", ctypes.c_int), ("region_ul_y", ctypes.c_int),
    ]


def build(force=False):
    """Compile every HIP translation unit for gfx950 into lib/libvwgpu.so (hipcc cross-compiles on CPU)."""
    args = ["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "visionworkbench_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; libvwgpu.so names the same SONAME (RUNPATH /opt/rocm).  Whichever
    # is loaded first serves both, and a process that ends up with two half-initialised HIP runtimes fails in
    # hipGetDeviceCount.  Import torch first so that torch tensors and our kernels share one runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    P, I, F, PD = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_ssize_t
    lib.vwgpu_abi_version.restype = I
    lib.vwgpu_create.argtypes = [ctypes.POINTER(P), I]
    lib.vwgpu_destroy.argtypes = [P]
    lib.vwgpu_destroy.restype = None
    lib.vwgpu_set_stream.argtypes = [P, P]
    lib.vwgpu_reset_stream.argtypes = [P]
    lib.vwgpu_synchronize.argtypes = [P]
    lib.vwgpu_trim.argtypes = [P, ctypes.POINTER(ctypes.c_size_t)]
    lib.vwgpu_strerror.argtypes = [I]
    lib.vwgpu_strerror.restype = ctypes.c_char_p
    lib.vwgpu_last_error.argtypes = [P]
    lib.vwgpu_last_error.restype = ctypes.c_char_p
    lib.vwgpu_force_path.argtypes = [P, I]
    lib.vwgpu_last_path.argtypes = [P]
    lib.vwgpu_set_option.argtypes = [P, I, I]
    lib.vwgpu_get_option.argtypes = [P, I, ctypes.POINTER(ctypes.c_int)]
    lib.vwgpu_profile_enable.argtypes = [P, I]
    lib.vwgpu_profile_reset.argtypes = [P]
    lib.vwgpu_profile_read.argtypes = [P, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(F), I]
    bm = [P, I, P, I, I, PD, P, I, I, PD, I, I, I, I, P, PD]
    lib.vwgpu_calc_disparity_dev.argtypes = bm
    lib.vwgpu_calc_disparity.argtypes = bm
    IP = ctypes.POINTER(ctypes.c_int)
    lib.vwgpu_comm_unique_id.argtypes = [P]
    lib.vwgpu_comm_create.argtypes = [P, P, I, I, ctypes.POINTER(P)]
    lib.vwgpu_comm_destroy.argtypes = [P]
    lib.vwgpu_halo_plan.argtypes = [I, I, I, I, I, IP, IP, IP, IP]
    lib.vwgpu_halo_headers_agree.argtypes = [ctypes.POINTER(ctypes.c_longlong), I, IP, IP]
    lib.vwgpu_mgm_front_count.argtypes = [I, I, I]
    lib.vwgpu_mgm_front_pixel.argtypes = [I, I, I, I, I, P, P, IP]
    lib.vwgpu_fetch_strip_window_dev.argtypes = [P, P, P, I, I, I, I, I, P, IP]
    bs = [P, P, I, I, PD, I, I, P, PD]
    lib.vwgpu_fast_box_sum_dev.argtypes = bs
    lib.vwgpu_fast_box_sum.argtypes = bs
    lr = [P, P, I, I, PD, P, I, I, PD, F]
    lib.vwgpu_cross_corr_consistency_check_dev.argtypes = lr
    lib.vwgpu_cross_corr_consistency_check.argtypes = lr
    lrd = lr + [P, I, I, PD, I, I]
    lib.vwgpu_cross_corr_consistency_check_diff_dev.argtypes = lrd
    lib.vwgpu_cross_corr_consistency_check_diff.argtypes = lrd
    lib.vwgpu_generate_gaussian_kernel.argtypes = [ctypes.c_double, I, P, I]
    sc = [P, P, I, I, PD, P, I, I, P, I, I, I, I, P, PD]
    lib.vwgpu_separable_convolution_dev.argtypes = sc
    lib.vwgpu_separable_convolution.argtypes = sc
    c2 = [P, P, I, I, PD, P, I, I, I, I, I, P, PD]
    lib.vwgpu_convolution_2d_dev.argtypes = c2
    lib.vwgpu_convolution_2d.argtypes = c2
    mk = [P, P, I, I, PD, P, PD]
    lib.vwgpu_subsample_mask_by_two_dev.argtypes = mk
    lib.vwgpu_subsample_mask_by_two.argtypes = mk
    pf = [P, P, I, I, PD, I, F, P, PD]
    lib.vwgpu_prefilter_image_dev.argtypes = pf
    lib.vwgpu_prefilter_image.argtypes = pf
    ps = [P, P, I, I, PD, P, PD, P, I, I, PD, I, F, I, I, P, PD]
    lib.vwgpu_parabola_subpixel_dev.argtypes = ps
    lib.vwgpu_parabola_subpixel.argtypes = ps
    D = ctypes.c_double
    df = [P, P, I, I, I, I, D, D, I, P]
    lib.vwgpu_disparity_filter_dev.argtypes = df
    lib.vwgpu_disparity_filter.argtypes = df
    dm = [P, P, I, I, P, P, I, I]
    lib.vwgpu_disparity_mask_dev.argtypes = dm
    lib.vwgpu_disparity_mask.argtypes = dm
    lib.vwgpu_subdivide_regions.argtypes = [P, I, I, I, I, P, I]
    lib.vwgpu_disparity_blob_filter_dev.argtypes = [P, P, I, I, I]
    lib.vwgpu_disparity_blob_filter.argtypes = [P, P, I, I, I]
    pc = [P, P, I, I, PD, P, I, I, PD, P, PD, P, PD, ctypes.POINTER(PyramidParams), I, I, I, I, P, PD]
    lib.vwgpu_pyramid_correlate_dev.argtypes = pc
    lib.vwgpu_pyramid_correlate.argtypes = pc
    pb = [P, P, I, I, PD, P, I, I, PD, P, PD, P, PD, ctypes.POINTER(PyramidParams), I, P, P, P, P, P, P]
    lib.vwgpu_pyramid_correlate_batch_dev.argtypes = pb
    lib.vwgpu_pyramid_correlate_batch.argtypes = pb
    IP = ctypes.POINTER(ctypes.c_int)
    sg = [P, ctypes.POINTER(SgmParams), P, I, I, PD, P, I, I, PD, I, I, P, I, I, P, I, I, P, I, I, P, P, ctypes.c_size_t, IP, IP]
    lib.vwgpu_calc_disparity_sgm_dev.argtypes = sg
    lib.vwgpu_calc_disparity_sgm.argtypes = sg
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError if the ABI and the header drifted apart
    _LIB = lib
    return lib
