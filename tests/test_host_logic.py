"""Host-side logic of libvwgpu.so that needs no GPU: the zone scheduler (vwgpu_subdivide_regions), the Gaussian tap
generator, argument validation (status codes + messages before any device work), and the ctypes struct layouts.  Runs in the
CPU suite; the zone scheduler and the taps are compared with the oracle's restatement of the reference."""
import ctypes

import numpy as np
import pytest

from visionworkbench_amd import _lib, filters, stereo
from visionworkbench_amd.core import BBox2i


def _random_disparity(rng, h, w, p_invalid=0.25):
    V = np.iinfo(np.int32).max
    d = np.zeros((h, w, 3), np.int32)
    d[..., 0] = rng.integers(-4, 5, (h, w)) + (np.arange(w)[None, :] // 16)
    d[..., 1] = rng.integers(-2, 3, (h, w))
    d[..., 2] = np.where(rng.random((h, w)) < p_invalid, 0, V)
    return d


def test_zone_scheduler_matches_oracle(oracle):
    """subdivide_regions (src/vw/Stereo/Correlation.cc:139-328): same zones, same order as the literal restatement."""
    rng = np.random.default_rng(9)
    for h, w, k in [(64, 96, (7, 7)), (33, 47, (5, 9)), (128, 200, (3, 3)), (17, 300, (11, 11)), (250, 250, (7, 7))]:
        d = _random_disparity(rng, h, w)
        d[: h // 2, : w // 3, 0] += 25
        d[h // 3:, w // 2:, 2] = 0
        got = [r.min + r.max + s.min + s.max for r, s in stereo.subdivide_regions(d, k)]
        assert got == [list(map(int, row)) for row in oracle.subdivide_regions(d, k)]
    # uniform image -> one zone with a [d, d+1) range; nothing valid -> no zone
    d = np.zeros((40, 60, 3), np.int32)
    d[..., 0], d[..., 1], d[..., 2] = 3, 1, np.iinfo(np.int32).max
    z = stereo.subdivide_regions(d, (7, 7))
    assert len(z) == 1 and z[0][0].min + z[0][0].max == [0, 0, 60, 40] and z[0][1].min + z[0][1].max == [3, 1, 4, 2]
    d[..., 2] = 0
    assert stereo.subdivide_regions(d, (7, 7)) == []


def test_gaussian_taps_match_oracle(oracle):
    """generate_gaussian_kernel<float> (src/vw/Image/Filter.tcc:37-78, size rule Filter.cc:32-37)."""
    for sigma, size in [(1.0, 5), (1.0, 4), (1.5, 0), (float(np.float32(1.4)), 0), (5.0, 0), (0.3, 0), (2.0, 9)]:
        assert np.array_equal(filters.generate_gaussian_kernel(sigma, size), oracle.generate_gaussian_kernel(sigma, size))
    assert len(filters.generate_gaussian_kernel(0.0)) == 0
    assert np.array_equal(filters.generate_pyramid_smoothing_kernel(), np.array([1, 4, 6, 4, 1], np.float32) / 16)


def test_struct_layouts_match_the_header():
    """The ctypes mirrors of vwgpu_pyramid_params / vwgpu_sgm_params must have the C layout (offsets from the header order)."""
    P = _lib.PyramidParams
    names = [f[0] for f in P._fields_]
    assert names == ["prefilter_mode", "prefilter_width", "search_min_x", "search_min_y", "search_max_x", "search_max_y", "kernel_x",
                     "kernel_y", "cost_type", "corr_timeout", "seconds_per_op", "consistency_threshold", "min_consistency_level",
                     "filter_half_kernel", "max_pyramid_levels", "algorithm", "blob_filter_area", "sgm_subpixel_mode",
                     "sgm_search_buffer_x", "sgm_search_buffer_y", "memory_limit_mb", "sgm_num_threads", "lr_disp_diff",
                     "lr_disp_diff_cols", "lr_disp_diff_rows", "lr_disp_diff_stride", "region_ul_x", "region_ul_y"]
    assert P.seconds_per_op.offset % 8 == 0 and P.memory_limit_mb.offset % 8 == 0 and ctypes.sizeof(P) % 8 == 0
    assert P.lr_disp_diff.offset % 8 == 0 and P.lr_disp_diff_stride.offset % 8 == 0
    S = _lib.SgmParams
    assert [f[0] for f in S._fields_] == ["cost_type", "use_mgm", "kernel_size", "subpixel_mode", "search_buffer_x", "search_buffer_y",
                                          "memory_limit_mb", "p1", "p2", "ternary_census_threshold", "num_threads"]
    assert S.memory_limit_mb.offset == 24 and ctypes.sizeof(S) == 48
    # the header spells the same field order
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vwgpu.h")).read()
    body = hdr[hdr.index("typedef struct vwgpu_pyramid_params {"):hdr.index("} vwgpu_pyramid_params;")]
    pos = [body.index(n) for n in ["prefilter_mode", "prefilter_width", "search_min_x", "kernel_x", "cost_type", "corr_timeout",
                                   "seconds_per_op", "consistency_threshold", "min_consistency_level", "filter_half_kernel",
                                   "max_pyramid_levels", "algorithm", "blob_filter_area", "sgm_subpixel_mode", "sgm_search_buffer_x",
                                   "memory_limit_mb", "sgm_num_threads", "float* lr_disp_diff;", "int lr_disp_diff_cols",
                                   "ptrdiff_t lr_disp_diff_stride", "int region_ul_x"]]
    assert pos == sorted(pos)


def test_no_context_functions_validate_arguments():
    lib = _lib.load()
    taps = np.zeros(4, np.float32)
    assert lib.vwgpu_generate_gaussian_kernel(1.0, 9, taps.ctypes.data, 4) < 0          # cap too small
    d = np.zeros((4, 4, 3), np.int32)
    z = np.zeros((8, 8), np.int32)
    assert lib.vwgpu_subdivide_regions(d.ctypes.data, 0, 4, 7, 7, z.ctypes.data, 8) < 0   # empty image
    assert lib.vwgpu_subdivide_regions(None, 4, 4, 7, 7, z.ctypes.data, 8) < 0
    assert lib.vwgpu_strerror(-1) and lib.vwgpu_strerror(-2) and lib.vwgpu_strerror(0)
    assert lib.vwgpu_abi_version() == 1
    with pytest.raises(Exception):
        stereo.subdivide_regions(np.zeros((4, 4), np.int32), (7, 7))


def test_halo_plan_matches_the_python_partition():
    """vwgpu_halo_plan (csrc/halo.hip, what the RCCL exchange of the C ABI works from) = the bounds fetch_strip_window uses."""
    from visionworkbench_amd import partition
    for world in (1, 2, 3, 8):
        for rows in (0, 7, 1000, 32768):
            for above, below in ((0, 0), (3, 9), (173, 173), (5000, 1)):
                for rank in range(world):
                    a, b = partition.row_strip(rank, world, rows)
                    want = (a, b, max(0, a - above), min(rows, b + below))
                    assert partition.halo_plan(rank, world, rows, above, below) == want
    import pytest
    with pytest.raises(ValueError):
        partition.halo_plan(2, 2, 10, 0, 0)
