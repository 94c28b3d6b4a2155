"""Host-side logic of libvwgpu.so that needs no GPU: the zone scheduler (vwgpu_subdivide_regions), the Gaussian tap
generator, argument validation (status codes + messages before any device work), and the ctypes struct layouts.  Runs in the
CPU suite; the zone scheduler and the taps are compared with the oracle's restatement of the reference."""
import ctypes
import os

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
                                          "memory_limit_mb", "p1", "p2", "ternary_census_threshold", "num_threads", "allow_block_cost"]
    assert S.memory_limit_mb.offset == 24 and S.allow_block_cost.offset == 48 and ctypes.sizeof(S) == 56
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
    assert lib.vwgpu_abi_version() == 3
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


def test_halo_request_check_is_collective():
    """csrc/halo.hip gathers the request header of EVERY rank (one all-gather) and each rank judges the same table
    (vwgpu_halo_headers_agree), so the ranks cannot disagree about entering the data exchange.  Three ranks, rank 0 differs:
    with the round-3 pairwise check rank 0 and rank 1 failed while rank 2 (whose header matched rank 1's) went on to wait for a
    rank 1 that had already returned; with the gathered table every rank — rank 2 included — reaches the same "no"."""
    import ctypes
    from visionworkbench_amd import _lib
    lib = _lib.load()

    def verdict(headers):
        flat = [v for h in headers for v in h]
        arr = (ctypes.c_longlong * len(flat))(*flat)
        a, b = ctypes.c_int(-1), ctypes.c_int(-1)
        return lib.vwgpu_halo_headers_agree(arr, len(headers), ctypes.byref(a), ctypes.byref(b)), a.value, b.value

    same = (4096, 3, 3, 4096 * 4)
    assert verdict([same, same, same])[0] == 1
    assert verdict([same])[0] == 1
    world = [(4096, 0, 3, 4096 * 4), same, same]                  # rank 0 passed another halo
    # what each rank sees after the all-gather is the same table: every rank returns the same verdict and names the same pair
    per_rank = [verdict(list(world)) for _rank in range(3)]
    assert per_rank[0] == per_rank[1] == per_rank[2] == (0, 0, 1)
    assert verdict([same, same, (4096, 3, 3, 4096 * 8)]) == (0, 0, 2)     # bytes per row differ on the last rank


# The eight SmoothPathAccumTask passes as the reference writes them (src/vw/Stereo/SGMAssist.h:911-1236), re-typed here from its text:
# (path predecessor, second predecessor, border test of the task, raster loops of the task).  Index = the engine's direction number.
_MGM_TASKS = [
    ("L",  (-1, 0), (0, -1), lambda c, r, lc, lr: r > 0 and c > 0,                 "rows_down_cols_right"),    # task_L  :911-955
    ("TL", (-1, -1), (1, -1), lambda c, r, lc, lr: r > 0 and c > 0 and c < lc,      "rows_down_cols_right"),    # task_TL :958-996
    ("R",  (1, 0), (0, 1),  lambda c, r, lc, lr: r < lr and c < lc,                "rows_up_cols_left"),       # task_R  :998-1033
    ("BR", (1, 1), (-1, 1), lambda c, r, lc, lr: r < lr and c > 0 and c < lc,      "rows_up_cols_left"),       # task_BR :1035-1071
    ("T",  (0, -1), (1, 0), lambda c, r, lc, lr: r > 0 and c < lc,                 "cols_left_rows_down"),     # task_T  :1147-1182
    ("BL", (-1, 1), (-1, -1), lambda c, r, lc, lr: r > 0 and r < lr and c > 0,      "cols_right_rows_up"),      # task_BL :1110-1145
    ("B",  (0, 1), (-1, 0), lambda c, r, lc, lr: r < lr and c > 0,                 "cols_right_rows_up"),      # task_B  :1073-1108
    ("TR", (1, -1), (1, 1), lambda c, r, lc, lr: r > 0 and r < lr and c < lc,      "cols_left_rows_down"),     # task_TR :1184-1219
]


def _raster(order, W, H):
    if order == "rows_down_cols_right":
        return [(c, r) for r in range(H) for c in range(W)]
    if order == "rows_up_cols_left":
        return [(c, r) for r in range(H - 1, -1, -1) for c in range(W - 1, -1, -1)]
    if order == "cols_right_rows_up":
        return [(c, r) for c in range(W) for r in range(H - 1, -1, -1)]
    return [(c, r) for c in range(W - 1, -1, -1) for r in range(H)]


@pytest.mark.parametrize("W,H", [(1, 1), (1, 7), (6, 1), (2, 2), (5, 3), (4, 9), (13, 13), (31, 8)])
def test_mgm_front_schedule_respects_the_tasks_of_the_reference(W, H):
    """vwgpu_mgm_front_pixel = what the MGM launcher enumerates (csrc/mgm_schedule.h).  For every direction: the fronts visit every pixel
    exactly once; predecessors and border test are the task's; a pixel that uses its predecessors finds both in the PREVIOUS front
    (so one launch per front is a valid order), and both precede it in the task's own raster loops (so the reference computes the same)."""
    lib = _lib.load()
    cr, pr, use = (ctypes.c_int * 2)(), (ctypes.c_int * 4)(), ctypes.c_int()
    assert lib.vwgpu_mgm_front_count(0, 4, 0) < 0 and lib.vwgpu_mgm_front_count(4, 4, 8) < 0
    for d, (name, A, B, test, order) in enumerate(_MGM_TASKS):
        nf = lib.vwgpu_mgm_front_count(W, H, d)
        assert nf == (W + H - 1 if name in "LRTB" else H if name in ("TL", "BR") else W), name
        front_of, uses = {}, {}
        for f in range(nf):
            i = 0
            while True:
                rc = lib.vwgpu_mgm_front_pixel(W, H, d, f, i, cr, pr, ctypes.byref(use))
                assert rc >= 0
                if rc == 0:
                    break
                c, r = cr[0], cr[1]
                assert 0 <= c < W and 0 <= r < H and (c, r) not in front_of, (name, c, r)
                front_of[(c, r)] = f
                uses[(c, r)] = bool(use.value)
                assert (pr[0] - c, pr[1] - r) == A and (pr[2] - c, pr[3] - r) == B, name
                assert bool(use.value) == bool(test(c, r, W - 1, H - 1)), (name, c, r)
                i += 1
            assert lib.vwgpu_mgm_front_pixel(W, H, d, f, i + 1, cr, pr, ctypes.byref(use)) == 0
        assert len(front_of) == W * H, name
        assert lib.vwgpu_mgm_front_pixel(W, H, d, nf, 0, cr, pr, ctypes.byref(use)) == 0
        position = {p: k for k, p in enumerate(_raster(order, W, H))}
        for (c, r), f in front_of.items():
            if not uses[(c, r)]:
                continue
            for dx, dy in (A, B):
                q = (c + dx, r + dy)
                assert q in front_of and front_of[q] == f - 1, (name, (c, r), q)
                assert position[q] < position[(c, r)], (name, (c, r), q)


def test_python_constants_match_the_header():
    """The option / path / error numbers of visionworkbench_amd.core are the enumerators of include/vwgpu.h."""
    import re
    from visionworkbench_amd import core
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vwgpu.h")).read()
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(VWGPU_(?:OPT|PATH)_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr)}
    assert len([k for k in enums if k.startswith("VWGPU_OPT_")]) >= 10 and len([k for k in enums if k.startswith("VWGPU_PATH_")]) >= 7
    for name, value in enums.items():
        py = name[len("VWGPU_"):]
        assert hasattr(core, py), "visionworkbench_amd.core lacks %s" % py
        assert getattr(core, py) == value, (name, value, getattr(core, py))


def test_mask_pyramid_is_zero_outside_the_halved_image_bounds(oracle):
    """A premise of the "cannot matter" certificate (csrc/bm_zones.hip ZEdge, csrc/pyramid.hip): a right-mask crop that is zero left of column
    v0 and from column v1 on (the part of the crop outside the image) stays zero, at pyramid level k of subsample_mask_by_two
    (CorrelationView.cc:38-63), left of v0 >> k and from ceil(v1 / 2^k) on — whatever the mask holds in between."""
    rng = np.random.default_rng(31)
    for _ in range(60):
        h, w = int(rng.integers(3, 70)), int(rng.integers(8, 150))
        v0 = int(rng.integers(0, w // 2)); v1 = int(rng.integers(max(v0 + 1, w // 2), w + 1))
        m = np.where(rng.random((h, w)) < 0.85, 255, 0).astype(np.uint8)
        m[:, :v0] = 0; m[:, v1:] = 0
        for level in range(1, 6):
            m = oracle.subsample_mask_by_two(m)
            if m.shape[0] < 1 or m.shape[1] < 1: break
            lo, hi = v0 >> level, (v1 + (1 << level) - 1) >> level
            assert not m[:, :lo].any(), (h, w, v0, v1, level)
            assert not m[:, hi:].any(), (h, w, v0, v1, level)


@pytest.mark.parametrize("cleanup", [0, 1])
def test_far_pointing_pixels_cannot_be_seen_after_filter_and_mask(oracle, cleanup):
    """The other premise of that certificate, on the oracle's own clean-up filter and mask pass (DisparityMap.h:357-441, :132-155): change
    the disparities (and the validity) of pixels that point half kernel + 4 or more columns outside the right mask's non-zero part to OTHER such
    values — the filtered, masked image does not change, for any neighbours."""
    rng = np.random.default_rng(77 + cleanup)
    V = np.iinfo(np.int32).max
    for trial in range(40):
        h, w = int(rng.integers(20, 60)), int(rng.integers(30, 90))
        hk = int(rng.choice([1, 2, 3, 5])); M = hk + 4
        m2w = w + int(rng.integers(10, 40)); v0 = int(rng.integers(0, 12)); v1 = m2w - int(rng.integers(0, 12))
        rmask = np.where(rng.random((h + 6, m2w)) < 0.95, 255, 0).astype(np.uint8); rmask[:, :v0] = 0; rmask[:, v1:] = 0
        lmask = np.where(rng.random((h, w)) < 0.95, 255, 0).astype(np.uint8)
        x = np.arange(w)[None, :].repeat(h, 0)
        d = np.zeros((h, w, 3), np.int32)
        d[..., 0] = rng.integers(-6, 14, (h, w)); d[..., 1] = rng.integers(-1, 3, (h, w))
        d[..., 2] = np.where(rng.random((h, w)) < 0.1, 0, V)
        far = rng.random((h, w)) < 0.25
        def far_values():
            left_side = rng.random((h, w)) < 0.5
            part = np.where(left_side, v0 - M - rng.integers(0, 20, (h, w)), v1 - 1 + M + rng.integers(0, 20, (h, w)))
            return (part - x).astype(np.int32)
        a, b = d.copy(), d.copy()
        a[..., 0][far] = far_values()[far]; b[..., 0][far] = far_values()[far]
        b[..., 1][far] = rng.integers(-3, 4, (h, w))[far]
        b[..., 2][far] = np.where(rng.random((h, w)) < 0.3, 0, V)[far]
        a[..., 2][far] = V
        outs = []
        for v in (a, b):
            f = oracle.disparity_filter(v, hk, hk, 3.0, 0.5, cleanup)
            outs.append(oracle.disparity_mask(f, lmask, rmask))
        assert np.array_equal(outs[0], outs[1]), (trial, hk, int((outs[0] != outs[1]).any(-1).sum()))
