"""vwgpu_pyramid_correlate_batch[_dev] (round 5): groups of equal-sized tiles go through the pyramid level loop together — every launch
serves the whole group, one host round trip per level (csrc/pyramid.hip, vwgpu_pyramid_group_impl).  Each tile of a batch must be IDENTICAL
to the single-tile entry on that tile (which the rest of the suite pins to the oracle), and a few batches are compared with the oracle directly
(src/vw/Stereo/CorrelationView.cc:273-886 restated)."""
import numpy as np
import pytest

import visionworkbench_amd as vwa
from visionworkbench_amd import core, stereo
from visionworkbench_amd.core import BBox2i

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    c = vwa.Context(0)
    yield c
    c.close()


def _scene(seed, H=420, W=600, floaty=False):
    rng = np.random.default_rng(seed)
    left = np.floor(rng.random((H, W)) * 256).astype(np.float32)
    if floaty:
        left = (left * np.float32(0.37) + rng.random((H, W)).astype(np.float32)).astype(np.float32)
    right = np.empty_like(left)
    for y0 in range(0, H, 70):
        right[y0:y0 + 70] = np.roll(left[y0:y0 + 70], int(rng.integers(-9, 10)), axis=1)
    return left, right


def _tiles(W, H, tw, th):
    return [BBox2i(x, y, min(tw, W - x), min(th, H - y)) for y in range(0, H, th) for x in range(0, W, tw)]


def _check(ctx, left, right, lm, rm, pf, pfw, search, kernel, cost, boxes, device=True, oracle=None, **kw):
    import torch
    box = BBox2i.from_corners(search[:2], search[2:])
    args = dict(consistency_threshold=2.0, filter_half_kernel=3, max_pyramid_levels=3)
    args.update(kw)
    if device:
        a = [torch.from_numpy(x).cuda() if x is not None else None for x in (left, right, lm, rm)]
    else:
        a = [left, right, lm, rm]
    got = stereo.pyramid_correlate_batch(a[0], a[1], a[2], a[3], pf, pfw, box, kernel, cost, boxes, ctx=ctx, **args)
    assert len(got) == len(boxes)
    for b, g in zip(boxes, got):
        one = stereo.pyramid_correlate(a[0], a[1], a[2], a[3], pf, pfw, box, kernel, cost, bbox=b, ctx=ctx, **args)
        g_, o_ = (g.cpu().numpy(), one.cpu().numpy()) if device else (g, one)
        assert g_.shape == (b.max[1] - b.min[1], b.max[0] - b.min[0], 3)
        assert np.array_equal(g_, o_), ("tile", b, int((g_ != o_).any(-1).sum()))
        if oracle is not None:
            w = oracle.pyramid_correlate(left, right, lm, rm, pf, pfw, search, kernel, cost, 0, 0.0, args["consistency_threshold"], args["filter_half_kernel"],
                                         args["max_pyramid_levels"], bbox=(b.min[0], b.min[1], b.max[0] - b.min[0], b.max[1] - b.min[1]))
            assert np.array_equal(g_, w), ("tile vs oracle", b, int((g_ != w).any(-1).sum()))
    return got


@pytest.mark.parametrize("cost,kernel,pf,floaty", [(0, (7, 7), 0, False), (2, (11, 11), 2, False), (1, (7, 7), 0, True), (2, (5, 5), 1, True), (0, (9, 9), 2, False)])
def test_batch_equals_single_tiles(ctx, oracle, cost, kernel, pf, floaty):
    """Integer SAD (order-free levels), LoG + NCC and float textures (certified levels, the "cannot matter" certificate on the border tiles),
    mean-subtracted floats: 4 x 3 tiles of 150 x 140 (the last column / row smaller: runs of equal-sized tiles form the groups)."""
    left, right = _scene(100 + cost * 7 + kernel[0], floaty=floaty)
    boxes = _tiles(600, 420, 150, 140)
    _check(ctx, left, right, None, None, pf, float(np.float32(1.4)) if pf else 0.0, (-12, -2, 13, 3), kernel, cost, boxes, oracle=oracle if cost != 1 else None)


def test_batch_with_masks_and_dead_tiles(ctx, oracle):
    """User masks (nodata mean fill per tile, one read-back for the group) including a tile whose left data is entirely masked (zeros)."""
    left, right = _scene(7)
    lm = np.full(left.shape, 255, np.uint8); rm = np.full(right.shape, 255, np.uint8)
    lm[30:200, 100:260] = 0; rm[:, -40:] = 0
    lm[140:420, 300:600] = 0                                         # covers the padded crop of the last tiles entirely
    boxes = _tiles(600, 420, 150, 140)
    _check(ctx, left, right, lm, rm, 2, float(np.float32(1.4)), (-10, -1, 11, 2), (7, 7), 2, boxes, oracle=oracle)


@pytest.mark.parametrize("kw", [dict(filter_half_kernel=0), dict(consistency_threshold=-1.0), dict(max_pyramid_levels=0), dict(max_pyramid_levels=5, filter_half_kernel=5),
                                dict(blob_filter_area=20), dict(corr_timeout=5, seconds_per_op=1e-12), dict(algorithm=1)])
def test_batch_options_and_fallbacks(ctx, kw):
    """No filter (no mask pass), no L/R check, a single level, deeper pyramids; options a group does not take (blob filter, time budget,
    SGM) must run tile by tile inside the call — same results."""
    left, right = _scene(31)
    boxes = _tiles(600, 420, 200, 210)
    cost, kernel = (3, (5, 5)) if kw.get("algorithm") else (0, (7, 7))
    _check(ctx, left, right, None, None, 0, 0.0, (-9, -1, 10, 2), kernel, cost, boxes, **kw)


def test_batch_group_sizes_and_host_entry(ctx):
    """1, 2, 16 and 17 equal tiles (a group holds at most 16), an odd tile between equal ones, the empty list; host pointers."""
    left, right = _scene(77, H=300, W=1000)
    for nt in (1, 2, 16, 17):
        boxes = [BBox2i(40 * i, 60, 96, 110) for i in range(nt)]          # overlapping tiles are fine: tiles are independent
        _check(ctx, left, right, None, None, 2, float(np.float32(1.4)), (-8, -1, 9, 2), (7, 7), 2, boxes, max_pyramid_levels=2)
    boxes = [BBox2i(0, 0, 128, 128), BBox2i(128, 0, 128, 128), BBox2i(256, 0, 100, 90), BBox2i(384, 0, 128, 128), BBox2i(512, 0, 128, 128)]
    _check(ctx, left, right, None, None, 0, 0.0, (-8, -1, 9, 2), (7, 7), 0, boxes)
    assert stereo.pyramid_correlate_batch(left, right, None, None, 0, 0.0, BBox2i.from_corners((-8, -1), (9, 2)), (7, 7), 0, [], ctx=ctx) == []
    _check(ctx, left, right, None, None, 0, 0.0, (-8, -1, 9, 2), (7, 7), 0, boxes, device=False)
    with pytest.raises(vwa.ArgumentErr):
        stereo.pyramid_correlate_batch(left, right, None, None, 0, 0.0, BBox2i.from_corners((-8, -1), (9, 2)), (6, 7), 0, boxes, ctx=ctx)


def test_host_entry_stages_per_run_of_tiles(ctx):
    """Host pointers, tiles far apart (round 6): the sources are staged per run of equal tiles, a run whose union window is much
    larger than its tiles' own windows goes tile by tile — the results are those of the single-tile entry either way."""
    left, right = _scene(91, H=900, W=1500)
    far = [BBox2i(10, 10, 96, 96), BBox2i(1300, 700, 96, 96), BBox2i(20, 780, 96, 96)]                 # one run, scattered: tile by tile
    near = [BBox2i(600 + 96 * i, 300, 96, 96) for i in range(4)]                                        # one run, adjacent: one window
    mixed = far[:1] + near + [BBox2i(1200, 40, 120, 80), BBox2i(1320, 40, 120, 80)] + far[1:]          # three runs
    for boxes in (far, near, mixed):
        _check(ctx, left, right, None, None, 0, 0.0, (-8, -1, 9, 2), (7, 7), 0, boxes, device=False, max_pyramid_levels=2)


def test_batch_at_benchmark_size(ctx, oracle):
    """Four 1024^2 tiles of the bench pair (interior, borders, corner) with the `correlate` defaults in one group: identical to the oracle."""
    import concurrent.futures
    import torch
    from visionworkbench_amd import synth
    left, right, _ = synth.stereo_pair(4096, 4096, 129, 1)
    right = np.ascontiguousarray(right[:, 64:64 + 4096])
    search = (-64, -1, 64, 1)
    boxes = [(1024, 1024, 1024, 1024), (0, 2048, 1024, 1024), (3072, 3072, 1024, 1024), (3072, 1024, 1024, 1024)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        futs = [pool.submit(oracle.pyramid_correlate, left, right, None, None, 2, float(np.float32(1.4)), search, (11, 11), 2, 0, 0.0, 2.0, 5, 5, bbox=b) for b in boxes]
        got = stereo.pyramid_correlate_batch(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), None, None, 2, float(np.float32(1.4)),
                                             BBox2i.from_corners(search[:2], search[2:]), (11, 11), 2, [BBox2i(*b) for b in boxes],
                                             consistency_threshold=2.0, filter_half_kernel=5, max_pyramid_levels=5, ctx=ctx)
        torch.cuda.synchronize()
        for b, g, f in zip(boxes, got, futs):
            w = f.result()
            g = g.cpu().numpy()
            assert np.array_equal(g, w), (b, int((g != w).any(-1).sum()))


@pytest.mark.parametrize("cost,kernel,pf", [(0, (7, 7), 0), (1, (7, 7), 0), (2, (9, 9), 0), (2, (7, 7), 2)])
def test_batch_mixed_level_classes(ctx, oracle, cost, kernel, pf):
    """Tiles of one group in DIFFERENT kernel classes on the same level: the left part of the pair is integer valued (order free, float32-exact
    under SAD), the middle a float texture (certified), one tile holds a NaN and an Inf (non-finite: the reference's order only).  Every
    class is one launch sequence over its tiles; the zone flags, "any" words and R->L buffers of the classes must not collide."""
    left, right = _scene(900 + cost, H=280, W=720)
    rng = np.random.default_rng(901)
    noise = rng.random(left.shape).astype(np.float32)
    left[:, 240:] = (left[:, 240:] * np.float32(0.37) + noise[:, 240:]).astype(np.float32)
    right[:, 240:] = (right[:, 240:] * np.float32(0.37) + noise[:, 240:][:, ::-1]).astype(np.float32)
    left[200, 650] = np.nan
    right[60, 600] = np.inf
    boxes = _tiles(720, 280, 120, 140)
    _check(ctx, left, right, None, None, pf, float(np.float32(1.4)) if pf else 0.0, (-10, -2, 11, 3), kernel, cost, boxes, oracle=oracle)
    ctx.set_option(core.OPT_CERTIFY, 0)                           # certified tiles become exact-only tiles: classes {0 / 1, 3}
    try:
        _check(ctx, left, right, None, None, pf, float(np.float32(1.4)) if pf else 0.0, (-10, -2, 11, 3), kernel, cost, boxes)
    finally:
        ctx.set_option(core.OPT_CERTIFY, 1)


@pytest.mark.parametrize("which", ["left", "right"])
def test_batch_with_one_mask_only(ctx, oracle, which):
    left, right = _scene(55)
    m = np.full(left.shape, 255, np.uint8)
    m[100:260, 200:420] = 0
    lm, rm = (m, None) if which == "left" else (None, m)
    _check(ctx, left, right, lm, rm, 0, 0.0, (-10, -1, 11, 2), (7, 7), 1, _tiles(600, 420, 150, 140), oracle=oracle)
