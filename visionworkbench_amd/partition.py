"""Row-strip partition of one stereo pair across the GPUs of a node.

Output tiles are independent units in the reference (each PyramidCorrelationView::prerasterize(bbox) works from
its own padded crop, src/vw/Stereo/CorrelationView.cc:89-97), so block matching shards with NO data-path
collective: rank g owns output rows [g*oh/N, (g+1)*oh/N) and needs the input rows of that strip plus the
ky-1 (+ sy-1 for the right image) halo rows below it, which are input data resident on that GPU.
"""


def row_strip(rank, world, out_rows):
    """Output row range [r0, r1) owned by `rank`."""
    if not (0 <= rank < world) or out_rows < 0:
        raise ValueError("bad strip request")
    return rank * out_rows // world, (rank + 1) * out_rows // world


def strip_inputs(rank, world, lh, ky, sy):
    """Input row ranges of the strip: (left rows [a, b), right rows [a, c)), for a left region of lh rows."""
    out_rows = lh - ky + 1
    r0, r1 = row_strip(rank, world, out_rows)
    return (r0, r1 + ky - 1), (r0, r1 + ky - 1 + sy - 1)


def owned_rows(rank, world, out_rows):
    """Input rows [a, b) whose HBM copy lives on `rank` when the SOURCE image itself is sharded (no overlap): the rows of the
    rank's output strip; the last rank also keeps the rows below its strip (the image's bottom ky-1 (+sy-1) rows)."""
    return row_strip(rank, world, out_rows)


def exchange_halo(owned, a, b, need_a, need_b, rank, world, bounds, group=None):
    """Halo exchange of a row-sharded image (the one real exchange step of the path): rank g holds rows
    [a, b) of an image in `owned` (rows x cols tensor) and needs rows [need_a, need_b) ⊇ [a, b) for its windows — ky-1
    (+sy-1) rows below for block matching, half_kernel * 2^levels rows on both sides (+ the search range) for a pyramid tile
    (src/vw/Stereo/CorrelationView.cc:89-97).  `bounds[r]` = (owned_a, owned_b, need_a, need_b) of every rank (sharded_bounds).  Point-to-point isend/irecv between
    the ranks whose ranges overlap — over xGMI one link per neighbour pair when the backend is nccl (= RCCL); works on
    gloo for the CPU tests.  Returns the (need_b - need_a) x cols tensor.  No collective: each rank talks only to the
    neighbours that own rows it needs."""
    import torch
    import torch.distributed as dist
    if need_a > a or need_b < b:
        raise ValueError("the needed range must contain the owned range")
    out = torch.empty((need_b - need_a, owned.shape[1]), dtype=owned.dtype, device=owned.device)
    out[a - need_a:b - need_a] = owned
    ops, keep = [], []
    for r in range(world):
        if r == rank:
            continue
        ar, br = bounds[r][0], bounds[r][1]
        # rows of rank r that I need
        lo, hi = max(need_a, ar), min(need_b, br)
        if lo < hi:
            buf = torch.empty((hi - lo, owned.shape[1]), dtype=owned.dtype, device=owned.device)
            keep.append((buf, lo, hi))
            ops.append(dist.P2POp(dist.irecv, buf, r, group=group))
    for r in range(world):
        if r == rank:
            continue
        na, nb = bounds[r][2], bounds[r][3]          # rank r's needed range
        lo, hi = max(na, a), min(nb, b)
        if lo < hi:
            ops.append(dist.P2POp(dist.isend, owned[lo - a:hi - a].contiguous(), r, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for buf, lo, hi in keep:
        out[lo - need_a:hi - need_a] = buf
    return out


def sharded_bounds(world, rows_total, out_rows, halo_above, halo_below):
    """bounds[r] = (owned_a, owned_b, need_a, need_b) for every rank: owned = the output strip's rows (the last rank also owns
    the image's remaining rows), needed = owned grown by the halos, clipped to the image."""
    bounds = []
    for r in range(world):
        a, b = row_strip(r, world, out_rows)
        if r == world - 1:
            b = rows_total
        bounds.append((a, b, max(0, a - halo_above), min(rows_total, (row_strip(r, world, out_rows)[1]) + halo_below)))
    # the last rank needs nothing below the image; make sure need >= owned everywhere
    return [(a, b, min(na, a), max(nb, b)) for (a, b, na, nb) in bounds]


# ---- pyramid / SGM tiles of a row-sharded pair ----------------------------------------------------------------------------

def pyramid_halo_rows(kernel_y, max_pyramid_levels, search_min_y, search_max_y, collar=0):
    """Rows above / below a tile that PyramidCorrelationView::prerasterize can touch (src/vw/Stereo/CorrelationView.cc:89-97:
    the tile grows by half_kernel * 2^levels, the right ROI additionally by the search range; the SGM branch's R->L runs by
    twice the search extent) — the window vwgpu_pyramid_correlate stages (csrc/pyramid.hip) — plus the collar of
    PyramidCorrelationView::rasterize (CorrelationView.h:123-133).  Returns (above, below)."""
    up = 1 << max(0, min(int(max_pyramid_levels), 12))
    sdy = max(0, search_max_y - search_min_y)
    pad = (kernel_y // 2) * up + 2 * sdy + 8 + int(collar)
    return pad - min(search_min_y, 0), pad + max(search_max_y, 0)


def strip_tiles(rank, world, rows, cols, tile=1024):
    """The output tiles (x, y, w, h) of rank's row strip: rows [rank*rows/world, (rank+1)*rows/world) cut into tile x tile
    blocks (tools/correlate.cc:266 uses 1024), raster order."""
    r0, r1 = row_strip(rank, world, rows)
    out = []
    for y in range(r0, r1, tile):
        for x in range(0, cols, tile):
            out.append((x, y, min(tile, cols - x), min(tile, r1 - y)))
    return out


def fetch_strip_window(owned, rank, world, rows_total, halo_above, halo_below, group=None):
    """The rows a rank's tiles can touch, for a source image whose rows are sharded by row_strip(): owned = rows
    [row_strip(rank)] of the image; returns (window tensor, first row of the window).  One isend/irecv per neighbour whose
    rows are needed (RCCL over xGMI on GPUs, gloo on CPU)."""
    bounds = []
    for r in range(world):
        a, b = row_strip(r, world, rows_total)
        bounds.append((a, b, max(0, a - halo_above), min(rows_total, b + halo_below)))
    a, b, na, nb = bounds[rank]
    return exchange_halo(owned, a, b, na, nb, rank, world, bounds, group=group), na


# ---- the same exchange through the engine's C ABI (csrc/halo.hip: RCCL send / recv, librccl.so opened at run time) -------------
# What bench.py --gpus N drives (the unique id travels over the process group that launched the ranks); C++ hosts that do not
# link torch use vw::engine::StripComm.  The torch.distributed functions above are the gloo-testable mirror and the fallback.

def halo_plan(rank, world, rows_total, halo_above, halo_below):
    """(owned_a, owned_b, need_a, need_b) of `rank` as vwgpu_halo_plan computes them (pure host arithmetic)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    v = [ctypes.c_int() for _ in range(4)]
    rc = lib.vwgpu_halo_plan(int(rank), int(world), int(rows_total), int(halo_above), int(halo_below), *[ctypes.byref(x) for x in v])
    if rc:
        raise ValueError("vwgpu_halo_plan: bad strip request")
    return tuple(x.value for x in v)


class EngineComm:
    """vwgpu_comm: an RCCL communicator owned by the engine (one per process / GPU)."""

    def __init__(self, ctx, unique_id, rank, world):
        import ctypes
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        ctx.check(ctx._lib.vwgpu_comm_create(ctx._h, buf, int(rank), int(world), ctypes.byref(self._h)))
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id():
        """128 bytes from ncclGetUniqueId; produced on one rank and handed to the others by the host application."""
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        rc = _lib.load().vwgpu_comm_unique_id(buf)
        if rc:
            raise RuntimeError("vwgpu_comm_unique_id failed (librccl.so not found?): rc=%d" % rc)
        return buf.raw

    def fetch_strip_window(self, owned, rows_total, halo_above, halo_below):
        """owned: this rank's rows (row_strip) of a row-sharded image, a contiguous CUDA tensor.  Returns (window, first row)."""
        import ctypes
        import torch
        a, b, na, nb = halo_plan(self.rank, self.world, rows_total, halo_above, halo_below)
        if owned.shape[0] != b - a or not owned.is_contiguous():
            raise ValueError("owned must hold rows [%d, %d) contiguously" % (a, b))
        win = torch.empty((nb - na, owned.shape[1]), dtype=owned.dtype, device=owned.device)
        first = ctypes.c_int()
        self._ctx.set_stream(torch.cuda.current_stream(owned.device).cuda_stream)
        self._ctx.check(self._ctx._lib.vwgpu_fetch_strip_window_dev(self._ctx._h, self._h, owned.data_ptr(), owned.shape[1], owned.element_size(),
                                                                     int(rows_total), int(halo_above), int(halo_below), win.data_ptr(),
                                                                     ctypes.byref(first)))
        return win, first.value

    def close(self):
        if getattr(self, "_h", None):
            self._ctx._lib.vwgpu_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass
