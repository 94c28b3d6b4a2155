"""N > 1 path of the benchmark / engine on CPU: world_size-2 `gloo` processes each take one row strip (with its
halo rows), compute it with the oracle standing in for the GPU kernel, and rank 0 reassembles the image, which must
equal the single-call result bit for bit.  This is the partition + halo arithmetic bench.py uses (no collective is
needed on the data path; the gather here only serves the check)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visionworkbench_amd import partition, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kernel, search, w, h, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32)
    (la, lb), (ra, rb) = partition.strip_inputs(rank, world, h, kernel[1], search[1])
    part = oracle.calc_disparity(0, left[la:lb], right[ra:rb], kernel, search)
    r0, r1 = partition.row_strip(rank, world, h - kernel[1] + 1)
    assert part.shape[0] == r1 - r0
    parts = [None] * world
    dist.all_gather_object(parts, part)
    dist.barrier()
    if rank == 0:
        q.put(np.concatenate(parts, axis=0))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_strips_reassemble_exactly(oracle, world):
    kernel, search, w, h = (7, 7), (17, 2), 96, 61
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kernel, search, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32)
    want = oracle.calc_disparity(0, left, right, kernel, search)
    assert np.array_equal(got, want)


def test_strip_ranges_cover_everything():
    for world in (1, 2, 4, 8):
        rows = [partition.row_strip(r, world, 4090) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == 4090
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        (la, lb), (ra, rb) = partition.strip_inputs(world - 1, world, 4096, 7, 1)
        assert lb == 4096 and rb == 4096


def _halo_worker(rank, world, port, kernel, search, w, h, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32)
    oh = h - kernel[1] + 1
    lb = partition.sharded_bounds(world, left.shape[0], oh, 0, kernel[1] - 1)
    rb = partition.sharded_bounds(world, right.shape[0], oh, 0, kernel[1] - 1 + search[1] - 1)
    a, b, na, nb = lb[rank]
    l_own = torch.from_numpy(left[a:b].copy())                    # this rank only ever sees its own rows
    l_ext = partition.exchange_halo(l_own, a, b, na, nb, rank, world, lb)
    a2, b2, na2, nb2 = rb[rank]
    r_own = torch.from_numpy(right[a2:b2].copy())
    r_ext = partition.exchange_halo(r_own, a2, b2, na2, nb2, rank, world, rb)
    assert np.array_equal(l_ext.numpy(), left[na:nb]) and np.array_equal(r_ext.numpy(), right[na2:nb2])
    r0, r1 = partition.row_strip(rank, world, oh)
    part = oracle.calc_disparity(0, l_ext.numpy()[:r1 - r0 + kernel[1] - 1], r_ext.numpy()[:r1 - r0 + kernel[1] - 1 + search[1] - 1], kernel, search)
    parts = [None] * world
    dist.all_gather_object(parts, part)
    dist.barrier()
    if rank == 0:
        q.put(np.concatenate(parts, axis=0))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_of_a_sharded_source_image(oracle, world):
    """The source image itself is row-sharded (no overlap); neighbours exchange the ky-1 (+sy-1) halo rows point to
    point (nccl/RCCL on GPUs, gloo here); results must reassemble to the single-call disparity."""
    kernel, search, w, h = (7, 7), (17, 3), 96, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, kernel, search, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    left, right, _ = synth.stereo_pair(w, h, search[0], search[1], block=32)
    assert np.array_equal(got, oracle.calc_disparity(0, left, right, kernel, search))


# ---- pyramid tiles and SGM strips of a row-sharded pair (BASELINE configs 4 and 5) -------------------------------------------

def _tile_worker(rank, world, port, sgm, collar, q):
    """Each rank owns a row strip of BOTH images, fetches the rows its tiles can touch from the neighbours (pyramid padding
    half_kernel * 2^levels + search, CorrelationView.cc:89-97; + the collar, CorrelationView.h:123-133) and correlates its
    tiles from that window alone."""
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, levels, k = 200, 240, 2, 7
    search = (-6, -1, 7, 2)
    left, right, _ = synth.stereo_pair(w, h, 13, 1, block=32)
    right = np.ascontiguousarray(right[:, 6:6 + w])
    a, b = partition.row_strip(rank, world, h)
    above, below = partition.pyramid_halo_rows(k, levels, search[1], search[3], collar)
    lwin, l0 = partition.fetch_strip_window(torch.from_numpy(left[a:b].copy()), rank, world, h, above, below)
    rwin, r0 = partition.fetch_strip_window(torch.from_numpy(right[a:b].copy()), rank, world, h, above, below)
    assert l0 == r0 == max(0, a - above)
    lwin, rwin = lwin.numpy(), rwin.numpy()
    assert np.array_equal(lwin, left[l0:l0 + lwin.shape[0]])
    parts = []
    for (x, y, tw, th) in partition.strip_tiles(rank, world, h, w, tile=96):
        # tile + collar semantics: correlate the collared box, keep its centre (the box may leave the image: edge extension)
        big = (x - collar, y - collar - l0, tw + 2 * collar, th + 2 * collar)
        if sgm:
            t = oracle.pyramid_correlate_sgm(lwin, rwin, None, None, search, k, 3, 2.0, 0, 3, levels, bbox=big, algorithm=int(sgm))
        else:
            t = oracle.pyramid_correlate(lwin, rwin, None, None, 0, 0.0, search, (k, k), 0, 0, 0.0, 2.0, 3, levels, bbox=big)
        parts.append(((x, y, tw, th), t[collar:collar + th, collar:collar + tw].copy()))
    allparts = [None] * world
    dist.all_gather_object(allparts, parts)
    dist.barrier()
    if rank == 0:
        out = np.zeros((h, w, 3), np.float32)
        for pl in allparts:
            for (x, y, tw, th), t in pl:
                out[y:y + th, x:x + tw] = t
        q.put(out)
    dist.destroy_process_group()


@pytest.mark.parametrize("sgm,collar", [(0, 0), (1, 16), (3, 16)])
def test_sharded_source_pyramid_tiles_and_sgm_collar(oracle, sgm, collar):
    """World 2: the reassembled tiles equal the same (collared) tiles computed from the whole pair.  sgm = the CorrelationAlgorithm
    (0 block matching, 1 SGM, 3 FINAL_MGM: the 2-D recurrence of MGM couples a tile no further than its collared box either)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, sgm, collar, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w, h, levels, k = 200, 240, 2, 7
    search = (-6, -1, 7, 2)
    left, right, _ = synth.stereo_pair(w, h, 13, 1, block=32)
    right = np.ascontiguousarray(right[:, 6:6 + w])
    want = np.zeros((h, w, 3), np.float32)
    for rank in range(world):
        for (x, y, tw, th) in partition.strip_tiles(rank, world, h, w, tile=96):
            big = (x - collar, y - collar, tw + 2 * collar, th + 2 * collar)
            if sgm:
                t = oracle.pyramid_correlate_sgm(left, right, None, None, search, k, 3, 2.0, 0, 3, levels, bbox=big, algorithm=int(sgm))
            else:
                t = oracle.pyramid_correlate(left, right, None, None, 0, 0.0, search, (k, k), 0, 0, 0.0, 2.0, 3, levels, bbox=big)
            want[y:y + th, x:x + tw] = t[collar:collar + th, collar:collar + tw]
    assert np.array_equal(got[..., 2], want[..., 2])
    assert np.abs(got - want).max() < 1e-6
    assert (got[..., 2] != 0).mean() > 0.5
