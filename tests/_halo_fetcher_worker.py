"""Worker of tests/test_bench_contract.py::test_halo_fetcher_decisions_are_collective (world 2, gloo, CPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import visionworkbench_amd as vwa  # noqa: E402
from visionworkbench_amd import partition  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)


class Dev:            # what HaloFetcher reads of a torch.device
    index = 0


dev = torch.device("cpu")
f = bench.HaloFetcher(torch, dist, vwa, partition, rank, world, dev)
assert f.comm is None and f.how.startswith("torch.distributed"), f.how          # no GPU: every rank agreed on the mirror
img = np.arange(101 * 7, dtype=np.float32).reshape(101, 7)
a, b = partition.row_strip(rank, world, 101)
win, first = f.fetch(torch.from_numpy(img[a:b].copy()), 101, 9, 4)
na, nb = max(0, a - 9), min(101, b + 4)
assert first == na and np.array_equal(win.numpy(), img[na:nb])
assert f.agree(True) and not f.agree(rank == 0)                                  # one dissenting rank decides for all
try:                                                                             # ranks that were handed different halos: EVERY rank raises,
    f.fetch(torch.from_numpy(img[a:b].copy()), 101, 9 if rank == 0 else 8, 4)    # before anyone posts a send or a receive
    raise SystemExit("a disagreeing request went through on rank %d" % rank)
except ValueError as e:
    assert "disagree" in str(e)
f.close()
dist.barrier()
print("halo fetcher ok")
dist.destroy_process_group()
