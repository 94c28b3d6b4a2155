"""Two-rank check of the engine's RCCL halo exchange (csrc/halo.hip) — needs TWO GPUs (RCCL refuses two ranks on one device:
"invalid usage"), so it cannot run on the 1-GPU gpurun box; on a multi-GPU node:
    python tools/rccl_two_rank_check.py 0 & python tools/rccl_two_rank_check.py 1
The unique id travels through /tmp/vwgpu_id.bin."""
import sys, os, time
sys.path.insert(0, ".")
import torch
import visionworkbench_amd as vwa
from visionworkbench_amd import partition
rank, world = int(sys.argv[1]), 2
idf = "/tmp/vwgpu_id.bin"
dev = rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
ctx = vwa.Context(dev)
if rank == 0:
    uid = partition.EngineComm.unique_id()
    open(idf + ".tmp", "wb").write(uid); os.rename(idf + ".tmp", idf)
else:
    while not os.path.exists(idf): time.sleep(0.05)
    uid = open(idf, "rb").read()
try:
    comm = partition.EngineComm(ctx, uid, rank, world)
    rows = 40
    full = torch.arange(rows * 16, dtype=torch.float32, device="cuda").reshape(rows, 16)
    a, b = partition.row_strip(rank, world, rows)
    win, first = comm.fetch_strip_window(full[a:b].contiguous(), rows, 4, 6)
    torch.cuda.synchronize()
    _, _, na, nb = partition.halo_plan(rank, world, rows, 4, 6)
    print("rank", rank, "ok", first == na and torch.equal(win, full[na:nb]))
    comm.close()
except Exception as e:
    print("rank", rank, "failed:", str(e)[:300])
