"""Timeline of workgroup 0 of the pipelined bm_sad_u8 kernel (needs a library built with -DVWX_STAMP).  GPU box only."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from visionworkbench_amd import stereo, synth, core, _lib
from visionworkbench_amd.core import BBox2i
W = 4096
sx = int(sys.argv[1]) if len(sys.argv) > 1 else 129
L, R, _ = synth.stereo_pair(W, W, 129, 1)
Lg, Rg = torch.from_numpy(L).cuda(), torch.from_numpy(R[:, :W + sx - 1].copy()).cuda()
for _ in range(3):
    stereo.calc_disparity(0, Lg, Rg, BBox2i(0, 0, W, W), (sx, 1), (7, 7))
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 2048)()
print("rc", lib.vwgpu_debug_read_stamps(buf, 2048))
a = np.frombuffer(buf, dtype=np.uint64)
for g in range(2):
    st = a[g * 1024:(g + 1) * 1024]
    tags = (st & np.uint64(0xff)).astype(int); t = (st >> np.uint64(8)).astype(np.int64)
    n = int(np.argmax(tags == 0)) if (tags == 0).any() else len(tags)
    t0 = t[0]
    print("group", g, "stamps", n)
    prev = t0
    for i in range(n):
        print("  %3d  tag %2d  t=%8.2f us  +%6.2f" % (i, tags[i], (t[i] - t0) / 100.0, (t[i] - prev) / 100.0))
        prev = t[i]
