"""Generates tests/golden/sgm_fixture.npz from the reference's own SGM test images
(src/vw/Stereo/tests/left.tif, left_const_offset.tif — uncompressed 8-bit strip TIFFs read with the 40-line parser
below).  Run in the build container, where /root/reference exists; the GPU box only sees the committed .npz.

    python tests/golden/make_sgm_fixture.py

Stored: the crops TestSGM.cxx:33-45 takes — left = [0,400)^2, right = rightRoi = [-4, 405)^2 with the out-of-image
border (4 px at the top/left) zero-filled."""
import os
import struct

import numpy as np

REF = "/root/reference/src/vw/Stereo/tests"


def read_tiff_u8(path):
    d = open(path, "rb").read()
    bo = "<" if d[:2] == b"II" else ">"
    off = struct.unpack(bo + "I", d[4:8])[0]
    n = struct.unpack(bo + "H", d[off:off + 2])[0]
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack(bo + "HHII", d[off + 2 + 12 * i:off + 14 + 12 * i])
        if typ == 3 and cnt == 1:
            val &= 0xffff
        tags[tag] = (typ, cnt, val)
    w, h = tags[256][2], tags[257][2]
    assert tags[258][2] == 8 and tags[259][2] == 1 and tags[277][2] == 1, "expected uncompressed 8-bit gray"
    cnt = tags[273][1]

    def arr(tag):
        typ, c, val = tags[tag]
        if c == 1:
            return [val]
        fmt = bo + ("I" if typ == 4 else "H") * c
        return list(struct.unpack(fmt, d[val:val + (4 if typ == 4 else 2) * c]))
    offs, lens = arr(273), arr(279)
    data = b"".join(d[o:o + l] for o, l in zip(offs, lens))
    return np.frombuffer(data[:w * h], np.uint8).reshape(h, w).copy()


def main():
    left = read_tiff_u8(os.path.join(REF, "left.tif"))
    right = read_tiff_u8(os.path.join(REF, "left_const_offset.tif"))
    l = left[:400, :400]
    r = np.zeros((409, 409), np.uint8)
    r[4:, 4:] = right[:405, :405]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sgm_fixture.npz")
    np.savez_compressed(out, left=l, right=r)
    print(out, l.shape, r.shape, os.path.getsize(out))


if __name__ == "__main__":
    main()
