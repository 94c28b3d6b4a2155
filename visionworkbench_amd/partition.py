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
