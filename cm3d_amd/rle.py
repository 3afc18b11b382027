"""Host-side COCO-RLE codec (the on-disk mask format of the hot path's input).

The reference reads `<frame>_masks.pkl`, a pickled list of COCO RLE dicts
{'size': [W_img, H_img], 'counts': bytes}, each encoding the (W,H)-transposed
mask in column-major order (reference src/nuscenes/gen_2d_masks_detic.py:468-472,
decoded at src/nuscenes/2d_to_3d.py:425).  Column-major over (h=W_img, w=H_img)
is row-major over the (H_img, W_img) image, so the run lengths walk the image
row by row -- which is how the kernels consume them (`cm3d_rle_*`).

This module only converts between the compressed string and uint32 run
lengths (vectorised numpy, no per-character Python loop) and builds run lengths
from row spans for the synthetic generator.  Expansion to pixels happens on the
GPU.
"""
import numpy as np


def string_to_counts(s: bytes) -> np.ndarray:
    """COCO compressed string -> uint32 run lengths (pycocotools rleFrString)."""
    if len(s) == 0:
        return np.zeros(0, np.uint32)
    b = np.frombuffer(s, np.uint8).astype(np.int64) - 48
    more = (b & 0x20) != 0
    ends = np.flatnonzero(~more)
    if ends.size == 0 or ends[-1] != b.size - 1:
        raise ValueError("malformed RLE string")
    starts = np.empty_like(ends)
    starts[0] = 0
    starts[1:] = ends[:-1] + 1
    grp = np.zeros(b.size, np.int64)          # index of the value each char belongs to
    grp[starts[1:]] = 1
    grp = np.cumsum(grp)
    k = np.arange(b.size) - starts[grp]       # position of the 5-bit group inside its value
    if k.max() > 12:
        raise ValueError("malformed RLE string")
    x = np.zeros(ends.size, np.int64)
    np.add.at(x, grp, (b & 0x1F) << (5 * k))
    nk = ends - starts + 1
    neg = (b[ends] & 0x10) != 0
    x = np.where(neg, x | (np.int64(-1) << (5 * nk)), x)
    # values from the 4th on are deltas against the value two positions back
    c = x.copy()
    if c.size > 3:
        odd = np.arange(1, c.size, 2)
        c[odd] = np.cumsum(x[odd])
        even = np.arange(2, c.size, 2)
        c[even] = np.cumsum(x[even])
    if (c < 0).any() or (c > 0xFFFFFFFF).any():
        raise ValueError("malformed RLE string")
    return c.astype(np.uint32)


def counts_to_string(cnts) -> bytes:
    """uint32 run lengths -> COCO compressed string (pycocotools rleToString)."""
    c = np.asarray(cnts, np.int64)
    if c.size == 0:
        return b""
    x = c.copy()
    if c.size > 3:
        x[3:] = c[3:] - c[1:-2]
    # number of 5-bit groups: smallest k with -2^(5k-1) <= x < 2^(5k-1)
    nk = np.ones(x.size, np.int64)
    for k in range(1, 8):
        lim = np.int64(1) << (5 * k - 1)
        nk = np.where((x >= lim) | (x < -lim), k + 1, nk)
    off = np.concatenate([[0], np.cumsum(nk)])
    total = int(off[-1])
    vid = np.repeat(np.arange(x.size), nk)
    k = np.arange(total) - off[vid]
    ch = (x[vid] >> (5 * k)) & 0x1F
    ch = np.where(k < nk[vid] - 1, ch | 0x20, ch)
    return (ch + 48).astype(np.uint8).tobytes()


def spans_to_counts(rows, x0, x1, W, H) -> np.ndarray:
    """Row spans (row y, inclusive columns x0..x1, sorted by y, one span per row)
    -> run lengths over the row-major (H,W) image, starting with a 0-run."""
    rows = np.asarray(rows, np.int64)
    p0 = rows * W + np.asarray(x0, np.int64)
    p1 = rows * W + np.asarray(x1, np.int64) + 1
    total = int(W) * int(H)
    if rows.size == 0:
        return np.array([total], np.uint32)
    # merge spans that touch in linear order (x1 == W-1 followed by x0 == 0)
    join = p0[1:] == p1[:-1]
    keep_start = np.concatenate([[True], ~join])
    keep_end = np.concatenate([~join, [True]])
    s, e = p0[keep_start], p1[keep_end]
    edges = np.empty(2 * s.size, np.int64)
    edges[0::2], edges[1::2] = s, e
    cnts = np.diff(np.concatenate([[0], edges]))
    tail = total - int(e[-1])
    if tail > 0:
        cnts = np.concatenate([cnts, [tail]])
    return cnts.astype(np.uint32)


def dense_to_counts(img_hw) -> np.ndarray:
    """(H,W) image (non-zero = set) -> run lengths."""
    flat = (np.asarray(img_hw).reshape(-1) != 0)
    if flat.size == 0:
        return np.zeros(0, np.uint32)
    chg = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], chg, [flat.size]])
    cnts = np.diff(edges)
    if flat[0]:
        cnts = np.concatenate([[0], cnts])
    return cnts.astype(np.uint32)


def counts_to_dense(cnts, W, H) -> np.ndarray:
    """run lengths -> (H,W) uint8 image of 0/1 (host reference expansion, used by
    tools and tests; the product path expands on the GPU)."""
    cnts = np.asarray(cnts, np.int64)
    if int(cnts.sum()) != W * H:
        raise ValueError("RLE run lengths do not match size")
    vals = (np.arange(cnts.size) & 1).astype(np.uint8)
    return np.repeat(vals, cnts).reshape(H, W)


def encode_mask(img_hw) -> dict:
    """(H,W) image -> COCO RLE dict as the reference's producer writes it
    (size = [W_img, H_img], i.e. the shape of the transposed array)."""
    H, W = np.asarray(img_hw).shape
    return {"size": [int(W), int(H)], "counts": counts_to_string(dense_to_counts(img_hw))}
