"""CPU restatement of COCO run-length encoding.  TEST INFRASTRUCTURE ONLY (tests/ and bench.py's cpu leg).

The reference calls `pycocotools.mask.encode(np.asfortranarray(seg.astype(np.uint8)))` (utils/common_utils.py:88-96).
pycocotools (cocoapi PythonAPI; un-vendored, no version pinned by the reference's README) is ABSENT from this image, so the
published algorithm of cocoapi `common/maskApi.c` is restated here — `rleEncode`, `rleToString`, `rleFrString`, `rleDecode` —
and **parity is unpinned by the reference**: it rests on hand-derived known-answer vectors (tests/test_oracle_rle.py) and on
the encode -> decode round trip.
"""
import numpy as np


def rle_counts(mask):
    """maskApi.c rleEncode: run lengths of the column-major (Fortran) pixel order, alternating 0s/1s, starting with 0s."""
    flat = np.asarray(mask).astype(bool).reshape(mask.shape[0], mask.shape[1]).T.reshape(-1)    # column-major order
    counts, prev, run = [], False, 0
    for v in flat:
        if v != prev:
            counts.append(run)
            run, prev = 0, v
        run += 1
    counts.append(run)
    return counts


def rle_counts_fast(mask):
    """Same as rle_counts, vectorised (for full-size masks)."""
    flat = np.asarray(mask).astype(bool).T.reshape(-1)
    change = np.flatnonzero(np.concatenate(([flat[0]], flat[1:] != flat[:-1])))   # positions where a new run starts (k=0 iff it is 1)
    edges = np.concatenate(([0], change, [flat.size]))
    return np.diff(edges).tolist()


def rle_to_string(counts):
    """maskApi.c rleToString (LEB128-like, 5 bits per char, delta against the count two places back from the 4th on)."""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5                                   # arithmetic shift, like the C `long`
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return out.decode('ascii')


def rle_from_string(s):
    """maskApi.c rleFrString."""
    counts, p, b = [], 0, s.encode('ascii')
    while p < len(b):
        x, k, more = 0, 0, True
        while more:
            c = b[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(counts, h, w):
    """maskApi.c rleDecode -> [h, w] uint8."""
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for c in counts:
        flat[pos:pos + c] = v
        pos += c
        v ^= 1
    return flat.reshape(w, h).T


def encode(mask):
    """-> {'size': [h, w], 'counts': str}: what the reference stores in its mask json."""
    h, w = mask.shape
    return {'size': [h, w], 'counts': rle_to_string(rle_counts_fast(mask))}
