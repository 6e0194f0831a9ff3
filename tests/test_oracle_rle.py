"""CPU: the oracle's COCO-RLE restatement (oracle/rle_ref.py) against hand-derived known answers (pycocotools is absent from the
image and the reference holds no RLE fixtures: parity unpinned by the reference) and the encode -> decode round trip."""
import numpy as np

from oracle import rle_ref as R


def test_known_answers():
    # 2x2, column-major order = (0,0),(1,0),(0,1),(1,1)
    m = np.array([[0, 1], [1, 1]])                    # column-major 0,1,1,1 -> runs [1,3]
    assert R.rle_counts(m) == [1, 3] and R.rle_to_string([1, 3]) == '13'
    m = np.array([[1, 0], [0, 0]])                    # starts with foreground: leading zero-length run
    assert R.rle_counts(m) == [0, 1, 3] and R.rle_to_string([0, 1, 3]) == '013'
    assert R.rle_counts(np.zeros((3, 4))) == [12] and R.rle_to_string([12]) == '<'       # 12 + 48 = '<'
    # multi-char count: 100 = 0b11_00100 -> low 5 bits 00100 | continuation 0x20 -> 36+48='T', then 3 -> '3'
    assert R.rle_to_string([100]) == 'T3'
    # bit 4 set in the last group needs an extra char: 16 = 0b10000 -> (16|32)+48 = '`', then 0 -> '0'
    assert R.rle_to_string([16]) == '`0'
    # 4th count is stored as a difference to the 2nd: [5, 7, 2, 4] -> 5, 7, 2, (4-7 = -3)
    # -3 & 31 = 29 (bit 4 set), -3 >> 5 = -1 -> terminates: 29 + 48 = 'M'
    assert R.rle_to_string([5, 7, 2, 4]) == '572M'
    assert R.rle_from_string('572M') == [5, 7, 2, 4] and R.rle_from_string('T3') == [100] and R.rle_from_string('`0') == [16]


def test_round_trip_and_fast_path():
    rng = np.random.default_rng(0)
    for h, w in ((1, 1), (7, 5), (33, 47), (120, 160)):
        for density in (0.0, 0.03, 0.5, 1.0):
            m = (rng.random((h, w)) < density).astype(np.uint8)
            c = R.rle_counts(m)
            assert c == R.rle_counts_fast(m) and sum(c) == h * w
            s = R.rle_to_string(c)
            assert R.rle_from_string(s) == c
            np.testing.assert_array_equal(R.rle_decode(c, h, w), m)
    blob = np.zeros((480, 640), dtype=np.uint8)
    blob[100:300, 200:420] = 1
    e = R.encode(blob)
    assert e['size'] == [480, 640]
    np.testing.assert_array_equal(R.rle_decode(R.rle_from_string(e['counts']), 480, 640), blob)
