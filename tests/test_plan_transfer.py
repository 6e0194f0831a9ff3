"""Plans for layer shapes without a row in the tuned table (yolact_minimal_amd/plan_transfer.py): host logic, no GPU.

The reference accepts any `--img_size` that is a multiple of 32 (config.py:75); the table is keyed on exact shapes.  A shape
without a row takes the row of the nearest tuned shape of its family, re-derived for its M; every transferred row has to be a
plan the library accepts (ym_conv2d_* host-side planning, which runs without a GPU)."""
import ctypes
import json
import os

import pytest

from yolact_minimal_amd import plan_transfer as PT

TABLE = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_minimal_amd',
                                    'tuned_gfx950.json')))


def test_every_table_key_parses():
    bad = [k for k in TABLE if PT.parse(k) is None]
    assert not bad, bad[:5]


def test_exact_row_wins(monkeypatch):
    monkeypatch.delenv('YM_TUNED_NEAREST', raising=False)
    sig = 'M1156_N1024_C256_k1_s1_seg1_r1'
    row, src = PT.lookup(TABLE, sig, 1156, 1024, 8)
    assert src == 'table' and row == TABLE[sig]


def test_nearest_is_the_same_layer_at_the_nearest_resolution(monkeypatch):
    monkeypatch.delenv('YM_TUNED_NEAREST', raising=False)
    # layer3 conv3 of a ResNet at 832 px (52 x 52 = 2704 pixels): no row; same N / C / filter / residual family
    sig = 'M2704_N1024_C256_k1_s1_seg1_r1'
    assert sig not in TABLE
    key, m_d = PT.nearest(TABLE, sig)
    fam, M, N, C = PT.parse(key)
    assert (N, C) == (1024, 256) and fam == PT.parse(sig)[0]
    others = [PT.parse(k)[1] for k in TABLE if PT.parse(k)[0] == fam and PT.parse(k)[2:] == (1024, 256)]
    import math
    assert abs(math.log2(M / 2704)) == min(abs(math.log2(m / 2704)) for m in others)
    row, src = PT.lookup(TABLE, sig, 2704, 1024, 8)
    assert src == 'nearest:' + key and row[0] in (32, 64, 128)


def test_no_donor_within_reach_means_heuristic(monkeypatch):
    monkeypatch.delenv('YM_TUNED_NEAREST', raising=False)
    row, src = PT.lookup(TABLE, 'M1156_N7777_C32_k5_s1_seg1_r0', 1156, 7777, 25)
    assert row is None and src == 'heuristic'
    # a stem (Cin = 4) never donates to / borrows from a Cin % 32 == 0 layer
    assert PT.parse('M73984_N64_C4_k7_s2_seg1_r0')[0] != PT.parse('M73984_N64_C32_k7_s2_seg1_r0')[0]


def test_switches(monkeypatch):
    sig = 'M1156_N1024_C256_k1_s1_seg1_r1'
    monkeypatch.setenv('YM_TUNED_NEAREST', '0')
    assert PT.lookup(TABLE, 'M2704_N1024_C256_k1_s1_seg1_r1', 2704, 1024, 8) == (None, 'heuristic')
    assert PT.lookup(TABLE, sig, 1156, 1024, 8)[1] == 'table'
    monkeypatch.setenv('YM_TUNED_NEAREST', 'only')
    row, src = PT.lookup(TABLE, sig, 1156, 1024, 8)
    assert src.startswith('nearest:') and src != 'nearest:' + sig


def test_k_split_keeps_the_workgroup_count():
    # donor: 64x64 tiles, 19 x 4 = 76 tiles, K split 6 = 456 workgroups; a quarter of the pixels -> 5 x 4 = 20 tiles
    row = PT.transfer_conv([64, 64, 6, 0, 22, 0, 0], 1156, 289, 256, 72)
    assert row[:2] == [64, 64] and row[3:5] == [0, 22]
    assert row[2] in (16, 24) and row[2] * 20 >= 300
    # never more slices than half the K tiles
    row = PT.transfer_conv([64, 64, 6, 0, 22, 0, 0], 1156, 289, 256, 8)
    assert row[2] <= 4


def test_tail_follows_the_donor_rule_and_the_tuner_limits():
    # donor 64x64: 145 x 4 = 580 tiles, tail 68 = 580 % 256; new M -> 100 x 4 = 400 tiles -> tail 144
    row = PT.transfer_conv([64, 64, 1, 0, 22, 68, 3], 9248, 6400, 256, 32)
    assert row[5:7] == [144, 3]
    # no arrival counters / segmented output / too few tiles: no tail
    assert PT.transfer_conv([64, 64, 1, 0, 22, 68, 3], 9248, 6400, 256, 32, counters=False)[5:7] == [0, 0]
    assert PT.transfer_conv([64, 64, 1, 0, 22, 68, 3], 9248, 6400, 256, 32, nseg=3)[5:7] == [0, 0]
    assert PT.transfer_conv([64, 64, 1, 0, 22, 68, 3], 9248, 1600, 256, 32)[5:7] == [0, 0]
    # a whole number of rounds: nothing to split
    assert PT.transfer_conv([64, 64, 1, 0, 22, 68, 3], 9248, 64 * 128, 256, 32)[5:7] == [0, 0]
    # the wave kernel's tail: 32x32 tiles, four K waves
    row = PT.transfer_conv([32, 32, 1, 4, 22, 40, 6], 1156, 1600, 256, 72)
    assert row[:5] == [32, 32, 1, 4, 22] and row[5:7] == [400 % 256, 6]


def test_zero_row_and_wgrad():
    assert PT.transfer_conv([0, 0, 0, 0, 0, 0, 0], 1156, 400, 256, 8) == [0, 0, 0, 0, 0, 0, 0]
    assert PT.transfer_wgrad([32, 22], 9248, 3200) == [25, 22]
    assert PT.transfer_wgrad([14, 24], 9248, 3200) == [14, 24]
    assert PT.transfer_wgrad([0, 22], 9248, 3200) == [0, 22]


def _lib():
    from yolact_minimal_amd import hip
    try:
        return hip, hip.lib()
    except Exception as e:       # the .so is built by __graft_entry__.build(); without it there is nothing to check here
        pytest.skip(f'libyolact_hip.so not loadable: {e}')


def _desc(hip, M_side, N, C, k, s, nseg, residual, transposed=False):
    d = hip.ConvDesc()
    d.inp = d.weight = 0x10000
    d.residual = 0x10000 if residual else None
    pad = k // 2
    if transposed:         # data gradient: (Ho, Wo) = dx, (H, W) = dy
        d.Ho = d.Wo = M_side
        d.H = d.W = (M_side + 2 * pad - k) // s + 1
    else:
        d.Ho = d.Wo = M_side
        d.H = d.W = (M_side - 1) * s + k - 2 * pad
    d.B, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = 1, C, N, k, k, s, pad
    d.k_pad = -(-(k * k * C) // 32) * 32
    d.nseg = nseg
    step = -(-N // nseg)
    for i in range(nseg):
        d.seg[i].n_begin, d.seg[i].n_end = i * step, min(N, (i + 1) * step)
        d.seg[i].out = 0x10000
        d.seg[i].batch_stride, d.seg[i].pitch = M_side * M_side * N, N
    d.transposed = int(transposed)
    d.tile_counters = 0x10000
    return d


def _launchable(row, nkt):
    """What the launch dispatch behind ym_conv2d_fwd builds (csrc/conv_wave.hip dispatch_kw / dispatch_dma, conv_mfma.hip): the
    host-side planner does not look at these, a launch would fail with YM_EINVAL."""
    tm, tn, ks, kw, st = row[:5]
    g = row[7] if len(row) > 7 else 0
    if (tm, tn) == (0, 0):
        return True
    if kw:
        if kw not in (1, 2, 4, 8) or kw > max(1, nkt):
            return False
        if 22 <= st <= 24:                                       # wave kernel with DMA rings
            if (tm, tn) not in ((32, 32), (64, 32), (32, 64)) or kw == 8 or g not in (0, 1, 2, 4) or (g and g < kw):
                return False
            return st < 24 or (tm, tn) == (32, 32)
        return (tm, tn) in ((32, 32), (64, 32), (32, 64), (64, 64)) and not (kw == 8 and tm * tn == 4096)
    if 52 <= st <= 54:
        return (tm, tn) in ((64, 256), (128, 128), (256, 64)) or (tm, tn) in ((64, 64), (128, 64), (64, 128))
    return (tm, tn) in ((64, 64), (128, 64), (64, 128), (128, 128)) and ks in (0,) + PT._KS_ALLOWED


def test_transferred_rows_are_plans_the_library_accepts(monkeypatch):
    """Every forward / data-gradient family of the table, at the layer sizes of 256 ... 800 px images, planned from the nearest
    row only (exact rows ignored): the host-side planner of the library must accept each one."""
    monkeypatch.setenv('YM_TUNED_NEAREST', 'only')
    hip, L = _lib()
    L.ym_conv2d_tile_counters.restype = ctypes.c_int
    checked = bad = 0
    fams = {}
    for key in TABLE:
        p = PT.parse(key)
        if p and p[0][0] in ('', 'T_') and not p[0][5] and p[0][6] in ('', '_st') and not p[0][7]:
            fams.setdefault((p[0], p[2], p[3]), key)
    # (+ the same families with half / twice the channels on either side: donors within 2x in N and C are in reach, e.g. Swin-T's
    #  C = 192 Linear layers at a size whose nearest measured shape has C = 384)
    variants = {}
    for (fam, N, C), key in fams.items():
        for n2, c2 in ((N, C), (N, C // 2), (N, C * 2), (N // 2, C), (N * 2, C)):
            if c2 % 32 == 0 and c2 >= 32 and n2 >= 16 and (int(fam[3] or 1) == 1 or n2 == N):
                variants.setdefault((fam, n2, c2), key)
    for (fam, N, C), key in sorted(variants.items(), key=str):
        pre, k, s, seg, r, lev, suf, stem = fam
        for side in (8, 10, 13, 16, 20, 23, 26, 32, 40, 46, 50, 64, 80, 92, 100, 128, 160, 184, 200):
            M = side * side
            sig = f'{pre}M{M}_N{N}_C{C}_k{k}_s{s}' + (f'_seg{seg}_r{r}' if seg else '') + suf
            nseg = int(seg) if seg else 1
            d = _desc(hip, side, N, C, k, s, nseg, r == '1', transposed=(pre == 'T_'))
            row, src = PT.lookup(TABLE, sig, M, N, d.k_pad // 32, nseg)
            if row is None:
                continue
            assert src.startswith('nearest:')
            d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = row[0], row[1], row[2], row[3], row[4]
            d.tail_tiles, d.tail_ksplit = row[5], row[6]
            d.grid_wgs = row[7] if len(row) > 7 else 0
            if d.kwaves and pre == 'T_':
                continue
            assert _launchable(row, d.k_pad // 32), (sig, src, row)
            # sentinel: a descriptor the planner refuses leaves its own message; an accepted plan leaves the sentinel in place
            bad_d = _desc(hip, side, N, C, k, s, nseg, False)
            bad_d.nseg = 0
            L.ym_conv2d_tile_counters(ctypes.byref(bad_d))
            sentinel = L.ym_last_error()
            n = L.ym_conv2d_tile_counters(ctypes.byref(d))
            err = L.ym_last_error()
            checked += 1
            if err != sentinel:
                bad += 1
                print(sig, src, row, err.decode())
            elif d.tail_tiles > 0 or d.ksplit > 1:
                assert n >= 0
    assert checked > 1000 and bad == 0, (checked, bad)


def test_user_cache_overlay_is_opt_in(tmp_path, monkeypatch):
    """YM_AUTOTUNE=1: rows measured on first use are kept in a per-user file and overlaid on the shipped table by later
    processes; without the switch nothing outside the package is read."""
    from yolact_minimal_amd import engine as E
    cache = tmp_path / 'sub' / 'rows.json'
    monkeypatch.setenv('YM_TUNED_CACHE', str(cache))
    monkeypatch.delenv('YM_NO_TUNED', raising=False)
    monkeypatch.setenv('YM_AUTOTUNE', '1')
    sig = 'M9_N256_C256_k3_s1_seg1_r0'
    assert sig not in TABLE
    E._store_user_rows({sig: [64, 64, 6, 0, 22, 0, 0]})
    E._store_user_rows({'M16_N256_C256_k3_s1_seg1_r0': [32, 32, 1, 4, 22, 0, 0]})        # merges, does not replace
    assert set(json.load(open(cache))) == {sig, 'M16_N256_C256_k3_s1_seg1_r0'}
    saved, E._tuned = E._tuned, None
    try:
        assert E.tuned_table()[sig] == [64, 64, 6, 0, 22, 0, 0]
        assert E._entry(sig, 9, 256, 72, 1, with_source=True) == ([64, 64, 6, 0, 22, 0, 0], 'table')
        monkeypatch.setenv('YM_AUTOTUNE', '0')
        E._tuned = None
        assert sig not in E.tuned_table()
        cache.write_text('{ torn')                     # a torn file is reported and ignored
        monkeypatch.setenv('YM_AUTOTUNE', '1')
        E._tuned = None
        assert sig not in E.tuned_table() and len(E.tuned_table()) == len(TABLE)
    finally:
        E._tuned = saved
