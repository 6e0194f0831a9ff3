"""Plans for layer shapes the tuned table has no row for.

`tuned_gfx950.json` is keyed on exact GEMM shapes (M = B * Ho * Wo output pixels, N = Cout, C = Cin, filter, stride, output
segments, residual).  The shipped rows cover the layer shapes of the configurations the table was measured on; any other
`--img_size` (the reference accepts every multiple of 32: config.py:75, detect.py:24, eval.py:18, train.py:25) or batch size has
the same layers with another M.  Instead of dropping such a launch to the planner's one-size-fits-all heuristic (64x64 register
ring, K split by workgroup count: 3.28 vs 2.31 ms per res101 544 px forward, bench `extra.other_sizes`), the row of the NEAREST
tuned shape of the same family is transferred:

* family = same table prefix (forward / `T_` data gradient / `W_` weight gradient), filter size, stride, segment count, residual
  flag, pyramid level count and suffix (`_st`, `_tp`, `_mma3`); stems (Cin = 4) only match stems;
* nearest = smallest |log2 M/M'| (x1.5 for a smaller donor) + 8 |log2 N/N'| + 8 |log2 C/C'| with M within 2.46x and N, C within 2x —
  in practice the SAME layer at the nearest measured resolution;
* what is kept: the kernel family (tile, wave / workgroup kernel, staging ring, persistent walker); what is re-derived for the new
  M: the K split (same number of workgroups in flight as the donor launch had), the tail split (the donor's rule — the tiles of
  the last partial round over 256 or 512 slots — applied to the new tile count, under the tuner's own validity limits), and
  nothing else.

Pure host logic (no GPU, no library): tests/test_plan_transfer.py.  `YM_TUNED_NEAREST=0` switches the transfer off (then a missing
row means the planner heuristic, as before round 6); `YM_TUNED_NEAREST=only` ignores exact rows (leave-one-out measurement of the
transfer itself: tools/size_bench.py)."""
import math
import os
import re

_SIG = re.compile(r'^(T_|W_)?M(\d+)_N(\d+)_C(\d+)_k(\d+)_s(\d+)(?:_seg(\d+)_r(\d))?(_L\d+)?(_st|_tp|_mma\d)?$')
_KS_ALLOWED = (1, 2, 3, 4, 6, 8, 12, 16, 24)
TILE_COUNTERS = 16384        # = hip.TILE_COUNTERS (int32 arrival counters the engines allocate)

M_REACH = 1.3               # |log2 M/M'| a donor may be away: 2.46x (res101 bs=8 training at 320 px, donors 2.9x away: 23.4 ms per step
                            # against 20.9 on the planner heuristic; at 384 px, 2.0x away: 24.9 against 29.0)

_index_cache = {}


def mode():
    v = os.environ.get('YM_TUNED_NEAREST', '1')
    return 'off' if v in ('0', 'off') else ('only' if v == 'only' else 'on')


def parse(sig):
    """(family, M, N, C) of a table key, or None."""
    m = _SIG.match(sig)
    if not m:
        return None
    pre, M, N, C, k, s, seg, r, lev, suf = m.groups()
    fam = (pre or '', int(k), int(s), seg, r, lev or '', suf or '', int(C) == 4)
    return fam, int(M), int(N), int(C)


def _index(table):
    key = (id(table), len(table))
    hit = _index_cache.get(key)
    if hit is None:
        hit = {}
        for sig in table:
            p = parse(sig)
            if p:
                hit.setdefault(p[0], []).append((p[1], p[2], p[3], sig))
        _index_cache.clear()
        _index_cache[key] = hit
    return hit


def nearest(table, sig, exclude_exact=False):
    """(donor key, donor M) of the nearest tuned shape of `sig`'s family, or None."""
    p = parse(sig)
    if not p:
        return None
    fam, M, N, C = p
    best = None
    for m2, n2, c2, key in _index(table).get(fam, ()):
        if exclude_exact and key == sig:
            continue
        dm, dn, dc = abs(math.log2(M / m2)), abs(math.log2(N / n2)), abs(math.log2(C / c2))
        if dm > M_REACH or dn > 1.0 or dc > 1.0:
            continue
        if m2 < M:
            dm *= 1.5        # a row measured on a LARGER launch scales down (K split re-derived) better than a small launch's choice
                             # (one-wave tiles, tail splits) scales up: res101 bs=8 at 320 px, 21.7 vs 20.9 ms per step without this
        dist = dm + 8.0 * dn + 8.0 * dc
        if best is None or dist < best[0]:
            best = (dist, key, m2)
    return (best[1], best[2]) if best else None


def _cdiv(a, b):
    return -(-a // b)


def transfer_conv(row, M_donor, M, N, nkt, nseg=1, counters=True):
    """Row of a forward / data-gradient launch ([tile_m, tile_n, ksplit, kwaves, stages, tail_tiles, tail_ksplit(, grid_wgs)]) of a
    donor with M_donor rows, re-derived for M rows (N output channels, nkt K tiles of 32, nseg output segments; `counters`: the
    caller provides arrival counters, without which a tail split is not possible)."""
    row = list(row) + [0] * (8 - len(row))
    tm, tn, ks, kw, st, tail_t, tail_k, g = row[:8]
    if tm == 0 or tn == 0:
        return [0, 0, 0, 0, 0, 0, 0]
    w_d, w = _cdiv(M_donor, tm) * _cdiv(N, tn), _cdiv(M, tm) * _cdiv(N, tn)
    out_ks, out_tail = ks, (0, 0)
    if kw == 0:
        if ks > 1:
            # the donor launch had w_d * ks workgroups: the same number for the new tile count, from the tuner's candidate list
            want = w_d * ks / w
            lim = max(1, nkt // 2)
            out_ks = min((k for k in _KS_ALLOWED if k <= lim), key=lambda k: abs(math.log2(k / want)) if want > 0 else k)
            if w >= 1024:
                out_ks = 1
        elif w < 192 and w_d >= 256:
            out_ks = 0                           # far fewer tiles than the donor had: the planner's own K split by workgroup count
        if tail_t > 0 and tail_k > 1 and out_ks == 1 and counters and nseg == 1 and 256 < w <= TILE_COUNTERS:
            mod = 512 if (w_d % 512 == tail_t and w_d % 256 != tail_t) else 256
            r, ts = w % mod, tail_k
            while ts > 1 and ts * 2 > nkt:
                ts -= 1
            if r > 0 and ts > 1 and r * ts <= 2048:
                out_tail = (r, ts)
    else:
        out_ks = 1
        if tail_t > 0 and tail_k > 1 and counters and nseg == 1 and (tm, tn) == (32, 32) and kw == 4 and 22 <= st <= 24 and \
                256 < w <= TILE_COUNTERS and g in (0, 4):
            ts = tail_k
            while ts > 1 and ts * 2 > nkt:
                ts -= 1
            if ts > 1:
                out_tail = (w % 256 or 256, ts)
        if kw > nkt:                                 # K waves of the wave kernel: 1 / 2 / 4 / 8, at most one per K tile
            kw = max(k for k in (1, 2, 4, 8) if k <= max(1, nkt))
    out = [tm, tn, out_ks, kw, st, out_tail[0], out_tail[1]]
    if g:
        out.append(g)
    return out


def transfer_wgrad(row, M_donor, M):
    """Row of a weight-gradient launch ([msplit, ring]): the pixel split follows the tile count of (Cout, K), not M; it is only
    capped so that a slice keeps at least 128 pixels."""
    ms = row[0]
    if ms > 1 and M // ms < 128:
        ms = max(1, M // 128)
    return [ms] + list(row[1:])


def lookup(table, sig, M, N, nkt, nseg=1, counters=True):
    """(row, source) for `sig`: the exact row ('table'), a transferred one ('nearest:<donor>') or (None, 'heuristic')."""
    md = mode()
    if md != 'only':
        hit = table.get(sig)
        if hit is not None:
            return hit, 'table'
    if md == 'off' or not table:
        return None, 'heuristic'
    nb = nearest(table, sig, exclude_exact=(md == 'only'))
    if nb is None:
        return None, 'heuristic'
    key, m_d = nb
    if key.startswith('W_'):
        return transfer_wgrad(table[key], m_d, M), 'nearest:' + key
    return transfer_conv(table[key], m_d, M, N, nkt, nseg, counters), 'nearest:' + key
