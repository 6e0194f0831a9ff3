"""CPU: the tuned-table plumbing of the inference engine (no GPU, no kernel launch)."""
import json
import os

import pytest


def test_throughput_mode_reads_tp_entries_first(monkeypatch):
    """`InferEngine(mode='throughput')` (the slots of a RequestPipeline with several requests in flight) reads `<sig>_tp` before
    `<sig>`; the default mode never sees the `_tp` rows."""
    from yolact_minimal_amd import engine as E
    table = {'M1_N1_C32_k1_s1_seg1_r0': [64, 64, 3, 0, 2, 0, 0], 'M1_N1_C32_k1_s1_seg1_r0_tp': [32, 32, 1, 4, 22, 0, 0],
             'M2_N1_C32_k1_s1_seg1_r0': [128, 64, 1, 0, 22, 8, 3]}
    monkeypatch.setattr(E, '_tuned', table)
    monkeypatch.setattr(E, '_build_mode', ['latency'])
    assert E._entry('M1_N1_C32_k1_s1_seg1_r0') == [64, 64, 3, 0, 2, 0, 0]
    monkeypatch.setattr(E, '_build_mode', ['throughput'])
    assert E._entry('M1_N1_C32_k1_s1_seg1_r0') == [32, 32, 1, 4, 22, 0, 0]
    assert E._entry('M2_N1_C32_k1_s1_seg1_r0') == [128, 64, 1, 0, 22, 8, 3]          # no _tp row: the latency choice
    assert E._entry('unknown') is None


def test_grid_wgs_field_is_validated():
    from yolact_minimal_amd import engine as E
    assert E._grid_wgs([64, 64, 1, 0, 43, 0, 0, 768]) == 768 and E._grid_wgs([32, 32, 1, 1, 22, 0, 0, 1]) == 1
    assert E._grid_wgs([64, 64, 1, 0, 2, 0, 0]) == 0
    with pytest.raises(ValueError):
        E._grid_wgs([64, 64, 1, 0, 43, 0, 0, 12.5])           # an old autotune detail row: a timing where grid_wgs belongs
    with pytest.raises(ValueError):
        E._grid_wgs([32, 32, 1, 4, 22, 0, 0, 768])            # a persistent-kernel grid on a wave-DMA row (there: waves per workgroup)
    with pytest.raises(ValueError):
        E._grid_wgs([32, 32, 1, 4, 22, 40, 4, 2])             # the wave kernel's tail split needs four-wave workgroups
    assert E._grid_wgs([32, 32, 1, 4, 22, 40, 4, 4]) == 4 and E._grid_wgs([32, 32, 1, 2, 23, 0, 0, 2]) == 2


def test_committed_table_is_well_formed():
    """Every row of yolact_minimal_amd/tuned_gfx950.json: conv rows have 7 or 8 integer fields (tile, ksplit, kwaves, stages, tail,
    [grid_wgs / waves per workgroup]), wave-kernel rows with DMA rings name a tile the kernel has, `_tp` rows shadow an existing
    shape, weight-gradient rows have two fields."""
    from yolact_minimal_amd import engine as E
    table = json.load(open(E.TUNED_PATH))
    assert len(table) > 400
    for key, row in table.items():
        assert all(isinstance(v, int) and not isinstance(v, bool) for v in row), (key, row)
        if key.startswith('W_'):
            assert len(row) == 2, (key, row)
            continue
        assert len(row) in (5, 7, 8), (key, row)
        if len(row) >= 7 and row[3] > 0 and 22 <= row[4] <= 24:                      # conv_wdma_f32
            assert (row[0], row[1]) in ((32, 32), (64, 32), (32, 64)) and row[3] in (1, 2, 4), (key, row)
            wpb = row[7] if len(row) > 7 else 0
            assert wpb in (0, 1, 2, 4) and (wpb == 0 or wpb >= row[3]), (key, row)
            # (K waves, ring depth, waves per workgroup) must be an instantiation csrc/conv_wave.hip builds (dispatch_dma)
            built = {(kw, ns, 4) for kw in (1, 2, 4) for ns in (2, 3)} | {(1, 2, 1), (1, 3, 1), (1, 2, 2), (1, 3, 2), (2, 2, 2), (2, 3, 2)}
            if (row[0], row[1]) == (32, 32):
                built |= {(1, 4, 4), (2, 4, 4), (4, 4, 4), (1, 4, 1), (1, 4, 2)}
            assert (row[3], row[4] - 20, wpb or 4) in built, (key, row)
            if row[5] or row[6]:                                                     # tail split: 32x32 tile, four K waves, <= 8 slices
                assert (row[0], row[1], row[3]) == (32, 32, 4) and 2 <= row[6] <= 8 and wpb in (0, 4), (key, row)
        if key.endswith('_tp'):
            assert key[:-3] in table or key[:-3].startswith('M'), key


def test_pipeline_checks_the_hardware_queue_count(monkeypatch):
    from yolact_minimal_amd.pipeline import hw_queues_ok
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')
    assert hw_queues_ok(4) and hw_queues_ok(7) and not hw_queues_ok(8)
    monkeypatch.delenv('GPU_MAX_HW_QUEUES')
    assert hw_queues_ok(3) and not hw_queues_ok(4)              # ROCm's default: 4 hardware queues
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', 'x')
    assert not hw_queues_ok(4)


def _desc_from_key(key, row):
    """A descriptor with the GEMM shape a forward-conv key names (M = B Ho Wo: any factorisation gives the same plan) and the
    row's tiling; pointers are placeholders (the planner never dereferences them)."""
    import re
    from yolact_minimal_amd.hip import ConvDesc
    m = re.match(r'M(\d+)_N(\d+)_C(\d+)_k(\d+)_s(\d+)_seg(\d+)_r(\d+)', key)
    M, N, C, k, s, nseg, res = map(int, m.groups())
    ho = max(h for h in range(1, 1200) if M % (h * h) == 0)
    d = ConvDesc()
    d.inp = d.weight = 0x1000
    d.residual = 0x1000 if res else None
    d.B, d.H, d.W, d.Cin, d.Cout = M // (ho * ho), ho * s, ho * s, C, N
    d.KH = d.KW = k
    d.stride, d.pad, d.Ho, d.Wo = s, k // 2, ho, ho
    d.k_pad = -(-(k * k * C) // 32) * 32
    cuts = [0, N] if nseg == 1 else [0, N - N // 3 - 12, N - N // 3, N][:nseg + 1]
    d.nseg = nseg
    for i in range(nseg):
        d.seg[i].n_begin, d.seg[i].n_end = cuts[i], cuts[i + 1]
        d.seg[i].out, d.seg[i].pitch, d.seg[i].batch_stride = 0x2000, cuts[i + 1] - cuts[i], ho * ho * (cuts[i + 1] - cuts[i])
    d.tile_m, d.tile_n, d.ksplit = row[0], row[1], row[2]
    d.kwaves = row[3] if len(row) > 3 else 0
    d.stages = row[4] if len(row) > 4 else 0
    d.tail_tiles, d.tail_ksplit = (row[5], row[6]) if len(row) > 6 else (0, 0)
    d.grid_wgs = row[7] if len(row) > 7 else 0
    d.tile_counters = 0x3000
    return d


def test_planner_accepts_every_forward_row_of_the_committed_table():
    """`ym_conv2d_workspace_bytes` / `ym_conv2d_tile_counters` run on the host: every inference row of the table (latency and
    throughput choices) must be a plan the C-ABI accepts for its shape -- a stale or mistyped row fails here, not on the GPU box.
    Rows for the segmented head (3 outputs) and the pyramid launch (`_L<n>`) are planned by the engine with their real segment
    tables and are covered by the GPU forward tests."""
    import ctypes
    import re
    from yolact_minimal_amd import engine as E, hip
    lib = hip.lib()
    table = json.load(open(E.TUNED_PATH))
    pat = re.compile(r'^M\d+_N\d+_C\d+_k\d+_s\d+_seg1_r[01](_tp)?$')
    checked = wave = split = 0
    for key, row in table.items():
        if not pat.match(key) or '_C4_' in key:                                   # (the 7x7 stem on the 4-channel image: its own mode)
            continue
        d = _desc_from_key(key, row)
        lib.ym_conv2d_workspace_bytes(None)                                        # plants a known message in ym_last_error()
        sentinel = lib.ym_last_error()
        nb = lib.ym_conv2d_workspace_bytes(ctypes.byref(d))
        assert lib.ym_last_error() == sentinel, (key, row, lib.ym_last_error())
        tiles = lib.ym_conv2d_tile_counters(ctypes.byref(d))
        assert 0 <= tiles <= hip.TILE_COUNTERS, (key, row, tiles)
        if row[0] > 0 and row[2] > 0:                  # (0 = the planner's own choice) scratch exactly when K slices meet in memory
            tail = len(row) > 6 and row[5] > 0
            in_workgroup = len(row) > 3 and row[3] > 0    # kwaves: the K split stays inside the workgroup
            assert (nb > 0) == (tail or (row[2] > 1 and not in_workgroup)), (key, row, nb)
        if len(row) > 4 and row[3] > 0 and 22 <= row[4] <= 24:
            assert d.Cin % 32 == 0, (key, row)
            wave += 1
        split += int(nb > 0)
        checked += 1
    assert checked > 150 and wave >= 15 and split > 50, (checked, wave, split)


def test_tuner_rows_reach_the_table_only_through_the_reference_digest_gate(tmp_path):
    """tools/table_gate.py: rows a tuner proposes (`tune_forward.py --inflight N --write` writes `<sig>_tp` rows) are merged into the
    table only after the 544 px reference-digest tests have run green against the CANDIDATE table; a red or missing run leaves
    the table as it was."""
    from tools import table_gate as G
    path = tmp_path / 'table.json'
    base = {'M1156_N256_C1024_k1_s1_seg1_r0': [32, 32, 1, 4, 22, 40, 4]}
    path.write_text(json.dumps(base))
    rows = {'M1156_N256_C1024_k1_s1_seg1_r0_tp': [32, 32, 1, 4, 22, 0, 0]}
    seen = []

    def red(candidate):
        seen.append(json.load(open(candidate)))
        return 1

    with pytest.raises(G.GateRefused):
        G.merge_rows(rows, str(path), runner=red)
    assert json.loads(path.read_text()) == base                         # untouched
    assert seen[0] == {**base, **rows}                                  # the tests saw the candidate, not the committed table
    assert not list(tmp_path.parent.glob('tuned_candidate_*'))
    merged = G.merge_rows(rows, str(path), runner=lambda candidate: 0)
    assert merged == {**base, **rows} and json.loads(path.read_text()) == merged
    # the gate's test list names tests that exist, parametrised over both plan modes
    import ast
    src = open(os.path.join(G.REPO, 'tests', 'test_gpu_forward.py')).read()
    names = {n.name for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef)}
    for t in G.GATE_TESTS:
        assert t.split('::')[1] in names, t
    assert "'throughput'" in src
    # and the tuner goes through it
    tool = open(os.path.join(G.REPO, 'tools', 'tune_forward.py')).read()
    assert 'merge_rows' in tool and 'json.dump(table, open(E.TUNED_PATH' not in tool
