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
