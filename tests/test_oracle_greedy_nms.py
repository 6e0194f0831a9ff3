"""Hand-derived known-answer cases for the C restatement of cython_nms.pyx:24-74 (the reference has none)."""
import numpy as np

from oracle import yolact_ref as R


def test_two_overlapping_one_far():
    # boxes 0 and 1 overlap heavily (IoU with +1 convention = 100*100.. see below), 2 is far away
    dets = np.array([[0, 0, 9, 9, 0.9],      # area 100
                     [0, 0, 9, 4, 0.8],      # area 50, inter 50 -> ovr = 50/(100+50-50) = 0.5 -> suppressed (>=)
                     [20, 20, 29, 29, 0.7]], np.float32)
    assert R.greedy_nms(dets, 0.5).tolist() == [0, 2]
    # just above the overlap -> kept
    assert R.greedy_nms(dets, 0.51).tolist() == [0, 1, 2]


def test_order_is_by_score_result_is_ascending_index():
    dets = np.array([[0, 0, 9, 4, 0.1],
                     [0, 0, 9, 9, 0.9],
                     [50, 50, 59, 59, 0.5]], np.float32)
    assert R.greedy_nms(dets, 0.5).tolist() == [1, 2]


def test_chain_suppression_is_greedy_not_transitive():
    # A suppresses B; B would suppress C but B is dead, so C survives
    a = [0, 0, 9, 9, 0.9]
    b = [4, 0, 13, 9, 0.8]     # inter with a: 6*10=60 -> 60/(100+100-60)=0.428
    c = [8, 0, 17, 9, 0.7]     # inter with b: 60 -> 0.428 ; with a: 2*10=20 -> 20/180 = 0.11
    dets = np.array([a, b, c], np.float32)
    assert R.greedy_nms(dets, 0.4).tolist() == [0, 2]


def test_touching_boxes_overlap_by_plus_one_convention():
    dets = np.array([[0, 0, 9, 9, 0.9], [10, 0, 19, 9, 0.8]], np.float32)   # w = 9-10+1 = 0 -> inter 0
    assert R.greedy_nms(dets, 0.01).tolist() == [0, 1]
    dets[1, 0] = 9                                                           # w = 1 -> inter 10, ovr = 10/(100+110-10)=0.05
    assert R.greedy_nms(dets, 0.05).tolist() == [0]


def test_empty():
    assert R.greedy_nms(np.zeros((0, 5), np.float32), 0.5).tolist() == []
