"""CPU: the property of torch's autograd engine that `train_engine.ResGradLink` / `GradJoin` rely on -- the nodes of one device run
in strictly DECREASING creation order (the ready queue is a priority queue on the node's sequence number) -- and the host-side
guard that turns a violation into an error instead of a silently missing gradient."""
import pytest
import torch


def test_autograd_runs_nodes_in_decreasing_creation_order():
    order, serial = [], [0]

    class Rec(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            serial[0] += 1
            ctx.tag = serial[0]
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            order.append(ctx.tag)
            return g

    # the shape of the backbone + FPN graph: a shared tensor h with a first-created consumer that heads a LONG chain (the next
    # stage), and later-created consumers (downsample conv, lateral conv) whose own consumers are created later still
    x = torch.ones(4, requires_grad=True)
    h = Rec.apply(x)                                  # 1
    first = Rec.apply(h)                              # 2  (conv1: the taker)
    chain = first
    for _ in range(5):
        chain = Rec.apply(chain)                      # 3..7
    second = Rec.apply(h)                             # 8  (downsample conv: passes)
    merged = Rec.apply(chain) + Rec.apply(second)     # 9, 10
    third = Rec.apply(h)                              # 11 (FPN lateral conv: passes first)
    tail = Rec.apply(Rec.apply(third))                # 12, 13
    (merged.sum() + tail.sum()).backward()
    assert order == sorted(order, reverse=True), order
    assert order.index(11) < order.index(8) < order.index(2)          # the passers of h run before its taker


def test_a_parked_gradient_that_nobody_takes_is_an_error():
    from yolact_minimal_amd import train_engine as T
    T._live_links.clear()
    j, r = T.GradJoin(), T.ResGradLink()
    assert T._live_links == [j, r]
    T.check_links_drained()                           # nothing parked: fine, and the list is reset
    assert T._live_links == []
    j = T.GradJoin()
    assert T._join_result(j, 'pass', torch.ones(2)) is None and j.grad is not None
    with pytest.raises(RuntimeError, match='never consumed'):
        T.check_links_drained()
    assert j.grad is None and T._live_links == []
    j = T.GradJoin()
    T._join_result(j, 'pass', torch.ones(2))
    assert torch.equal(T._join_add(j), torch.ones(2)) and j.grad is None      # a taker consumed it
    T.check_links_drained()
