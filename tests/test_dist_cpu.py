"""world_size-2 gloo tests (CPU) of the multi-process host logic: batch sharding, max-over-ranks timing, the loss
all-reduce, and the DDP identity the trainer relies on (mean of per-shard gradients == single-process gradient of
the per-shard-normalised losses' mean) checked with the CPU oracle's autograd."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from yolact_minimal_amd import trainer
    r, w, lr = trainer.init_distributed(backend='gloo')
    assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
    # sharding: disjoint, complete
    mine = list(trainer.shard_batch(8, rank, world))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    assert sorted(sum(gathered, [])) == list(range(8))
    # timing: MAX over ranks
    assert trainer.reduce_max(1.0 + rank) == float(world)
    # loss logging all-reduce (SUM, like train.py:122)
    t = torch.tensor([1.0, 2.0, 3.0, 4.0]) * (rank + 1)
    dist.all_reduce(t)
    assert torch.equal(t, torch.tensor([1.0, 2.0, 3.0, 4.0]) * sum(range(1, world + 1)))
    # DDP identity on a tiny conv net: averaged per-shard grads == grad of mean of per-shard losses
    torch.manual_seed(0)
    w_ = torch.randn(4, 3, 3, 3, requires_grad=True)
    x = torch.randn(8, 3, 9, 9)
    y = torch.nn.functional.conv2d(x[mine], w_, padding=1).pow(2).mean()
    y.backward()
    g = w_.grad.clone()
    dist.all_reduce(g)
    g /= world
    w2 = w_.detach().clone().requires_grad_()
    full = sum(torch.nn.functional.conv2d(x[list(trainer.shard_batch(8, k, world))], w2, padding=1).pow(2).mean()
               for k in range(world)) / world
    full.backward()
    torch.testing.assert_close(g, w2.grad, rtol=1e-5, atol=1e-6)
    _check_flat_reducer(rank, world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, 'ok'))


def _check_flat_reducer(rank, world):
    """FlatGradReducer (the trainer's zero-copy bucketed all-reduce) on a toy net: averaged gradients in the flat buffer equal
    the mean of the per-rank gradients, with a parameter used twice, one never used, several buckets, two steps."""
    from yolact_minimal_amd.trainer import FlatSGD, FlatGradReducer, flatten_buffers
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.Conv2d(8, 4, 1))
    unused = torch.nn.Parameter(torch.randn(5))
    params = list(net.parameters()) + [unused]
    opt = FlatSGD(params, lr=0.1)
    red = FlatGradReducer(opt, world, bucket_bytes=150)
    assert len(red.buckets) >= 3 and sum(len(b[2]) for b in red.buckets) == len(opt.params)
    assert red.buckets[0][1] == opt.flat.numel() and red.buckets[-1][0] == 0          # reverse order, complete cover
    for step in range(2):
        g = torch.Generator().manual_seed(10 * step + rank)
        x = torch.randn(2, 3, 6, 6, generator=g)
        h = net[2](net[1](net[0](x)))
        y = net[4](net[3](net[3](h)))                                                  # net[3] is used twice
        opt.zero_grad()
        y.pow(2).mean().backward()
        local = torch.zeros_like(opt.grad)                     # same (64-byte aligned, zero padded) layout as the flat buffer
        for p, (a_, b_) in zip(opt.params, opt.offsets):
            if p.grad is not None:
                local[a_:b_] = p.grad.reshape(-1)
        assert all(p.grad is None or p.grad.data_ptr() == p._ym_grad_slot.data_ptr() for p in opt.params)
        red.finish()
        want = local.clone()
        dist.all_reduce(want)
        want /= world
        torch.testing.assert_close(opt.grad, want, rtol=1e-6, atol=1e-7)
        assert float(opt.grad[opt.offsets[-1][0]:].abs().sum()) == 0.0                 # the unused parameter reduces zeros
    assert red.launches == 2 * len(red.buckets)
    # BN running statistics: one flat broadcast from rank 0
    flat = flatten_buffers(net)
    net[1].running_mean.fill_(float(rank + 1))
    dist.broadcast(flat, 0)
    assert float(net[1].running_mean[0]) == 1.0 and flat.numel() == 16


def test_two_rank_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


def test_lr_schedule_matches_reference_formula():
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.trainer import lr_at
    cfg = build_cfg('res101_coco', 'train', 544, train_bs=16, bs_per_gpu=8)
    assert cfg.lr == pytest.approx(0.002) and cfg.lr_steps[1] == 140000
    assert lr_at(cfg, 0) == pytest.approx(cfg.lr)                   # 0 is in lr_steps: the reference trains step 0 at the full rate
    assert lr_at(cfg, 1) == pytest.approx((cfg.lr - cfg.warmup_init) / 500 + cfg.warmup_init)
    assert lr_at(cfg, 250) == pytest.approx((cfg.lr - cfg.warmup_init) * 0.5 + cfg.warmup_init)
    assert lr_at(cfg, 501) == pytest.approx(cfg.lr)
    assert lr_at(cfg, 140000) == pytest.approx(cfg.lr * 0.1)
    assert lr_at(cfg, 280001) == pytest.approx(cfg.lr * 0.01)

    # against a literal simulation of the reference loop's stateful updates (train.py:103-109), incl. an lr_step inside the warm-up
    class C:
        lr, warmup_init, warmup_until, lr_steps = 0.01, 0.001, 50, (0, 20, 80, 120)
    lr = C.lr
    for step in range(200):
        if C.warmup_until > 0 and step <= C.warmup_until:
            lr = (C.lr - C.warmup_init) * (step / C.warmup_until) + C.warmup_init
        if step in C.lr_steps:
            lr = C.lr * 0.1 ** C.lr_steps.index(step)
        assert lr_at(C, step) == pytest.approx(lr), step


def test_bench_spawns_one_rank_per_gpu_when_launched_bare(monkeypatch):
    """`python bench.py --gpus N` without a launcher must become N ranks (the driver's scaling run may use either form):
    the re-exec goes through torch.distributed.run on 127.0.0.1 with this command line's own arguments."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('MASTER_PORT', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # under a launcher (WORLD_SIZE set) a mismatching --gpus is an error, not a silent 1-rank run
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=2' in str(e.value.code)


def test_ddp_block_reports_what_the_scaling_record_needs():
    """`bench.py --gpus N` (N > 1) keeps `value` = inference replicas and carries the north star's DDP figures under `extra.ddp`:
    whole-job and per-GPU training img/s, the 1-GPU reference of the committed bench line, bucket layout, how many buckets were
    handed to the collective during backward, and the exposed all-reduce / buffer-broadcast times."""
    import types
    import bench
    red = types.SimpleNamespace(buckets=[(100, 200, [3, 2]), (0, 100, [1, 0])], last_launch_log=[(0, 2, False), (1, 4, True)])
    tr = types.SimpleNamespace(reducer=red, ddp=True)
    blk = bench.ddp_block(tr, dict(allreduce_exposed_ms=1.25, buffer_broadcast_ms=0.05, backward_ms=30.0), 1400.0, 8, 'res101_coco', 8)
    assert blk['train_img_s'] == 1400.0 and blk['per_gpu'] == 175.0 and blk['buckets'] == 2 and blk['buckets_launched_during_backward'] == 1
    assert blk['allreduce_exposed_ms'] == 1.25 and blk['buffer_broadcast_ms'] == 0.05 and blk['world_size_seen'] == 1
    ref = blk['vs_1gpu_reference_img_s']
    assert ref is None or (ref > 0 and abs(blk['scaling_vs_1gpu'] - round(1400.0 / ref, 3)) < 1e-9 and blk['reference_source'].startswith('profiles/'))


def _worker8(rank, world, port, q):
    """One of 8 gloo ranks: res101_coco's REAL parameter list (50 M floats -> the trainer's 8 gradient buckets), gradient hooks firing
    in a DIFFERENT order on every rank, rank 0 late for the second step (train.py:162-174: only the main rank runs `evaluate` at
    val_interval while the others walk into the next step's collectives)."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from yolact_minimal_amd import trainer
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    trainer.init_distributed(backend='gloo')
    assert dist.get_world_size() == 8
    torch.manual_seed(100 + rank)                                    # different initial weights per rank
    net = Yolact(build_cfg('res101_coco', 'train', 544, train_bs=64, bs_per_gpu=8))
    opt = trainer.FlatSGD(net.parameters(), lr=1e-3)
    dist.broadcast(opt.flat, 0)                                      # what Trainer.__init__ does
    flat_buffers = trainer.flatten_buffers(net)
    red = trainer.FlatGradReducer(opt, world)                        # default 25 MB buckets
    sizes = [round((e - a) * 4 / 2 ** 20, 1) for a, e, _ in red.buckets]
    assert len(red.buckets) == 8 and min(sizes[:-1]) >= 25 and sum(len(b[2]) for b in red.buckets) == len(opt.params), sizes      # (the tail: stem + layer1, 4 MB)
    n = len(opt.params)
    coef = lambda r, i, step: float(((r + 1) * 37 + i * 11 + step * 5) % 23) / 7.0 - 1.0      # noqa: E731  d(loss)/d(p_i) on rank r
    for step in range(2):
        if step == 1 and rank == 0:
            time.sleep(2.0)                                          # rank 0 is still in `evaluate`; the others are already here
        dist.broadcast(flat_buffers, 0)                              # Trainer.step: BN running statistics follow rank 0
        opt.zero_grad()
        # every rank builds its loss terms in its own order, so AccumulateGrad (and the reducer's hooks) fire in a different order on
        # every rank; the bucket collectives must still be issued in the same order everywhere
        order = torch.randperm(n, generator=torch.Generator().manual_seed(1000 * step + rank)).tolist()
        loss = sum(opt.params[i].sum() * coef(rank, i, step) for i in order)
        all_loss = torch.stack([loss.detach()] * 4)
        dist.all_reduce(all_loss)                                    # the 16-byte logging collective sits between forward and backward
        loss.backward()
        red.finish()
        log = red.last_launch_log
        assert [b for b, _, _ in log] == list(range(8)), log         # bucket order, whatever order the hooks came in
        for i in (0, 1, n // 3, n // 2, n - 2, n - 1):               # averaged gradient of a few parameters, exactly
            want = sum(coef(r, i, step) for r in range(world)) / world
            got = opt.params[i]._ym_grad_slot
            assert abs(float(got.reshape(-1)[0]) - want) < 1e-6 and abs(float(got.reshape(-1)[-1]) - want) < 1e-6, (i, want)
        with torch.no_grad():                                        # (the HIP optimizer launch has no CPU path: a plain SGD update)
            opt.flat.add_(opt.grad, alpha=-1e-3)
    digest = torch.stack([opt.flat.double().sum(), opt.flat.double().abs().sum(), flat_buffers.double().sum()])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert all(torch.equal(gathered[0], g) for g in gathered)        # replicas identical after two steps
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sizes))


def test_eight_rank_gloo_reducer_with_res101_bucket_layout():
    """The first 8-GPU run's host logic, without the hardware: 8 processes, res101's real 200 MB / 8-bucket gradient layout, uneven
    hook order per rank, a late rank 0 — no dead-lock, bucket order identical on every rank, exact averages, identical replicas."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = [q.get(timeout=5) for _ in range(8)]
    assert sorted(r for r, _ in got) == list(range(8)) and len({tuple(s) for _, s in got}) == 1


def test_module_train_state_surface():
    """Host logic of train_state.ModuleTrainState (what a Yolact brings along for the reference's own loop), on the CPU: slot layout,
    the one parameter left to torch's DDP, the ignore list DDP reads, release()."""
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    from yolact_minimal_amd.train_state import ModuleTrainState, DDP_KEEPS
    torch.manual_seed(0)
    net = Yolact(build_cfg('res50_coco', 'train', 64))
    names = dict(net.named_parameters())
    bufs = dict(net.named_buffers())
    assert not net._ddp_wrapped
    ignore = net._ddp_params_and_buffers_to_ignore                       # (what DDP's constructor does)
    assert net._ddp_wrapped and DDP_KEEPS in names and DDP_KEEPS not in ignore
    assert set(ignore) == (set(names) - {DDP_KEEPS}) | set(bufs)
    assert not hasattr(Yolact(build_cfg('res50_coco', 'val', 64)), '_ddp_params_and_buffers_to_ignore')
    before = {k: v.clone() for k, v in net.state_dict().items()}
    st = ModuleTrainState(net, torch.device('cpu'))
    assert len(st.params) == len(names) - 1 and all(p is not names[DDP_KEEPS] for p in st.params)
    assert all(a % 16 == 0 for a, _ in st.offsets) and st.offsets[-1][1] <= st.grad.numel()
    assert all(p._ym_grad_slot.data_ptr() == st.grad.data_ptr() + 4 * a and p._ym_grad_slot.shape == p.shape
               for p, (a, _) in zip(st.params, st.offsets))
    assert not hasattr(names[DDP_KEEPS], '_ym_grad_slot')
    # BatchNorm statistics / counters became views of flat tensors without changing a value or a state-dict key
    after = net.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    assert st.buffers_flat.numel() == sum(b.numel() for b in net.buffers() if b.is_floating_point())
    bn = net.backbone.bn1
    st.buffers_flat.fill_(3.0)
    assert float(bn.running_mean[0]) == 3.0 and bn._ym_nbt_flat
    st.after_forward()
    assert int(bn.num_batches_tracked) == 1
    assert not st.distributed(True)                                       # no process group: the module never touches one
    st.release()
    assert not bn._ym_nbt_flat and not any(hasattr(p, '_ym_grad_slot') or hasattr(p, '_ym_auto') for p in net.parameters())


def _worker_state(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from yolact_minimal_amd import trainer
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    from yolact_minimal_amd.train_state import ModuleTrainState
    trainer.init_distributed(backend='gloo')
    torch.manual_seed(100 + rank)                                         # different initial weights per rank
    net = Yolact(build_cfg('res50_coco', 'train', 64))
    net.backbone.bn1.running_mean.fill_(float(rank + 1))
    st = ModuleTrainState(net, torch.device('cpu'))
    st.sync_before_forward(True)                                          # first forward under DDP: parameters + buffers follow rank 0
    assert st.reducer is not None and st.synced_params and float(net.backbone.bn1.running_mean[0]) == 1.0
    n = len(st.params)
    coef = lambda r, i: float(((r + 1) * 13 + i * 7) % 11) - 5.0          # noqa: E731
    # gradient accumulation: a backward inside DDP.no_sync() stays local ...
    st.begin_forward()
    st.sync_before_forward(True, grad_sync=False)
    sum(st.params[i].sum() * coef(rank, i) for i in range(n)).backward()
    st.end_backward()
    assert st.reducer.launches == 0 and abs(float(st.params[3].grad.reshape(-1)[0]) - coef(rank, 3)) < 1e-6
    # ... the next one accumulates into it and the SUM is averaged over the ranks
    st.begin_forward()
    st.sync_before_forward(True)
    order = torch.randperm(n, generator=torch.Generator().manual_seed(rank)).tolist()
    sum(st.params[i].sum() * coef(rank, i) for i in order).backward()
    st.end_backward()
    assert st.reducer.launches == len(st.reducer.buckets)
    for i in (0, 3, n // 2, n - 1):
        want = sum(2 * coef(r, i) for r in range(world)) / world
        assert abs(float(st.params[i].grad.reshape(-1)[-1]) - want) < 1e-5, (i, want)
    digest = torch.stack([torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double().sum(), st.buffers_flat.double().sum()])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, 'ok'))


def test_module_train_state_two_rank_gloo():
    """The module's own DDP plumbing with two gloo ranks on the CPU: initial parameter broadcast, per-forward buffer broadcast, local
    accumulation under no_sync, bucketed averaging of the accumulated sum, identical replicas."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_state, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]
