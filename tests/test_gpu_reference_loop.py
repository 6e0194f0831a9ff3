"""SURVEY §8b, training leg of the drop-in boundary: the reference's own loop — `optim.SGD(net.parameters(), ...)` (AdamW for
swin_tiny_coco), `DDP(net.cuda(), [local_rank], output_device=local_rank, broadcast_buffers=True)` on a one-rank RCCL group,
`net(images, targets, masks)`, `dist.all_reduce(all_loss)`, `optimizer.zero_grad()`, `loss_total.backward()`, `optimizer.step()`,
then the `evaluate`-style eval forward on `net.module` (/root/reference/train.py:44-48,60-63,76,102-130,165-166; restated in
dropin/reference_loops.py, bound through `dropin/` exactly as `dropin/run.py train.py` binds it) — on the HIP path, against

  (i)  the build's own `Trainer.step` on the same seed and inputs (same kernels, other plumbing: flat gradient buffer, side
       stream, deferred slab reductions, FlatGradReducer, one-launch optimizer), and
  (ii) the REAL reference running the same statements on the CPU (tests/golden/loop_*.npz, oracle/make_golden_loop.py).

Each case runs in a process of its own (tests/run_reference_loop.py): train.py's `get_config` joins a process group."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, cfg, size, bs, seed, steps=3, extra=()):
    out = str(tmp_path / f'loop_{cfg}_{size}.npz')
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'YM_FORCE_DIST'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tests', 'run_reference_loop.py'), '--cfg', cfg, '--img_size', str(size),
                        '--train_bs', str(bs), '--steps', str(steps), '--seed', str(seed), '--out', out, *extra],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and 'REFERENCE_LOOP_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    return np.load(out)


def _check_against_trainer(d, exact=True):
    """(i) the reference loop and Trainer.step agree.  SGD: BIT FOR BIT at every step — losses, every parameter, the BatchNorm
    buffers, the eval forward on the trained weights: both paths run the same kernels on the same numbers (gradients land in fresh
    tensors + DDP bucket copies + torch.optim.SGD on one side, in the flat buffer + FlatGradReducer + `ym_sgd_step`, which repeats
    torch.optim.SGD's roundings, on the other).  AdamW (Swin-T): the parameters to 2e-5 of max|p| at every step and the losses to
    1e-5; the UPDATES are not comparable element by element — the key bias of every attention block has a mathematically zero
    gradient (softmax is invariant to it), i.e. pure rounding noise, which Adam's g / (|g| + eps) turns into +-lr steps, and the
    relative-position-bias gradient is summed with float atomics (run-to-run rounding differences feed that noise)."""
    assert str(d['wrapper']) == 'DistributedDataParallel' and int(d['grads_none']) == 0
    q = np.quantile(d['update_rel_diff'], [0.5, 0.9, 1.0])
    print(f'reference loop vs Trainer: losses max rel {np.abs(d["losses_loop"] / d["losses_trainer"] - 1).max():.2e}; per step: update diff / '
          f'max|update| {d["step_update_diff"]}, parameter diff / max|p| {d["step_param_diff"]}; per tensor after the last step: update diff '
          f'median {q[0]:.2e} p90 {q[1]:.2e} max {q[2]:.2e} ({d["keys"][int(d["update_rel_diff"].argmax())]}); buffers '
          f'{float(d["buffer_rel_diff"]):.2e}; eval outputs {d["eval_loop_vs_trainer"]}')
    assert int(d['num_batches_tracked']) == int(d['num_batches_tracked_trainer']) == int(d['end_step']) or int(d['num_batches_tracked']) == -1
    if exact:
        assert np.array_equal(d['losses_loop'], d['losses_trainer'])
        assert float(d['step_update_diff'].max()) == 0.0 and float(d['step_param_diff'].max()) == 0.0
        assert float(d['buffer_rel_diff']) == 0.0 and float(d['eval_loop_vs_trainer'].max()) == 0.0
    else:
        np.testing.assert_allclose(d['losses_loop'], d['losses_trainer'], rtol=1e-5)
        assert float(d['step_param_diff'].max()) <= 2e-5
        assert q[0] <= 1e-3, q
        assert float(d['eval_loop_vs_trainer'].max()) <= 1e-4


def _check_against_reference(d, g, loss_rtol, digest_tol, sample_tol):
    """(ii) the REAL reference's loop on the CPU: learning rates exactly, losses step by step, every parameter's UPDATE (what the
    three optimizer steps did to it) by robust norms and strided samples, the stem's running statistics, the eval forward."""
    assert [str(k) for k in g['keys']] == [str(k) for k in d['keys']]
    np.testing.assert_allclose(d['lrs'], g['lrs'], rtol=1e-12)
    for s, tol in enumerate(loss_rtol):
        np.testing.assert_allclose(d['losses_loop'][s], g['losses'][s], rtol=tol, err_msg=f'step {s}')
    bad, errs = [], []
    for i, k in enumerate(d['keys']):
        dg, rg = d['update_digest'][i], g['update_digest'][i]
        e = np.abs(d['update_sample'][i] - g['update_sample'][i]).max() / (float(g['update_absmax'][i]) + 1e-30)
        errs.append(e)
        if abs(dg[1] - rg[1]) > digest_tol * rg[1] + 1e-12 or abs(dg[2] - rg[2]) > 2 * digest_tol * rg[2] + 1e-20 or e > sample_tol:
            bad.append((str(k), float(e), dg.tolist(), rg.tolist()))
    print(f'reference loop vs the REAL reference (CPU): losses rel {np.abs(d["losses_loop"] / g["losses"] - 1).max(axis=1)}, parameter-update '
          f'samples / max|update|: median {np.median(errs):.2e} max {np.max(errs):.2e}; {len(bad)} of {len(errs)} tensors outside the bar')
    assert not bad, (len(bad), [(b[0], round(b[1], 4), [round(x / y - 1, 4) for x, y in zip(b[2][1:], b[3][1:])]) for b in bad[:8]])
    if int(g['num_batches_tracked']) >= 0:
        assert int(d['num_batches_tracked']) == int(g['num_batches_tracked'])
        np.testing.assert_allclose(d['run_mean_stem'], g['run_mean_stem'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(d['run_var_stem'], g['run_var_stem'], rtol=1e-4, atol=1e-6)


def test_reference_train_loop_res50_256(tmp_path, golden_dir):
    d = _run(tmp_path, 'res50_coco', 256, 4, 71, extra=('--wellcond',))
    assert str(d['optimizer']) == 'SGD'
    _check_against_trainer(d)
    g = np.load(os.path.join(golden_dir, 'loop_res50_coco_256_b4.npz'))
    _check_against_reference(d, g, (1e-5, 1e-4, 5e-4), 0.05, 0.10)


def test_reference_train_loop_res101_544_bs8(tmp_path, golden_dir):
    """BASELINE config 3's per-GPU shape, the step bench.py times as `extra.train_reference_loop`."""
    d = _run(tmp_path, 'res101_coco', 544, 8, 72, extra=('--wellcond',))
    assert str(d['optimizer']) == 'SGD'
    _check_against_trainer(d)
    g = np.load(os.path.join(golden_dir, 'loop_res101_coco_544_b8.npz'))
    _check_against_reference(d, g, (1e-5, 5e-4, 1e-2), 0.05, 0.15)


def test_reference_train_loop_swin_tiny_adamw(tmp_path, golden_dir):
    """train.py:62-63: AdamW(weight_decay=0.05) for swin_tiny_coco (stochastic depth off: the reference draws DropPath masks from
    the global generator of its own device, which no other device reproduces)."""
    d = _run(tmp_path, 'swin_tiny_coco', 128, 2, 73, extra=('--no_drop_path',))
    assert str(d['optimizer']) == 'AdamW'
    _check_against_trainer(d, exact=False)
    g = np.load(os.path.join(golden_dir, 'loop_swin_tiny_coco_128_b2.npz'))
    # AdamW's first steps move every weight by ~lr * sign(g): the update of an element whose gradient is inside the rounding noise
    # flips sign, so samples are held loosely and the robust norms carry the comparison
    _check_against_reference(d, g, (3e-4, 1e-2, 2e-2), 0.05, 2.5)


def test_partial_backward_fails_loudly_instead_of_dropping_gradients():
    """GradJoin 'pass' consumers park their gradient for a 'take' consumer; a backward over part of the graph
    (`loss_s.backward()`) never reaches the taker.  Outside Trainer.step that used to drop the parked gradient in silence: now every
    backward that parks a gradient ends with the drain check (a callback on the autograd engine)."""
    import torch
    from oracle import yolact_ref as R
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    cfg = build_cfg('res50_coco', 'train', 64)
    torch.manual_seed(0)
    net = Yolact(cfg).train().cuda()
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    boxes, masks = R.synth_targets(2, 64, seed=3)
    losses = net(img, [b.cuda() for b in boxes], [m.cuda() for m in masks])
    with pytest.raises(RuntimeError, match='never consumed'):
        losses[3].backward()                                 # semantic loss alone: P3's taker (protonet) is not in this graph
    # the whole graph still works, and twice in a row (the guard re-arms per backward)
    for _ in range(2):
        net.zero_grad()
        losses = net(img, [b.cuda() for b in boxes], [m.cuda() for m in masks])
        sum(losses).backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize('cfg_name', ['res50_coco', 'swin_tiny_coco'])
def test_reference_loop_two_ranks_keep_replicas_identical(cfg_name):
    """The reference's loop with TWO real ranks (torch.distributed.run, gloo so that both may share this box's GPU): torch's DDP
    wraps the module, which reduces its own gradients and broadcasts its own buffers (train_state.py; DDP is told to ignore them).
    Different shards and different initial seeds per rank -> after 3 steps both ranks hold bit-identical parameters, optimizer state
    and BatchNorm statistics (weights broadcast from rank 0 at the first forward, gradients averaged bucket by bucket, running
    statistics following rank 0)."""
    env = dict(os.environ, YM_DIST_BACKEND='gloo', YM_CHECK_CFG=cfg_name, YM_CHECK_LOOP='reference', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29553', os.path.join(REPO, 'tools', 'ddp_check.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith('DDP_CHECK')][-1]
    assert ' OK ' in line and 'world 2' in line and 'reference loop' in line, line


def test_launcher_runs_a_training_script_end_to_end(tmp_path):
    """`python dropin/run.py <script>`: a script that sits in a checkout with its OWN `modules/`, `utils/`, `config.py` (which Python
    would import first), written with train.py's import lines and statements, trains two steps on the GPU through the launcher —
    the hot-path imports resolve to yolact_minimal_amd, `GPU_MAX_HW_QUEUES` stays at its default for a training script (8 is for the
    serving scripts: `dropin/run.py::hw_queues_for`), the module's own training state is in place under torch's DDP."""
    for pkg in ('utils', 'modules'):
        (tmp_path / pkg).mkdir()
        (tmp_path / pkg / '__init__.py').write_text('')
    (tmp_path / 'modules' / 'yolact.py').write_text('raise RuntimeError("the checkout\'s own yolact was imported")\n')
    (tmp_path / 'config.py').write_text('raise RuntimeError("the checkout\'s own config was imported")\n')
    (tmp_path / 'train_like.py').write_text('''
import argparse, os
import torch
import torch.optim as optim
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from utils import timer
from modules.yolact import Yolact
from config import get_config
assert os.environ.get('GPU_MAX_HW_QUEUES') is None          # training keeps the default 4 hardware queues (dropin/run.py)
parser = argparse.ArgumentParser()
parser.add_argument('--local_rank', type=int, default=None)
parser.add_argument('--cfg', default='res101_coco')
parser.add_argument('--train_bs', type=int, default=8)
parser.add_argument('--img_size', default=544, type=int)
parser.add_argument('--resume', default=None, type=str)
parser.add_argument('--val_interval', default=4000, type=int)
parser.add_argument('--val_num', default=-1, type=int)
parser.add_argument('--traditional_nms', default=False, action='store_true')
parser.add_argument('--coco_api', action='store_true')
args = parser.parse_args()
cfg = get_config(args, mode='train')
net = Yolact(cfg)
net.train()
optimizer = optim.SGD(net.parameters(), lr=cfg.lr, momentum=0.9, weight_decay=5e-4)
net = DDP(net.cuda(), [args.local_rank or 0], output_device=args.local_rank or 0, broadcast_buffers=True)
from yolact_minimal_amd.utils.synthetic import synth_targets
images = torch.randn(2, 3, 64, 64)
targets, masks = synth_targets(2, 64, seed=1)
timer.reset()
hist = []
for step in range(2):
    im = images.cuda().detach()
    tg = [ann.cuda().detach() for ann in targets]
    mk = [mask.cuda().detach() for mask in masks]
    with timer.counter('for+loss'):
        loss_c, loss_b, loss_m, loss_s = net(im, tg, mk)
        all_loss = torch.stack([loss_c, loss_b, loss_m, loss_s], dim=0)
        dist.all_reduce(all_loss)
    with timer.counter('backward'):
        loss_total = loss_c + loss_b + loss_m + loss_s
        optimizer.zero_grad()
        loss_total.backward()
    with timer.counter('update'):
        optimizer.step()
    hist.append(float(loss_total))
    if step == 0:
        timer.start()
st = net.module._train_state
assert st is not None and net.module._ddp_wrapped
adopted = sum(1 for p in st.params if p.grad is not None and p.grad.data_ptr() == p._ym_grad_slot.data_ptr())
assert adopted >= 0.9 * len(st.params), (adopted, len(st.params))
assert all(h == h for h in hist) and hist[1] != hist[0]
net.eval()
with torch.no_grad():
    out = net.module(images[:1].cuda())
assert all(bool(torch.isfinite(o).all()) for o in out)
print('LAUNCHER_TRAIN_OK', hist)
''')
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29557')
    env.pop('GPU_MAX_HW_QUEUES', None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'dropin', 'run.py'), 'train_like.py', '--cfg', 'res50_coco', '--img_size', '64',
                        '--train_bs', '2'], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LAUNCHER_TRAIN_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_eval_loop_branches_agree(tmp_path):
    """eval.py:36-69 through `dropin/reference_loops.eval_loop` on a detecting network: the `--coco_api` branch as the reference writes
    it (dense masks over PCIe, one `add_mask` per detection) and the device-RLE variant produce the SAME records (category ids, boxes,
    scores, RLE strings), and the `prep_metrics` branch equals the CPU oracle's AP bookkeeping on the loop's own detections."""
    code = r'''
import os, sys, json
sys.path[:0] = [os.path.join(REPO, 'dropin'), REPO]
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import reference_loops as L
import bench
from oracle import metrics_ref as M
from yolact_minimal_amd.utils.synthetic import synth_eval_case
dev = torch.device('cuda:0')
net, cfg, img = bench.detecting_net('res50_coco', 256, dev)
h, w = 96, 128
_, _, _, _, gt, gt_masks, _, _ = synth_eval_case(1, 40, 7, h, w, 10)
loader = lambda: [(img, gt.clone(), gt_masks, h, w) for _ in range(2)]
_, mj_dense, seen, _ = L.eval_loop(net, cfg, loader(), coco_api=True)
_, mj_dev, seen2, _ = L.eval_loop(net, cfg, loader(), coco_api='device')
assert seen == seen2 == 2 and len(mj_dense.mask_data) > 10
assert mj_dense.bbox_data == mj_dev.bbox_data
assert mj_dense.mask_data == mj_dev.mask_data
ap, _, seen3, _ = L.eval_loop(net, cfg, loader(), coco_api=False)
# the oracle on the same detections (taken from one more pass of the hot path)
from utils.output_utils import nms, after_nms
with torch.no_grad():
    o = net(img)
d = nms(*o, net.anchors, cfg)
ids, sc, boxes, masks = after_nms(*d, h, w)
ref = M.new_ap_data(len(cfg.class_names), len(L.IOU_THRES))
for _ in range(2):
    M.prep_metrics(ref, [int(i) for i in ids.cpu()], [float(s) for s in sc.cpu()], boxes.cpu(), masks.cpu(), gt.clone(), gt_masks, h, w, L.IOU_THRES)
for kind in ('box', 'mask'):
    for k in range(len(L.IOU_THRES)):
        for c in range(len(cfg.class_names)):
            a, b = ap[kind][k][c], ref[kind][k][c]
            assert a.num_gt_positives == b.num_gt_positives and [p[1] for p in a.data_points] == [p[1] for p in b.data_points], (kind, k, c)
print('EVAL_LOOP_OK', len(mj_dense.mask_data), seen3)
'''
    r = subprocess.run([sys.executable, '-c', f'REPO = {REPO!r}\n' + code], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'EVAL_LOOP_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_ddp_no_sync_accumulates_locally(tmp_path):
    """`with ddp.no_sync():` (gradient accumulation) under the module-owned reducer: the backward inside leaves the gradients local
    (no collective is launched), the next synchronised backward accumulates into them and all-reduces the sum — torch's own contract.
    One RCCL rank with the reducer forced on; the accumulated gradient must equal g(batch A) + g(batch B) from two separate runs."""
    code = r'''
import os, sys, socket
sys.path[:0] = [os.path.join(REPO, 'dropin'), REPO]
os.environ['YM_FORCE_DIST'] = '1'
import torch, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from yolact_minimal_amd.config import build_cfg
from yolact_minimal_amd.modules.yolact import Yolact
from yolact_minimal_amd.utils.synthetic import synth_targets
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
cfg = build_cfg('res50_coco', 'train', 64, train_bs=2, bs_per_gpu=2)
torch.manual_seed(0)
net = Yolact(cfg); net.train()
ddp = DDP(net.cuda(), [0], output_device=0, broadcast_buffers=True)
def batch(seed):
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(seed)).cuda()
    b, m = synth_targets(2, 64, seed=seed)
    return img, [x.cuda() for x in b], [x.cuda() for x in m]
def grads():
    return {n: p.grad.detach().clone() for n, p in net.named_parameters()}
singles = []
for seed in (1, 2):                                   # each batch on its own (BatchNorm statistics are per forward either way)
    net.zero_grad()
    sum(ddp(*batch(seed))).backward()
    singles.append(grads())
red = net._train_state.reducer
assert red is not None and red.launches == 2 * len(red.buckets)
net.zero_grad()
before = red.launches
with ddp.no_sync():
    sum(ddp(*batch(1))).backward()
assert red.launches == before, 'a collective was launched inside no_sync()'
local = grads()
assert all(torch.equal(local[n], singles[0][n]) for n in local)
sum(ddp(*batch(2))).backward()                        # synchronised: accumulates, then reduces
assert red.launches == before + len(red.buckets)
acc = grads()
worst = max(float((acc[n] - (singles[0][n] + singles[1][n])).abs().max() / ((singles[0][n] + singles[1][n]).abs().max() + 1e-30)) for n in acc)
assert worst < 1e-6, worst
print('NO_SYNC_OK', worst)
dist.destroy_process_group()
'''
    r = subprocess.run([sys.executable, '-c', f'REPO = {REPO!r}\n' + code], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'NO_SYNC_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
