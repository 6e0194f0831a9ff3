"""Runs the reference's training loop (dropin/reference_loops.py = the statements of /root/reference/train.py:44-48,60-63,76,
102-130,165-166) on cuda:0 through `dropin/`, then the build's own `Trainer` on the same seed and inputs, and writes what both
produced to an .npz.  A process of its own because train.py's `get_config(args, mode='train')` joins a process group (one RCCL
rank here).  Used by tests/test_gpu_reference_loop.py; not collected by pytest.

    python tests/run_reference_loop.py --cfg res50_coco --img_size 128 --train_bs 2 --steps 3 --seed 71 --out /tmp/x.npz
"""
import argparse
import os
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(REPO, 'dropin'), REPO]          # what dropin/run.py puts in front of the checkout

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
if 'MASTER_PORT' not in os.environ:
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    os.environ['MASTER_PORT'] = str(s.getsockname()[1])
    s.close()
os.environ.setdefault('RANK', '0')
os.environ.setdefault('LOCAL_RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')

import numpy as np  # noqa: E402
import torch  # noqa: E402

# --- the reference scripts' import lines (train.py:14-16, eval.py:13) ---
from modules.yolact import Yolact  # noqa: E402
from config import get_config  # noqa: E402
import reference_loops as L  # noqa: E402

from oracle import yolact_ref as R  # noqa: E402   (synthetic targets: the generator the goldens were made with)


def digest(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def sample(t):
    f = t.reshape(-1)
    return f[:: max(1, f.numel() // 64)][:64]


def pad64(a):
    return np.pad(a, (0, 64 - a.shape[0]))


def main():
    # train.py:21-31: the reference's flags (defaults kept), + what this harness needs
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
    parser.add_argument('--steps', type=int, default=3)
    parser.add_argument('--seed', type=int, default=71)
    parser.add_argument('--out', required=True)
    parser.add_argument('--no_drop_path', action='store_true')
    parser.add_argument('--wellcond', action='store_true', help='the well-conditioned weights of the loop goldens (oracle/make_golden_loop.py)')
    parser.add_argument('--backbone', default=None, help='a backbone-only checkpoint for net.backbone.init_backbone (train.py:55)')
    args = parser.parse_args()
    steps, seed, out_path = args.steps, args.seed, args.out
    args.local_rank = int(os.environ['LOCAL_RANK'])          # torch.distributed.launch used to pass --local_rank

    cfg = get_config(args, mode='train')                     # train.py:44 (joins the RCCL group: config.py:229-232)
    assert cfg.cuda and torch.distributed.get_world_size() == 1

    def build():
        torch.manual_seed(seed)
        net = Yolact(cfg)                                    # train.py:47-48
        net.train()
        if args.backbone:
            net.backbone.init_backbone(args.backbone)        # train.py:55
        if args.wellcond:
            sd = net.state_dict()
            R.damp_residual_branches_(sd, seed + 400)
            R.shift_bn_bias_(sd, 3.0)
            net.load_state_dict(sd)
        if args.no_drop_path:
            for blk in net.modules():
                if hasattr(blk, 'drop_prob'):
                    blk.drop_prob = 0.0
        return net

    size, batch = cfg.img_size, cfg.bs_per_gpu
    img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
    boxes, masks = R.synth_targets(batch, size, seed=seed)
    data_loader = [(img, [b.clone() for b in boxes], [m.clone() for m in masks]) for _ in range(steps)]      # CPU tensors, like train_collate

    # ---- A. the reference's loop: torch optimizer + torch DDP around the module -------------------------------------------------
    os.environ['YM_FORCE_DIST'] = '1'                        # one rank: the module's own gradient reducer still takes part (train_state.py)
    net = build()
    p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    optimizer = L.make_optimizer(net, cfg)                   # train.py:60-65
    net = L.wrap_ddp(net, args.local_rank)                   # train.py:76
    lrs = []
    hist = []

    snaps_a = []                                             # the parameters after every step (flat copy)

    def flat_of(module):
        return torch.cat([p.detach().reshape(-1) for p in module.parameters()])

    def on_step(step, losses, lr):
        hist.append([float(l.detach()) for l in losses])
        lrs.append(lr)
        snaps_a.append(flat_of(net.module))
    _, end_step = L.train_loop(net, optimizer, cfg, data_loader, max_steps=steps, on_step=on_step)
    torch.cuda.synchronize()
    mod = net.module
    pa = {k: v.detach().clone() for k, v in mod.named_parameters()}
    sda = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    net.eval()                                               # train.py:165-166: evaluate(net.module, ...) -> eval forward
    with torch.no_grad():
        out_a = [o.clone() for o in mod(img[:1].cuda())]
    net.train()
    grads_none = sum(1 for p in mod.parameters() if p.grad is None)

    # ---- B. the build's own Trainer on the same seed / inputs -------------------------------------------------------------------
    from yolact_minimal_amd.trainer import Trainer
    net_b = build()
    tr = Trainer(net_b, cfg, torch.device('cuda', args.local_rank), world=1, local_rank=args.local_rank)
    img_d, boxes_d, masks_d = img.cuda(), [b.cuda() for b in boxes], [m.cuda() for m in masks]
    hist_b, snaps_b = [], []
    for _ in range(steps):
        losses = tr.step(img_d, [b.clone() for b in boxes_d], masks_d)
        hist_b.append([float(l.detach()) for l in losses])
        snaps_b.append(flat_of(net_b))
    torch.cuda.synchronize()
    flat0 = torch.cat([p0[k].reshape(-1) for k in p0]).cuda()
    # per step: max |update_loop - update_trainer| / max |update| over the whole parameter vector, and the same for the parameters
    step_update_diff = [float(((a - flat0).double() - (b - flat0).double()).abs().max() / ((a - flat0).double().abs().max() + 1e-30))
                        for a, b in zip(snaps_a, snaps_b)]
    step_param_diff = [float((a.double() - b.double()).abs().max() / (a.double().abs().max() + 1e-30)) for a, b in zip(snaps_a, snaps_b)]
    pb = {k: v.detach().clone() for k, v in net_b.named_parameters()}
    sdb = net_b.state_dict()
    net_b.eval()
    with torch.no_grad():
        out_b = [o.clone() for o in net_b(img_d[:1])]

    keys = list(pa.keys())
    rel = []
    for k in keys:
        da, db = (pa[k] - p0[k].cuda()).double(), (pb[k] - p0[k].cuda()).double()
        rel.append(float((da - db).abs().max() / (da.abs().max() + 1e-30)))
    stem = 'backbone.bn1' if 'backbone.bn1.running_mean' in sda else None
    buf_diff = max([float((sda[k].double() - sdb[k].double()).abs().max() / (sda[k].double().abs().max() + 1e-30))
                    for k in sda if k not in pa and sda[k].is_floating_point()] or [0.0])
    np.savez_compressed(
        out_path, losses_loop=np.array(hist), losses_trainer=np.array(hist_b), lrs=np.array(lrs), keys=np.array(keys),
        update_rel_diff=np.array(rel), step_update_diff=np.array(step_update_diff), step_param_diff=np.array(step_param_diff), buffer_rel_diff=np.array(buf_diff), grads_none=np.array(grads_none), end_step=np.array(end_step),
        param_digest=np.stack([digest(pa[k]) for k in keys]),
        update_digest=np.stack([digest(pa[k] - p0[k].cuda()) for k in keys]),
        update_sample=np.stack([pad64(sample(pa[k] - p0[k].cuda()).double().cpu().numpy()) for k in keys]),
        run_mean_stem=sda[f'{stem}.running_mean'].cpu().numpy() if stem else np.zeros(0),
        run_var_stem=sda[f'{stem}.running_var'].cpu().numpy() if stem else np.zeros(0),
        num_batches_tracked=np.array(int(sda[f'{stem}.num_batches_tracked']) if stem else -1),
        num_batches_tracked_trainer=np.array(int(sdb[f'{stem}.num_batches_tracked']) if stem else -1),
        eval_digest=np.stack([digest(o) for o in out_a]),
        eval_class_sample=out_a[0][0, ::37].cpu().numpy(), eval_proto_sample=out_a[3][0, ::5, ::5].cpu().numpy(),
        eval_loop_vs_trainer=np.array([float((a.double() - b.double()).abs().max() / (a.double().abs().max() + 1e-30))
                                       for a, b in zip(out_a, out_b)]),
        optimizer=np.array(type(optimizer).__name__), wrapper=np.array(type(net).__name__))
    print('REFERENCE_LOOP_OK', out_path, flush=True)
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
