import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import yolact_ref as R
from yolact_minimal_amd.config import build_cfg
from yolact_minimal_amd.modules.yolact import Yolact
for size, dt in ((64, torch.float32), (128, torch.float32), (128, torch.float64)):
    cfg = build_cfg('res50_coco', 'train', size)
    torch.manual_seed(41)
    net = Yolact(cfg).train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    img = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(341))
    boxes, masks = R.synth_targets(2, size, seed=41)
    params = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k, _ in net.named_parameters():
        params[k].requires_grad_(True)
    out = R.TrainNet(params).forward(img.to(dt))
    anchors = torch.tensor(net.anchors).reshape(-1, 4).to(dt)
    torch.set_default_dtype(dt)
    ref_losses = R.compute_loss(*out, [b.to(dt) for b in boxes], [m.to(dt) for m in masks], anchors)
    torch.set_default_dtype(torch.float32)
    sum(ref_losses).backward()
    net = net.to('cuda:0')
    losses = net(img.cuda(), [b.cuda() for b in boxes], [m.cuda() for m in masks])
    sum(losses).backward()
    print(size, dt, [float(l) for l in losses], [float(l) for l in ref_losses])
    errs = []
    for k, p in net.named_parameters():
        a, b = p.grad.cpu().double(), params[k].grad.double()
        errs.append((k, (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)))
    for k, e in errs[::12]:
        print(f'   {k:50s} {e:.2e}')
    print('   max', max(errs, key=lambda t: t[1]))
