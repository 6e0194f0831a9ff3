import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import yolact_ref as R
from yolact_minimal_amd.config import build_cfg
from yolact_minimal_amd.modules.yolact import Yolact
size = 64
cfg = build_cfg('res50_coco', 'train', size)
keys = ['fpn.pred_layers.0.0.bias', 'fpn.pred_layers.0.0.weight', 'fpn.pred_layers.1.0.bias', 'proto_net.proto1.0.bias',
        'semantic_seg_conv.weight', 'prediction_layers.upfeature.0.bias', 'fpn.lat_layers.0.bias']
for which in range(4):
    torch.manual_seed(41)
    net = Yolact(cfg).train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    img = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(341))
    boxes, masks = R.synth_targets(2, size, seed=41)
    dt = torch.float64
    params = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k, _ in net.named_parameters():
        params[k].requires_grad_(True)
    out = R.TrainNet(params).forward(img.to(dt))
    anchors = torch.tensor(net.anchors).reshape(-1, 4).to(dt)
    torch.set_default_dtype(dt)
    ref = R.compute_loss(*out, [b.to(dt) for b in boxes], [m.to(dt) for m in masks], anchors)
    torch.set_default_dtype(torch.float32)
    ref[which].backward()
    net = net.to('cuda:0')
    losses = net(img.cuda(), [b.cuda() for b in boxes], [m.cuda() for m in masks])
    losses[which].backward()
    print('loss', which, float(losses[which].detach()), float(ref[which].detach()))
    for k in keys:
        p = dict(net.named_parameters())[k]
        if p.grad is None or params[k].grad is None:
            print('   ', k, 'no grad'); continue
        a, b = p.grad.cpu().double(), params[k].grad
        print(f'    {k:42s} {((a-b).abs().max()/(b.abs().max()+1e-30)).item():.2e}  |ref|max {b.abs().max().item():.3e}')
