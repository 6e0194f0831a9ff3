import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from yolact_minimal_amd.train_engine import ConvBias, Bilinear2x
DEV='cuda:0'
def nhwc(t): return t.permute(0,2,3,1).contiguous()
def nchw(t): return t.permute(0,3,1,2).contiguous()
g = torch.Generator().manual_seed(1)
b, hw = 2, 8
x = torch.randn(b,256,hw,hw,generator=g)
w1 = torch.randn(256,256,3,3,generator=g)*0.02; b1 = torch.randn(256,generator=g)*0.1
w2 = torch.randn(80,256,1,1,generator=g)*0.05; b2 = torch.randn(80,generator=g)*0.1
w3 = torch.randn(256,256,3,3,generator=g)*0.02; b3 = torch.randn(256,generator=g)*0.1
def run_ref(dt):
    xs = [t.to(dt).clone().requires_grad_() for t in (x,w1,b1,w2,b2,w3,b3)]
    p3 = F.relu(F.conv2d(xs[0], xs[1], xs[2], 1, 1))
    seg = F.conv2d(p3, xs[3], xs[4])
    up = F.relu(F.conv2d(p3, xs[5], xs[6], 1, 1))
    up2 = F.interpolate(up, scale_factor=2, mode='bilinear', align_corners=True)
    loss = (seg.sin()).sum() + (up2.cos()*up2).sum()
    loss.backward()
    return [t.grad for t in xs]
r64 = run_ref(torch.float64); r32 = run_ref(torch.float32)
xg = nhwc(x).to(DEV).requires_grad_()
ps = [t.to(DEV).requires_grad_() for t in (w1,b1,w2,b2,w3,b3)]
p3 = ConvBias.apply(xg, ps[0], ps[1], 1, 1, 1, 256, None)
seg = ConvBias.apply(p3, ps[2], ps[3], 1, 0, 0, 96, None)[..., :80].permute(0,3,1,2)
up = ConvBias.apply(p3, ps[4], ps[5], 1, 1, 1, 256, None)
up2 = Bilinear2x.apply(up, True)
loss = (seg.sin()).sum() + (up2.cos()*up2).sum()
loss.backward()
got = [nchw(xg.grad).cpu()] + [p.grad.cpu() for p in ps]
for name, a, b64, b32 in zip(['x','w1','b1','w2','b2','w3','b3'], got, r64, r32):
    e = ((a.double()-b64).abs().max()/b64.abs().max()).item(); e32 = ((b32.double()-b64).abs().max()/b64.abs().max()).item()
    print(f'{name}: gpu err {e:.2e}   cpu32 err {e32:.2e}')
