import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from yolact_minimal_amd.train_engine import ConvBias
DEV='cuda:0'
for (cin,cout,k,stride,hw,act,cout_pad) in [(256,256,3,1,4,1,256),(256,351,3,1,4,0,352),(256,256,3,1,2,1,256),(256,256,3,2,4,1,256),(256,256,1,1,4,0,256),(256,256,3,1,8,1,256)]:
    g = torch.Generator().manual_seed(cin+cout+hw)
    x = torch.randn(2, cin, hw, hw, generator=g); w = torch.randn(cout, cin, k, k, generator=g)*(1/(cin*k*k)**0.5); b = torch.randn(cout, generator=g)*0.1
    pad=k//2; ho=(hw+2*pad-k)//stride+1
    dy = torch.randn(2, cout, ho, ho, generator=g)
    xc, wc, bc = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    y = F.conv2d(xc, wc, bc, stride, pad)
    if act==1: y = F.relu(y)
    (y*dy.double()).sum().backward()
    xg = x.permute(0,2,3,1).contiguous().to(DEV).requires_grad_(); wg = w.to(DEV).requires_grad_(); bg = b.to(DEV).requires_grad_()
    out = ConvBias.apply(xg, wg, bg, stride, pad, act, cout_pad, None)
    dyp = torch.zeros(2, ho, ho, cout_pad); dyp[..., :cout] = dy.permute(0,2,3,1)
    (out*dyp.to(DEV)).sum().backward()
    def rel(a,b): return float((a.double()-b).abs().max()/(b.abs().max()+1e-12))
    print((cin,cout,k,stride,hw), 'y', rel(out.detach().cpu()[..., :cout].permute(0,3,1,2), y.detach()), 'dx', rel(xg.grad.cpu().permute(0,3,1,2), xc.grad), 'dw', rel(wg.grad.cpu(), wc.grad), 'db', rel(bg.grad.cpu(), bc.grad))
