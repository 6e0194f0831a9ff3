import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from yolact_minimal_amd.train_engine import ConvBias
DEV='cuda:0'
def nhwc(t): return t.permute(0,2,3,1).contiguous()
def nchw(t): return t.permute(0,3,1,2).contiguous()
for (cin,cout,k,hw,b) in [(256,80,1,8,2),(256,32,1,16,2),(256,80,1,12,2),(256,96,1,8,2),(256,64,1,8,2),(256,80,3,8,2),(64,80,1,8,2)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b,cin,hw,hw,generator=g); w = torch.randn(cout,cin,k,k,generator=g)*0.05; bias=torch.randn(cout,generator=g)
    xc = x.double().requires_grad_(); wc = w.double().requires_grad_()
    y = F.conv2d(xc, wc, bias.double(), 1, k//2)
    gy = torch.randn(y.shape, generator=g).double()
    y.backward(gy)
    xg = nhwc(x).to(DEV).requires_grad_(); wg = w.to(DEV).requires_grad_(); bg = bias.to(DEV).requires_grad_()
    cp = (cout+31)//32*32
    yg = ConvBias.apply(xg, wg, bg, 1, k//2, 0, cp, None)
    gyp = torch.zeros(b,hw,hw,cp); gyp[...,:cout] = nhwc(gy.float())
    yg.backward(gyp.to(DEV))
    ex = ((nchw(xg.grad).cpu().double()-xc.grad).abs().max()/xc.grad.abs().max()).item()
    ew = ((wg.grad.cpu().double()-wc.grad).abs().max()/wc.grad.abs().max()).item()
    print((cin,cout,k,hw,b), f'dx err {ex:.2e}  dw err {ew:.2e}')
