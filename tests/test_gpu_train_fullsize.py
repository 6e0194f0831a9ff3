"""GPU: every training-time conv launch of the res101_coco / res50_coco 544 px plans (batch 8 and 16 per GPU: BASELINE configs 3
and 4) at its REAL shape and under the tile / split / staging choice tuned_gfx950.json selects for it -- the data-gradient convs
(MODE 2, with the fused BatchNorm-backward sums of their epilogue) and the weight-gradient GEMMs (every msplit / 64-wide tile /
DMA-ring variant the table picks) -- against fp64 evaluations of the same sums on random operands.

The end-to-end gradient goldens of test_gpu_train.py cannot be tight (a random-init BatchNorm ResNet amplifies fp32 rounding to
~5e-2 of max|g| in the reference itself); per LAUNCH the arithmetic is a plain fp32 dot product of <= 4608 (dgrad) or
<= 1.2 M (wgrad) terms, and is held to 1e-4 * max|reference| here.  The fp64 reference is evaluated on sampled outputs (2048 per
data gradient, 256 per weight gradient: a full fp64 convolution of every shape would take minutes); a wrong tile, a dropped K
slice or a ring hazard corrupts whole 32x32 / 64x64 blocks, which 2048 uniform samples of <= 38 M outputs still hit with
near certainty because every launch has < 40 k tiles -- and the column sums of the BatchNorm epilogue see EVERY output element.

Reference: modules/yolact.py:166-203 + train.py:124-127 (loss.backward() through every nn.Conv2d / BatchNorm2d of the net).
"""
import math
import re

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CHAIN = [544, 272, 136, 68, 34, 17, 9, 5]


def _keys(prefix):
    from yolact_minimal_amd.engine import tuned_table
    out = []
    for k in sorted(tuned_table()):
        m = re.match(rf'{prefix}_M(\d+)_N(\d+)_C(\d+)_k(\d+)_s(\d+)$', k)
        if m:
            out.append((k,) + tuple(int(v) for v in m.groups()))
    return out


def _geometry(M):
    for b in (8, 16):
        s = math.isqrt(M // b)
        if b * s * s == M and s in CHAIN:
            return b, s
    return None


def _shards(items, n=4):
    return [items[i::n] for i in range(n)]


@pytest.mark.parametrize('shard', range(4))
def test_every_tuned_data_gradient_launch_at_full_size(shard):
    """`T_M{b h w}_N{cin}_C{cout_pad}_k_s` = dx [b,h,w,cin] from dz [b,ho,wo,cout_pad]: 2048 sampled outputs vs fp64, and the two
    BatchNorm-backward column sums of the epilogue (sum dz, sum dz * xhat under the ReLU mask) vs fp64 over the WHOLE dx."""
    from yolact_minimal_amd import train_engine as T
    keys = _shards(_keys('T'))[shard]
    assert keys
    g = torch.Generator(device=DEV).manual_seed(100 + shard)
    worst = 0.0
    for key, M, cin, cout_pad, k, stride in keys:
        geo = _geometry(M)
        assert geo is not None, key
        b, h = geo
        pad = k // 2
        ho = (h + 2 * pad - k) // stride + 1
        dz = torch.randn(b, ho, ho, cout_pad, device=DEV, generator=g)
        w = torch.randn(cout_pad, cin, k, k, device=DEV, generator=g) * (1.0 / math.sqrt(cout_pad * k * k))
        # synthetic BatchNorm context of the tensor whose gradient this launch writes (ym_conv_desc.bnb_*)
        y = torch.randn(b, h, h, cin, device=DEV, generator=g)
        mean = torch.randn(cin, device=DEV, generator=g) * 0.1
        invstd = torch.rand(cin, device=DEV, generator=g) + 0.5
        gamma = torch.rand(cin, device=DEV, generator=g) + 0.5
        beta = torch.randn(cin, device=DEV, generator=g) * 0.2
        link = T.BnGradLink()
        link.y, link.out, link.mean, link.invstd = y.data_ptr(), None, mean.data_ptr(), invstd.data_ptr()
        link.gamma, link.beta, link.relu, link.c, link.m = gamma.data_ptr(), beta.data_ptr(), 1, cin, b * h * h
        T._stats_pool.begin(torch.device(DEV))
        dx = T._conv_dgrad(dz, w, cout_pad, (b, h, h, cin), stride, pad, bn_bwd=link)
        # ---- sampled fp64 reference of dx ----
        n = 2048
        sb = torch.randint(0, b, (n,), device=DEV, generator=g)
        sy = torch.randint(0, h, (n,), device=DEV, generator=g)
        sx = torch.randint(0, h, (n,), device=DEV, generator=g)
        sc = torch.randint(0, cin, (n,), device=DEV, generator=g)
        ref = torch.zeros(n, device=DEV, dtype=torch.float64)
        for kh in range(k):
            for kw in range(k):
                th, tw = sy + pad - kh, sx + pad - kw
                ok = (th >= 0) & (tw >= 0) & (th % stride == 0) & (tw % stride == 0)
                oh, ow = th // stride, tw // stride
                ok &= (oh < ho) & (ow < ho)
                rows = dz[sb, oh.clamp(0, ho - 1), ow.clamp(0, ho - 1)].double()            # [n, cout_pad]
                wv = w[:, sc, kh, kw].t().double()                                          # [n, cout_pad]
                ref += torch.where(ok, (rows * wv).sum(1), torch.zeros_like(ref))
        got = dx[sb, sy, sx, sc].double()
        err = float((got - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err <= 1e-4, (key, err)
        # ---- the fused BatchNorm-backward sums over every element of dx ----
        if link.stats is not None and link.dout_ptr == dx.data_ptr():
            stats = link.stats.double() if link.stats.dtype != torch.float64 else link.stats
            dxd, yd = dx.double().reshape(-1, cin), y.double().reshape(-1, cin)
            xhat32 = (y.reshape(-1, cin) - mean) * invstd                                    # fp32, the kernel's operation order
            # sign of fma(xhat, gamma, beta) (bn_affine): the product of two floats is exact in fp64, so this is the sign of the exact value
            mask = (xhat32.double() * gamma.double() + beta.double()) > 0
            dzm = torch.where(mask, dxd, torch.zeros_like(dxd))
            s0 = dzm.sum(0)
            s1 = (dzm * xhat32.double()).sum(0)
            e0 = float((stats[:cin] - s0).abs().max() / s0.abs().max())
            e1 = float((stats[cin:2 * cin] - s1).abs().max() / s1.abs().max())
            assert e0 <= 1e-4 and e1 <= 1e-4, (key, e0, e1)
        del dz, w, y, dx
    print(f'data gradients, shard {shard}: {len(keys)} launches, worst sampled error {worst:.2e} of max|ref|')


@pytest.mark.parametrize('shard', range(4))
def test_every_tuned_weight_gradient_launch_at_full_size(shard):
    """`W_M{b ho wo}_N{cout_pad}_C{cin_pad}_k_s` = dw [cout,cin,k,k] = sum over the b*ho*wo output pixels: 256 sampled filter
    taps vs fp64 (each is a dot product over up to 1.2 M pixels)."""
    from yolact_minimal_amd import train_engine as T
    keys = _shards(_keys('W'))[shard]
    assert keys
    g = torch.Generator(device=DEV).manual_seed(200 + shard)
    worst = 0.0
    for key, M, cout_pad, cin_p, k, stride in keys:
        geo = _geometry(M)
        assert geo is not None, key
        b, ho = geo
        pad = k // 2
        if stride == 1:
            h = ho
        elif k == stride:                                # Swin's patch embedding: 4x4 / 4, no padding (modules/swin_transformer.py:419-433)
            pad, h = 0, ho * stride
        else:
            h = CHAIN[CHAIN.index(ho) - 1]
            assert (h + 2 * pad - k) // stride + 1 == ho, key
        cin = 3 if cin_p == 4 else cin_p
        x = torch.randn(b, h, h, cin_p, device=DEV, generator=g)
        if cin_p == 4:
            x[..., 3] = 0
        dz = torch.randn(b, ho, ho, cout_pad, device=DEV, generator=g) * (1.0 / math.sqrt(M))
        dw = T._conv_wgrad_now(x, dz, (cout_pad, cin, k, k), stride, pad)
        torch.cuda.synchronize()
        n = 256
        sco = torch.randint(0, cout_pad, (n,), device=DEV, generator=g)
        sci = torch.randint(0, cin, (n,), device=DEV, generator=g)
        skh = torch.randint(0, k, (n,), device=DEV, generator=g)
        skw = torch.randint(0, k, (n,), device=DEV, generator=g)
        xp = F.pad(x, (0, 0, pad, pad, pad, pad))
        ref = torch.zeros(n, device=DEV, dtype=torch.float64)
        for kh in range(k):
            for kw in range(k):
                sel = ((skh == kh) & (skw == kw)).nonzero().flatten()
                if sel.numel() == 0:
                    continue
                xs = xp[:, kh:kh + stride * (ho - 1) + 1:stride, kw:kw + stride * (ho - 1) + 1:stride][..., sci[sel]].double()
                ds = dz[..., sco[sel]].double()
                ref[sel] = (xs * ds).sum(dim=(0, 1, 2))
        got = dw[sco, sci, skh, skw].double()
        err = float((got - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err <= 1e-4, (key, err)
        del x, dz, dw, xp
    print(f'weight gradients, shard {shard}: {len(keys)} launches, worst sampled error {worst:.2e} of max|ref|')


@pytest.mark.parametrize('b,h,c', [(8, 136, 256), (8, 68, 512), (16, 34, 1024), (8, 17, 2048), (8, 272, 64)])
def test_bn_pool_upsample_backward_at_full_size(b, h, c):
    """The HBM-bound backward passes at the plan's sizes against torch's own fp64 autograd on the device: train-mode BatchNorm
    (+ ReLU) backward (`ym_bn_train_bwd`), and for the stem / FPN sizes max-pool 3x3/2 and bilinear x2 backward."""
    from yolact_minimal_amd import train_engine as T
    g = torch.Generator(device=DEV).manual_seed(b + h + c)
    conv = torch.nn.Conv2d(c, c, 1, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(c, device=DEV).reshape(c, c, 1, 1))
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, device=DEV, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, device=DEV, generator=g) * 0.2)
    x = torch.randn(b, h, h, c, device=DEV, generator=g)
    gy = torch.randn(b, h, h, c, device=DEV, generator=g)
    xg = x.clone().requires_grad_()
    T._stats_pool.begin(torch.device(DEV))
    out = T._conv_bn(xg, conv, bn, relu=True)                     # identity 1x1 conv: the BatchNorm sees x itself
    out.backward(gy)
    xd = x.double().requires_grad_()
    wd, bd = bn.weight.detach().double().requires_grad_(), bn.bias.detach().double().requires_grad_()
    ref = F.relu(F.batch_norm(xd.permute(0, 3, 1, 2), None, None, wd, bd, True, 0.1, bn.eps))
    ref.backward(gy.double().permute(0, 3, 1, 2))
    for got, want, name in ((xg.grad, xd.grad, 'dx'), (bn.weight.grad, wd.grad, 'dgamma'), (bn.bias.grad, bd.grad, 'dbeta')):
        err = float((got.double() - want).abs().max() / want.abs().max())
        assert err <= 1e-4, (name, err)
    if h in (272, 68, 34):
        xq = torch.randn(b, h, h, min(c, 256), device=DEV, generator=g)
        if h == 272:                                              # the stem's max-pool (modules/resnet.py:91), post-ReLU ties included
            xq = F.relu(xq)
            a = xq.clone().requires_grad_()
            y = T.MaxPool.apply(a)
            gq = torch.randn_like(y)
            y.backward(gq)
            # (ties among post-ReLU zeros: ATen's CPU kernel keeps the first maximum in scan order, which is the rule the HIP kernel
            # reproduces -- torch's own device kernel may pick another one, so this reference runs on the host)
            r = xq.cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()
            F.max_pool2d(r, 3, 2, 1).backward(gq.cpu().permute(0, 3, 1, 2).contiguous())
            want = r.grad.to(DEV).double()
        else:                                                     # FPN top-down x2 (align_corners=False, modules/yolact.py:70-71)
            a = xq.clone().requires_grad_()
            y = T.Bilinear2x.apply(a, False)
            gq = torch.randn_like(y)
            y.backward(gq)
            r = xq.double().permute(0, 3, 1, 2).requires_grad_()
            F.interpolate(r, scale_factor=2, mode='bilinear', align_corners=False).backward(gq.double().permute(0, 3, 1, 2))
            want = r.grad
        err = float((a.grad.double() - want.permute(0, 2, 3, 1)).abs().max() / want.abs().max())
        assert err <= 1e-4, ('pool / upsample', err)
