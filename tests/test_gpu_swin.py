"""GPU parity of the Swin-T path (LayerNorm, patch merge, shifted-window attention on MFMA, GELU epilogue) against the
CPU oracle and the real reference's golden vectors (swin_tiny_coco)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import yolact_ref as R
from yolact_minimal_amd.config import build_cfg
from yolact_minimal_amd.modules.yolact import Yolact

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_layernorm_and_patch_merge():
    from yolact_minimal_amd import hip
    g = torch.Generator().manual_seed(0)
    for c in (96, 192, 768, 1536):
        x = torch.randn(37, c, generator=g) * 3 + 1
        w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
        out = torch.empty(37, c, device=DEV)
        hip.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out)
        torch.testing.assert_close(out.cpu(), F.layer_norm(x, (c,), w, b, 1e-5), rtol=1e-5, atol=1e-5)
    for (h, w_, c) in ((8, 8, 96), (7, 9, 192), (17, 17, 384)):
        x = torch.randn(2, h, w_, c, generator=g)
        gm, bt = torch.rand(4 * c, generator=g) + 0.5, torch.randn(4 * c, generator=g)
        sd = {'m.norm.weight': gm, 'm.norm.bias': bt, 'm.reduction.weight': torch.eye(4 * c)}
        want = R.swin_merge(x.reshape(2, h * w_, c), h, w_, sd, 'm')       # identity reduction -> the LN output
        out = torch.empty(2, (h + 1) // 2, (w_ + 1) // 2, 4 * c, device=DEV)
        hip.patch_merge_layernorm(x.to(DEV), gm.to(DEV), bt.to(DEV), 1e-5, out)
        torch.testing.assert_close(out.cpu().reshape(2, -1, 4 * c), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('h,w,heads,shift', [(14, 14, 3, 0), (14, 14, 3, 3), (8, 10, 6, 3), (17, 17, 24, 3), (5, 6, 12, 0)])
def test_window_attention_matches_oracle(h, w, heads, shift):
    """Between qkv and proj of one block, including pad / roll / partition / mask / reverse / un-roll / crop."""
    from yolact_minimal_amd import hip
    g = torch.Generator().manual_seed(h * 100 + w + heads + shift)
    c, ws, b = heads * 32, 7, 2
    xn = torch.randn(b, h, w, c, generator=g)                       # "norm1 output"
    qkv_w, qkv_b = torch.randn(3 * c, c, generator=g) * c ** -0.5, torch.randn(3 * c, generator=g) * 0.3
    table = torch.randn(169, heads, generator=g) * 0.5
    sd = {'a.qkv.weight': qkv_w, 'a.qkv.bias': qkv_b, 'a.proj.weight': torch.eye(c), 'a.proj.bias': torch.zeros(c),
          'a.relative_position_bias_table': table, 'a.relative_position_index': R.swin_rel_index(ws)}
    pr, pb = (ws - w % ws) % ws, (ws - h % ws) % ws
    y = F.pad(xn, (0, 0, 0, pr, 0, pb))
    hp, wp = y.shape[1:3]
    mask = R.swin_shift_mask(hp, wp, ws, ws // 2) if shift else None
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    a = R.swin_attention(R.swin_windows(y, ws), sd, 'a', heads, ws, mask)
    y = R.swin_unwindows(a, ws, hp, wp)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    want = y[:, :h, :w, :].contiguous()
    qkv = F.linear(xn, qkv_w, qkv_b).reshape(b * h * w, 3 * c).contiguous()
    out = torch.full((b * h * w, c), float('nan'), device=DEV)
    hip.swin_window_attention(qkv.to(DEV), qkv_b.to(DEV), table.to(DEV), b, h, w, c, heads, ws, shift, out)
    torch.testing.assert_close(out.cpu().reshape(b, h, w, c), want, rtol=1e-4, atol=1e-5)


def _make_swin(size, seed):
    from oracle.make_golden_swin import randomize_swin_
    cfg = build_cfg('swin_tiny_coco', 'val', size)
    torch.manual_seed(seed)
    net = Yolact(cfg).eval()
    sd = net.state_dict()
    randomize_swin_(sd, seed + 100)
    R.randomize_bias_(sd, seed + 200)
    net.load_state_dict(sd)
    return net, cfg


def _close(got, want, name, atol=1e-4, rtol=1e-4):
    got = got.cpu()
    want = want if torch.is_tensor(want) else torch.from_numpy(want)
    err = (got - want).abs()
    assert bool((err <= atol + rtol * want.abs()).all()), f'{name}: max err {err.max().item():.3e}'


@pytest.mark.parametrize('graph', ['0', '1'])
def test_swin_forward_128_matches_golden(golden_dir, graph, monkeypatch):
    monkeypatch.setenv('YM_GRAPH', graph)
    g = np.load(os.path.join(golden_dir, 'forward_swin_tiny_coco_128_b2.npz'))
    seed = int(g['seed'])
    net, cfg = _make_swin(128, seed)
    img = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(seed + 300))
    with torch.no_grad():
        ref = R.forward_eval_any(img, net.state_dict())
    net = net.to(DEV)
    with torch.no_grad():
        out = net(img.to(DEV))
    for t, key, r in zip(out, ('class_pred', 'box_pred', 'coef_pred', 'proto_out'), ref):
        _close(t, g[key], key + ' vs reference golden')
        _close(t, r, key + ' vs oracle')


def test_swin_forward_544_digest(golden_dir):
    g = np.load(os.path.join(golden_dir, 'forward_swin_tiny_coco_544_digest.npz'))
    seed = int(g['seed'])
    net, cfg = _make_swin(544, seed)
    img = torch.randn(1, 3, 544, 544, generator=torch.Generator().manual_seed(seed + 300))
    net = net.to(DEV)
    with torch.no_grad():
        cls, box, coef, proto = net(img.to(DEV))
    _close(cls[0, ::37], g['class_sample'], 'class sample')
    _close(box[0, ::37], g['box_sample'], 'box sample')
    _close(coef[0, ::37], g['coef_sample'], 'coef sample')
    _close(proto[0, ::5, ::5], g['proto_sample'], 'proto sample')
