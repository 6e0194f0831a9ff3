"""GPU parity of Yolact.forward (HIP engine) against the CPU oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import yolact_ref as R
from tests.test_oracle_golden import make_net

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _close(got, want, name, atol=1e-4, rtol=1e-4):
    # north_star tolerance: mask/conf fp32 within 1e-4 (absolute); raw tensors also checked relatively
    got, want = got.cpu(), want.cpu() if torch.is_tensor(want) else torch.from_numpy(want)
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    assert bool((err <= bound).all()), f'{name}: max err {err.max().item():.3e} (max |ref| {want.abs().max().item():.3e})'


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('name,size,batch', [('res50_coco', 64, 1), ('res50_coco', 96, 2), ('res101_coco', 128, 1)])
def test_forward_matches_golden_and_oracle(golden_dir, name, size, batch, graph, monkeypatch):
    monkeypatch.setenv('YM_GRAPH', '1' if graph else '0')
    g = np.load(os.path.join(golden_dir, f'forward_{name}_{size}_b{batch}.npz'))
    seed = int(g['seed'])
    net, cfg = make_net(name, size, seed)
    img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
    with torch.no_grad():
        feats = R.features(img, net.state_dict())
    net = net.to(DEV)
    with torch.no_grad():
        out = net(img.to(DEV))
        out2 = net(img.to(DEV))         # second call exercises graph replay / buffer reuse
    torch.cuda.synchronize()
    for a, b in zip(out, out2):
        assert torch.equal(a, b)
    for t, key in zip(out, ('class_pred', 'box_pred', 'coef_pred', 'proto_out')):
        _close(t, g[key], key + ' vs reference golden')
    # pre-softmax logits (softmax of a random-init net is nearly flat, so check logits relatively too)
    eng = net._engine(img.to(DEV))
    _close(eng.class_logits, feats[0], 'class logits', atol=1e-4, rtol=1e-4)
    _close(eng.box_pred, feats[1], 'box', atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('name', ['res50_coco', 'res101_coco'])
def test_forward_544_digest(golden_dir, name):
    """Full-size (the "550-class" config is really 544): sampled slices + sums from the reference."""
    g = np.load(os.path.join(golden_dir, f'forward_{name}_544_digest.npz'))
    seed = int(g['seed'])
    net, cfg = make_net(name, 544, seed)
    img = torch.randn(1, 3, 544, 544, generator=torch.Generator().manual_seed(seed + 300))
    net = net.to(DEV)
    with torch.no_grad():
        cls, box, coef, proto = net(img.to(DEV))
    assert cls.shape == (1, 18525, 81) and proto.shape == (1, 136, 136, 32)
    _close(cls[0, ::37], g['class_sample'], 'class sample')
    _close(box[0, ::37], g['box_sample'], 'box sample')
    _close(coef[0, ::37], g['coef_sample'], 'coef sample')
    _close(proto[0, ::5, ::5], g['proto_sample'], 'proto sample')
    for t, key in ((cls, 'class_digest'), (box, 'box_digest'), (coef, 'coef_digest'), (proto, 'proto_digest')):
        d = t.double()
        got = np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()])
        np.testing.assert_allclose(got, g[key], rtol=2e-4)


def test_batch_equals_per_image():
    """bs=8-style batching: image i of a batch equals the bs=1 result."""
    net, cfg = make_net('res50_coco', 64, 77)
    net = net.to(DEV)
    imgs = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        full = net(imgs)
        for i in range(3):
            one = net(imgs[i:i + 1])
            for a, b in zip(full, one):
                torch.testing.assert_close(a[i:i + 1], b, rtol=1e-5, atol=1e-6)


def test_weight_reload_invalidates_packed_weights():
    net, cfg = make_net('res50_coco', 64, 5)
    net = net.to(DEV)
    img = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        a = net(img)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd['prediction_layers.bbox_layer.bias'] += 1.0
        net.load_state_dict(sd)
        b = net(img)
    torch.testing.assert_close(b[1], a[1] + 1.0, rtol=1e-5, atol=1e-5)
