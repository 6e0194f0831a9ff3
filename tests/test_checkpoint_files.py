"""Checkpoint FILES at the boundary (the "published .pth loads" half of north_star's acceptance that can be shown
without the published weights): `Yolact.load_weights(path, cuda)` (modules/yolact.py:127-139) and `backbone.init_backbone(path)`
(modules/resnet.py:100-104, modules/swin_transformer.py:486-498) on files that carry the REAL reference's key names, order, shapes
and dtypes (tests/golden/checkpoint_keys.json, oracle/make_golden_checkpoint.py), for res50_coco / res101_coco / swin_tiny_coco.

CPU part: the key surface.  GPU part: write the file, load it the way eval.py / train.py do, run the HIP forward, compare with the
oracle on the file's tensors (1e-4)."""
import json
import os

import numpy as np
import pytest
import torch

from yolact_minimal_amd.config import build_cfg
from yolact_minimal_amd.modules.yolact import Yolact

CFGS = ('res50_coco', 'res101_coco', 'swin_tiny_coco')


@pytest.fixture(scope='module')
def surface(golden_dir):
    with open(os.path.join(golden_dir, 'checkpoint_keys.json')) as f:
        return json.load(f)


def _reference_keyed_state(name, surface, seed):
    """A state dict with the reference's train-mode keys / shapes / dtypes in the reference's order, filled from a seeded stream
    (non-trivial BatchNorm / LayerNorm statistics and biases), i.e. what `save_latest(net, ...)` of the reference would have written."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    template = None
    for key, shape, dtype in surface[name]['train_keys']:
        if dtype == 'torch.int64' and shape:                     # Swin-T's registered `relative_position_index` tables: constants
            if template is None:
                template = Yolact(build_cfg(name, 'train', 64)).state_dict()
            t = template[key].clone()
        elif dtype == 'torch.int64':
            t = torch.tensor(7, dtype=torch.int64)               # num_batches_tracked
        elif key.endswith('running_var'):
            t = torch.rand(shape, generator=g) + 0.5
        elif key.endswith('running_mean') or key.endswith('.bias'):
            t = torch.randn(shape, generator=g) * 0.1
        elif key.endswith('relative_position_bias_table'):
            t = torch.randn(shape, generator=g) * 0.5
        elif key.endswith('bn3.weight'):                         # (keeps the 33 residual blocks of res101 from saturating the softmax)
            t = torch.rand(shape, generator=g) * 0.2 + 0.1
        elif len(shape) == 1:                                    # BatchNorm / LayerNorm scale
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        else:                                                    # conv / linear weight: xavier-like
            fan = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = (torch.rand(shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5
        sd[key] = t
    return sd


@pytest.mark.parametrize('name', CFGS)
def test_state_dict_surface_is_the_references(name, surface):
    """Train mode: every key, in order, with the reference's shape and dtype (incl. `num_batches_tracked`, the registered
    `relative_position_index` buffers of Swin-T and `semantic_seg_conv.*`); val mode: the same minus `semantic_seg_conv.*`; the
    backbone's own state dict = what `init_backbone` loads; parameter count of the reference's print-out (train.py:94)."""
    torch.manual_seed(0)
    net = Yolact(build_cfg(name, 'train', 64))
    got = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
    assert got == surface[name]['train_keys']
    assert [[k, list(v.shape), str(v.dtype)] for k, v in net.backbone.state_dict().items()] == surface[name]['backbone_keys']
    assert sum(p.numel() for p in net.parameters()) == surface[name]['n_parameters']
    val = Yolact(build_cfg(name, 'val', 64))
    assert sorted(set(net.state_dict()) - set(val.state_dict())) == surface[name]['val_dropped']


@pytest.mark.parametrize('name', CFGS)
def test_load_weights_reads_a_reference_file_strictly(name, surface, tmp_path):
    """`load_weights(path, cuda=False)` in val mode drops the train-only keys and loads the rest strictly; a file with a missing or
    an unexpected key is refused (strict=True, reference :137), as is a train-mode load of a file without `semantic_seg_conv.*`."""
    sd = _reference_keyed_state(name, surface, 5)
    path = str(tmp_path / f'best_30.5_{name}_392000.pth')
    torch.save(sd, path)
    val = Yolact(build_cfg(name, 'val', 64))
    val.load_weights(path, False)
    own = val.state_dict()
    assert all(torch.equal(own[k], sd[k]) for k in own) and len(own) == len(sd) - 2
    train = Yolact(build_cfg(name, 'train', 64))
    train.load_weights(path, False)
    assert all(torch.equal(v, sd[k]) for k, v in train.state_dict().items())
    broken = dict(sd)
    broken.pop('fpn.lat_layers.0.weight')
    torch.save(broken, path)
    with pytest.raises(RuntimeError, match='Missing key'):
        Yolact(build_cfg(name, 'val', 64)).load_weights(path, False)
    extra = dict(sd, **{'not.a.key': torch.zeros(1)})
    torch.save(extra, path)
    with pytest.raises(RuntimeError, match='Unexpected key'):
        Yolact(build_cfg(name, 'val', 64)).load_weights(path, False)
    no_seg = {k: v for k, v in sd.items() if not k.startswith('semantic_seg_conv')}
    torch.save(no_seg, path)
    with pytest.raises(RuntimeError, match='Missing key'):
        Yolact(build_cfg(name, 'train', 64)).load_weights(path, False)


@pytest.mark.gpu
@pytest.mark.parametrize('name,size', [('res50_coco', 128), ('res101_coco', 128), ('swin_tiny_coco', 128), ('res101_coco', 544)])
def test_checkpoint_file_round_trip_on_gpu(name, size, surface, tmp_path):
    """eval.py:118-125 on a reference-keyed file: `Yolact(cfg)`, `load_weights(cfg.weight, cfg.cuda)` (torch.load WITHOUT
    map_location, as the reference does with cuda), `.eval()`, `.cuda()`, forward — against the oracle evaluated on the file's
    tensors.  Then train.py:55: a backbone-only file through `net.backbone.init_backbone(path)` into a train-mode net, whose eval
    forward must equal the oracle on ITS state dict (heads from the seeded construction, backbone from the file)."""
    from oracle import yolact_ref as R
    sd = _reference_keyed_state(name, surface, 11)
    path = str(tmp_path / f'best_30.5_{name}_392000.pth')
    torch.save(sd, path)
    cfg = build_cfg(name, 'val', size)
    net = Yolact(cfg)
    net.load_weights(path, True)
    net.eval()
    net = net.cuda()
    assert not any(k.startswith('semantic_seg_conv') for k in net.state_dict())
    img = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        out = net(img.cuda())
        ref = R.forward_eval_any(img, {k: v for k, v in sd.items() if not k.startswith('semantic_seg_conv')})
    for a, b, what in zip(out, ref, ('class', 'box', 'coef', 'proto')):
        err = float((a.cpu() - b).abs().max())
        assert err <= 1e-4 * max(1.0, float(b.abs().max())), (what, err)

    # backbone-only checkpoint (weights/backbone_res101.pth, weights/swin_tiny.pth in the reference's README)
    bb = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    assert [[k, list(v.shape), str(v.dtype)] for k, v in bb.items()] == surface[name]['backbone_keys']
    bpath = str(tmp_path / ('swin_tiny.pth' if name.startswith('swin') else f'backbone_{name[:-5]}.pth'))
    torch.save(bb, bpath)
    torch.manual_seed(13)
    tnet = Yolact(build_cfg(name, 'train', size))
    tnet.train()
    tnet.backbone.init_backbone(bpath)
    own = tnet.state_dict()
    assert all(torch.equal(own['backbone.' + k], v) for k, v in bb.items())
    tnet.eval()
    tnet = tnet.cuda()
    with torch.no_grad():
        out = tnet(img.cuda())
        ref = R.forward_eval_any(img, {k: v.cpu() for k, v in tnet.state_dict().items() if not k.startswith('semantic_seg_conv')})
    for a, b, what in zip(out, ref, ('class', 'box', 'coef', 'proto')):
        err = float((a.cpu() - b).abs().max())
        assert err <= 1e-4 * max(1.0, float(b.abs().max())), ('init_backbone', what, err)
