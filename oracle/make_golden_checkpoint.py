"""Checkpoint surface of the REAL reference (imported from /root/reference): for res50_coco / res101_coco / swin_tiny_coco in TRAIN
mode, the ordered key list + shapes + dtypes of `net.state_dict()` (what `save_latest` / `save_best` write,
utils/common_utils.py:41-63, and what `Yolact.load_weights` reads back strictly, modules/yolact.py:127-139) and of
`net.backbone.state_dict()` (what `init_backbone` loads, modules/resnet.py:100-104, modules/swin_transformer.py:486-498).
Also round-trips a reference-written file through the reference's own `load_weights` in val mode (the `semantic_seg_conv.*` drop).

TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python -m oracle.make_golden_checkpoint   -> tests/golden/checkpoint_keys.json
"""
import json
import os
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.make_golden import import_reference, ref_cfg, OUT  # noqa: E402


def main():
    ref_config, ref_yolact, _, _ = import_reference()
    out = {}
    for name in ('res50_coco', 'res101_coco', 'swin_tiny_coco'):
        torch.manual_seed(1)
        net = ref_yolact.Yolact(ref_cfg(ref_config, name, 64, mode='train'))
        sd = net.state_dict()
        bb = net.backbone.state_dict()
        path = os.path.join(tempfile.mkdtemp(), f'latest_{name}_7.pth')
        torch.save(sd, path)                                            # utils/common_utils.py:62
        val = ref_yolact.Yolact(ref_cfg(ref_config, name, 64, mode='val'))
        val.load_weights(path, False)                                   # strict, after dropping semantic_seg_conv.* (:133-137)
        vsd = val.state_dict()
        assert all(torch.equal(vsd[k], sd[k]) for k in vsd) and set(sd) - set(vsd) == {'semantic_seg_conv.weight', 'semantic_seg_conv.bias'}
        out[name] = dict(train_keys=[[k, list(v.shape), str(v.dtype)] for k, v in sd.items()],
                         backbone_keys=[[k, list(v.shape), str(v.dtype)] for k, v in bb.items()],
                         val_dropped=sorted(set(sd) - set(vsd)), n_parameters=sum(p.numel() for p in net.parameters()))
        print(name, len(sd), 'state-dict entries,', len(bb), 'backbone entries,', out[name]['n_parameters'], 'parameters')
    with open(os.path.join(OUT, 'checkpoint_keys.json'), 'w') as f:
        json.dump(out, f)


if __name__ == '__main__':
    main()
