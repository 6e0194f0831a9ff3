#!/usr/bin/env python3
"""Tune only the pyramid-head conv shapes (sig ..._L5) for bs 1 and 8 and merge them into the tuned table."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net  # noqa: E402
from yolact_minimal_amd import engine as E  # noqa: E402

dev = torch.device('cuda:0')
out = {}
net, cfg = build_net('res50_coco', 544, dev)
for b in (1, 8):
    img = torch.randn(b, 3, 544, 544, device=dev)
    eng = net._engine(img)
    keep = [c for c in eng.convs if c.sig.endswith('_L5')]
    allc = eng.convs
    eng.convs = keep
    res = eng.autotune(10, verbose=True)
    eng.convs = allc
    for k, v in res.items():
        out[k] = v[:7]
    net._engines.clear()
table = json.load(open(E.TUNED_PATH))
table.update(out)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(table, open('gpurun_out/tuned_pyr.json', 'w'), indent=0, sort_keys=True)
print(out)
