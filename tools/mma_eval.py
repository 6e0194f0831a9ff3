"""Evaluation of the split-bf16 conv modes (ym_conv_desc.mma = 3 / 6) against the f32 MFMA parity mode:
accuracy on the reference's 544 px goldens (bs=1 and bs=8, the north-star 1e-4 bar) and forward throughput.
    python tools/mma_eval.py [cfg ...]        (needs an MI355X; prints one JSON line per configuration)"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, 'tests', 'golden')


def make_net(name, seed):
    from oracle import yolact_ref as R          # (test-infrastructure generator of the golden's weights; evaluation tool only)
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    cfg = build_cfg(name, 'val', 544)
    torch.manual_seed(seed)
    net = Yolact(cfg).eval()
    sd = net.state_dict()
    if name.startswith('swin'):
        from oracle.make_golden_swin import randomize_swin_
        randomize_swin_(sd, seed + 100)
    else:
        R.randomize_bn_(sd, seed + 100)
    R.randomize_bias_(sd, seed + 200)
    net.load_state_dict(sd)
    return net


def bar(got, want):
    """max over elements of |got - want| / (1e-4 + 1e-4 |want|): <= 1 passes the parity tests' bound."""
    want = torch.as_tensor(want).to(got.device)
    return float(((got - want).abs() / (1e-4 + 1e-4 * want.abs())).max()), float((got - want).abs().max())


def main():
    dev = torch.device('cuda:0')
    names = sys.argv[1:] or ['res101_coco', 'res50_coco', 'swin_tiny_coco']
    flops = {'res101_coco': 157.2e9, 'res50_coco': 113.4e9, 'swin_tiny_coco': 119.2e9}
    for name in names:
        g8 = np.load(os.path.join(GOLD, f'forward_{name}_544_b8_digest.npz'))
        seed = int(g8['seed'])
        net = make_net(name, seed).to(dev)
        img = torch.randn(8, 3, 544, 544, generator=torch.Generator().manual_seed(seed + 300)).to(dev)
        base = None
        for batch in [int(b) for b in os.environ.get('YM_EVAL_BATCHES', '8,1').split(',')]:
            x = img[:batch].contiguous()
            for mma in [int(m) for m in os.environ.get('YM_EVAL_MMA', '0,3,6').split(',')]:
                with torch.no_grad():
                    net(x)
                eng = net._engine(x)
                eng.set_mma(mma)
                n_split = sum(1 for c in eng.convs if c.mma)
                with torch.no_grad():
                    out = [t.clone() for t in net(x)]
                torch.cuda.synchronize()
                rec = dict(cfg=name, batch=batch, mma=mma, convs_split=n_split, convs=len(eng.convs))
                if batch == 8:
                    worst = 0.0
                    for t, key, sl in ((out[0], 'class_sample', (slice(None), slice(None, None, 97))), (out[1], 'box_sample', (slice(None), slice(None, None, 97))),
                                       (out[2], 'coef_sample', (slice(None), slice(None, None, 97))),
                                       (out[3], 'proto_sample', (slice(None), slice(None, None, 9), slice(None, None, 9)))):
                        r, a = bar(t[sl], g8[key])
                        rec[key.replace('_sample', '') + '_vs_reference'] = dict(ratio_to_bar=round(r, 3), max_abs=a)
                        worst = max(worst, r)
                    rec['passes_bar'] = worst <= 1.0
                if mma == 0:
                    base = out
                elif base is not None:
                    rec['vs_f32_mode'] = {k: dict(ratio_to_bar=round(bar(o, b)[0], 3), max_abs=bar(o, b)[1])
                                          for k, o, b in zip(('class', 'box', 'coef', 'proto'), out, base)}
                # throughput: graph replay of the forward
                for _ in range(5):
                    eng.run(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                iters = 30
                for _ in range(iters):
                    eng.run(x)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
                rec.update(ms=round(dt * 1e3, 3), img_s=round(batch / dt, 1), tflops_f32_equiv=round(flops[name] * batch / dt / 1e12, 1))
                if os.environ.get('YM_EVAL_LAYERS'):
                    from bench import conv_roofline
                    _, _, _, layers = conv_roofline(eng, x, iters=3)
                    top = sorted(layers, key=lambda l: -l['ms'])[:12]
                    rec['conv_ms_total'] = round(sum(l['ms'] for l in layers), 3)
                    rec['top_layers'] = [(l['name'], round(l['ms'] * 1e3, 1), round(l['gflop'] / l['ms'], 1)) for l in top]
                print(json.dumps(rec), flush=True)
            net._engines.clear() if hasattr(net, '_engines') else None
        del net
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
