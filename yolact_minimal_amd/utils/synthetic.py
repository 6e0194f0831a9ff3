"""Seeded synthetic inputs for the path (there is no dataset and no checkpoint offline): the shapes SURVEY.md §8d prescribes
for bench.py, `__graft_entry__.smoke()` and the tests.  Pure input generators — no part of the computation."""
import numpy as np
import torch
import torch.nn.functional as F


def synth_head_outputs(n_anchors, num_classes=81, proto_hw=136, seed=1, bg_bias=4.0, spread=2.5):
    """softmax(randn*spread + bg_bias*e_bg), randn*0.5 boxes, tanh(randn) coefs, relu(randn) protos."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(1, n_anchors, num_classes, generator=g) * spread
    logits[..., 0] += bg_bias
    cls = F.softmax(logits, -1)
    box = torch.randn(1, n_anchors, 4, generator=g) * 0.5
    coef = torch.tanh(torch.randn(1, n_anchors, 32, generator=g))
    proto = F.relu(torch.randn(1, proto_hw, proto_hw, 32, generator=g))
    return cls, box, coef, proto


def synth_targets(batch, img_size, n_gt=4, num_classes=80, seed=0):
    """SURVEY.md §8d training inputs: n_gt boxes uniform in [0.1,0.9] with min side 0.1, rectangular float masks."""
    boxes, masks = [], []
    for i in range(batch):
        g = torch.Generator().manual_seed(seed + i)
        xy = torch.rand(n_gt, 2, generator=g) * 0.6 + 0.1
        wh = torch.rand(n_gt, 2, generator=g) * 0.25 + 0.1
        x2y2 = torch.clamp(xy + wh, max=0.9)
        cls = torch.randint(0, num_classes, (n_gt, 1), generator=g).float()
        boxes.append(torch.cat([xy, x2y2, cls], 1))
        m = torch.zeros(n_gt, img_size, img_size)
        for j in range(n_gt):
            x1, y1, x2, y2 = (torch.cat([xy[j], x2y2[j]]) * img_size).round().long().tolist()
            m[j, y1:y2, x1:x2] = 1.0
        masks.append(m)
    return boxes, masks


def synth_eval_case(seed, n=40, g=7, h=48, w=64, num_classes=6):
    """Synthetic detections vs ground truth: rectangular gt masks, predictions = jittered copies (some duplicates, some
    wrong-class, some empty masks) so that every branch of the matching is exercised.  Returns the prep_metrics arguments."""
    rng = torch.Generator().manual_seed(seed)
    gt_box = torch.zeros(g, 5)
    gt_masks = torch.zeros(g, h, w)
    for j in range(g):
        x1, y1 = torch.rand(2, generator=rng).mul(0.55).tolist()
        bw, bh = (torch.rand(2, generator=rng) * 0.3 + 0.12).tolist()
        gt_box[j] = torch.tensor([x1, y1, x1 + bw, y1 + bh, float(torch.randint(0, num_classes, (1,), generator=rng))])
        gt_masks[j, int(y1 * h):int((y1 + bh) * h) + 1, int(x1 * w):int((x1 + bw) * w) + 1] = 1.0
    ids, scores, boxes, masks = [], [], torch.zeros(n, 4, dtype=torch.int32), torch.zeros(n, h, w)
    for i in range(n):
        j = int(torch.randint(0, g, (1,), generator=rng))
        jit = (torch.rand(4, generator=rng) - 0.5) * (0.02 + 0.3 * float(torch.rand(1, generator=rng)))
        b = (gt_box[j, :4] + jit).clamp(0, 1)
        x1, y1, x2, y2 = int(b[0] * w), int(b[1] * h), int(b[2] * w), int(b[3] * h)
        boxes[i] = torch.tensor([x1, y1, x2, y2], dtype=torch.int32)
        if i % 11 != 10:                                             # every 11th prediction has an empty mask
            masks[i, y1:y2 + 1, x1:x2 + 1] = 1.0
        cls = int(gt_box[j, 4]) if i % 5 else int(torch.randint(0, num_classes, (1,), generator=rng))
        ids.append(cls)
        scores.append(float(torch.rand(1, generator=rng)))
    order = sorted(range(n), key=lambda k: -scores[k])             # after_nms returns detections by descending score
    return ([ids[k] for k in order], [scores[k] for k in order], boxes[order], masks[order], gt_box, gt_masks, h, w)


def synth_sample(seed, h, w, n):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    boxes, masks = [], []
    for _ in range(n):
        x1, y1 = rng.uniform(0, w * 0.6), rng.uniform(0, h * 0.6)
        bw, bh = rng.uniform(12, w * 0.4), rng.uniform(12, h * 0.4)
        x2, y2 = min(x1 + bw, w - 1), min(y1 + bh, h - 1)
        boxes.append([x1, y1, x2, y2])
        m = np.zeros((h, w), np.uint8)
        m[int(y1):int(y2) + 1, int(x1):int(x2) + 1] = 1
        masks.append(m)
    return img, np.stack(masks), np.array(boxes), rng.integers(0, 80, n)


def synth_polygons(seed, h, w, n=6):
    """Random annotations for the tests / bench: per annotation 1-3 polygons (star-shaped, 3-40 vertices, fractional
    coordinates, some reaching outside the image), like COCO's 'segmentation' lists."""
    rng = np.random.default_rng(seed)
    anns = []
    for _ in range(n):
        polys = []
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(3, 41))
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            rad = rng.uniform(0.5, max(2.0, 0.45 * min(h, w)))
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            r = rad * rng.uniform(0.4, 1.0, k)
            xs = np.round(cx + r * np.cos(ang), 2)
            ys = np.round(cy + r * np.sin(ang), 2)
            if rng.random() < 0.7:                      # COCO polygons are clipped to the image; keep some that are not
                xs, ys = np.clip(xs, 0, w), np.clip(ys, 0, h)
            polys.append(np.stack([xs, ys], 1).reshape(-1).tolist())
        anns.append(polys)
    return anns
