"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A plain PyTorch-CPU fp32 restatement of the reference hot path (feiyuhuahuo/Yolact_minimal), written
functionally over a state-dict so that it does not depend on either the reference's or the
product's module classes.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this file, and only as the checker / reported baseline.

Pinned by: `oracle/make_golden.py` imports the real reference from /root/reference in the build
container, runs it and this restatement on identical seeded inputs, asserts equality and writes the
vectors under `tests/golden/` (the reference itself has no tests or golden vectors — SURVEY.md §4).
The arithmetic that is not in the reference's own files (conv / batch-norm / softmax / sort /
interpolate) is PyTorch's CPU implementation (un-pinned by the reference: "PyTorch >= 1.1"); the
goldens are therefore tied to torch 2.10.0 CPU kernels.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------
# network forward (eval mode)
# ------------------------------------------------------------------------------------------------
def _conv(x, sd, name, stride=1, padding=0):
    return F.conv2d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride=stride, padding=padding)


def _bn(x, sd, name):
    # eval-mode BatchNorm2d, eps 1e-5 (torch default, modules/resnet.py:8-14)
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'], False, 0.0, 1e-5)


def bottleneck(x, sd, p, stride):
    """modules/resnet.py:20-40 — stride on the 3x3, residual add then ReLU."""
    y = F.relu(_bn(_conv(x, sd, p + '.conv1'), sd, p + '.bn1'))
    y = F.relu(_bn(_conv(y, sd, p + '.conv2', stride=stride, padding=1), sd, p + '.bn2'))
    y = _bn(_conv(y, sd, p + '.conv3'), sd, p + '.bn3')
    if (p + '.downsample.0.weight') in sd:
        x = _bn(_conv(x, sd, p + '.downsample.0', stride=stride), sd, p + '.downsample.1')
    return F.relu(y + x)


def resnet(x, sd, layers, p='backbone'):
    """modules/resnet.py:86-98 — stem 7x7/2 + BN + ReLU + maxpool 3x3/2, then 4 stages."""
    x = F.relu(_bn(_conv(x, sd, p + '.conv1', stride=2, padding=3), sd, p + '.bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nblk in enumerate(layers):
        for bi in range(nblk):
            x = bottleneck(x, sd, f'{p}.layers.{li}.{bi}', 2 if (bi == 0 and li > 0) else 1)
        outs.append(x)
    return outs


def fpn(c3, c4, c5, sd, p='fpn'):
    """modules/yolact.py:73-89 — note P6 is computed from P5 *after* its pred conv."""
    def up(t):
        return F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False)
    p5_1 = _conv(c5, sd, p + '.lat_layers.2')
    p4_1 = _conv(c4, sd, p + '.lat_layers.1') + up(p5_1)
    p3_1 = _conv(c3, sd, p + '.lat_layers.0') + up(p4_1)
    p5 = F.relu(_conv(p5_1, sd, p + '.pred_layers.2.0', padding=1))
    p4 = F.relu(_conv(p4_1, sd, p + '.pred_layers.1.0', padding=1))
    p3 = F.relu(_conv(p3_1, sd, p + '.pred_layers.0.0', padding=1))
    p6 = F.relu(_conv(p5, sd, p + '.downsample_layers.0.0', stride=2, padding=1))
    p7 = F.relu(_conv(p6, sd, p + '.downsample_layers.1.0', stride=2, padding=1))
    return [p3, p4, p5, p6, p7]


def protonet(p3, sd, p='proto_net'):
    """modules/yolact.py:49-53 — the only align_corners=True upsample in the net."""
    x = p3
    for i in (0, 2, 4):
        x = F.relu(_conv(x, sd, f'{p}.proto1.{i}', padding=1))
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    x = F.relu(_conv(x, sd, p + '.proto2.0', padding=1))
    x = F.relu(_conv(x, sd, p + '.proto2.2'))
    return x


def head(x, sd, num_classes, coef_dim=32, p='prediction_layers'):
    """modules/yolact.py:26-31 — anchor index = (y*W + x)*3 + a."""
    b = x.shape[0]
    x = F.relu(_conv(x, sd, p + '.upfeature.0', padding=1))
    conf = _conv(x, sd, p + '.conf_layer', padding=1).permute(0, 2, 3, 1).reshape(b, -1, num_classes)
    box = _conv(x, sd, p + '.bbox_layer', padding=1).permute(0, 2, 3, 1).reshape(b, -1, 4)
    coef = torch.tanh(_conv(x, sd, p + '.coef_layer.0', padding=1)).permute(0, 2, 3, 1).reshape(b, -1, coef_dim)
    return conf, box, coef


def resnet_layers_from_sd(sd):
    n = [0, 0, 0, 0]
    for k in sd:
        if k.startswith('backbone.layers.') and k.endswith('.conv1.weight'):
            li, bi = int(k.split('.')[2]), int(k.split('.')[3])
            n[li] = max(n[li], bi + 1)
    return tuple(n)


def features(img, sd):
    """Backbone + FPN + protonet + head logits, before the eval softmax (modules/yolact.py:141-157)."""
    layers = resnet_layers_from_sd(sd)
    c2, c3, c4, c5 = resnet(img, sd, layers)
    levels = fpn(c3, c4, c5, sd)
    proto = protonet(levels[0], sd).permute(0, 2, 3, 1).contiguous()
    num_classes = sd['prediction_layers.conf_layer.weight'].shape[0] // 3
    confs, boxes, coefs = zip(*(head(lv, sd, num_classes) for lv in levels))
    return (torch.cat(confs, 1), torch.cat(boxes, 1), torch.cat(coefs, 1), proto,
            dict(c2=c2, c3=c3, c4=c4, c5=c5, levels=levels))


def forward_eval(img, sd):
    """modules/yolact.py:141-164, eval branch: returns (class_pred softmaxed, box, coef, proto NHWC)."""
    conf, box, coef, proto, _ = features(img, sd)
    return F.softmax(conf, -1), box, coef, proto


# ------------------------------------------------------------------------------------------------
# anchors
# ------------------------------------------------------------------------------------------------
def anchors_for(img_size, scales, aspect_ratios=(1, 0.5, 2)):
    """modules/yolact.py:111-114 + utils/box_utils.py:86-101 (float64 list -> fp32 tensor [N,4])."""
    out = []
    for stride, scale in zip((8, 16, 32, 64, 128), scales):
        n = math.ceil(img_size / stride)
        for j in range(n):
            for i in range(n):
                for ar in aspect_ratios:
                    r = math.sqrt(ar)
                    out += [(i + 0.5) / n, (j + 0.5) / n, scale * r / img_size, scale / r / img_size]
    return torch.tensor(out).reshape(-1, 4)


# ------------------------------------------------------------------------------------------------
# post-processing
# ------------------------------------------------------------------------------------------------
def pairwise_iou(a, b):
    """utils/box_utils.py:8-37 — inter / (area_a + area_b - inter), no +1, 0/0 -> NaN. a:[n,A,4] b:[n,B,4]."""
    lo = torch.max(a[:, :, None, :2], b[:, None, :, :2])
    hi = torch.min(a[:, :, None, 2:], b[:, None, :, 2:])
    wh = torch.clamp(hi - lo, min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]))[:, :, None]
    area_b = ((b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]))[:, None, :]
    return inter / (area_a + area_b - inter)


_expf_lib = None


def expf_cr(x):
    """exp(x) rounded to nearest float32 (oracle/expf_cr.c: IEEE double, fixed operation sequence, no libm)."""
    global _expf_lib
    if _expf_lib is None:
        path = os.path.join(_HERE, 'libexpf_cr.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle`')
        _expf_lib = ctypes.CDLL(path)
        _expf_lib.oracle_expf_cr_array.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    x = x.detach().float().contiguous()
    y = torch.empty_like(x)
    _expf_lib.oracle_expf_cr_array(x.data_ptr(), y.data_ptr(), x.numel())
    return y


def decode(box_p, anchors, exp='torch'):
    """utils/output_utils.py:148-153 — centre-size decode to corners, clipped to [0,1].
    exp='torch': the reference's own call (MKL VML on this torch build: host-ISA dependent, 1 ulp off the correctly rounded
    value in ~1.1 % of inputs).  exp='cr': the correctly rounded exp of oracle/expf_cr.c — the host-independent anchor the
    HIP kernel is held to bit for bit."""
    cxcy = anchors[:, :2] + box_p[:, :2] * 0.1 * anchors[:, 2:]
    if torch.is_tensor(exp):      # FROZEN outputs of the reference's own torch.exp for exactly these inputs (tests/golden/exp_decode_frozen.npz)
        assert exp.shape == box_p[:, 2:].shape and exp.dtype == torch.float32
        e = exp
    else:
        e = expf_cr(box_p[:, 2:] * 0.2) if exp == 'cr' else torch.exp(box_p[:, 2:] * 0.2)
    wh = anchors[:, 2:] * e
    x1y1 = cxcy - wh / 2
    x2y2 = wh + x1y1
    return torch.clip(torch.cat((x1y1, x2y2), 1), min=0., max=1.)


def _sort_desc(t, dim, stable):
    # torch.sort(descending=True) is unstable for n > 16 on CPU; `stable=True` pins ties to "lower index
    # first", which is what the HIP kernels implement (the reference's tie order is implementation-defined).
    return t.sort(dim=dim, descending=True, stable=True) if stable else t.sort(dim, descending=True)


def fast_nms(boxes, coefs, scores, top_k=200, iou_thre=0.5, max_det=100, stable=False):
    """utils/output_utils.py:11-43. scores:[C,K] boxes:[K,4] coefs:[K,32]."""
    scores, idx = _sort_desc(scores, 1, stable)
    idx, scores = idx[:, :top_k], scores[:, :top_k]
    ncls, ndet = idx.shape
    bx = boxes[idx.reshape(-1)].reshape(ncls, ndet, 4)
    cf = coefs[idx.reshape(-1)].reshape(ncls, ndet, -1)
    iou = pairwise_iou(bx, bx)
    iou.triu_(diagonal=1)                 # zero-fills (does not multiply): lower-triangle NaNs vanish
    iou_max, _ = iou.max(dim=1)           # NaN propagates -> `<=` below is False -> dropped
    keep = iou_max <= iou_thre
    cls = torch.arange(ncls)[:, None].expand_as(keep)[keep]
    bx, cf, sc = bx[keep], cf[keep], scores[keep]
    sc, order = _sort_desc(sc, 0, stable)
    order, sc = order[:max_det], sc[:max_det]
    return bx[order], cf[order], cls[order], sc


_greedy_lib = None


def _greedy():
    global _greedy_lib
    if _greedy_lib is None:
        path = os.path.join(_HERE, 'libgreedy_nms.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle` (done by __graft_entry__.build())')
        lib = ctypes.CDLL(path)
        lib.oracle_greedy_nms.restype = ctypes.c_int
        lib.oracle_greedy_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _greedy_lib = lib
    return _greedy_lib


def greedy_nms(dets, thresh):
    """cython_nms.pyx:24-74 via the C restatement oracle/greedy_nms.c. dets float32 [n,5] -> kept idx (ascending)."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    keep = np.empty(dets.shape[0], dtype=np.int64)
    n = _greedy().oracle_greedy_nms(dets.ctypes.data, dets.shape[0], ctypes.c_float(thresh), keep.ctypes.data)
    return keep[:n]


def traditional_nms(boxes, coefs, scores, img_size, score_thre=0.05, iou_thre=0.5, max_det=100, stable=False):
    """utils/output_utils.py:84-123 — per-class greedy NMS on boxes scaled by img_size."""
    boxes = boxes * img_size
    idx_l, cls_l, scr_l = [], [], []
    for c in range(scores.shape[0]):
        s = scores[c]
        m = s > score_thre
        if int(m.sum()) == 0:
            continue
        cand = torch.arange(s.shape[0])[m]
        dets = torch.cat([boxes[m], s[m][:, None]], 1).numpy()
        keep = torch.from_numpy(greedy_nms(dets, iou_thre)).long()
        idx_l.append(cand[keep])
        cls_l.append(torch.full_like(keep, c))
        scr_l.append(s[m][keep])
    idx, cls, scr = torch.cat(idx_l), torch.cat(cls_l), torch.cat(scr_l)
    scr, order = _sort_desc(scr, 0, stable)
    order, scr = order[:max_det], scr[:max_det]
    idx, cls = idx[order], cls[order]
    return boxes[idx] / img_size, coefs[idx], cls, scr


def nms(class_pred, box_pred, coef_pred, proto_out, anchors, score_thre=0.05, iou_thre=0.5, top_k=200,
        max_det=100, traditional=False, img_size=544, stable=False, exp='torch'):
    """utils/output_utils.py:126-163 (batch of one). Returns (ids, scores, boxes, coefs, proto) or 5x None."""
    cls = class_pred.squeeze(0).transpose(1, 0).contiguous()[1:]       # [C-1, N], background dropped
    box_p, coef_p, proto = box_pred.squeeze(0), coef_pred.squeeze(0), proto_out.squeeze(0)
    keep = cls.max(dim=0)[0] > score_thre
    cls_k = cls[:, keep]
    boxes = decode(box_p[keep], anchors[keep], exp)
    coefs = coef_p[keep]
    if cls_k.shape[1] == 0:
        return None, None, None, None, None
    if traditional:
        bx, cf, ids, sc = traditional_nms(boxes, coefs, cls_k, img_size, score_thre, iou_thre, max_det, stable)
    else:
        bx, cf, ids, sc = fast_nms(boxes, coefs, cls_k, top_k, iou_thre, max_det, stable)
    return ids, sc, bx, cf, proto


def crop_window(boxes, w, h, padding=1):
    """utils/box_utils.py:117-132,147-153 — float window [x1,x2) x [y1,y2) in prototype pixels."""
    def span(a, b, size):
        a, b = a * size, b * size
        lo, hi = torch.min(a, b), torch.max(a, b)
        return torch.clamp(lo - padding, min=0), torch.clamp(hi + padding, max=size)
    x1, x2 = span(boxes[:, 0], boxes[:, 2], w)
    y1, y2 = span(boxes[:, 1], boxes[:, 3], h)
    return x1, x2, y1, y2


def crop(masks, boxes, padding=1):
    """utils/box_utils.py:147-168. masks [h,w,n]."""
    h, w, n = masks.shape
    x1, x2, y1, y2 = crop_window(boxes, w, h, padding)
    xs = torch.arange(w, dtype=x1.dtype).view(1, -1, 1)
    ys = torch.arange(h, dtype=x1.dtype).view(-1, 1, 1)
    inside = (xs >= x1.view(1, 1, -1)) & (xs < x2.view(1, 1, -1)) & (ys >= y1.view(1, 1, -1)) & (ys < y2.view(1, 1, -1))
    return masks * inside.float()


def assemble_masks(proto, coefs, boxes, do_crop=True):
    """utils/output_utils.py:217-222 — sigmoid(proto @ coef^T), crop, -> [n, Hp, Wp] (pre-resize, soft)."""
    m = torch.sigmoid(torch.matmul(proto, coefs.t()))
    if do_crop:
        m = crop(m, boxes)
    return m.permute(2, 0, 1).contiguous()


def after_nms(ids, scores, boxes, coefs, proto, img_h, img_w, visual_thre=0.0, do_crop=True, return_soft=False):
    """utils/output_utils.py:200-233. `boxes` is NOT mutated here (the reference scales it in place)."""
    if ids is None:
        return None, None, None, None
    if visual_thre > 0:
        k = scores >= visual_thre
        if not bool(k.any()):
            return None, None, None, None
        ids, scores, boxes, coefs = ids[k], scores[k], boxes[k], coefs[k]
    soft = assemble_masks(proto, coefs, boxes, do_crop)
    size = max(img_h, img_w)
    up = F.interpolate(soft.unsqueeze(0), (size, size), mode='bilinear', align_corners=False).squeeze(0)
    masks = (up > 0.5).float()
    masks = masks[:, :img_h, :] if img_h < img_w else masks[:, :, :img_w]
    boxes_px = (boxes * size).int()
    if return_soft:
        up = up[:, :img_h, :] if img_h < img_w else up[:, :, :img_w]
        return ids, scores, boxes_px, masks, soft, up
    return ids, scores, boxes_px, masks


# ------------------------------------------------------------------------------------------------
# synthetic inputs shared by goldens, tests and bench
# ------------------------------------------------------------------------------------------------
from yolact_minimal_amd.utils.synthetic import synth_head_outputs, synth_targets  # noqa: E402,F401  (input generators)


def detections_match(a, b, tol=1e-4, cut_margin=None):
    """Do two `nms` results (ids, scores, boxes, ...) describe the same detections up to `tol`?  Detections are ordered by score,
    and scores that differ by less than the arithmetic noise of two implementations may swap places or fall on different sides
    of the top-`max_det` cut, so the comparison is on SETS: every detection of one result whose score is safely above the other
    result's cut (by `cut_margin`, default 2 tol) must have a partner of the same class with score and box within `tol`; partners
    are used once.  Returns (ok, message, pairs) with pairs = [(index in a, index in b)]."""
    cut_margin = 2 * tol if cut_margin is None else cut_margin
    ia, sa, ba = a[0], a[1], a[2]
    ib, sb, bb = b[0], b[1], b[2]
    if ia is None or ib is None:
        return (ia is None and ib is None), 'one result is empty', []
    used, pairs = set(), []
    cut_a, cut_b = float(sa.min()), float(sb.min())
    for i in range(ia.numel()):
        best = None
        for j in range(ib.numel()):
            if j in used or int(ia[i]) != int(ib[j]):
                continue
            if abs(float(sa[i]) - float(sb[j])) <= tol and float((ba[i] - bb[j]).abs().max()) <= tol:
                best = j
                break
        if best is None:
            if float(sa[i]) > cut_b + cut_margin:
                return False, f'detection {i} of a (class {int(ia[i])}, score {float(sa[i]):.6f}) has no partner in b', pairs
        else:
            used.add(best)
            pairs.append((i, best))
    for j in range(ib.numel()):
        if j not in used and float(sb[j]) > cut_a + cut_margin:
            return False, f'detection {j} of b (class {int(ib[j])}, score {float(sb[j]):.6f}) has no partner in a', pairs
    return True, '', pairs


def randomize_bn_(sd, seed=7):
    """Give every BatchNorm non-trivial statistics/affine (default init is identity-like and would not test folding)."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd):
        if k.endswith('running_mean'):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.1)
        elif k.endswith('running_var'):
            sd[k].copy_(torch.rand(sd[k].shape, generator=g) * 0.5 + 0.75)
        elif '.bn' in k or 'downsample.1' in k:
            if k.endswith('.weight'):
                sd[k].copy_(torch.rand(sd[k].shape, generator=g) * 0.2 + 0.4)
            elif k.endswith('.bias'):
                sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.05)
    return sd


def damp_residual_branches_(sd, seed=13, scale=0.05):
    """Every Bottleneck's last BatchNorm gets gamma in [scale/2, 3*scale/2] ("zero-init-residual" style): the blocks are close to
    the identity, which makes a random-init net well enough conditioned in backward for a meaningful gradient comparison
    (with gamma = 1 the reference's OWN CPU gradients differ by 2e-2 of max|g| between fp32 and fp64, and by 7e-3 between an
    8-thread and a 1-thread fp32 run — DESIGN.md "Oracle and parity")."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd):
        if k.endswith('bn3.weight'):
            sd[k].copy_(torch.rand(sd[k].shape, generator=g) * scale + scale / 2)
    return sd


def shift_bn_bias_(sd, k=2.0):
    """Every backbone BatchNorm gets beta = k * |gamma|: the ReLU behind it switches at -k sigma of its input instead of at the mode.
    Why: what makes a random-init YOLACT step "ill-conditioned" in fp32 is not rounding that grows with depth but DISCRETE ReLU
    sign flips — a unit whose pre-activation lies inside the forward rounding noise (~1e-6 of the scale; ~0.4 * 1e-6 of all units
    when the zero crossing sits at the mode of a normalised activation) comes out on the other side in another fp32 implementation,
    which changes that channel's BatchNorm-backward sums by 1 / (samples per channel) and everything below it.  Measured with the
    CPU oracle (fp32 vs fp64, damped res50 256 px bs=4): the error is 6e-6 of max|g| in layer4's last block, jumps to 8e-2 in ONE
    tensor (layers.3.1.conv2) and is 5e-4 (median) in every tensor below; with the crossings at -2 sigma (2.3 % of the units still
    switch off, so the mask paths of backward stay exercised) the whole backbone is at 2e-6.  Used by the well-conditioned
    full-size goldens (oracle/make_golden_fullsize.py train544wellcond, oracle/make_golden_loop.py)."""
    for key in sorted(sd):
        if key.startswith('backbone') and key.endswith('.bias') and key[:-4] + 'running_mean' in sd:
            sd[key].copy_(k * sd[key[:-4] + 'weight'].abs())
    return sd


def randomize_bias_(sd, seed=11):
    """Conv biases are zero-initialised by the reference; make them non-zero so the bias path is exercised."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd):
        if k.endswith('.bias') and sd[k].dim() == 1 and '.bn' not in k and 'downsample.1' not in k:
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.05)
    return sd


# ------------------------------------------------------------------------------------------------
# training: forward with batch-statistics BN + the YOLACT loss (modules/yolact.py:160-313)
# ------------------------------------------------------------------------------------------------
class TrainNet:
    """Functional train-mode forward over a dict of leaf tensors (requires_grad as set by the caller).
    BatchNorm uses batch statistics and updates `running_*` in place (momentum 0.1), like nn.BatchNorm2d."""

    def __init__(self, params):
        self.p = params

    def conv(self, x, name, stride=1, padding=0):
        return F.conv2d(x, self.p[name + '.weight'], self.p.get(name + '.bias'), stride=stride, padding=padding)

    def bn(self, x, name):
        return F.batch_norm(x, self.p[name + '.running_mean'], self.p[name + '.running_var'],
                            self.p[name + '.weight'], self.p[name + '.bias'], True, 0.1, 1e-5)

    def bottleneck(self, x, p, stride):
        y = F.relu(self.bn(self.conv(x, p + '.conv1'), p + '.bn1'))
        y = F.relu(self.bn(self.conv(y, p + '.conv2', stride, 1), p + '.bn2'))
        y = self.bn(self.conv(y, p + '.conv3'), p + '.bn3')
        if (p + '.downsample.0.weight') in self.p:
            x = self.bn(self.conv(x, p + '.downsample.0', stride), p + '.downsample.1')
        return F.relu(y + x)

    def forward(self, img):
        P = self.p
        layers = resnet_layers_from_sd(P)
        x = F.relu(self.bn(self.conv(img, 'backbone.conv1', 2, 3), 'backbone.bn1'))
        x = F.max_pool2d(x, 3, 2, 1)
        outs = []
        for li, nblk in enumerate(layers):
            for bi in range(nblk):
                x = self.bottleneck(x, f'backbone.layers.{li}.{bi}', 2 if (bi == 0 and li > 0) else 1)
            outs.append(x)
        levels = fpn(outs[1], outs[2], outs[3], P)
        proto = protonet(levels[0], P).permute(0, 2, 3, 1).contiguous()
        nc = P['prediction_layers.conf_layer.weight'].shape[0] // 3
        confs, boxes, coefs = zip(*(head(lv, P, nc) for lv in levels))
        seg = F.conv2d(levels[0], P['semantic_seg_conv.weight'], P['semantic_seg_conv.bias'])
        return torch.cat(confs, 1), torch.cat(boxes, 1), torch.cat(coefs, 1), proto, seg


def encode_offsets(matched, priors):
    """utils/box_utils.py:104-114."""
    g_cxcy = ((matched[:, :2] + matched[:, 2:]) / 2 - priors[:, :2]) / (0.1 * priors[:, 2:])
    g_wh = torch.log((matched[:, 2:] - matched[:, :2]) / priors[:, 2:]) / 0.2
    return torch.cat([g_cxcy, g_wh], 1)


def match_anchors(box_gt, anchors, class_gt, pos_thre=0.5, neg_thre=0.4):
    """utils/box_utils.py:57-83. box_gt [g,4] corners, anchors [N,4] centre-size, class_gt [g] int64."""
    corners = torch.cat((anchors[:, :2] - anchors[:, 2:] / 2, anchors[:, :2] + anchors[:, 2:] / 2), 1)
    ov = pairwise_iou(box_gt[None], corners[None])[0]           # [g, N]
    gt_best_anchor = ov.max(1)[1]
    anchor_best, anchor_gt = ov.max(0)
    anchor_best = anchor_best.clone()
    anchor_gt = anchor_gt.clone()
    anchor_best.index_fill_(0, gt_best_anchor, 2)
    for j in range(gt_best_anchor.shape[0]):                    # later gt wins, like the reference loop :72-73
        anchor_gt[gt_best_anchor[j]] = j
    matched = box_gt[anchor_gt]
    conf = class_gt[anchor_gt] + 1
    conf[anchor_best < pos_thre] = -1
    conf[anchor_best < neg_thre] = 0
    return encode_offsets(matched, anchors), conf, matched, anchor_gt


def ohem_class_loss(class_p, conf_gt, pos, conf_alpha=1.0, ratio=3, stable=False):
    """modules/yolact.py:205-232."""
    nc = class_p.shape[-1]
    flat = class_p.reshape(-1, nc)
    mx = flat.max()
    mark = torch.log(torch.sum(torch.exp(flat - mx), 1)) + mx - flat[:, 0]
    mark = mark.reshape(class_p.shape[0], -1).clone()
    mark[pos] = 0
    mark[conf_gt < 0] = 0
    _, idx = _sort_desc(mark, 1, stable)
    _, rank = idx.sort(1)
    num_pos = pos.long().sum(1, keepdim=True)
    num_neg = torch.clamp(ratio * num_pos, max=pos.shape[1] - 1)
    neg = rank < num_neg.expand_as(rank)
    neg[pos] = 0
    neg[conf_gt < 0] = 0
    sel = pos | neg
    return conf_alpha * F.cross_entropy(class_p[sel].reshape(-1, nc), conf_gt[sel], reduction='sum') / num_pos.sum()


def box_reg_loss(box_p, offsets, pos, bbox_alpha=1.5):
    """modules/yolact.py:234-239."""
    return bbox_alpha * F.smooth_l1_loss(box_p[pos, :], offsets[pos, :], reduction='sum') / pos.sum()


def mask_loss(pos, anchor_gt, coef_p, proto_p, mask_gt, anchor_box, mask_alpha=6.125, masks_to_train=100):
    """modules/yolact.py:241-291 (without the random sub-sampling branch: callers keep <= masks_to_train positives)."""
    ph, pw = proto_p.shape[1:3]
    total = 0
    for i in range(coef_p.shape[0]):
        ds = F.interpolate(mask_gt[i].unsqueeze(0), (ph, pw), mode='bilinear', align_corners=False).squeeze(0)
        ds = ds.permute(1, 2, 0).contiguous().gt(0.5).to(proto_p.dtype)
        idx = anchor_gt[i][pos[i]]
        bx = anchor_box[i][pos[i]]
        cf = coef_p[i][pos[i]]
        if idx.shape[0] == 0:
            continue
        assert cf.shape[0] <= masks_to_train, 'oracle does not restate the randperm sub-sampling (yolact.py:261-267)'
        gt = ds[:, :, idx]
        mp = crop(torch.sigmoid(proto_p[i] @ cf.t()), bx)
        l = F.binary_cross_entropy(torch.clamp(mp, 0, 1), gt, reduction='none')
        area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
        total = total + torch.sum(l.sum(dim=(0, 1)) / area)
    return mask_alpha * total / ph / pw / pos.sum()


def semantic_loss(seg_p, mask_gt, class_gt, semantic_alpha=1.0):
    """modules/yolact.py:293-313."""
    b, nc, mh, mw = seg_p.shape
    total = 0
    for i in range(b):
        ds = F.interpolate(mask_gt[i].unsqueeze(0), (mh, mw), mode='bilinear', align_corners=False).squeeze(0)
        ds = ds.gt(0.5).to(seg_p.dtype)
        tgt = torch.zeros_like(seg_p[i])
        for j in range(ds.shape[0]):
            tgt[class_gt[i][j]] = torch.max(tgt[class_gt[i][j]], ds[j])
        total = total + F.binary_cross_entropy_with_logits(seg_p[i], tgt, reduction='sum')
    return semantic_alpha * total / mh / mw / b


def compute_loss(class_p, box_p, coef_p, proto_p, seg_p, box_class, mask_gt, anchors, stable=False):
    """modules/yolact.py:166-203. box_class: list of [g,5] (x1,y1,x2,y2,cls); mask_gt: list of [g,H,W]."""
    b, n = box_p.shape[:2]
    offs = torch.zeros(b, n, 4, dtype=box_p.dtype)
    conf = torch.zeros(b, n, dtype=torch.int64)
    abox = torch.zeros(b, n, 4, dtype=box_p.dtype)
    aidx = torch.zeros(b, n, dtype=torch.int64)
    cls_gt = []
    for i in range(b):
        cls_gt.append(box_class[i][:, -1].long())
        offs[i], conf[i], abox[i], aidx[i] = match_anchors(box_class[i][:, :-1], anchors, cls_gt[i])
    pos = conf > 0
    return (ohem_class_loss(class_p, conf, pos, stable=stable), box_reg_loss(box_p, offs, pos),
            mask_loss(pos, aidx, coef_p, proto_p, mask_gt, abox), semantic_loss(seg_p, mask_gt, cls_gt))


# ------------------------------------------------------------------------------------------------
# Swin-T backbone (modules/swin_transformer.py), eval mode
# ------------------------------------------------------------------------------------------------
def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], 1e-5)


def _lin(x, sd, name):
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def swin_windows(x, ws):
    """window_partition (:99-111): [B,H,W,C] -> [B*nW, ws*ws, C]."""
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).contiguous()
    return x.view(-1, ws * ws, c)


def swin_unwindows(win, ws, h, w):
    """window_reverse (:114-128)."""
    b = win.shape[0] // ((h // ws) * (w // ws))
    x = win.view(b, h // ws, w // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous()
    return x.view(b, h, w, -1)


def swin_shift_mask(hp, wp, ws, shift):
    """BasicLayer.forward :366-383 — region ids -> 0 / -100 additive mask [nW, N, N]."""
    img = torch.zeros(1, hp, wp, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = swin_windows(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_rel_index(ws):
    """WindowAttention.__init__ :152-162."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_attention(xw, sd, p, heads, ws, mask):
    """WindowAttention.forward :172-200."""
    bw, n, c = xw.shape
    qkv = _lin(xw, sd, p + '.qkv').reshape(bw, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (c // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + '.relative_position_bias_table'][sd[p + '.relative_position_index'].view(-1)]
    attn = attn + bias.view(n, n, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(bw // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    attn = F.softmax(attn, -1)
    return _lin((attn @ v).transpose(1, 2).reshape(bw, n, c), sd, p + '.proj')


def swin_block(x, h, w, sd, p, heads, ws, shift, mask):
    """SwinTransformerBlock.forward :234-289 (eval: DropPath is the identity)."""
    b, l, c = x.shape
    short = x
    y = _ln(x, sd, p + '.norm1').view(b, h, w, c)
    pr, pb = (ws - w % ws) % ws, (ws - h % ws) % ws
    y = F.pad(y, (0, 0, 0, pr, 0, pb))
    hp, wp = y.shape[1:3]
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    a = swin_attention(swin_windows(y, ws), sd, p + '.attn', heads, ws, mask if shift > 0 else None)
    y = swin_unwindows(a, ws, hp, wp)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y[:, :h, :w, :].contiguous().view(b, h * w, c)
    x = short + y
    m = _lin(F.gelu(_lin(_ln(x, sd, p + '.norm2'), sd, p + '.mlp.fc1')), sd, p + '.mlp.fc2')
    return x + m


def swin_merge(x, h, w, sd, p):
    """PatchMerging.forward :299-325."""
    b, l, c = x.shape
    x = x.view(b, h, w, c)
    if h % 2 or w % 2:
        x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(b, -1, 4 * c)
    return _lin(_ln(x, sd, p + '.norm'), sd, p + '.reduction')


def swin_backbone(img, sd, p='backbone', depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), ws=7):
    """SwinTransformer.forward :500-518 — returns the 4 stage maps NCHW (stages 1-3 layer-normed)."""
    b, _, hh, ww = img.shape
    if ww % 4:
        img = F.pad(img, (0, 4 - ww % 4))
    if hh % 4:
        img = F.pad(img, (0, 0, 0, 4 - hh % 4))
    x = F.conv2d(img, sd[p + '.patch_embed.proj.weight'], sd[p + '.patch_embed.proj.bias'], stride=4)
    h, w = x.shape[2:]
    x = _ln(x.flatten(2).transpose(1, 2), sd, p + '.patch_embed.norm')
    outs = []
    for li, depth in enumerate(depths):
        hp, wp = math.ceil(h / ws) * ws, math.ceil(w / ws) * ws
        mask = swin_shift_mask(hp, wp, ws, ws // 2)
        for bi in range(depth):
            x = swin_block(x, h, w, sd, f'{p}.layers.{li}.blocks.{bi}', heads[li], ws, 0 if bi % 2 == 0 else ws // 2, mask)
        xo = x
        if li in (1, 2, 3):
            xo = _ln(xo, sd, f'{p}.norm{li}')
        outs.append(xo.view(b, h, w, -1).permute(0, 3, 1, 2).contiguous())
        if li < len(depths) - 1:
            x = swin_merge(x, h, w, sd, f'{p}.layers.{li}.downsample')
            h, w = (h + 1) // 2, (w + 1) // 2
    return outs


def features_any(img, sd):
    """Backbone-agnostic version of `features` (ResNet or Swin-T, chosen from the state-dict keys)."""
    if 'backbone.patch_embed.proj.weight' in sd:
        outs = swin_backbone(img, sd)
    else:
        outs = resnet(img, sd, resnet_layers_from_sd(sd))
    levels = fpn(outs[1], outs[2], outs[3], sd)
    proto = protonet(levels[0], sd).permute(0, 2, 3, 1).contiguous()
    num_classes = sd['prediction_layers.conf_layer.weight'].shape[0] // 3
    confs, boxes, coefs = zip(*(head(lv, sd, num_classes) for lv in levels))
    return torch.cat(confs, 1), torch.cat(boxes, 1), torch.cat(coefs, 1), proto


def forward_train_any(img, sd):
    """Train-branch outputs of Yolact.forward (modules/yolact.py:141-161) for a backbone WITHOUT batch-norm state, i.e. Swin-T
    with DropPath inactive (rate 0): LayerNorm / Linear / attention behave identically in train and eval mode, so the functional
    eval restatement is also the train-mode one and autograd through it gives the reference gradients.
    Returns (class logits, box, coef, proto, semantic-seg logits)."""
    assert 'backbone.patch_embed.proj.weight' in sd, 'ResNet training goes through TrainNet (batch-statistics BatchNorm)'
    outs = swin_backbone(img, sd)
    levels = fpn(outs[1], outs[2], outs[3], sd)
    proto = protonet(levels[0], sd).permute(0, 2, 3, 1).contiguous()
    num_classes = sd['prediction_layers.conf_layer.weight'].shape[0] // 3
    confs, boxes, coefs = zip(*(head(lv, sd, num_classes) for lv in levels))
    seg = F.conv2d(levels[0], sd['semantic_seg_conv.weight'], sd['semantic_seg_conv.bias'])
    return torch.cat(confs, 1), torch.cat(boxes, 1), torch.cat(coefs, 1), proto, seg


def forward_eval_any(img, sd):
    conf, box, coef, proto = features_any(img, sd)
    return F.softmax(conf, -1), box, coef, proto


# ------------------------------------------------------------------------------------------------
# pre-processing (utils/augmentations.py:219-227)
# ------------------------------------------------------------------------------------------------
def val_aug(img_hwc_bgr, val_size, mean=(103.94, 116.78, 123.68), std=(57.38, 57.12, 58.40)):
    """pad_to_square (:138-165, fill = norm_mean, image at the top-left) -> cv2.resize to val_size (restated as bilinear
    align_corners=False: cv2 is not importable here — PARITY UNPINNED by cv2) -> (x-mean)/std -> BGR->RGB -> CHW."""
    img = img_hwc_bgr.float()
    h, w, _ = img.shape
    mean_t, std_t = torch.tensor(mean), torch.tensor(std)
    if h != w:
        p = max(h, w)
        pad = mean_t.view(1, 1, 3).expand(p, p, 3).clone()
        pad[:h, :w] = img
        img = pad
    x = img.permute(2, 0, 1).unsqueeze(0)
    x = F.interpolate(x, (val_size, val_size), mode='bilinear', align_corners=False).squeeze(0)
    x = (x - mean_t.view(3, 1, 1)) / std_t.view(3, 1, 1)
    return x[[2, 1, 0]].contiguous()
