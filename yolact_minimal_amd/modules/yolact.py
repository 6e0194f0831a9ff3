"""`Yolact` with the reference's constructor / forward / state-dict surface, executed by HIP kernels.

Reference: `/root/reference/modules/yolact.py` — PredictionModule `:12-31`, ProtoNet `:34-53`,
FPN `:56-89`, Yolact `:92-164` (ctor `:93-125`, load_weights `:127-139`, forward `:141-164`).

What is kept identical:
  * `Yolact(cfg)`; backbone chosen from the cfg *class name*; `net.anchors` (flat python list);
  * every parameter / buffer name and OIHW fp32 shape, created in the reference's order, so a seeded
    construction is bit-identical and published `.pth` files load with `strict=True`;
  * `forward(img, box_classes=None, masks_gt=None)`; eval returns
    `(class_pred[B,N,C] softmaxed, box_pred[B,N,4], coef_pred[B,N,32], proto_out[B,Hp,Wp,32])`.

What is different: no layer here computes anything in PyTorch.  `forward` hands the image to
`yolact_minimal_amd.engine.InferEngine`, which runs NHWC fp32 implicit-GEMM convolutions on the
f32 MFMA pipe with BN/bias/ReLU/residual/tanh fused in the epilogue, and writes the five head
levels straight into the concatenated `[B, N, C]` outputs.  There is no eager/CPU fallback: a CPU
tensor or a missing `libyolact_hip.so` raises.
"""
import math

import torch
import torch.nn as nn

from .resnet import ResNet
from ..utils.box_utils import make_anchors


def _conv_relu(cin, cout, k, stride=1, padding=0):
    # nn.ReLU is kept only so that state-dict indices match (`pred_layers.N.0`, `proto1.{0,2,4}`).
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=padding), nn.ReLU(inplace=True))


class PredictionModule(nn.Module):
    """Shared head: upfeature 3x3+ReLU then bbox / conf / coef(tanh) 3x3 (reference :13-24)."""

    def __init__(self, cfg, coef_dim=32):
        super().__init__()
        self.num_classes = cfg.num_classes
        self.coef_dim = coef_dim
        na = len(cfg.aspect_ratios)
        self.upfeature = _conv_relu(256, 256, 3, padding=1)
        self.bbox_layer = nn.Conv2d(256, na * 4, 3, padding=1)
        self.conf_layer = nn.Conv2d(256, na * self.num_classes, 3, padding=1)
        self.coef_layer = nn.Sequential(nn.Conv2d(256, na * coef_dim, 3, padding=1), nn.Tanh())


class ProtoNet(nn.Module):
    """3x(3x3+ReLU) -> bilinear x2 (align_corners=True) -> 3x3+ReLU -> 1x1+ReLU (reference :35-47)."""

    def __init__(self, coef_dim):
        super().__init__()
        seq = []
        for _ in range(3):
            seq += list(_conv_relu(256, 256, 3, padding=1))
        self.proto1 = nn.Sequential(*seq)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.proto2 = nn.Sequential(*_conv_relu(256, 256, 3, padding=1), *_conv_relu(256, coef_dim, 1))


class FPN(nn.Module):
    """Lateral 1x1, top-down bilinear x2 (align_corners=False) add, 3x3 pred, 2 stride-2 extras (:57-71)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.lat_layers = nn.ModuleList(nn.Conv2d(c, 256, 1) for c in in_channels)
        self.pred_layers = nn.ModuleList(_conv_relu(256, 256, 3, padding=1) for _ in in_channels)
        self.downsample_layers = nn.ModuleList(_conv_relu(256, 256, 3, stride=2, padding=1)
                                               for _ in range(2))


class Yolact(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.coef_dim = 32

        name = cfg.__class__.__name__
        if name.startswith('res101'):
            self.backbone = ResNet(layers=(3, 4, 23, 3))
            self.fpn = FPN(in_channels=(512, 1024, 2048))
        elif name.startswith('res50'):
            self.backbone = ResNet(layers=(3, 4, 6, 3))
            self.fpn = FPN(in_channels=(512, 1024, 2048))
        elif name.startswith('swin_tiny'):
            from .swin_transformer import SwinTransformer
            self.backbone = SwinTransformer()
            self.fpn = FPN(in_channels=(192, 384, 768))
        else:
            raise ValueError(f'cannot derive a backbone from cfg class {name!r}')

        self.proto_net = ProtoNet(coef_dim=self.coef_dim)
        self.prediction_layers = PredictionModule(cfg, coef_dim=self.coef_dim)

        # anchors: flat python list [cx, cy, w, h] * N, level-major, as the reference keeps them
        self.fpn_fm_shape = [math.ceil(cfg.img_size / stride) for stride in (8, 16, 32, 64, 128)]
        self.anchors = []
        for size, scale in zip(self.fpn_fm_shape, cfg.scales):
            self.anchors += make_anchors(cfg, size, size, scale)

        if cfg.mode == 'train':
            self.semantic_seg_conv = nn.Conv2d(256, cfg.num_classes - 1, kernel_size=1)

        for module in self.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.xavier_uniform_(module.weight.data)
                if module.bias is not None:
                    module.bias.data.zero_()

        self._engines = {}
        self._weights_epoch = 0
        self._train_state = None          # train_state.ModuleTrainState, created at the first train-mode forward on the GPU
        self._ddp_wrapped = False
        self._ddp_wrapper = None

    # ---- torch.nn.parallel.DistributedDataParallel around this module (reference train.py:76) ------------------------------------
    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """The attribute torch's DDP constructor reads to learn which tensors of a module it must NOT manage.  This module reduces
        its own gradients (flat buffer, >= 25 MB buckets all-reduced on the RCCL stream during backward) and broadcasts its own
        BatchNorm statistics (one message): `train_state.ModuleTrainState`.  Reading the attribute is also how the module learns that
        it has been wrapped — without a wrapper it never touches a process group.  With YM_AUTO_FLAT=0 the attribute does not exist
        and torch's DDP manages everything itself."""
        from ..train_state import AUTO, DDP_KEEPS
        if not AUTO or self.cfg.mode != 'train':
            raise AttributeError('_ddp_params_and_buffers_to_ignore')
        self._ddp_wrapped = True
        # the wrapper itself (the caller of this property is its constructor), kept weakly: `DDP.no_sync()` works by clearing
        # `require_backward_grad_sync` on it, and the module's reducer has to honour that (train_state.ModuleTrainState.sync_wanted)
        try:
            import inspect
            import weakref
            caller = inspect.currentframe().f_back.f_locals.get('self')
            # (DDP.__init__ reads this before it sets `self.module`; anything else that merely probes the attribute is not a wrapper)
            if caller is not None and type(caller).__name__ == 'DistributedDataParallel':
                self._ddp_wrapper = weakref.ref(caller)
        except Exception:
            pass
        return [n for n, _ in self.named_parameters() if n != DDP_KEEPS] + [n for n, _ in self.named_buffers()]

    def _drop_train_state(self):
        if self._train_state is not None:
            self._train_state.release()
            self._train_state = None

    # ---- weights -----------------------------------------------------------------------------
    def mark_weights_changed(self):
        """Packed HBM weight images are rebuilt on the next forward."""
        self._weights_epoch += 1

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.mark_weights_changed()
        return out

    def _apply(self, fn, *args, **kwargs):
        if getattr(self, '_train_state', None) is not None:
            self._drop_train_state()             # parameters and buffers are about to be re-allocated
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, '_engines'):
            self._engines.clear()
            self.mark_weights_changed()
        return out

    def load_weights(self, weight, cuda):
        """Same contract as reference `:127-139`: strict load, train-only keys dropped in eval."""
        state_dict = torch.load(weight) if cuda else torch.load(weight, map_location='cpu')
        if self.cfg.mode != 'train':
            for key in [k for k in state_dict if k.startswith('semantic_seg_conv')]:
                del state_dict[key]
        self.load_state_dict(state_dict, strict=True)
        print(f'Model loaded with {weight}.\n')
        print(f'Number of all parameters: {sum(p.numel() for p in self.parameters())}\n')

    def _own_train_state(self, device):
        """The module's own training plumbing (flat gradient slots, side stream, gradient reducer: train_state.py) unless a
        `Trainer` owns the parameters (its FlatSGD installed the slots) or gradients are off."""
        from ..train_state import AUTO, ModuleTrainState
        st = self._train_state
        if st is not None:
            return st
        if not AUTO or not torch.is_grad_enabled():
            return None
        first = next(self.parameters())
        if getattr(first, '_ym_grad_slot', None) is not None:
            return None                          # trainer-owned
        st = self._train_state = ModuleTrainState(self, device)
        return st

    # ---- forward -----------------------------------------------------------------------------
    CONV_MODES = {'f32': 0, 'bf16x3': 3, 'bf16x6': 6}

    def set_conv_mode(self, mode):
        """Matrix pipe of the inference convolutions (not part of the reference's surface; `ym_conv_desc.mma`): 'f32' = exact
        fp32 products on the f32 MFMA (default, the parity mode); 'bf16x3' / 'bf16x6' = fp32 operands split into 2 / 3 bf16 terms
        on the bf16 MFMA with fp32 accumulation (bf16x3: ~1.6x the bs=8 throughput, within 1e-4 of the reference on its 544 px
        goldens; bf16x6: fp32-grade).  Tensors, weights and the state dict stay fp32."""
        if mode not in self.CONV_MODES:
            raise ValueError(f'conv mode must be one of {sorted(self.CONV_MODES)}, got {mode!r}')
        self._conv_mma = self.CONV_MODES[mode]
        for eng in self._engines.values():
            eng.set_mma(self._conv_mma)

    PLAN_MODES = ('latency', 'throughput')

    def set_plan_mode(self, mode):
        """Which rows of the tuned table the engines behind `forward` read (not part of the reference's surface): 'latency' (default:
        one request at a time has the chip to itself) or 'throughput' (`<shape>_tp` rows first: the plan the slots of a
        `pipeline.RequestPipeline` with several requests in flight run, i.e. the plan behind bench.py's `value`).  Same kernels, same
        arithmetic per launch; tile / K-split choices differ, so results agree to fp32 summation order."""
        if mode not in self.PLAN_MODES:
            raise ValueError(f'plan mode must be one of {self.PLAN_MODES}, got {mode!r}')
        if mode != getattr(self, '_plan_mode', 'latency'):
            self._plan_mode = mode
            self._engines.clear()

    def _engine(self, img):
        from ..engine import InferEngine
        key = (img.device.index, img.shape[0], img.shape[2], img.shape[3])
        eng = self._engines.get(key)
        if eng is None:
            eng = InferEngine(self, batch=img.shape[0], height=img.shape[2], width=img.shape[3],
                              device=img.device, mode=getattr(self, '_plan_mode', 'latency'))
            if getattr(self, '_conv_mma', None) is not None:
                eng.set_mma(self._conv_mma)
            self._engines[key] = eng
        return eng

    def forward(self, img, box_classes=None, masks_gt=None):
        if not img.is_cuda:
            raise RuntimeError('yolact_minimal_amd.Yolact runs on an MI355X only: got a CPU tensor and '
                               'there is no CPU fallback (the CPU restatement lives in oracle/ and is '
                               'test infrastructure).')
        if self.training:
            # train branch of the reference forward (:158-161): head logits + semantic-seg conv + compute_loss
            from ..train_engine import train_features, weights_changed
            from ..loss import compute_loss
            if isinstance(self.anchors, list):
                self.anchors = torch.tensor(self.anchors, device=img.device).reshape(-1, 4)    # like reference :171-172
            state = self._own_train_state(img.device)
            if state is not None:
                weights_changed()                # a torch optimizer stepped since the last forward: one batched re-pack
                state.begin_forward()
                wrapper = self._ddp_wrapper() if self._ddp_wrapper is not None else None
                state.sync_before_forward(self._ddp_wrapped, getattr(wrapper, 'require_backward_grad_sync', True))
            class_p, box_p, coef_p, proto_p, seg_p = train_features(self, img)
            if state is not None:
                state.after_forward()
            self.mark_weights_changed()          # the optimizer is about to change them; eval engines must repack
            return compute_loss(self.cfg, self.anchors, class_p, box_p, coef_p, proto_p, seg_p, box_classes, masks_gt)
        return self._engine(img).forward(img)
