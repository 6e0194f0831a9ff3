"""Configuration surface of the YOLACT hot path (host side, pure Python).

Mirrors the attribute names and the `get_config(args, mode)` entry point of the
reference (`/root/reference/config.py:70-137` for the cfg attributes,
`config.py:222-253` for `get_config`) so that callers written against the
reference (`eval.py`, `detect.py`, `train.py`) find the same fields.  This
module is not a kernel: it only carries numbers the kernels are launched with.

Differences from the reference, on purpose:
  * no directories are created at import time (`config.py:6-15` does that);
  * the process group is initialised with the RCCL backend through
    `yolact_minimal_amd.dist_utils` and reads LOCAL_RANK from the environment
    (torchrun) as well as `args.local_rank` (torch.distributed.launch).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

COCO_CLASSES = (
    'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat',
    'traffic light', 'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog',
    'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe', 'backpack', 'umbrella',
    'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite',
    'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
    'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange',
    'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant',
    'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone',
    'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors',
    'teddy bear', 'hair drier', 'toothbrush')

PASCAL_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair',
                  'cow', 'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant',
                  'sheep', 'sofa', 'train', 'tvmonitor')

CUSTOM_CLASSES = ('dog', 'person', 'bear', 'sheep')

# COCO category ids are 1..90 with ten holes; the map sends them to 1..80.
_COCO_ID_HOLES = (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)
COCO_LABEL_MAP = {cid: k + 1 for k, cid in
                  enumerate(c for c in range(1, 91) if c not in _COCO_ID_HOLES)}

# BGR mean / std used by the reference pre-processing (config.py:66-67).
def _palette(n=81):
    """Drawing colours for `draw_img` (host-side visualisation, outside the hot path): index 0 = background black, then n-1 well
    separated BGR colours from a golden-angle walk round the hue circle.  Own palette — the values do not influence any result."""
    import colorsys
    cols = [[0, 0, 0]]
    for k in range(n - 1):
        r, g, b = colorsys.hsv_to_rgb((k * 0.61803398875) % 1.0, 0.65 + 0.3 * ((k % 3) / 2), 0.95 - 0.25 * ((k % 4) / 3))
        cols.append([int(b * 255), int(g * 255), int(r * 255)])
    return np.array(cols)


COLORS = _palette()

norm_mean = np.array([103.94, 116.78, 123.68], dtype=np.float32)
norm_std = np.array([57.38, 57.12, 58.40], dtype=np.float32)

_COCO_SCALES = (24, 48, 96, 192, 384)
_PASCAL_SCALES = (32, 64, 128, 256, 512)


class res101_coco:
    """cfg object; the class *name* selects the backbone (modules/yolact.py:98-106)."""
    _backbone_file = 'weights/backbone_res101.pth'

    def __init__(self, args):
        self.mode = args.mode
        self.cuda = args.cuda
        self.gpu_id = args.gpu_id
        if args.img_size % 32 != 0:
            raise AssertionError(f'Img_size must be divisible by 32, got {args.img_size}.')
        self.img_size = args.img_size
        self.class_names = COCO_CLASSES
        self.num_classes = len(self.class_names) + 1
        self.continuous_id = COCO_LABEL_MAP
        self.scales = [int(self.img_size / 544 * s) for s in _COCO_SCALES]
        self.aspect_ratios = [1, 1 / 2, 2]
        self.data_root = '/home/feiyu/Data/'

        training = self.mode == 'train'
        if training:
            self.weight = args.resume if args.resume else self._backbone_file
            self._init_train(args)
        else:
            self.weight = getattr(args, 'weight', None)

        if self.mode in ('train', 'val'):
            self.val_imgs = self.data_root + 'coco2017/val2017/'
            self.val_ann = self.data_root + 'coco2017/annotations/instances_val2017.json'
            self.val_bs = 1
            self.val_num = getattr(args, 'val_num', -1)
            self.coco_api = getattr(args, 'coco_api', False)

        # post-processing knobs read by utils.output_utils.nms / fast_nms
        self.traditional_nms = getattr(args, 'traditional_nms', False)
        self.nms_score_thre = 0.05
        self.nms_iou_thre = 0.5
        self.top_k = 200
        self.max_detections = 100

        if self.mode == 'detect':
            for key, value in vars(args).items():
                setattr(self, key, value)

    def _init_train(self, args):
        self.train_imgs = self.data_root + 'coco2017/train2017/'
        self.train_ann = self.data_root + 'coco2017/annotations/instances_train2017.json'
        self.train_bs = args.train_bs
        self.bs_per_gpu = args.bs_per_gpu
        self.val_interval = getattr(args, 'val_interval', 4000)

        self.bs_factor = self.train_bs / 8
        self.lr = 0.001 * self.bs_factor
        self.warmup_init = self.lr * 0.1
        self.warmup_until = 500
        self.lr_steps = tuple(int(s / self.bs_factor) for s in (0, 280000, 560000, 620000, 680000))

        self.pos_iou_thre = 0.5
        self.neg_iou_thre = 0.4
        self.conf_alpha = 1
        self.bbox_alpha = 1.5
        self.mask_alpha = 6.125
        self.semantic_alpha = 1
        self.masks_to_train = 100

    def print_cfg(self):
        print()
        print('-' * 30 + self.__class__.__name__ + '-' * 30)
        for key, value in vars(self).items():
            if key not in ('continuous_id', 'data_root', 'cfg'):
                print(f'{key}: {value}')
        print()


class res50_coco(res101_coco):
    _backbone_file = 'weights/backbone_res50.pth'


class swin_tiny_coco(res101_coco):
    _backbone_file = 'weights/swin_tiny.pth'

    def _init_train(self, args):
        super()._init_train(args)
        self.lr = 0.00005 * self.bs_factor


class _ContiguousIds:
    def _set_classes(self, names):
        self.class_names = names
        self.num_classes = len(names) + 1
        self.continuous_id = {k + 1: k + 1 for k in range(len(names))}


class res50_pascal(res101_coco, _ContiguousIds):
    _backbone_file = 'weights/backbone_res50.pth'

    def __init__(self, args):
        super().__init__(args)
        self._set_classes(PASCAL_CLASSES)
        self.use_square_anchors = False
        if self.mode == 'train':
            self.train_imgs = self.data_root + 'pascal_sbd/img'
            self.train_ann = self.data_root + 'pascal_sbd/pascal_sbd_train.json'
            self.lr_steps = tuple(int(s / self.bs_factor) for s in (0, 60000, 100000, 120000))
            self.scales = [int(self.img_size / 544 * s) for s in _PASCAL_SCALES]
        if self.mode in ('train', 'val'):
            self.val_imgs = self.data_root + 'pascal_sbd/img'
            self.val_ann = self.data_root + 'pascal_sbd/pascal_sbd_val.json'


class res101_custom(res101_coco, _ContiguousIds):
    def __init__(self, args):
        super().__init__(args)
        self._set_classes(CUSTOM_CLASSES)
        if self.mode == 'train':
            self.train_imgs = 'custom_dataset/'
            self.train_ann = 'custom_dataset/custom_ann.json'
            self.warmup_until = 100
            self.lr_steps = (0, 1200, 1600, 2000)
        if self.mode in ('train', 'val'):
            self.val_imgs = ''
            self.val_ann = ''


class res50_custom(res101_custom):
    _backbone_file = 'weights/backbone_res50.pth'


def _local_rank(args):
    lr = getattr(args, 'local_rank', None)
    if lr is None:
        lr = int(os.environ.get('LOCAL_RANK', 0))
    return lr


def get_config(args, mode):
    """Same contract as the reference `get_config` (config.py:222-253).

    With a visible GPU and mode == 'train' this joins the RCCL process group
    (backend name 'nccl' is RCCL on ROCm) exactly where the reference joins NCCL.
    """
    args.cuda = torch.cuda.is_available()
    args.mode = mode

    if args.cuda:
        visible = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES')
        args.gpu_id = visible if visible else '0'
        if mode == 'train':
            torch.cuda.set_device(_local_rank(args))
            if not dist.is_initialized():
                dist.init_process_group(backend='nccl', init_method='env://')
            world = int(os.environ['WORLD_SIZE'])
            if args.train_bs % world != 0:
                raise AssertionError('Total training batch size must be divisible by GPU number.')
            args.bs_per_gpu = args.train_bs // world
        elif not args.gpu_id.isdigit():
            raise AssertionError(f'Only one GPU can be used in val/detect mode, got {args.gpu_id}.')
    else:
        args.gpu_id = None
        if mode == 'train':
            args.bs_per_gpu = args.train_bs
            print('\n-----No GPU found, training on CPU.-----')
        else:
            print('\n-----No GPU found, validate on CPU.-----')

    cfg = globals()[args.cfg](args)

    if (not args.cuda) or mode != 'train' or dist.get_rank() == 0:
        cfg.print_cfg()
    return cfg


def make_args(cfg='res101_coco', img_size=544, **kw):
    """Small helper for tests/bench: an argparse-like namespace with the reference's flag names."""
    import argparse
    ns = argparse.Namespace(cfg=cfg, img_size=img_size, weight=None, resume=None, train_bs=8,
                            bs_per_gpu=8, val_interval=4000, val_num=-1, coco_api=False,
                            traditional_nms=False, local_rank=None)
    for key, value in kw.items():
        setattr(ns, key, value)
    return ns


def build_cfg(name='res101_coco', mode='val', img_size=544, **kw):
    """Construct a cfg without touching torch.distributed (tests, bench, oracle)."""
    ns = make_args(cfg=name, img_size=img_size, **kw)
    ns.mode = mode
    ns.cuda = torch.cuda.is_available()
    ns.gpu_id = '0' if ns.cuda else None
    return globals()[name](ns)
