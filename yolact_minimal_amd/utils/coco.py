"""The COCO reader feeding the hot path.

Reference: `/root/reference/utils/coco.py` — `COCODetection :47-134`, `train_collate :14-28`, `val_collate :31-35`,
`detect_collate :38-40`; it leans on `pycocotools.coco.COCO` (annotation index, `annToMask`) and `cv2.imread`, neither of which
exists in the build image.

What is rebuilt and how:

* `COCO` — the annotation index (`imgToAnns`, `getAnnIds`, `loadAnns`, `loadImgs`, `annToMask`), same method names; plain
  JSON bookkeeping on the host, like pycocotools' own Python class.
* `anns_to_masks` / `COCO.annToMask` — polygon / RLE annotation -> dense uint8 mask **on the device** (`ym_poly_to_mask`,
  `ym_runs_to_mask`): the reference rasterises every instance on the CPU and later ships n x H x W floats over PCIe; here
  only the vertices cross and the masks are born in HBM, where `train_aug` / `val_aug` / the loss kernels consume them.
* image decode — PIL (libjpeg) instead of `cv2.imread`; BGR channel order and EXIF orientation handled like cv2.  Host I/O,
  "parity unpinned" (no cv2 here to compare with).
* `COCODetection.__getitem__` — same three modes and return shapes as the reference, with CUDA tensors for the image and the
  masks (boxes / labels stay small host arrays).  Because samples are produced on the GPU there are no DataLoader worker
  processes: `BatchLoader` shards indices per rank like `DistributedSampler`, decodes JPEGs on a small thread pool (PIL releases
  the GIL) and collates with the reference's rules (`train_collate`: a rejected sample is replaced by a repeated valid one).
"""
import glob
import json
import os.path as osp
import random as _random
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

import ctypes

import numpy as np
import torch

from .. import hip
from .augmentations import train_aug, val_aug


def _rle_string_to_counts(s):
    """cocoapi rleFrString: 5 data bits per character (+48), bit 5 = continuation, sign-extended, delta-coded from the third on."""
    if isinstance(s, bytes):
        s = s.decode('ascii')
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def anns_to_masks(segmentations, height, width, device='cuda'):
    """[n] COCO `segmentation` entries (polygon lists or RLE dicts) of one image -> uint8 CUDA tensor [n, height, width]
    (what `np.stack([coco.annToMask(a) for a in anns])` is in the reference, utils/coco.py:96,102)."""
    n = len(segmentations)
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError('yolact_minimal_amd.utils.coco rasterises masks with HIP kernels: a CUDA/HIP device is required')
    masks = torch.empty(n, height, width, dtype=torch.uint8, device=dev)
    if n == 0:
        return masks
    poly_idx = [i for i, s in enumerate(segmentations) if isinstance(s, list)]
    run_idx = [i for i in range(n) if i not in set(poly_idx)]
    L = hip.lib()
    with torch.cuda.device(dev):
        def _ws(count):
            nbytes = L.ym_ann_to_mask_workspace_bytes(count, height, width)
            return torch.empty(max(nbytes, 4) // 4 + 1, dtype=torch.int32, device=dev), nbytes

        def _dev(arr, dtype):
            return torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(dev)

        for idx, is_poly in ((poly_idx, True), (run_idx, False)):
            if not idx:
                continue
            whole = len(idx) == n
            out = masks if whole else torch.empty(len(idx), height, width, dtype=torch.uint8, device=dev)
            ws, ws_bytes = _ws(len(idx))
            if is_poly:
                xy, poly_off, ann_off = [], [0], [0]
                for i in idx:
                    for poly in segmentations[i]:
                        k = len(poly) // 2                       # frPyObjects: k = len / 2, a trailing odd value is ignored
                        xy.extend(poly[:2 * k])
                        poly_off.append(poly_off[-1] + k)
                    ann_off.append(len(poly_off) - 1)
                d_xy = _dev(xy if xy else [0.0], np.float64)
                d_po, d_ao = _dev(poly_off, np.int32), _dev(ann_off, np.int32)
                hip.check(L.ym_poly_to_mask(ctypes.c_void_p(d_xy.data_ptr()), ctypes.c_void_p(d_po.data_ptr()),
                                            ctypes.c_void_p(d_ao.data_ptr()), len(idx), height, width, ctypes.c_void_p(out.data_ptr()),
                                            ctypes.c_void_p(ws.data_ptr()), ws_bytes, hip.stream_ptr()), 'ym_poly_to_mask')
            else:
                counts, run_off = [], [0]
                for i in idx:
                    seg = segmentations[i]
                    c = seg['counts']
                    c = _rle_string_to_counts(c) if isinstance(c, (str, bytes)) else list(c)
                    if list(seg.get('size', [height, width])) != [height, width]:
                        raise ValueError(f"RLE annotation of size {seg.get('size')} in a {height} x {width} image")
                    counts.extend(c)
                    run_off.append(len(counts))
                d_c, d_ro = _dev(counts if counts else [0], np.uint32), _dev(run_off, np.int32)
                hip.check(L.ym_runs_to_mask(ctypes.c_void_p(d_c.data_ptr()), ctypes.c_void_p(d_ro.data_ptr()), len(idx), height, width,
                                            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes, hip.stream_ptr()),
                          'ym_runs_to_mask')
            if not whole:
                masks[torch.as_tensor(idx, device=dev)] = out
    return masks


class COCO:
    """The part of `pycocotools.coco.COCO` the reader uses, same attribute / method names."""

    def __init__(self, annotation_file=None, device='cuda'):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        self.device = device
        if annotation_file is not None:
            with open(annotation_file, 'r') as f:
                dataset = json.load(f)
            assert type(dataset) == dict, f'annotation file format {type(dataset)} not supported'
            self.dataset = dataset
            self.createIndex()

    def createIndex(self):
        for ann in self.dataset.get('annotations', []):
            self.imgToAnns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
        for img in self.dataset.get('images', []):
            self.imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            self.cats[cat['id']] = cat
        if 'annotations' in self.dataset and 'categories' in self.dataset:
            for ann in self.dataset['annotations']:
                self.catToImgs[ann['category_id']].append(ann['image_id'])

    @staticmethod
    def _as_list(v):
        return v if isinstance(v, (list, tuple)) else [v]

    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
        imgIds, catIds = self._as_list(imgIds), self._as_list(catIds)
        if len(imgIds) == len(catIds) == len(areaRng) == 0:
            anns = self.dataset['annotations']
        else:
            if len(imgIds) > 0:
                anns = [a for i in imgIds if i in self.imgToAnns for a in self.imgToAnns[i]]
            else:
                anns = self.dataset['annotations']
            if len(catIds) > 0:
                anns = [a for a in anns if a['category_id'] in catIds]
            if len(areaRng) > 0:
                anns = [a for a in anns if areaRng[0] < a['area'] < areaRng[1]]
        if iscrowd is not None:
            return [a['id'] for a in anns if a['iscrowd'] == iscrowd]
        return [a['id'] for a in anns]

    def loadAnns(self, ids=[]):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadImgs(self, ids=[]):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]

    def annToMask(self, ann):
        """Dense uint8 mask [h, w] of one annotation, on the device (h, w from the image record, like annToRLE)."""
        rec = self.imgs[ann['image_id']]
        return anns_to_masks([ann['segmentation']], rec['height'], rec['width'], self.device)[0]

    def annsToMasks(self, anns):
        """All the annotations of ONE image in one launch -> [n, h, w] uint8 CUDA tensor."""
        rec = self.imgs[anns[0]['image_id']]
        return anns_to_masks([a['segmentation'] for a in anns], rec['height'], rec['width'], self.device)


def imread_bgr(path):
    """`cv2.imread(path)`: HWC uint8, BGR, EXIF orientation applied (cv2's IMREAD_COLOR default)."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im).convert('RGB')
        rgb = np.asarray(im, dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def train_collate(batch):
    """utils/coco.py:14-28: rejected samples (None) are replaced by repeating valid ones; images are stacked."""
    imgs, targets, masks = [], [], []
    valid_batch = [aa for aa in batch if aa[0] is not None]
    lack_len = len(batch) - len(valid_batch)
    for i in range(lack_len):
        valid_batch.append(valid_batch[i])
    for img, boxes, m in valid_batch:
        imgs.append(img)
        targets.append(torch.as_tensor(boxes, dtype=torch.float32).to(img.device))
        masks.append(m.to(torch.float32))
    return torch.stack(imgs, 0), targets, masks


def val_collate(batch):
    img, boxes, masks, h, w = batch[0]
    return img.unsqueeze(0), torch.as_tensor(boxes, dtype=torch.float32), masks.to(torch.float32), h, w


def detect_collate(batch):
    return batch[0][0].unsqueeze(0), batch[0][1], batch[0][2]


class COCODetection:
    """`COCODetection(cfg, mode)` of the reference (utils/coco.py:47-134), samples produced on `device`."""

    def __init__(self, cfg, mode='train', device='cuda', rng=_random):
        self.mode, self.cfg, self.device, self.rng = mode, cfg, torch.device(device), rng
        if mode in ('train', 'val'):
            self.image_path = cfg.train_imgs if mode == 'train' else cfg.val_imgs
            self.coco = COCO(cfg.train_ann if mode == 'train' else cfg.val_ann, device=self.device)
            self.ids = list(self.coco.imgToAnns.keys())
        elif mode == 'detect':
            self.image_path = sorted(glob.glob(cfg.image + '/*.jpg'))
        self.continuous_id = cfg.continuous_id

    def __len__(self):
        if self.mode == 'train':
            return len(self.ids)
        if self.mode == 'val':
            return len(self.ids) if self.cfg.val_num == -1 else min(self.cfg.val_num, len(self.ids))
        return len(self.image_path)

    def read(self, index):
        """Host half of `__getitem__`: file read + JPEG decode + annotation bookkeeping (thread-safe; BatchLoader runs it on its
        thread pool).  Returns a record for `finish`."""
        if self.mode == 'detect':
            img_name = self.image_path[index]
            return {'img': imread_bgr(img_name), 'name': img_name.split(osp.sep)[-1]}
        img_id = self.ids[index]
        target = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id))
        target = [aa for aa in target if not aa['iscrowd']]
        file_name = self.coco.loadImgs(img_id)[0]['file_name']
        img_path = osp.join(self.image_path, file_name)
        assert osp.exists(img_path), f'Image path does not exist: {img_path}'
        img = imread_bgr(img_path)
        assert len(target) > 0, 'No annotation in this image!'
        box_list, label_list, kept = [], [], []
        for aa in target:
            bbox = aa['bbox']
            if self.mode == 'train' and (bbox[0] < 0 or bbox[1] < 0 or bbox[2] < 4 or bbox[3] < 4):
                continue                                         # "some boxes are wrong, ignore them" (:86-88)
            box_list.append(np.array([bbox[0], bbox[1], bbox[0] + bbox[2], bbox[1] + bbox[3]]))
            label_list.append(self.continuous_id[aa['category_id']] - 1)
            kept.append(aa)
        return {'img': img, 'img_id': img_id, 'boxes': box_list, 'labels': label_list, 'anns': kept}

    def finish(self, rec):
        """Device half: H2D of the decoded image, mask rasterisation, augmentation (and the `random` draws, in sample order)."""
        img = rec['img']
        img_dev = torch.from_numpy(img).to(self.device)
        if self.mode == 'detect':
            return val_aug(img_dev, self.cfg.img_size), img, rec['name']
        height, width, _ = img.shape
        if len(rec['boxes']) == 0:
            if self.mode == 'val':
                raise RuntimeError('Error, no valid object in this image.')
            print(f"No valid object in image: {rec['img_id']}. Use a repeated image in this batch.")
            return None, None, None
        boxes, labels = np.array(rec['boxes']), np.array(rec['labels'])
        masks = self.coco.annsToMasks(rec['anns'])
        assert tuple(masks.shape) == (boxes.shape[0], height, width), 'Unmatched annotations.'
        if self.mode == 'train':
            img_out, masks_out, boxes, labels = train_aug(img_dev, masks, boxes, labels, self.cfg.img_size, self.rng)
            if img_out is None:
                return None, None, None
            return img_out, np.hstack((boxes, np.expand_dims(labels, axis=1))), masks_out
        img_out = val_aug(img_dev, self.cfg.img_size)
        boxes = boxes / np.array([width, height, width, height])     # to 0~1 scale
        return img_out, np.hstack((boxes, np.expand_dims(labels, axis=1))), masks, height, width

    def __getitem__(self, index):
        return self.finish(self.read(index))


class BatchLoader:
    """The DataLoader + DistributedSampler pair of train.py:78-81 / eval.py:30 for GPU-resident samples: rank-sharded
    (optionally shuffled, seeded per epoch) indices, `batch_size` samples per step, `dataset.read` (file + JPEG decode) one batch
    ahead on `threads` host threads, `dataset.finish` (HIP launches, `random` draws) in sample order on the caller's thread,
    collated by `collate_fn`."""

    def __init__(self, dataset, batch_size, collate_fn, shuffle=False, rank=0, world_size=1, seed=0, threads=4, drop_last=False):
        self.dataset, self.batch_size, self.collate_fn = dataset, batch_size, collate_fn
        self.shuffle, self.rank, self.world_size, self.seed, self.threads, self.drop_last = shuffle, rank, world_size, seed, threads, drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        """DistributedSampler: permutation seeded by seed + epoch, padded to a multiple of world_size, strided by rank."""
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        total = -(-n // self.world_size) * self.world_size
        order += order[:total - n]
        return order[self.rank:total:self.world_size]

    def __len__(self):
        n = len(self.indices())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        idx = self.indices()
        batches = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        with ThreadPoolExecutor(max_workers=max(1, self.threads)) as pool:
            pending = [pool.submit(self.dataset.read, i) for i in batches[0]] if batches else []
            for b in range(len(batches)):
                ahead = [pool.submit(self.dataset.read, i) for i in batches[b + 1]] if b + 1 < len(batches) else []
                samples = [self.dataset.finish(f.result()) for f in pending]    # GPU work + random draws: consumer thread, in order
                pending = ahead
                yield self.collate_fn(samples)
