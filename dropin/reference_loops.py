"""The reference scripts' own loops, statement for statement, for tests and `bench.py` — nothing in here is product code.

`train.py` / `eval.py` cannot run on a box without COCO, tensorboardX and pycocotools, so the two functions below repeat the
statements those scripts execute around the hot path (every block cites the lines it repeats) with the data loader handed in by
the caller.  All names resolve through `dropin/` (`modules.yolact`, `utils.output_utils`, `utils.common_utils`, `utils.timer`,
`config`), i.e. exactly what `python dropin/run.py train.py ...` binds: `torch.optim.SGD` / `AdamW` over `net.parameters()`,
`torch.nn.parallel.DistributedDataParallel` around the `Yolact` module, `loss.backward()`, `optimizer.step()` — NOT the build's
own `Trainer` (flat gradient buffer, `FlatGradReducer`), which is what `tests/test_gpu_reference_loop.py` compares against.
"""
import time

import torch
import torch.distributed as dist
import torch.optim as optim
from torch.nn.parallel import DistributedDataParallel as DDP

from utils import timer                                                       # noqa: E402  (dropin/)
from utils.output_utils import after_nms, nms                                 # noqa: E402
from utils.common_utils import APDataObject, MakeJson, prep_metrics, calc_map  # noqa: E402
from yolact_minimal_amd.utils.common_utils import rle_encode                 # noqa: E402  (coco_api='device' only)


def make_optimizer(net, cfg):
    """train.py:60-65."""
    if 'res' in cfg.__class__.__name__:
        optimizer = optim.SGD(net.parameters(), lr=cfg.lr, momentum=0.9, weight_decay=5e-4)
    elif cfg.__class__.__name__ == 'swin_tiny_coco':
        optimizer = optim.AdamW(net.parameters(), lr=cfg.lr, weight_decay=0.05)
    else:
        raise ValueError('Unrecognized cfg.')
    return optimizer


def wrap_ddp(net, local_rank):
    """train.py:76 (the process group was joined by `get_config(args, mode='train')`, config.py:229-232)."""
    return DDP(net.cuda(), [local_rank], output_device=local_rank, broadcast_buffers=True)


def train_loop(net, optimizer, cfg, data_loader, start_step=0, max_steps=None, on_step=None, val_interval=-1, evaluate=None,
               fences=True):
    """train.py:88-90,102-136 (+ :162-177 when `val_interval` > 0): `net` is what train.py:76 left in `net` (the DDP wrapper when
    cfg.cuda).  Returns the per-step loss 4-tuples as python floats (one host read per step, as train.py:147-150 does every
    tenth step) and the step counter.

    `fences`: the reference starts its timer at the second iteration (train.py:176-177: `step == val_step + 1`), and from then on
    every `timer.counter` block synchronizes the device on entry and exit (utils/timer.py:63-76) — the loop as published never has
    two phases of a step in flight at once.  False leaves the timer stopped for the whole run (the fences are the only difference)."""
    step, history = start_step, []
    val_step = start_step
    timer.reset()
    for images, targets, masks in data_loader:
        if cfg.warmup_until > 0 and step <= cfg.warmup_until:  # warm up learning rate.
            for param_group in optimizer.param_groups:
                param_group['lr'] = (cfg.lr - cfg.warmup_init) * (step / cfg.warmup_until) + cfg.warmup_init

        if step in cfg.lr_steps:  # learning rate decay.
            for param_group in optimizer.param_groups:
                param_group['lr'] = cfg.lr * 0.1 ** cfg.lr_steps.index(step)

        if cfg.cuda:
            images = images.cuda().detach()
            targets = [ann.cuda().detach() for ann in targets]
            masks = [mask.cuda().detach() for mask in masks]

        with timer.counter('for+loss'):
            loss_c, loss_b, loss_m, loss_s = net(images, targets, masks)

            if cfg.cuda:
                # use .all_reduce() to get the summed loss from all GPUs
                all_loss = torch.stack([loss_c, loss_b, loss_m, loss_s], dim=0)
                dist.all_reduce(all_loss)

        with timer.counter('backward'):
            loss_total = loss_c + loss_b + loss_m + loss_s
            optimizer.zero_grad()
            loss_total.backward()

        with timer.counter('update'):
            optimizer.step()

        if on_step is not None:
            on_step(step, (loss_c, loss_b, loss_m, loss_s), optimizer.param_groups[0]['lr'])
        else:
            history.append([float(l.detach()) for l in (loss_c, loss_b, loss_m, loss_s)])

        time_this = time.time()
        if step > start_step:
            batch_time = time_this - time_last
            timer.add_batch_time(batch_time)
        time_last = time_this

        if val_interval > 0 and step % val_interval == 0 and step != start_step and evaluate is not None:
            val_step = step
            net.eval()
            evaluate(net.module if cfg.cuda else net, cfg, step)
            net.train()
            timer.reset()  # training timer and val timer share the same Obj, so reset it to avoid conflict

        if fences and step == val_step + 1:
            timer.start()  # the first iteration after validation should not be included

        step += 1
        if max_steps is not None and step - start_step >= max_steps:
            break
    return history, step


IOU_THRES = [x / 100 for x in range(50, 100, 5)]                              # eval.py:24


def eval_loop(net, cfg, data_loader, image_ids=None, coco_api=False, make_json=None, sync_stages=True):
    """eval.py:35-69 for every `(img, gt, gt_masks, img_h, img_w)` of `data_loader`, one image at a time.  `coco_api`: the
    `--coco_api` branch (eval.py:60-67: boxes and the dense fp32 masks cross PCIe, `MakeJson.add_bbox/add_mask`; 'device' = the
    same records with the RLE strings made on the GPU); otherwise
    `prep_metrics` on the device tensors (eval.py:69).  `sync_stages`: the reference's `timer.counter` fences every stage with a
    device synchronize (utils/timer.py:63-76); False leaves the fences out (the loop is otherwise unchanged).
    Returns (ap_data, make_json, images with detections, seconds)."""
    ap_data = {'box': [[APDataObject() for _ in cfg.class_names] for _ in IOU_THRES],
               'mask': [[APDataObject() for _ in cfg.class_names] for _ in IOU_THRES]}
    if coco_api and make_json is None:
        make_json = MakeJson()
    timer.reset()
    if sync_stages:
        timer.start()
    counter = timer.counter if sync_stages else _no_counter
    seen = 0
    t0 = time.perf_counter()
    for i, (img, gt, gt_masks, img_h, img_w) in enumerate(data_loader):
        if cfg.cuda:
            img, gt, gt_masks = img.cuda(), gt.cuda(), gt_masks.cuda()

        with torch.no_grad(), counter('forward'):
            class_p, box_p, coef_p, proto_p = net(img)

        with counter('nms'):
            ids_p, class_p, box_p, coef_p, proto_p = nms(class_p, box_p, coef_p, proto_p, net.anchors, cfg)

        with counter('after_nms'):
            ids_p, class_p, boxes_p, masks_p = after_nms(ids_p, class_p, box_p, coef_p, proto_p, img_h, img_w)
            if ids_p is None:
                continue

        with counter('metric'):
            ids_p = list(ids_p.cpu().numpy().astype(int))
            class_p = list(class_p.cpu().numpy().astype(float))

            if coco_api == 'device':
                # the build's variant of the same branch: RLE strings are made on the GPU for the image's masks at once
                # (`ym_rle_encode`), so a few hundred bytes per mask cross PCIe instead of img_h * img_w * 4
                boxes_p = boxes_p.cpu().numpy()
                rles = rle_encode(masks_p)
                for j in range(len(rles)):
                    if (boxes_p[j, 3] - boxes_p[j, 1]) * (boxes_p[j, 2] - boxes_p[j, 0]) > 0:
                        image_id = image_ids[i] if image_ids is not None else i
                        make_json.add_bbox(image_id, ids_p[j], boxes_p[j, :], class_p[j])
                        make_json.add_mask(image_id, ids_p[j], rles[j], class_p[j])
            elif coco_api:
                boxes_p = boxes_p.cpu().numpy()
                masks_p = masks_p.cpu().numpy()

                for j in range(masks_p.shape[0]):
                    if (boxes_p[j, 3] - boxes_p[j, 1]) * (boxes_p[j, 2] - boxes_p[j, 0]) > 0:
                        image_id = image_ids[i] if image_ids is not None else i
                        make_json.add_bbox(image_id, ids_p[j], boxes_p[j, :], class_p[j])
                        make_json.add_mask(image_id, ids_p[j], masks_p[j, :, :], class_p[j])
            else:
                prep_metrics(ap_data, ids_p, class_p, boxes_p, masks_p, gt, gt_masks, img_h, img_w, IOU_THRES)
        seen += 1
    if cfg.cuda:
        torch.cuda.synchronize()
    return ap_data, make_json, seen, time.perf_counter() - t0


class _no_counter:
    def __init__(self, name):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def table(ap_data, cfg, step=None):
    """eval.py:106."""
    return calc_map(ap_data, IOU_THRES, len(cfg.class_names), step=step)
