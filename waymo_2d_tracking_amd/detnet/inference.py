"""Detection inference CLI - mirrors /root/reference/inference.py + /root/reference/detnet/inference.py
(flags :26-60, PredictModel :77-130, inference() :151-199) and the JSON export
(/root/reference/detnet/export.py:159-165 -> /root/reference/detnet/data/coco.py:229-252).

    python -m waymo_2d_tracking_amd.detnet.inference -m detectron2:Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml \
        -i IMAGES_DIR --export submission.json [--tta x1.5,hflip] [--batch-size 1]
    torchrun --nproc-per-node 8 -m waymo_2d_tracking_amd.detnet.inference ...     # one process per GPU

Multi-GPU: the reference spawns `-j N` processes with contiguous dataset shards and merges shelve files
(detnet/trainer/test.py:227-270); here torchrun starts one process per GPU, each takes its contiguous shard
(distributed.contiguous_split) and rank 0 receives the rows through one gather (RCCL) - no files, no collective
inside the detection loop.
"""
import argparse
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

IMG_EXT = ('.jpg', '.jpeg', '.png', '.bmp')


def arg2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ('yes', 'true', 't', 'y', '1')


def add_test_argument(parser):
    """detnet/inference.py:26-46 (flags that do not apply to the Cascade R-CNN path are accepted and ignored)."""
    parser.add_argument('--device', default='auto', choices=['auto', 'cuda', 'cpu', 'half'])
    parser.add_argument("-m", "--model", type=str, default='detectron2:Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml',
                        help='pre/trained model file')
    parser.add_argument("--threshold", type=float, default=0.01, help='threshold for accepting detection')
    parser.add_argument("--max-bbox", type=int, default=0, help='maximum number of bbox output per image if positive')
    parser.add_argument('--max-bbox-per-class', type=int, default=4000)
    parser.add_argument("--nms-thresh", type=float, default=0.15)
    parser.add_argument('--soft-nms', action='store_true')
    parser.add_argument('--bbox-voting', type=float, default=0)
    parser.add_argument('--tta', type=str, default='', help='Test Time Augmentation, e.g. x1.5,hflip')
    parser.add_argument("--batch-size", type=int, default=0)
    parser.add_argument("--resize", type=str)
    parser.add_argument("--max-image-size", type=int)
    parser.add_argument("--auto-contrast", type=arg2bool)
    parser.add_argument("--clahe", type=arg2bool)


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_test_argument(parser)
    parser.add_argument("-i", "--input", type=str, help='root directory of input images')
    parser.add_argument("-o", "--output", type=str, help='root directory of output')
    parser.add_argument("--exclusive", type=str)
    parser.add_argument('-j', '--jobs', type=int, default=1, help='accepted for compatibility; use torchrun for N GPUs')
    parser.add_argument('--resume', type=str)
    parser.add_argument('--eval', action='store_true')
    parser.add_argument('--export', type=str, help='path of export file')
    parser.add_argument('--export-format', type=str, choices=('json',), default='json')
    parser.add_argument('--profile', action='store_true')
    parser.add_argument('--cudnn-benchmark', default=True, type=arg2bool)
    parser.add_argument('--data-root', type=str, default=None)
    parser.add_argument('--data-include-empty', action='store_true')
    parser.add_argument('--data-bgr', action='store_true')
    return parser


class PredictModel(torch.nn.Module):
    """detnet/inference.py:77-130: TTA wrap + optional global top-k over classes."""

    def __init__(self, model, detect_args):
        super().__init__()
        self.model = model
        if detect_args.get('tta'):
            from .nn.tta import TTA
            self.model = TTA(model, detect_args['tta'])
        self.detect_args = detect_args

    def forward(self, sample):
        return [self.post_process(d) for d in self.model.predict(sample)]

    def post_process(self, detections):
        max_bbox = self.detect_args.get('max_bbox', 0)
        if max_bbox > 0:
            labels = np.concatenate([[i] * len(b) for i, b in enumerate(detections)]).astype(np.int64) if detections else np.zeros(0, np.int64)
            allb = np.vstack(detections)
            if len(labels) > max_bbox:
                top = np.argsort(-allb[:, 0], kind='stable')[:max_bbox]
                allb, labels = allb[top], labels[top]
                detections = [allb[labels == i] for i in range(len(detections))]
        return detections


def load_prediction(image_sizes, classnames, predictions, category_ids=None):
    """COCODetection.load_prediction (detnet/data/coco.py:229-252): normalised per-class [score,cx,cy,w,h] ->
    [{image_id, category_id, bbox [int x, y, w, h], score (5 decimals)}].  category ids default to 1..C
    (waymo_to_coco.py:19,36)."""
    results = []
    for image_id, (width, height) in image_sizes.items():
        det = predictions[str(image_id)]
        scale = np.asarray([1, width, height, width, height])
        for cls, bbox in enumerate(det):
            bbox = bbox * scale
            bbox[:, 1:3] -= (bbox[:, 3:5] / 2)
            cat = category_ids[cls] if category_ids else cls + 1
            for box in bbox:
                results.append(dict(image_id=image_id, category_id=cat, bbox=[int(v) for v in box[1:5]],
                                    score=round(float(box[0]), 5)))
    return results


def list_images(root):
    root = Path(root)
    files = sorted(p for p in root.rglob('*') if p.suffix.lower() in IMG_EXT)
    return [(str(p.relative_to(root).with_suffix('')), p) for p in files]       # image_id = "<segment>/<ts>/<CAMERA>"


def inference(args):
    """detnet/inference.py:151-199 for an image folder."""
    from PIL import Image
    from . import nn as detnn
    from .. import distributed as D
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    if args.cudnn_benchmark:
        from ..tuning import enable_gemm_tuning
        enable_gemm_tuning()                       # library-GEMM counterpart of --cudnn-benchmark
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('the detector runs on the GPU only (HIP kernels, no CPU fallback)')
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    start = time.time()
    model = detnn.load(args.model).cuda().eval()
    tta = [t for t in args.tta.split(',') if t] if args.tta else None
    predict = PredictModel(model, dict(tta=tta, max_bbox=args.max_bbox))
    images = list_images(args.input)
    lo, hi = D.contiguous_split(len(images), world)[rank]
    sizes, preds = {}, {}
    with torch.no_grad():
        for image_id, path in images[lo:hi]:
            img = Image.open(path).convert('RGB')                                   # ToRGB
            x = torch.as_tensor(np.asarray(img, dtype=np.float32).transpose(2, 0, 1)).unsqueeze(0)   # ToTensor(scaling=False)
            preds[image_id] = predict(x.cuda())[0]
            sizes[image_id] = (img.width, img.height)
    rows = load_prediction(sizes, model.classnames, preds)
    gathered = D.gather_object_rank0(rows)
    if rank == 0:
        rows = [r for part in gathered for r in part]
        if args.export:
            out = Path(args.export).with_suffix('.json')
            out.parent.mkdir(parents=True, exist_ok=True)
            with open(out, 'wt') as fp:
                json.dump(rows, fp)
        print(f'inference done in {time.time() - start:.1f}s, {len(images)} images, {len(rows)} detections')
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return rows if rank == 0 else None


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.output and not args.eval and not args.export:
        raise UserWarning("Please specify at least one path for output / evaluation / export")
    inference(args)


if __name__ == '__main__':
    main()
