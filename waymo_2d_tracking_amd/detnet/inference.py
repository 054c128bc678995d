"""Detection inference CLI - mirrors /root/reference/inference.py + /root/reference/detnet/inference.py
(flags :26-60, PredictModel :77-130, inference() :151-199), the test driver (/root/reference/detnet/trainer/test.py:189-279:
`-j N`, `--resume`, `-o`) and the JSON export (/root/reference/detnet/export.py:159-165 -> detnet/data/coco.py:229-252).

    python -m waymo_2d_tracking_amd.detnet.inference -m MODEL_FILE -i IMAGES_DIR --export submission.json \
        [--tta x1.5,hflip] [--auto-contrast=1] [-j 8] [-o OUT_DIR] [--resume OUT_DIR/detections.pkl] [--eval --annotations GT.json]
    torchrun --nproc-per-node 8 -m waymo_2d_tracking_amd.detnet.inference ...     # the same, started by torchrun

MODEL_FILE is a `{args, kwargs, state_dict}` file of the reference or of this package (nn/__init__.py); the string
`detectron2:<yaml>` loads the detectron2 COCO checkpoint like the reference does (it must be available locally).

Multi-GPU: the reference spawns `-j N` processes with contiguous dataset shards and merges per-process shelve files
(trainer/test.py:227-270).  Here `-j N` spawns N processes (one per GPU) - or torchrun does - each takes its contiguous shard
(distributed.contiguous_split), fills a columnar prediction store (trainer/predictions.py) and rank 0 receives all rows in one
tensor gather over RCCL - no files, no collective inside the detection loop.  Image decoding (PIL, like the reference's
dataset workers) runs in a thread pool ahead of the GPU; AutoContrast runs on the GPU.
"""
import argparse
import os
import time
from pathlib import Path

import threading

import numpy as np
import torch

from .. import _lib

IMG_EXT = ('.jpg', '.jpeg', '.png', '.bmp')


def arg2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ('yes', 'true', 't', 'y', '1')


def add_test_argument(parser):
    """detnet/inference.py:26-46; flags this path cannot honour raise in check_supported()."""
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
    parser.add_argument("--inflight", type=int, default=2,
                        help="(extension) detector passes in flight: each image size is captured once per lane as a hipGraph and replayed on its own "
                             "stream, results are collected one image behind; 0 = one eager call per image")
    parser.add_argument("--decoder", type=str, default='gpu', choices=('gpu', 'pil'),
                        help="JPEG decoding: 'gpu' = HIP decoder (bit-exact with PIL), 'pil' = host decode like the reference")
    parser.add_argument("--auto-contrast", type=arg2bool)
    parser.add_argument("--clahe", type=arg2bool)


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_test_argument(parser)
    parser.add_argument("-i", "--input", type=str, help='root directory of input images')
    parser.add_argument("-o", "--output", type=str, help='root directory of output')
    parser.add_argument("--exclusive", type=str)
    parser.add_argument('-j', '--jobs', type=int, default=1, help='number of processes = GPUs (one process per GPU)')
    parser.add_argument('--annotations', type=str, help='COCO-format ground truth for --eval')
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

    # ---- round 6: the same results for a STREAM of images with `lanes` detector passes in flight (captured hipGraphs, nn.GraphLanePredictor) ----
    def stream(self, lanes=2):
        """A (submit, collect) pair or None when this model / TTA plan has no static-shape pass (the loop then calls forward() per image)."""
        from .nn.tta import TTA, undo_plan
        if isinstance(self.model, TTA):
            gl, plan = self.model.graph_lanes(lanes), self.model.plan
        elif hasattr(self.model, 'predict_padded') and not getattr(self.model, 'tta_min_sizes', None):
            from .nn.detectron2_det import GraphLanePredictor
            gl, plan = GraphLanePredictor(self.model, lanes=lanes), []
        else:
            gl = None
        if gl is None:
            return None

        def collect(ticket):
            det = [[b.copy() for b in gl.collect(ticket)]]
            return self.post_process(undo_plan(det, plan)[0])
        return gl.submit, collect

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
    (waymo_to_coco.py:19,36).  Dict form of export.detection_rows (kept for API parity)."""
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


def autocontrast_(img_u8):
    """ImageOps.autocontrast(image) of PIL (cutoff 0; the reference's AutoContrast transform, trainer/transforms/vision.py:
    1069-1075) on a (H, W, 3) uint8 DEVICE tensor: per channel lo / hi = darkest / brightest value present; if hi <= lo the
    channel is unchanged, else lut[v] = clamp(int(v * 255 / (hi - lo) - lo * 255 / (hi - lo)), 0, 255).  Bit-exact with PIL
    (tests/test_gpu_detector.py); float64 like PIL's Python arithmetic."""
    if img_u8.is_cuda and img_u8.dim() == 3 and img_u8.shape[2] == 3:
        from .nn import ops
        return ops.autocontrast_(img_u8.contiguous().clone())          # two HIP launches; the torch formulation below is the CPU-tensor twin
    flat = img_u8.reshape(-1, img_u8.shape[-1])
    lo = flat.amin(0).double()
    hi = flat.amax(0).double()
    v = torch.arange(256, dtype=torch.float64, device=img_u8.device).unsqueeze(1)
    span = torch.where(hi > lo, hi - lo, torch.ones_like(hi))
    scale = torch.full_like(span, 255.0) / span          # true division: `255.0 / tensor` is reciprocal * 255 in torch (one ulp off PIL's scale)
    lut = torch.trunc(v * scale + (-lo * scale)).clamp_(0, 255)
    lut = torch.where((hi > lo).unsqueeze(0), lut, v.expand(-1, lut.shape[1])).to(torch.uint8)      # (256, C)
    out = torch.gather(lut, 0, flat.long())
    return out.reshape(img_u8.shape)


def resize_size(w, h, size, max_size=None):
    """Resize.compute_scaled_image_size (trainer/transforms/vision.py:168-190) for an int size: (out_h, out_w)."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    if max_size:
        if oh > max_size:
            ow, oh = int(max_size / oh * ow), max_size
        if ow > max_size:
            oh, ow = int(max_size / ow * oh), max_size
    return oh, ow


class ImageLoader(object):
    """Decode ahead of the detector in a thread pool.  JPEG files: the file bytes go to the GPU and are decoded there
    (ops.jpeg_decode: Huffman, IDCT, upsampling, colour conversion in HIP kernels; bit-exact with PIL's decode, ~1 MB
    instead of 7.4 MB over PCIe per 1920x1280 frame).  With --resize, or for other formats: PIL decode + ToRGB (+ Resize,
    PIL bilinear like the reference), pinned staging buffer, asynchronous H2D of the uint8 HWC image.  Yields
    (image_id, device uint8 (H, W, 3), (width, height) of the ORIGINAL decoded image): the detector's boxes are normalised,
    and the reference's export scales them by the untransformed data set's image size (export.py:159-165 ->
    coco.py:243-246 with `dataset.dataset`), not by the resized one.

    Transform order of the reference (detnet/inference.py:170-178): ToRGB, AutoContrast, [CLAHE], Resize.  Without --resize
    AutoContrast runs on the GPU (autocontrast_, bit-exact with PIL); with --resize it has to precede the resize, so the
    loader thread applies PIL's own ImageOps.autocontrast before resizing (`auto_contrast_in_loader`)."""
    _warned = False                 # the PIL hand-over of an unsupported JPEG flavour is announced once


    def __init__(self, items, resize=None, max_image_size=None, workers=4, depth=4, auto_contrast=False, decoder='gpu'):
        self.items, self.resize, self.max_size = items, resize, max_image_size
        self.workers, self.depth = workers, depth
        self.auto_contrast_in_loader = bool(auto_contrast and resize)
        assert decoder in ('gpu', 'pil')
        # JPEG files are decoded by the HIP decoder (csrc/jpeg_decode.hip, bit-exact with PIL) unless the image has to pass
        # through PIL's own resampling filter first (--resize: the reference resizes the PIL image, vision.py:137) or the
        # caller asks for the host decoder; other formats (PNG ...) are read by PIL as in the reference
        self.decoder = 'pil' if resize else decoder
        self._local = threading.local()

    def _decode_pil(self, path):
        from PIL import Image, ImageOps
        img = Image.open(path).convert('RGB')                                       # ToRGB (vision.py:954)
        size = img.size
        if self.auto_contrast_in_loader:
            img = ImageOps.autocontrast(img)                                        # AutoContrast precedes Resize
        if self.resize:
            oh, ow = resize_size(img.width, img.height, int(self.resize), self.max_size)
            img = img.resize((ow, oh), Image.BILINEAR)
        return np.asarray(img, dtype=np.uint8), size

    def _decode(self, path):
        if self.decoder == 'gpu':
            with open(path, 'rb') as f:
                data = f.read()
            if data[:2] == b'\xff\xd8':
                from .nn import ops
                if getattr(self._local, 'stream', None) is None:
                    self._local.stream = torch.cuda.Stream()                        # one decode stream per loader thread
                try:
                    with torch.cuda.stream(self._local.stream):
                        t = ops.jpeg_decode(data)                                   # returns with the image complete
                    return t, (t.shape[1], t.shape[0])
                except _lib.WaymoTrackError as e:
                    # a flavour the HIP decoder does not implement (progressive, CMYK, ...) is read the way PNG / BMP files are:
                    # by PIL, like the reference - said once, never silently; corrupt or truncated files still raise
                    if 'unsupported' not in str(e):
                        raise
                    if not ImageLoader._warned:
                        ImageLoader._warned = True
                        print('ImageLoader: %s: %s -> such files are decoded by PIL on the host' % (path, e), flush=True)
        return self._decode_pil(path)

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.workers) as pool:
            pending = []
            it = iter(self.items)
            for image_id, path in it:
                pending.append((image_id, pool.submit(self._decode, path)))
                if len(pending) >= self.depth:
                    break
            while pending:
                image_id, fut = pending.pop(0)
                nxt = next(it, None)
                if nxt is not None:
                    pending.append((nxt[0], pool.submit(self._decode, nxt[1])))
                t, size = fut.result()
                if isinstance(t, np.ndarray):
                    t = torch.from_numpy(np.array(t, copy=True)).pin_memory().cuda(non_blocking=True)
                else:
                    t.record_stream(torch.cuda.current_stream())                    # allocated on a loader thread's stream
                yield image_id, t, size


def check_supported(args):
    """Flags of the reference that this path cannot honour fail loudly instead of being ignored."""
    if args.clahe:
        raise NotImplementedError('--clahe needs scikit-image equalize_adapthist (vision.py:1088-1092), which is not available')
    if args.device in ('cpu', 'half'):
        raise NotImplementedError('--device %s: the detector runs in fp32 on the GPU only (HIP kernels, no CPU fallback)' % args.device)
    if args.soft_nms or args.bbox_voting:
        raise NotImplementedError('--soft-nms / --bbox-voting belong to the SSD detect layer of the reference, not to Cascade R-CNN')
    if args.resize is not None:
        try:
            int(args.resize)
        except ValueError:
            raise NotImplementedError('--resize takes the length of the shorter edge (int); (h, w) pairs are not supported')
    if args.eval and not args.annotations:
        raise ValueError('--eval needs --annotations GT.json (COCO-format ground truth; the reference reads it from --data-root)')
    if args.export_format != 'json':
        raise NotImplementedError('--export-format json only')


def run_rank(args, world, rank):
    """One process = one GPU: detect the shard, collate, and on rank 0 save / evaluate / export."""
    from . import nn as detnn
    from .trainer import Predictions
    from .. import distributed as D
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    if args.cudnn_benchmark:
        from ..tuning import enable_gemm_tuning
        enable_gemm_tuning()                       # library-GEMM counterpart of --cudnn-benchmark
    start = time.time()
    model = detnn.load(args.model).cuda().eval()
    tta = [t for t in args.tta.split(',') if t] if args.tta else None
    predict = PredictModel(model, dict(tta=tta, max_bbox=args.max_bbox))
    images = list_images(args.input)
    if args.exclusive:
        images = [(i, p) for i, p in images if args.exclusive not in i]
    image_ids = [i for i, _ in images]
    resumed = None
    if args.resume:                                # trainer/test.py:190-195: skip the samples already tested
        resumed = Predictions.open(args.resume)
        done = set(resumed.keys())
        if rank == 0:
            print(f"resuming {len(done)} tested samples")
        images = [(i, p) for i, p in images if i not in done]
    lo, hi = D.contiguous_split(len(images), world)[rank]
    store = Predictions(model.classnames, image_ids)
    sizes = np.zeros((len(image_ids), 2), np.int32)
    index = {k: i for i, k in enumerate(image_ids)}
    with torch.no_grad():
        loader = ImageLoader(images[lo:hi], args.resize, args.max_image_size, auto_contrast=args.auto_contrast, decoder=args.decoder)
        # round 6: `--inflight` detector passes in flight, each a captured hipGraph per image size (the Waymo cameras have two); results are collected
        # one image behind.  --inflight 0: the per-image eager call of the earlier rounds (and the only form for TTA plans / models without a
        # static-shape pass)
        lanes = predict.stream(args.inflight) if args.inflight > 0 else None
        pending = []
        for image_id, img, (w, h) in loader:
            if args.auto_contrast and not loader.auto_contrast_in_loader:
                img = autocontrast_(img)
            sizes[index[image_id]] = (w, h)
            if lanes is None:
                store[image_id] = predict(img.unsqueeze(0))[0]        # uint8 HWC -> fused pre-processing kernel
                continue
            if len(pending) >= args.inflight:                         # the oldest image's lane is the one this image will take
                k, t = pending.pop(0)
                store[k] = lanes[1](t)
            pending.append((image_id, lanes[0](img)))
        for k, t in pending:
            store[k] = lanes[1](t)
    cols, tested = store.shard_columns()
    allc = D.gather_columns_rank0(cols)                               # the one exchange: result rows -> rank 0
    meta = D.gather_columns_rank0(dict(image=np.nonzero(tested)[0].astype(np.int32), size=sizes[tested]))
    rows = None
    if rank == 0:
        mask = np.zeros(len(image_ids), bool)
        mask[meta['image']] = True
        sizes[meta['image']] = meta['size']
        predictions = Predictions.from_shards(model.classnames, image_ids, [allc], [mask])
        if resumed is not None:
            predictions.update(resumed)
        if args.output:                                              # trainer/test.py:272-276
            out_dir = Path(args.output)
            out_dir.mkdir(parents=True, exist_ok=True)
            print('saving', out_dir / 'detections.pkl')
            predictions.save(out_dir / 'detections.pkl')
        def image_sizes_in_order():
            """{image_id: (width, height)} of every tested image in data-set order (the reference iterates coco.imgs); resumed
            samples were not decoded in this run: their sizes come from the file headers."""
            out, paths = {}, None
            for i, k in enumerate(image_ids):
                if not predictions.tested[predictions._index[k]]:
                    continue
                if sizes[i, 0] > 0:
                    out[k] = (int(sizes[i, 0]), int(sizes[i, 1]))
                else:
                    from PIL import Image
                    paths = paths or dict(list_images(args.input))
                    with Image.open(paths[k]) as im:
                        out[k] = im.size
            return out
        image_sizes = image_sizes_in_order() if (args.eval or args.export) else {}
        if args.eval:
            from .data.metric import evaluate_detections
            evaluate_detections(predictions, args.annotations, image_sizes, print_fn=print)
        if args.export:
            from .export import export, detection_rows
            export(predictions, args.export, image_sizes, args.export_format, args.threshold)
            rows = detection_rows(predictions, image_sizes)
        print(f'inference done in {time.time() - start:.1f}s, {len(image_ids)} images, {len(allc["image"])} detections')
    return rows


def _spawned(rank, world, port, argv):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    main(argv, _spawned_child=True)


def inference(args, argv=None):
    """detnet/inference.py:151-199 for an image folder."""
    check_supported(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1 and args.jobs > 1:                                  # -j N: one process per GPU (trainer/test.py:227-250)
        import socket
        import torch.multiprocessing as mp
        n_gpu = torch.cuda.device_count()
        if args.jobs > n_gpu:
            raise RuntimeError(f'-j {args.jobs} but only {n_gpu} GPUs are visible (one process per GPU)')
        with socket.socket() as sck:
            sck.bind(('127.0.0.1', 0))
            port = sck.getsockname()[1]
        mp.spawn(_spawned, args=(args.jobs, port, list(argv or [])), nprocs=args.jobs, join=True)
        return None
    if not torch.cuda.is_available():
        raise RuntimeError('the detector runs on the GPU only (HIP kernels, no CPU fallback)')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))))
    try:
        rows = run_rank(args, world, rank)
    except BaseException:
        if world > 1:                                                 # no barrier here: the peers sit in a collective and a
            import torch.distributed as dist                         # barrier would mismatch it; tear the group down so they
            try:                                                     # fail fast instead of waiting for the RCCL timeout
                dist.destroy_process_group()
            except Exception:
                pass
        raise
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return rows


def main(argv=None, _spawned_child=False):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if not args.output and not args.eval and not args.export:
        raise UserWarning("Please specify at least one path for output / evaluation / export")
    return inference(args, argv)


if __name__ == '__main__':
    main()
