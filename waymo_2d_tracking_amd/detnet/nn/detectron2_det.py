"""Detector wrapper - mirrors /root/reference/detnet/nn/detectron2_det/__init__.py (class Detectron2Det :63-139).

``predict(x)`` keeps the reference contract: ``x`` is a PIL image or a (B,3,H,W) float tensor 0..255 in RGB order;
the result is ``[per image][per class] ndarray (n,5) float32 [score, cx/W, cy/H, w/W, h/H]`` with empty ``(0,5)``
arrays for classes without detections (:119-135).  The graph is the MI355X-native Cascade R-CNN of cascade_rcnn.py
instead of a detectron2 model; ``predict_device`` is the device-resident form used by the end-to-end pipeline.
"""
import os

import numpy as np
import torch
from torch.nn import Module

from . import ops
from .cascade_rcnn import CascadeRCNN, PIXEL_MEAN, PIXEL_STD

WAYMO_CLASSNAMES = ('vehicle', 'pedestrian', 'sign', 'cyclist')      # category ids 1..4 (waymo_to_coco.py:19,36)


# the detectron2 model-zoo checkpoint the reference downloads for pretrained='coco' (logs/12442/job.log:1226); there is no
# network here: the file has to be provided
COCO_CHECKPOINT = 'Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv/18131413/model_0039999_e76410.pkl'


def find_pretrained_checkpoint():
    """$WAYMO_DETECTRON2_WEIGHTS (a file), or the detectron2 / fvcore cache locations of the model-zoo file."""
    cands = [os.environ.get('WAYMO_DETECTRON2_WEIGHTS', '')]
    for root in (os.environ.get('FVCORE_CACHE', ''), os.path.expanduser('~/.torch/fvcore_cache'), os.path.expanduser('~/.torch/iopath_cache')):
        if root:
            cands.append(os.path.join(root, 'detectron2', COCO_CHECKPOINT))
    for c in cands:
        if c and os.path.isfile(c):
            return c
    return None


class Detectron2Det(Module):
    """detectron2_det/__init__.py:63-68, same constructor arguments.  `pretrained == 'coco'` loads the detectron2 model-zoo
    checkpoint (found through find_pretrained_checkpoint(); a missing file raises - there is no silent random-weight
    fallback); any other value builds the graph with seeded random weights (benchmarks, tests, or a later load_state_dict).
    `freeze_pretrained` = cfg.MODEL.BACKBONE.FREEZE_AT (:44)."""

    def __init__(self, arch='Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml', classnames=WAYMO_CLASSNAMES,
                 freeze_pretrained=2, frozen_bn=True, pretrained=False, seed=0):
        super().__init__()
        if 'X_152' not in arch:
            raise NotImplementedError('only the Cascade R-CNN X-152-32x8d-FPN dconv graph of the Waymo solution is built')
        # frozen_bn False = MODEL.RESNETS.NORM "BN" (:45): in eval mode BatchNorm applies its running statistics exactly like
        # FrozenBatchNorm, so inference is identical; TRAINING with trainable BatchNorm is not built (loss() raises)
        self.frozen_bn = bool(frozen_bn)
        self.arch = arch
        self.classnames = list(classnames) if classnames else list(WAYMO_CLASSNAMES)
        self.freeze_at = int(freeze_pretrained)
        self.model = CascadeRCNN(num_classes=len(self.classnames), seed=seed)
        self.tta_min_sizes = None
        if pretrained == 'coco':
            path = find_pretrained_checkpoint()
            if path is None:
                raise FileNotFoundError(
                    "pretrained='coco' needs the detectron2 checkpoint %s: set WAYMO_DETECTRON2_WEIGHTS to the file "
                    "(no network access here; random weights are only used when pretrained is not 'coco')" % COCO_CHECKPOINT)
            self.load_detectron2(path)

    def load_detectron2(self, path_or_state_dict, strict=None):
        """Load a detectron2 `.pkl` / `.pth` or a reference `{args, kwargs, state_dict}` file (or a state dict).  The COCO
        checkpoint has 80-class predictors: like DetectionCheckpointer the mismatching cls_score / bbox_pred tensors are
        skipped (left at their initialisation) when strict is None."""
        from . import weights
        sd = path_or_state_dict if isinstance(path_or_state_dict, dict) else weights.load_checkpoint_file(path_or_state_dict)
        sd = weights._strip(sd)
        if strict is None:
            want = {n: s for n, s, _ in weights.detectron2_layout(self.model.num_classes)}
            drop = [n for n in sd if n in want and list(sd[n].shape) != want[n] and '.box_predictor.' in n]
            for n in drop:
                del sd[n]
            mask_keys = [n for n in sd if n.startswith('roi_heads.mask_')]      # MASK_ON False (:54)
            for n in mask_keys:
                del sd[n]
            return weights.load_state_dict_detectron2(self.model, sd, strict=False)
        return weights.load_state_dict_detectron2(self.model, sd, strict=strict)

    def enable_tta(self, min_sizes=None):
        """detectron2_det/__init__.py:188-196: wrap the model in detectron2's GeneralizedRCNNWithTTA with TEST.AUG.FLIP False:
        predict() then runs one pass per TEST.AUG.MIN_SIZES entry (shortest edge resized, max size 4000 - detectron2's
        defaults (400, 500, 600, 700, 800, 900, 1000, 1100, 1200)) and merges all boxes with the final per-class NMS."""
        if self.tta_min_sizes is None:
            self.tta_min_sizes = list(min_sizes) if min_sizes else [400, 500, 600, 700, 800, 900, 1000, 1100, 1200]
            print(f'TEST.AUG.MIN_SIZES={self.tta_min_sizes}')

    def forward(self, x):
        """detectron2_det/__init__.py:70-74: RGB -> BGR (INPUT.FORMAT == 'BGR')."""
        return x[:, [2, 1, 0]]

    @torch.no_grad()
    def predict_device(self, x, scale=1.0, hflip=False, vflip=False):
        """(B,3,H,W) float RGB 0..255 or (B,H,W,3) uint8 RGB -> list of (boxes xyxy px, scores, classes) device tensors.
        One fused HIP kernel does TTA.pre_process (tta.py:147-190: resize by `scale`, flips), the RGB->BGR swap of
        forward() (:70-74), detectron2's normalisation and the padding to a multiple of 32; boxes are in pixels of the
        transformed (resized / flipped) image, like the reference's detector sees it."""
        dev = next(self.parameters()).device
        if self.tta_min_sizes:
            return self._predict_multiscale(x.to(dev), scale, hflip, vflip)
        xn, (ho, wo) = ops.preprocess(x.to(dev), scale, hflip, vflip, True, PIXEL_MEAN, PIXEL_STD, 32)
        self.last_input_size = (ho, wo)
        return [self.model.forward_normalized(xn[i:i + 1], ho, wo) for i in range(xn.shape[0])]

    @torch.no_grad()
    def predict_padded(self, x, scale=1.0, hflip=False, vflip=False):
        """One image, static shapes, no host synchronisation (hipGraph-capturable): (boxes (100,4), scores (100), classes (100),
        count int32[1]) in pixels of the transformed image; rows >= count are padding."""
        xn, (ho, wo) = ops.preprocess(x, scale, hflip, vflip, True, PIXEL_MEAN, PIXEL_STD, 32)
        self.last_input_size = (ho, wo)
        return self.model.forward_padded(xn, ho, wo)

    def predict_padded_bottom_up(self, x, scale=1.0, hflip=False, vflip=False):
        """First half of predict_padded: pre-processing + stem .. res5 -> [c2, c3, c4, c5] (static shapes)."""
        xn, (ho, wo) = ops.preprocess(x, scale, hflip, vflip, True, PIXEL_MEAN, PIXEL_STD, 32)
        self.last_input_size = (ho, wo)
        return self.model.backbone.bottom_up(xn)

    def predict_padded_heads(self, bottom_up_feats):
        """Second half: FPN, RPN, cascade heads, tail - same outputs as predict_padded."""
        ho, wo = self.last_input_size
        return self.model.forward_padded_from(bottom_up_feats, ho, wo)

    def _predict_multiscale(self, x, scale, hflip, vflip, max_size=4000):
        """enable_tta(): GeneralizedRCNNWithTTA restated (detectron2 0.1.3 modeling/test_time_augmentation.py, FLIP False): one
        detector pass per min size (ResizeShortestEdge), boxes rescaled to the base image, all detections merged by the model's
        own score threshold + per-class NMS + top-k.  Arithmetic unpinned (detectron2 absent), structure from the call site."""
        h0 = x.shape[2] if x.dtype != torch.uint8 else x.shape[1]
        w0 = x.shape[3] if x.dtype != torch.uint8 else x.shape[2]
        hb, wb = int(h0 * scale), int(w0 * scale)                 # the image the reference's TTA wrapper would be handed
        self.last_input_size = (hb, wb)
        out = []
        for i in range(x.shape[0]):
            bs, ss, cs = [], [], []
            for ms in self.tta_min_sizes:
                f = ms / float(min(hb, wb))
                if max(hb, wb) * f > max_size:
                    f = max_size / float(max(hb, wb))
                xn, (ho, wo) = ops.preprocess(x[i:i + 1], scale * f, hflip, vflip, True, PIXEL_MEAN, PIXEL_STD, 32)
                b, s, c = self.model.forward_normalized(xn, ho, wo)
                bs.append(b * torch.tensor([wb / wo, hb / ho, wb / wo, hb / ho], device=b.device)); ss.append(s); cs.append(c)
            b, s, c = torch.cat(bs), torch.cat(ss), torch.cat(cs)
            keep = ops.batched_nms(b, s, c.to(torch.int32), self.model.nms_thresh)[: self.model.topk] if s.numel() else s.new_zeros(0, dtype=torch.long)
            out.append((b[keep], s[keep], c[keep]))
        return out

    def criterion(self, args=None):
        """detectron2_det/__init__.py:141-142"""
        return self.loss

    def loss(self, images, target):
        """detectron2_det/__init__.py:144-186: images (B,3,H,W) RGB 0..255, target = {'labels': [LongTensor (n_i) 1-based],
        'boxes': [Tensor (n_i,4) xyxy pixels]} -> dict of loss tensors (summed over the batch / B)."""
        from . import training
        if not self.frozen_bn:
            raise NotImplementedError('training with MODEL.RESNETS.NORM = "BN" (frozen_bn=False) is not built: the solution trains with FrozenBN')
        x = self(images.to(next(self.parameters())))
        total = {}
        for i in range(x.shape[0]):
            boxes = target['boxes'][i].to(x.device).float()
            classes = (target['labels'][i].to(x.device).long() - 1)
            for k, v in training.losses(self.model, x[i:i + 1], boxes, classes).items():
                total[k] = total.get(k, 0) + v / x.shape[0]
        return total

    def predict(self, x, scale=1.0, hflip=False, vflip=False):
        """detectron2_det/__init__.py:76-139.  `scale / hflip / vflip` (extension) fold TTA.pre_process into the fused
        pre-processing kernel; the returned boxes are normalised in the transformed image, as if the reference's
        predict had been handed TTA.pre_process(x)."""
        single = False
        if not torch.is_tensor(x):                              # PIL image (:103-107)
            x = torch.as_tensor(np.float32(x).transpose(2, 0, 1)).unsqueeze(0)
            single = True
        results = self.predict_device(x, scale, hflip, vflip)
        h, w = self.last_input_size
        output = []
        for boxes, scores, classes in results:
            bbox_cls = []
            if scores.numel():
                bx = boxes.clone()
                bx[:, 0::2] *= 1.0 / w                           # Boxes.scale(1/W, 1/H) (:124)
                bx[:, 1::2] *= 1.0 / h
                center = (bx[:, :2] + bx[:, 2:]) / 2
                wh = bx[:, 2:4] - bx[:, 0:2]
                bbox = torch.cat((scores.unsqueeze(1), center, wh), dim=1)
                for c in range(len(self.classnames)):
                    bbox_cls.append(bbox[classes == c].cpu().numpy())
            else:
                bbox_cls = [np.empty((0, 5)) for _ in self.classnames]
            output.append(bbox_cls)
        return output[0] if single else output


class GraphLanePredictor:
    """Detectron2Det.predict for a STREAM of equally sized images (inference.py's loop, detnet/inference.py:151 of the reference), round 6: the detector
    pass of an image size is captured once per lane as a hipGraph with static shapes (predict_padded + the box normalisation of predict(), no host
    synchronisation) and `lanes` frames are in flight on separate streams - the serial tails of one frame run under the other frame's kernels
    (profiles/r06_inflight_ab.txt).  submit() enqueues an image and returns a ticket; collect(ticket) waits for THAT image only and returns what
    predict() returns for one image: per class an (n, 5) float32 array [score, cx, cy, w, h], normalised in the transformed image.
    One set of lanes per (height, width): the Waymo cameras have two sizes."""

    MAX_SIZES = 4          # distinct image sizes that get their own captured lanes (graph pools are a few GB each at 1920x1280); further sizes run eagerly

    def __init__(self, det, scale=1.0, hflip=False, vflip=False, lanes=2):
        self.det, self.pre = det, (float(scale), bool(hflip), bool(vflip))
        self.n_lanes = max(1, int(lanes))
        self._sets = {}
        self._turn = 0

    def _lane_set(self, h, w, dev):
        key = (h, w, dev)
        if key in self._sets:
            return self._sets[key]
        det, lanes = self.det, []
        saved, ops.EVENT_LOG = ops.EVENT_LOG, None
        torch.cuda.synchronize(dev)
        for _ in range(self.n_lanes):
            lane = dict(stream=torch.cuda.Stream(device=dev), gin=torch.zeros((1, h, w, 3), dtype=torch.uint8, device=dev), ticket=None)

            def core(lane=lane):
                boxes, scores, classes, cnt = det.predict_padded(lane['gin'], *self.pre)
                ho, wo = det.last_input_size
                bx = boxes.clone()                          # Detectron2Det.predict (:119-131): the same float32 operations in the same order
                bx[:, 0::2] *= 1.0 / wo
                bx[:, 1::2] *= 1.0 / ho
                center = (bx[:, :2] + bx[:, 2:]) / 2
                wh = bx[:, 2:4] - bx[:, 0:2]
                return torch.cat((scores.unsqueeze(1), center, wh), dim=1), classes, cnt

            with torch.cuda.stream(lane['stream']), torch.no_grad():
                for _ in range(2):                          # library kernel selection, per-stream scratch
                    core()
            lane['stream'].synchronize()
            lane['graph'] = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(lane['graph'], stream=lane['stream'], capture_error_mode='thread_local'):
                lane['out'] = core()
            lane['host'] = tuple(torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in lane['out'])
            lanes.append(lane)
        torch.cuda.synchronize(dev)
        ops.EVENT_LOG = saved
        self._sets[key] = lanes
        return lanes

    def submit(self, img):
        """img: (1, H, W, 3) or (H, W, 3) uint8 RGB on the device.  Returns a ticket for collect()."""
        if img.dim() == 3:
            img = img.unsqueeze(0)
        assert img.dtype == torch.uint8 and img.is_cuda and img.shape[0] == 1 and img.shape[3] == 3, 'GraphLanePredictor takes one uint8 HWC image on the GPU'
        if (img.shape[1], img.shape[2], img.device) not in self._sets and len(self._sets) >= self.MAX_SIZES:
            return dict(lane=None, done=None, result=self.det.predict(img, *self.pre)[0])      # a folder of many sizes: eager, synchronous
        lanes = self._lane_set(img.shape[1], img.shape[2], img.device)
        lane = lanes[self._turn % self.n_lanes]
        self._turn += 1
        if lane['ticket'] is not None and lane['ticket'].get('result') is None:
            self.collect(lane['ticket'])                    # its pinned buffers are about to be reused
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(img.device))          # whatever produced / touched the image on the caller's stream (auto-contrast)
        with torch.cuda.stream(lane['stream']):
            lane['stream'].wait_event(ready)
            lane['gin'].copy_(img)
            img.record_stream(lane['stream'])
            lane['graph'].replay()
            for h, d in zip(lane['host'], lane['out']):
                h.copy_(d, non_blocking=True)
            done = torch.cuda.Event()
            done.record(lane['stream'])
        ticket = dict(lane=lane, done=done, result=None)
        lane['ticket'] = ticket
        return ticket

    def collect(self, ticket):
        if ticket['result'] is None:
            ticket['done'].synchronize()
            bbox, classes, cnt = (t.numpy() for t in ticket['lane']['host'])
            k = int(cnt[0])
            bbox, classes = bbox[:k].copy(), classes[:k].copy()
            ticket['result'] = [bbox[classes == c] for c in range(len(self.det.classnames))]
        return ticket['result']


def detections_to_wire(boxes, scores, classes, width, height, out_width=None, out_height=None):
    """Device-side twin of Detectron2Det.predict (:119-131) + COCODetection.load_prediction
    (/root/reference/detnet/data/coco.py:229-252): the [x, y, w, h] integers, 5-decimal score and category id that
    the detection JSON carries between inference.py and tracking/track.py.  float32 box arithmetic, float64 scaling
    and truncation toward zero, exactly in that order.  (round(score, 5) is rint(score*1e5)/1e5 here; Python's
    correctly-rounded round() differs only on exact decimal half-ways.)"""
    if boxes.is_cuda:                                      # one launch (csrc/det_tail.hip wire_kernel), same arithmetic
        xywhs, cat = ops.detections_to_wire(boxes, scores, classes, None, width, height, out_width or width, out_height or height)
        return xywhs[:4].t(), xywhs[4], cat
    bx = boxes.clone()
    bx[:, 0::2] *= 1.0 / width
    bx[:, 1::2] *= 1.0 / height
    center = (bx[:, :2] + bx[:, 2:]) / 2
    wh = bx[:, 2:4] - bx[:, 0:2]
    # TTA: the boxes are normalised in the transformed (resized) image and scaled by the ORIGINAL size (tta.py: boxes are
    # normalised, so ResizeTTA needs no undo; coco.py:249 multiplies by the image's own width / height)
    scale = torch.tensor([out_width or width, out_height or height], dtype=torch.float64, device=boxes.device)
    c64 = center.double() * scale
    wh64 = wh.double() * scale
    lt = c64 - wh64 / 2
    xywh = torch.trunc(torch.cat((lt, wh64), dim=1))
    score = torch.round(scores.double() * 1e5) / 1e5
    return xywh, score, (classes + 1).to(torch.int32)
