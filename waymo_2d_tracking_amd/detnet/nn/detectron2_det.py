"""Detector wrapper - mirrors /root/reference/detnet/nn/detectron2_det/__init__.py (class Detectron2Det :63-139).

``predict(x)`` keeps the reference contract: ``x`` is a PIL image or a (B,3,H,W) float tensor 0..255 in RGB order;
the result is ``[per image][per class] ndarray (n,5) float32 [score, cx/W, cy/H, w/W, h/H]`` with empty ``(0,5)``
arrays for classes without detections (:119-135).  The graph is the MI355X-native Cascade R-CNN of cascade_rcnn.py
instead of a detectron2 model; ``predict_device`` is the device-resident form used by the end-to-end pipeline.
"""
import numpy as np
import torch
from torch.nn import Module

from . import ops
from .cascade_rcnn import CascadeRCNN, PIXEL_MEAN, PIXEL_STD

WAYMO_CLASSNAMES = ('vehicle', 'pedestrian', 'sign', 'cyclist')      # category ids 1..4 (waymo_to_coco.py:19,36)


class Detectron2Det(Module):
    def __init__(self, arch='Misc/cascade_mask_rcnn_X_152_32x8d_FPN_IN5k_gn_dconv.yaml', classnames=WAYMO_CLASSNAMES,
                 freeze_pretrained=2, frozen_bn=True, pretrained=False, seed=0):
        super().__init__()
        if 'X_152' not in arch:
            raise NotImplementedError('only the Cascade R-CNN X-152-32x8d-FPN dconv graph of the Waymo solution is built')
        self.arch = arch
        self.classnames = list(classnames)
        self.model = CascadeRCNN(num_classes=len(self.classnames), seed=seed)

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
        xn, (ho, wo) = ops.preprocess(x.to(dev), scale, hflip, vflip, True, PIXEL_MEAN, PIXEL_STD, 32)
        self.last_input_size = (ho, wo)
        return [self.model.forward_normalized(xn[i:i + 1], ho, wo) for i in range(xn.shape[0])]

    def criterion(self, args=None):
        """detectron2_det/__init__.py:141-142"""
        return self.loss

    def loss(self, images, target):
        """detectron2_det/__init__.py:144-186: images (B,3,H,W) RGB 0..255, target = {'labels': [LongTensor (n_i) 1-based],
        'boxes': [Tensor (n_i,4) xyxy pixels]} -> dict of loss tensors (summed over the batch / B)."""
        from . import training
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


def detections_to_wire(boxes, scores, classes, width, height, out_width=None, out_height=None):
    """Device-side twin of Detectron2Det.predict (:119-131) + COCODetection.load_prediction
    (/root/reference/detnet/data/coco.py:229-252): the [x, y, w, h] integers, 5-decimal score and category id that
    the detection JSON carries between inference.py and tracking/track.py.  float32 box arithmetic, float64 scaling
    and truncation toward zero, exactly in that order.  (round(score, 5) is rint(score*1e5)/1e5 here; Python's
    correctly-rounded round() differs only on exact decimal half-ways.)"""
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
