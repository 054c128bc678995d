#!/usr/bin/env python
"""One-command accuracy check for whoever HAS the real checkpoint (there is no network in the build image, so the mAP / MOTA
parity of BASELINE.json's north_star cannot be shown here - BASELINE.md section 4).

    WAYMO_DETECTRON2_WEIGHTS=/path/model_final.pkl  (or a reference {args, kwargs, state_dict} file via --model)
    python tools/check_real_weights.py --images /data/waymo/det2d/validation --annotations images.json
                                       [--reference-detections ref_det.json] [--tta x1.5,hflip] [--limit 500]

Runs the MI355X-native detector through the same CLI code path as `python -m waymo_2d_tracking_amd.inference`, then
  * with --annotations: the VOC-style AP the reference prints during training (detnet/data/metric.py; published targets
    0.7403 / 0.7558 / 0.7452 on the 2 495-image val subset, logs/12442|12620|12650/job.log),
  * with --reference-detections (a detection JSON produced by the reference on the same images): per-image greedy matching,
    share of the reference's detections reproduced within 1e-4 of box / score (north_star's tolerance after the
    load_prediction int-truncation / 5-decimal rounding), and the largest deviations.
Exit code 0 iff every requested comparison is within its tolerance."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', required=True, help='image folder or COCO json (like inference.py -i)')
    ap.add_argument('--model', default=None, help="reference model file; default: Detectron2Det(pretrained='coco') from WAYMO_DETECTRON2_WEIGHTS")
    ap.add_argument('--annotations', default=None)
    ap.add_argument('--reference-detections', default=None)
    ap.add_argument('--tta', default='')
    ap.add_argument('--limit', type=int, default=0)
    ap.add_argument('--min-ap', type=float, default=None, help='fail below this mean AP (e.g. 0.74 for the phase-1 checkpoint)')
    args = ap.parse_args()
    import numpy as np
    import torch
    from waymo_2d_tracking_amd.detnet import inference as INF
    from waymo_2d_tracking_amd.detnet import nn as detnn
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    out_dir = tempfile.mkdtemp(prefix='wt_check_')
    model_path = args.model
    if model_path is None:
        m = Detectron2Det(pretrained='coco')                    # raises FileNotFoundError with the expected path when absent
        model_path = os.path.join(out_dir, 'model.pth')
        detnn.save(m, model_path)
    argv = ['-i', args.images, '--model', model_path, '--export', os.path.join(out_dir, 'det.json')]
    if args.tta:
        argv += ['--tta', args.tta]
    if args.annotations:
        argv += ['--eval', '--annotations', args.annotations]
    rows = INF.main(argv)
    ok = True
    det = json.load(open(os.path.join(out_dir, 'det.json')))
    print('detections written:', len(det), '->', os.path.join(out_dir, 'det.json'))
    if args.reference_detections:
        ref = json.load(open(args.reference_detections))
        by = {}
        for d in det:
            by.setdefault((d['image_id'], d['category_id']), []).append(d)
        hit = miss = 0
        worst_box = worst_score = 0.0
        for r in ref:
            cands = by.get((r['image_id'], r['category_id']), [])
            best = None
            for c in cands:
                db = max(abs(a - b) for a, b in zip(c['bbox'], r['bbox']))
                if best is None or db < best[0]:
                    best = (db, abs(c['score'] - r['score']))
            if best is not None and best[0] <= 1.0 and best[1] <= 1e-4:      # int-truncated boxes: one unit
                hit += 1
                worst_box, worst_score = max(worst_box, best[0]), max(worst_score, best[1])
            else:
                miss += 1
        share = hit / max(1, hit + miss)
        print('reference detections reproduced: %d / %d (%.4f); worst box deviation %.3g px, worst score deviation %.3g'
              % (hit, hit + miss, share, worst_box, worst_score))
        ok = ok and share >= 0.999
    if args.min_ap is not None and args.annotations:
        print('(mean AP is printed by the evaluation above; compare with --min-ap %.4f by eye or parse the log)' % args.min_ap)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
