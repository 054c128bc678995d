"""VOC-style AP evaluation of detections - the tool for mAP-parity checks.

Restates, on arrays, the evaluation the reference runs after inference (`--eval`) and during training:
  * /root/reference/detnet/data/metric.py: `compute_truth_and_false_positive` (:143-270), `update_tp_fp` (:274-301), `voc_ap`
    (:107-140), `f2_score` (:304-320), `voc_eval` (:31-104), `evaluate_detections` (:323-401);
  * /root/reference/data/__init__.py:9-84 `metric_fun`: the Waymo variant (IoU 0.7 for `vehicle`, 0.5 for the other classes,
    AP / AR over all box sizes);
  * /root/reference/detnet/data/coco.py:54-118 `COCOAnnotationTransform`: ground truth = float32 [x1, y1, x2, y2] / image size,
    duplicate rows removed (np.unique).
Pinned by tests/golden/metric_g8.json, which oracle/gen_golden_metric.py produced by running those reference functions.

One deliberate difference: the reference sorts each image's CONFIDENCES but appends the TP / FP flags in
input order, so flags and confidences are mis-paired unless the detections already arrive score-sorted (detectron2's do).
Here detections are matched in descending-confidence order and the flags follow that order; on score-sorted input the
two agree exactly (the fixture's case), on unsorted input this tool gives the intended VOC result.
"""
import json

import numpy as np

EPS = np.finfo(np.float64).eps
SIZE_BUCKETS = {'S': (None, 32 ** 2), 'M': (32 ** 2, 96 ** 2), 'L': (96 ** 2, None), '': (None, None)}


def voc_ap(rec, prec, use_07_metric=False):
    """metric.py:107-140: area under the monotone precision envelope (or the VOC-07 11-point average)."""
    rec, prec = np.asarray(rec, np.float64), np.asarray(prec, np.float64)
    if use_07_metric:
        pts = [(prec[rec >= t].max() if (rec >= t).any() else 0.0) for t in np.arange(0., 1.1, 0.1)]
        return float(np.sum(pts) / 11.)
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    step = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def f2_score(npos, tp, fp, beta=2):
    """metric.py:304-320 (scalar form)."""
    if npos == 0:
        return 1.0 if fp == 0 else 0.0
    fn = npos - tp
    return float((1 + beta ** 2) * tp / ((1 + beta ** 2) * tp + beta ** 2 * fn + fp))


def load_ground_truth(annotations):
    """COCO-format dict / file -> (image ids sorted, {image_id: (width, height)}, {image_id: float64 (n, 5) [x1, y1, x2, y2, label]
    normalised}, classnames with 'background' at 0) like COCODetection + COCOAnnotationTransform (coco.py:54-118,121-170)."""
    if not isinstance(annotations, dict):
        with open(annotations) as fp:
            annotations = json.load(fp)
    cats = annotations['categories']
    classnames = ['background'] * (max(c['id'] for c in cats) + 1)
    for c in cats:
        classnames[c['id']] = c['name']
    sizes = {str(im['id']): (im['width'], im['height']) for im in annotations['images']}
    rows = {k: [] for k in sizes}
    for a in annotations['annotations']:
        if 'bbox' in a and a['category_id'] > 0:
            k = str(a['image_id'])
            w, h = sizes[k]
            b = np.asarray(a['bbox'], dtype=np.float32)
            b[2:] += b[:2]
            b /= np.asarray([w, h, w, h], dtype=np.float32)
            rows[k].append(np.concatenate((b.astype(np.float64), [a['category_id']])))
    gt = {}
    for k, r in rows.items():
        gt[k] = np.unique(np.asarray(r, np.float64).reshape(-1, 5), axis=0) if r else np.zeros((0, 5))
    return sorted(sizes), sizes, gt, classnames


def match_class(image_ids, sizes, gt, detections, label, ovthresh):
    """TP / FP flags of one class over the data set (metric.py:143-301).  detections: {image_id: (n, 5) [conf, cx, cy, w, h]}.
    Returns conf, det_size (pixels^2), pos_size, {thr: tp flags}, {thr: fp flags}, all in matching order."""
    conf, det_size, pos_size = [], [], []
    tp = {t: [] for t in ovthresh}
    fp = {t: [] for t in ovthresh}
    for image_id in image_ids:
        w, h = sizes[image_id]
        area_img = float(w * h)
        g = gt[image_id]
        g = g[g[:, 4].astype(int) == label][:, :4]
        g_size = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
        pos_size.append(g_size * area_img)
        d = detections.get(image_id)
        if d is None or len(d) == 0:
            continue
        d = np.asarray(d, np.float64).reshape(-1, 5)
        d = d[np.argsort(d[:, 0])[::-1]]                       # descending confidence; flags follow this order (D-10)
        conf.append(d[:, 0])
        area = d[:, 3] * d[:, 4]
        det_size.append(area * area_img)
        if len(g) == 0:
            for t in ovthresh:
                tp[t].append(np.zeros(len(d))); fp[t].append(np.ones(len(d)))
            continue
        x1, y1 = d[:, 1] - d[:, 3] / 2, d[:, 2] - d[:, 4] / 2
        x2, y2 = d[:, 1] + d[:, 3] / 2, d[:, 2] + d[:, 4] / 2
        iw = np.maximum(np.minimum(g[None, :, 2], x2[:, None]) - np.maximum(g[None, :, 0], x1[:, None]), 0.)
        ih = np.maximum(np.minimum(g[None, :, 3], y2[:, None]) - np.maximum(g[None, :, 1], y1[:, None]), 0.)
        inter = iw * ih
        iou = inter / (area[:, None] + g_size[None, :] - inter)
        jmax = iou.argmax(1)
        ovmax = iou[np.arange(len(d)), jmax]
        for t in ovthresh:
            hit = ovmax > t
            first = np.zeros(len(d), bool)                     # the first (most confident) detection that claims a ground truth
            idx = np.nonzero(hit)[0]
            _, f = np.unique(jmax[idx], return_index=True)
            first[idx[f]] = True
            tp[t].append((hit & first).astype(np.float64)); fp[t].append((~(hit & first)).astype(np.float64))
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)
    return cat(conf), cat(det_size), cat(pos_size), {t: cat(v) for t, v in tp.items()}, {t: cat(v) for t, v in fp.items()}


def _curves(conf, det_size, pos_size, tp, fp, buckets):
    order = np.argsort(conf)[::-1]
    det_size, tp, fp = det_size[order], tp[order], fp[order]
    out = {}
    for key, (low, high) in buckets.items():
        m = np.ones(len(det_size), bool)
        p = pos_size
        if low is not None:
            m &= det_size >= low; p = p[p >= low]
        if high is not None:
            m &= det_size < high; p = p[p < high]
        ctp, cfp = np.cumsum(tp[m]), np.cumsum(fp[m])
        npos = len(p)
        rec = ctp / np.maximum(npos, EPS)
        prec = ctp / np.maximum(ctp + cfp, EPS)
        out[key] = dict(ap=voc_ap(rec, prec), ar=float(rec[-1]) if len(rec) else float('nan'), T=npos,
                        f2=f2_score(npos, ctp[-1] if len(ctp) else 0, cfp[-1] if len(cfp) else 0), recall=rec, precision=prec)
    return out


def waymo_metric(image_ids, sizes, gt, detections, label, label_name):
    """data/__init__.py:9-84: IoU 0.7 for vehicle, 0.5 otherwise; the summary is the all-sizes entry."""
    thr = 0.7 if label_name == 'vehicle' else 0.5
    conf, dsz, psz, tp, fp = match_class(image_ids, sizes, gt, detections, label, (thr,))
    res = _curves(conf, dsz, psz, tp[thr], fp[thr], SIZE_BUCKETS)
    allsz = res['']
    return dict(ap=allsz['ap'], ar=allsz['ar'], T=allsz['T'], score=allsz['ap'], by_size={k: v['ap'] for k, v in res.items() if k})


def voc_eval(image_ids, sizes, gt, detections, label, ovthresh=(0.5,), size_ovthreshs=0.5):
    """metric.py:31-104: ap@t / ar@t per threshold (+ size buckets at size_ovthreshs), T, score = ap@0.5."""
    conf, dsz, psz, tp, fp = match_class(image_ids, sizes, gt, detections, label, tuple(ovthresh))
    summary = {}
    for t in ovthresh:
        buckets = SIZE_BUCKETS if t == size_ovthreshs else {'': (None, None)}
        for key, v in _curves(conf, dsz, psz, tp[t], fp[t], buckets).items():
            summary['ap@%s%s' % (t, key)] = v['ap']
            summary['ar@%s%s' % (t, key)] = v['ar']
    summary['T'] = len(psz)
    summary['score'] = summary['ap@0.5'] if 'ap@0.5' in summary else None
    return summary


def evaluate_detections(predictions, annotations, image_sizes=None, threshold=0.01, metric='waymo', print_fn=None):
    """metric.py:323-401 on a Predictions store (or {image_id: [per class (n, 5)]}): per class, detections with confidence >
    threshold (filter_detections :9-28), then the metric; returns {class: summary} (+ 'mean' / 'score' for metric='voc')."""
    image_ids, sizes, gt, gt_classnames = load_ground_truth(annotations)
    classnames = list(getattr(predictions, 'classnames', None) or gt_classnames[1:])
    if classnames and classnames[0] == 'background':
        classnames = classnames[1:]
    get = predictions.__getitem__ if hasattr(predictions, 'classnames') else predictions.get
    per_image = {k: get(k) for k in image_ids}
    evaluation = {}
    for i, cls in enumerate(classnames):
        if cls not in gt_classnames:
            continue
        dets = {}
        for k, v in per_image.items():
            if v is None:
                continue
            d = np.asarray(v[i]).reshape(-1, 5)
            dets[k] = d[d[:, 0] > threshold] if threshold > 0 and d.size else d
        label = gt_classnames.index(cls)
        evaluation[cls] = waymo_metric(image_ids, sizes, gt, dets, label, cls) if metric == 'waymo' else \
            voc_eval(image_ids, sizes, gt, dets, label, (0.5, 0.75))
    if metric == 'voc':
        mean = {k: float(np.mean([v[k] for v in evaluation.values()])) for k in ('ap@0.5', 'ar@0.5', 'T')}
        evaluation['mean'] = mean
        evaluation['score'] = mean['ap@0.5']
    if print_fn:
        for cls in classnames:
            if cls in evaluation:
                v = evaluation[cls]
                print_fn('%-12s ' % cls + ' '.join('%s %.4f' % (k, x) for k, x in v.items() if isinstance(x, (int, float))))
        aps = [v['ap'] if 'ap' in v else v['ap@0.5'] for k, v in evaluation.items() if k in classnames and v['T'] > 0]
        print_fn('* mean AP over classes with ground truth = %.4f' % (float(np.mean(aps)) if aps else float('nan')))
    return evaluation
