"""EXPLORATORY: the res4 deformable conv as three bf16 MFMAs per tile and tap (2-way bfloat16 split of weights and
samples, f32 accumulation) instead of sixteen f32 MFMAs.  NOT fp32 (about 16 mantissa bits per operand): never the benchmarked value.
Reports (a) kernel time and output error against the fp32 kernel, (b) the drift of the detector's end boxes / scores on synthetic
frames when all 35 res4 layers run in that mode.  Each mode runs in its own process (the switch is read once per process).
    python tools/bf16x3_experiment.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from waymo_2d_tracking_amd.detnet.nn import ops
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det
    torch.manual_seed(0)
    C, H, W = 1024, 80, 120
    x = torch.randn(1, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, 32, 3, 3, device='cuda') / (3 * 32 ** 0.5)
    pw = ops.deform_pack_weight(w, 32)
    out = {}
    for std in (0.2, 1.0):
        off = (torch.randn(1, 18, H, W, device='cuda') * std).contiguous(memory_format=torch.channels_last)
        f = lambda: ops.deform_conv3x3(x, off, pw, 32, 1, 1)
        y = f()
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            f()
        e1.record()
        torch.cuda.synchronize()
        out['res4_std%.1f' % std] = dict(us=e0.elapsed_time(e1) / 30 * 1e3, y=y.double().cpu())
    # end boxes of the whole detector on two synthetic frames (small enough for a quick run, all 35 res4 layers included)
    m = Detectron2Det(seed=0).cuda().eval()
    g = torch.Generator().manual_seed(1)
    dets = []
    for i in range(2):
        img = torch.randint(0, 256, (1, 3, 640, 960), generator=g).float().cuda()
        boxes, scores, classes = m.predict_device(img)[0]
        dets.append(dict(boxes=boxes.double().cpu(), scores=scores.double().cpu(), classes=classes.cpu()))
    torch.save(dict(kernel=out, dets=dets), sys.argv[2])


def main():
    res = {}
    for mode in ('0', '1'):
        path = '/tmp/bf16x3_%s.pt' % mode
        env = dict(os.environ, WD_DEFORM_BF16X3=mode, WT_EXPERIMENT='1')
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', path], env=env, check=True)
        import torch
        res[mode] = torch.load(path)
    import torch
    rep = {}
    for k in res['0']['kernel']:
        a, b = res['0']['kernel'][k], res['1']['kernel'][k]
        d = (a['y'] - b['y']).abs()
        rep[k] = dict(fp32_us=a['us'], bf16x3_us=b['us'], speedup=a['us'] / b['us'], max_abs_diff=float(d.max()),
                      max_rel_to_output_rms=float(d.max() / a['y'].pow(2).mean().sqrt()), rms_diff=float(d.pow(2).mean().sqrt()))
    drift = []
    for a, b in zip(res['0']['dets'], res['1']['dets']):
        n = min(len(a['boxes']), len(b['boxes']))
        # greedy match by IoU, same class
        ab, bb = a['boxes'], b['boxes']
        area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
        lt = torch.max(ab[:, None, :2], bb[None, :, :2]); rb = torch.min(ab[:, None, 2:], bb[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        iou = inter / (area(ab)[:, None] + area(bb)[None, :] - inter)
        iou[a['classes'][:, None] != b['classes'][None, :]] = 0
        best, idx = iou.max(1)
        ok = best > 0.5
        drift.append(dict(n_fp32=len(ab), n_bf16x3=len(bb), matched=int(ok.sum()),
                          max_box_drift_px=float((ab[ok] - bb[idx[ok]]).abs().max()) if ok.any() else None,
                          mean_box_drift_px=float((ab[ok] - bb[idx[ok]]).abs().mean()) if ok.any() else None,
                          max_score_drift=float((a['scores'][ok] - b['scores'][idx[ok]]).abs().max()) if ok.any() else None))
    rep['end_boxes_960x640'] = drift
    rep['note'] = ('2-way bfloat16 split (hi.hi + hi.lo + lo.hi), f32 accumulate: about 16 mantissa bits per operand - not fp32; an exploratory '
                   'secondary line, never the benchmarked value; north_star tolerance on boxes / scores is 1e-4')
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child()
    else:
        main()
