"""Whole-detector drift of the split-operand graph against the all-exact-f32 graph ("whole-detector boxes vs the exact-f32 graph").

    python tools/split_box_drift.py [--height 1280 --width 1920] [--seeds 3] [--out profiles/r05_split_box_drift.txt]

Same random-weight X-152 Cascade R-CNN, same random image, two graphs: cascade_rcnn.SPLIT_GEMM on (1x1 / dense 3x3 convolutions and the FC layers on the
3 x bf16 exact-split matrix-core kernel) and off (hipBLASLt f32 GEMMs / MIOpen f32 convolutions).  Three levels, so that a rounding difference is not
mistaken for a different experiment:
  1. FPN features p2..p6: max |a - b| / max |b|;
  2. the SAME proposal list (the exact graph's RPN output) through both cascades: decoded boxes (pixels) and mean class scores, row by row;
  3. end to end (each graph's own RPN top-k / NMS decisions): detections matched by nearest neighbour.
Neither graph is "the truth": both are float32 evaluations of the same network whose errors against float64 are of the same size (per-GEMM table:
profiles/r05_split_gemm_error.txt; the split kernel's is the smaller one).  The library graph is not run-to-run deterministic either (split-K atomics,
algorithm choice by timing), so the exact graph is also compared with ITSELF run twice - that row is the noise floor the other rows are read against.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from waymo_2d_tracking_amd.detnet.nn import cascade_rcnn
from waymo_2d_tracking_amd.detnet.nn.detectron2_det import Detectron2Det


def run(model, img, split, proposals=None):
    cascade_rcnn.SPLIT_GEMM = split
    inter = {}
    b, s, c = model(img, proposals=proposals, intermediates=inter)
    torch.cuda.synchronize()
    return (b, s, c), inter


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def match(b0, s0, c0, b1, s1, c1):
    """Nearest detection of the other graph (same class) for every detection: (max box distance px over the matched, max score diff, unmatched)."""
    if b0.shape[0] == 0 or b1.shape[0] == 0:
        return 0.0, 0.0, int(b0.shape[0] + b1.shape[0])
    d = (b0.double()[:, None, :] - b1.double()[None, :, :]).abs().amax(-1)
    d = d + (c0[:, None] != c1[None, :]).double() * 1e9
    dist, j = d.min(dim=1)
    ok = dist < 0.5
    worst = float(dist[ok].max()) if bool(ok.any()) else 0.0
    ds = float((s0.double()[ok] - s1.double()[j[ok]]).abs().max()) if bool(ok.any()) else 0.0
    return worst, ds, int((~ok).sum()) + abs(int(b0.shape[0]) - int(b1.shape[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=1280)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--seeds', type=int, default=3)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    lines = ['split-operand graph vs all-exact-f32 graph, %d x %d, random weights (Detectron2Det(seed)), random uint8 image' % (args.width, args.height),
             'rows "exact vs exact": the library graph against itself, second run (noise floor of a float32 evaluation whose GEMM algorithms use atomics)',
             '']
    worst = dict(feat=0.0, box=0.0, score=0.0, floor_box=0.0)
    for seed in range(args.seeds):
        m = Detectron2Det(seed=seed).eval().cuda()
        net = m.model
        g = torch.Generator().manual_seed(100 + seed)
        img = torch.randint(0, 256, (1, 3, args.height, args.width), generator=g).float().cuda()
        (eb, es, ec), ei = run(net, img, False)
        (eb2, es2, ec2), ei2 = run(net, img, False)
        (sb, ss, sc), si = run(net, img, True)
        lines.append('seed %d: %d detections (exact), %d (split); %d proposals' % (seed, eb.shape[0], sb.shape[0], int(ei['n_proposals'].item())))
        for lvl in range(len(ei['feats'])):
            r, r0 = rel(si['feats'][lvl], ei['feats'][lvl]), rel(ei2['feats'][lvl], ei['feats'][lvl])
            worst['feat'] = max(worst['feat'], r)
            lines.append('  FPN p%d  split vs exact: max|d|/max|x| = %.3e      exact vs exact: %.3e' % (lvl + 2, r, r0))
        n = int(ei['n_proposals'].item())
        props = ei['proposals'][:n].clone()
        (_, _, _), pe = run(net, img, False, proposals=props)
        (_, _, _), pe2 = run(net, img, False, proposals=props)
        (_, _, _), ps = run(net, img, True, proposals=props)
        db = float((ps['boxes'].double() - pe['boxes'].double()).abs().max())
        db0 = float((pe2['boxes'].double() - pe['boxes'].double()).abs().max())
        dsc = float((ps['scores'].double() - pe['scores'].double()).abs().max())
        dsc0 = float((pe2['scores'].double() - pe['scores'].double()).abs().max())
        med = float((ps['boxes'].double() - pe['boxes'].double()).abs().amax(-1).median())
        worst['box'], worst['score'], worst['floor_box'] = max(worst['box'], db), max(worst['score'], dsc), max(worst['floor_box'], db0)
        lines.append('  same %d proposals through the 3 cascade stages: decoded boxes max |d| = %.3e px (median row %.3e px), scores max |d| = %.3e'
                     % (n, db, med, dsc))
        lines.append('  %s exact vs exact:                           decoded boxes max |d| = %.3e px,                        scores max |d| = %.3e'
                     % (' ' * len(str(n)), db0, dsc0))
        w, ds, un = match(sb, ss, sc, eb, es, ec)
        w0, ds0, un0 = match(eb2, es2, ec2, eb, es, ec)
        lines.append('  end to end (own RPN / NMS decisions): matched detections max |d| = %.3e px, scores %.3e, unmatched %d      exact vs exact: %.3e px, %.3e, %d'
                     % (w, ds, un, w0, ds0, un0))
        del m, net
        torch.cuda.empty_cache()
    lines += ['', 'worst over seeds: FPN features %.3e relative, same-proposal boxes %.3e px (exact-vs-exact floor %.3e px), scores %.3e'
              % (worst['feat'], worst['box'], worst['floor_box'], worst['score']),
              'float32 spacing at a 1000-px coordinate: 6.1e-05 px']
    text = '\n'.join(lines) + '\n'
    print(text)
    if args.out:
        with open(args.out, 'wt') as f:
            f.write(text)


if __name__ == '__main__':
    main()
