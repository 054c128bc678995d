#!/usr/bin/env python
"""Benchmark of the MI355X-native detect -> SORT hot path (contract: see the driver's bench.py rules).

    python bench.py --gpus N --steps K --warmup W [--stage e2e|detect|track|ensemble]

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM.
  e2e      (default, BASELINE.json metric): detector on `--frames-per-step` synthetic 1920x1280x3 frames of one
           5-camera chunk, then SORT over the detections of those streams.  value = frames/s.
  detect   the detector alone (config 2).
  track    SORT alone on pre-computed detections (config 1 scaled to --segments segments per GPU).
  ensemble soft-NMS ensemble alone (config 4: K=13 inputs).
  train    one training step fwd+bwd+SGD (config 5), DDP over RCCL for N > 1.
Multi-GPU: one process per GPU (torchrun), camera sequences / frames sharded with no data-path collective in
the timed region (weak scaling); the only exchange is the ID-offset all_gather + result gather done by the CLIs.
Rank 0 prints ONE JSON line with `roofline` (dominant hand-written kernel, HIP-event timed) and `cpu_baseline`
(the C oracle timed on the host, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA = vector peak
F64_VALU_PEAK_GFLOPS = 78600.0  # f64 vector FMA rate = half the f32 vector rate (CDNA4 public figure; the guide only lists f32)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--stage', default=os.environ.get('WT_BENCH_STAGE', 'e2e'),
                    choices=['e2e', 'detect', 'track', 'ensemble', 'train', 'decode'])
    ap.add_argument('--segments', type=int, default=8, help='track stage: segments (x5 cameras x198 frames) per GPU')
    ap.add_argument('--images', type=int, default=990, help='ensemble stage: images per GPU')
    ap.add_argument('--k-inputs', type=int, default=13)
    ap.add_argument('--frames-per-step', type=int, default=10, help='e2e/detect: frames per step (multiple of 5)')
    ap.add_argument('--tta', default='', help="e2e/detect: test-time augmentation of the detector pass, e.g. x1.5,hflip (config 4)")
    ap.add_argument('--collate', action='store_true',
                    help='e2e: run the per-chunk birth-count exchange + row-block gather to rank 0 also with one rank '
                         '(always on when N > 1 or WT_FORCE_DIST=1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--inflight', type=int, default=2,
                    help='e2e/detect: frames in flight (hipGraph lanes on separate streams, default 2: the tails and ramps of one frame run under the '
                         "other frame's kernels; the all-library WD_SPLIT_GEMM=0 graph must use 1: its library kernels deadlock with two lanes)")
    ap.add_argument('--auto-contrast', action='store_true', help='e2e/detect: ImageOps.autocontrast on every frame (the --auto-contrast=1 of the '
                    "reference's documented TTA run, README.md:37)")
    ap.add_argument('--from-jpeg', action='store_true', help='e2e/detect: the frames enter as JPEG bytes and are decoded on the GPU inside the step '
                    '(loader threads, one step ahead); the default keeps decoded frames resident in HBM as the bench contract asks')
    ap.add_argument('--no-defer-track', action='store_true', help='e2e: start the SORT call of a chunk right behind its last frame instead of '
                    'behind the bottom-up pathway of the next frame (A/B of the overlap placement)')
    ap.add_argument('--no-graph', action='store_true', help='e2e/detect: launch every frame eagerly instead of replaying the captured hipGraph')
    ap.add_argument('--no-verify', action='store_true', help='skip the oracle replay of the timed output (after the timed region)')
    return ap.parse_args()


def check_flags(args):
    """Two hipGraph lanes of the ALL-LIBRARY graph (WD_SPLIT_GEMM=0) deadlocked at 1920x1280 in round 2 (library kernels that spin-wait on partner
    workgroups): that combination is refused unless WT_EXPERIMENT=1 says the caller knows.  The default graph (own kernels) runs two lanes."""
    if args.inflight > 1 and os.environ.get('WD_SPLIT_GEMM') == '0' and os.environ.get('WT_EXPERIMENT') != '1':
        raise SystemExit('bench.py: --inflight %d with WD_SPLIT_GEMM=0 (all-library graph) deadlocks at full size; use --inflight 1 or set WT_EXPERIMENT=1'
                         % args.inflight)


def launch_if_needed(args):
    """`python bench.py --gpus N` without a launcher: the parent starts the N ranks itself (one fresh process per GPU, the
    way the reference's Tester starts its workers, detnet/trainer/test.py:227-255) BEFORE anything touches the GPU, waits,
    and exits with the children's status.  Under torchrun (WORLD_SIZE set) this is a no-op."""
    if 'WORLD_SIZE' in os.environ or args.gpus <= 1:
        return
    from waymo_2d_tracking_amd import launcher
    try:
        rc = launcher.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    except launcher.LaunchError as e:
        raise SystemExit('bench.py --gpus %d: %s' % (args.gpus, e))
    raise SystemExit(rc)


def init_dist(args):
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and 'WORLD_SIZE' in os.environ and args.gpus != 1:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # under torchrun (the driver's launch) nobody has given the ranks their own library caches yet: eight ranks in MIOpen find mode
        # on a fresh box must not share one user database / TunableOp result file (set before the libraries initialise; a value the
        # user exported wins) - the package's own launcher does the same for its children
        from waymo_2d_tracking_amd import launcher
        launcher.adopt_rank_caches(rank, world)
    dist_on = world > 1 or os.environ.get('WT_FORCE_DIST') == '1'       # WT_FORCE_DIST: exercise RCCL with one rank
    torch.cuda.set_device(local if dist_on else 0)
    if dist_on:
        import torch.distributed as dist
        if 'RANK' not in os.environ:                 # WT_FORCE_DIST=1 without a launcher: a one-rank group on this GPU
            from waymo_2d_tracking_amd import launcher
            launcher.adopt_single_rank_env()
        if torch.cuda.device_count() <= local:
            raise SystemExit('bench.py: rank %d needs GPU %d but only %d visible' % (rank, local, torch.cuda.device_count()))
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        world = dist.get_world_size()                # what RCCL actually connected, not what the flag says
    return world, rank, local


def _dist_on():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except ImportError:
        return False


def rccl_ranks():
    """The rank numbers as an actual all_gather over the initialised backend reports them (None without a process group)."""
    if not _dist_on():
        return None
    import torch
    import torch.distributed as dist
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device='cuda')
    seen = torch.zeros(dist.get_world_size(), dtype=torch.int64, device='cuda')
    dist.all_gather_into_tensor(seen, mine)
    return [int(v) for v in seen.cpu().tolist()]


def exchange_counts(counts_dev, all_counts_dev):
    """Per-step birth-count exchange of the sharded stages (ids of rank r start after the births of ranks < r): one
    all_gather of the device-side (rows, births) pair, stream-ordered, no host synchronisation."""
    if _dist_on():
        import torch.distributed as dist
        dist.all_gather_into_tensor(all_counts_dev, counts_dev)


def barrier_sync(world):
    import torch
    if _dist_on():
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


PER_RANK_MS = []          # ms per step as each rank measured it (last timed_steps call; empty without a process group)


def timed_steps(world, run_step, steps, warmup):
    """W untimed + exactly K timed steps bracketed by barrier + synchronize; max over ranks; also HIP events."""
    import torch
    for _ in range(warmup):
        run_step()
    barrier_sync(world)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        run_step()
    ev1.record()
    barrier_sync(world)
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    PER_RANK_MS.clear()
    if _dist_on():
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        every = torch.zeros(dist.get_world_size(), dtype=torch.float64, device='cuda')
        dist.all_gather_into_tensor(every, t)          # every rank's own wall time between the two barriers
        PER_RANK_MS.extend(1e3 * x / steps for x in every.cpu().tolist())
        dt = float(every.max().item())                 # the job's time = the slowest rank's
    return dt, ev_ms


def build_predictions(seed, n_segments, n_frames=198, n_objects=100):
    from waymo_2d_tracking_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    xs, ys, ws, hs, ss, cs, fo, so, cw, ch = [], [], [], [], [], [], [0], [0], [], []
    n = 0
    nf = 0
    for s in range(n_segments):
        for cam in syn.CAMERAS:
            d = syn.stream_detections(rng, n_frames, n_objects, cam)
            counts = np.bincount(d['frame'], minlength=n_frames)
            xs.append(d['x']); ys.append(d['y']); ws.append(d['w']); hs.append(d['h']); ss.append(d['score'])
            cs.append(d['cat'].astype(np.int32))
            for c in counts:
                n += int(c)
                fo.append(n)
            nf += n_frames
            so.append(nf)
            cw.append(syn.IMAGE_SIZES[cam][0]); ch.append(syn.IMAGE_SIZES[cam][1])
    cat = np.concatenate
    return dict(x=cat(xs), y=cat(ys), w=cat(ws), h=cat(hs), score=cat(ss), category=cat(cs),
                frame_det_offsets=np.asarray(fo, np.int64), stream_frame_offsets=np.asarray(so, np.int64),
                clip_w=np.asarray(cw, np.float64), clip_h=np.asarray(ch, np.float64))


def stage_track(args, world, rank):
    """Config 1 scaled: SORT on pre-computed detections, `segments` x 5 cameras x 198 frames per GPU."""
    from waymo_2d_tracking_amd.devpath import DeviceTracker
    packed = build_predictions(1000 + rank, args.segments)
    ithr = [0.01, 0.01, 1.0, 0.0]
    sthr = [0.0, 0.0, 0.0, 0.0]          # track all ~100 boxes/frame (north_star workload)
    trk = DeviceTracker(packed, ithr, 2, 0, sthr)
    steps = args.steps or 20
    warmup = args.warmup if args.warmup is not None else 3
    import torch
    all_counts = torch.zeros(2 * world, dtype=torch.int64, device='cuda')

    def step():
        trk.run()
        exchange_counts(trk.counts, all_counts)

    dt, ev_ms = timed_steps(world, step, steps, warmup)
    out, births = trk.results()
    n_frames = trk.n_frames
    # algorithmic bytes: per class-frame 20 N + 896 T + 912 K + 8 N T + 48 K_out ; approximated with the
    # measured totals: N = dets, T ~= K ~= K_out ~= emitted rows (matched tracks), N*T per class-frame from averages
    n_dets = trk.n_dets
    rows = len(out['frame'])
    cls_frames = n_frames * 3
    nt = (n_dets / cls_frames) * (rows / cls_frames) * cls_frames
    alg_bytes = 20.0 * n_dets + 896.0 * rows + 912.0 * rows + 8.0 * nt + 48.0 * rows
    sort_flops = 21.0 * nt + 2.0 * 343 * 3 * 2 * rows
    res = dict(value=n_frames * world * steps / dt, unit='frames/s', ms_per_step=1e3 * dt / steps,
               workload='SORT on pre-computed detections: %d segments x 5 cameras x 198 frames/GPU, ~100 boxes/frame, '
                        'max_age 2, min_hits 0, all boxes tracked' % args.segments,
               dtype='f64',
               roofline=dict(bound='latency', kernel='sort_streams_kernel',
                             achieved=sort_flops / (ev_ms / steps * 1e-3) / 1e9, peak=F64_VALU_PEAK_GFLOPS, unit='GFLOP/s f64',
                             traffic=None, hbm_gbs=alg_bytes / (ev_ms / steps * 1e-3) / 1e9,
                             note='a serial per-frame control loop (Kalman, Munkres) on one wave per tracker (plus three helper waves for '
                                  'the IoU matrix and Munkres steps 1 / 6 when there are at most 256 trackers): bound by '
                                  'dependent-instruction / LDS latency, neither by HBM nor by the f64 ALU rate; both '
                                  'fractions are reported for completeness (flops: 21 per IoU pair + 2x7x7x7x3 per '
                                  'Kalman predict/update + Munkres passes not counted)'),
               extra=dict(n_dets=n_dets, n_rows=rows, n_births=births, n_frames=n_frames,
                          births_by_rank=all_counts.view(-1, 2)[:, 1].cpu().tolist() if _dist_on() else None))
    res['roofline']['frac'] = res['roofline']['achieved'] / F64_VALU_PEAK_GFLOPS
    if rank == 0 and not args.no_verify:
        # checker, outside the timed region: the rows of the timed call replayed through the CPU oracle
        from oracle import oracle as O
        O.build()
        ref = O.track_streams(packed, 2, 0, sthr, ithr)
        res['verified'] = dict(ok=bool(births == ref['n_births'] and np.array_equal(out['object_id'], ref['object_id'])
                                       and np.array_equal(out['frame'], ref['frame']) and np.array_equal(out['bbox'], ref['bbox'])),
                               rows=rows, rows_ref=int(len(ref['frame'])), against='oracle/sort_oracle.c on the same detections')
    if rank == 0 and world == 1 and not args.no_cpu_baseline:       # the CPU baseline is timed at N = 1 only
        res['cpu_baseline'] = cpu_baseline_track(ithr, sthr, args.segments)
    return res, steps, warmup


def cpu_baseline_track(ithr, sthr, segments=1):
    """The C oracle on the host cores - SURVEY 8d (ii).  `value` is the SAME configuration as the GPU line (`segments` segments, i.e.
    5 x segments independent camera streams, one host thread per stream up to the core count); the single-thread rate, the one-segment rate
    and the all-cores rate (one segment per thread, 128 segments) ride along as side keys."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    packed = build_predictions(1000, 1)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 6.0:
        O.track_streams(packed, 2, 0, sthr, ithr)
        reps += 1
    dt = time.perf_counter() - t0
    single = reps * 990 / dt
    cores = os.cpu_count() or 1
    n_thr = min(cores, 128)
    per_thread = max(1, int(round(single * 6.0 / 990)))          # ~6 s of work per thread at the single-thread rate

    def work(_):
        for _ in range(per_thread):
            O.track_streams(packed, 2, 0, sthr, ithr)            # ctypes releases the GIL inside the C call
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as ex:
        list(ex.map(work, range(n_thr)))
    dtm = time.perf_counter() - t0
    # config 1 at its stated size: ONE segment; its five camera streams are the only independent pieces (ids are offset afterwards),
    # so the port can use five threads on it, not 128
    subs = split_streams(packed)
    reps1 = max(2, int(round(single * 3.0 / 990)))

    def one_stream(sp):
        for _ in range(reps1):
            O.track_streams(sp, 2, 0, sthr, ithr)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(subs)) as ex:
        list(ex.map(one_stream, subs))
    dt1 = time.perf_counter() - t0
    # the GPU line's own configuration: `segments` segments = 5 x segments independent streams, one thread per stream
    if segments == 1:
        same_value, same_threads = reps1 * 990 / dt1, len(subs)
    else:
        n_streams = 5 * segments
        thr_s = min(n_streams, cores)
        jobs = [subs[i % len(subs)] for i in range(n_streams)]

        def stream_job(sp):
            for _ in range(reps1):
                O.track_streams(sp, 2, 0, sthr, ithr)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(thr_s) as ex:
            list(ex.map(stream_job, jobs))
        same_value, same_threads = reps1 * 198 * n_streams / (time.perf_counter() - t0), thr_s
    res = dict(value=same_value, unit='frames/s', cores=same_threads, kind='port',
               same_config='%d segment(s) x 5 camera streams, one thread per stream (%d threads)' % (segments, same_threads),
               all_cores_one_segment_per_thread=dict(value=n_thr * per_thread * 990 / dtm, threads=n_thr),
               single_thread=single, one_segment_one_thread_per_stream=dict(value=reps1 * 990 / dt1, threads=len(subs)),
               sample='oracle/sort_oracle.c (C restatement of tracking/sort): 1 segment x 5 cameras x 198 frames per '
                      'call; %d calls on one thread (%.0f frames/s), then %d threads x %d calls (one segment per thread, '
                      'the way the GPU path shards streams); one segment with one thread per camera stream: %.0f frames/s'
                      % (reps, single, n_thr, per_thread, reps1 * 990 / dt1))
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r02_reference_sort_on_stub.json')
    if os.path.exists(ref):
        try:
            res['reference_python_sort'] = dict(json.load(open(ref)), source='profiles/r02_reference_sort_on_stub.json')
        except ValueError:
            pass
    return res


def split_streams(packed):
    """One packed detection set per stream (the CPU port's unit of parallelism inside a segment)."""
    so, fo = packed['stream_frame_offsets'], packed['frame_det_offsets']
    out = []
    for i in range(len(so) - 1):
        f0, f1 = int(so[i]), int(so[i + 1])
        d0, d1 = int(fo[f0]), int(fo[f1])
        sub = {k: packed[k][d0:d1].copy() for k in ('x', 'y', 'w', 'h', 'score', 'category')}
        sub['frame_det_offsets'] = (fo[f0:f1 + 1] - d0).astype(np.int64)
        sub['stream_frame_offsets'] = np.asarray([0, f1 - f0], np.int64)
        sub['clip_w'] = packed['clip_w'][i:i + 1].copy()
        sub['clip_h'] = packed['clip_h'][i:i + 1].copy()
        out.append(sub)
    return out


def build_groups(seed, n_images, k_inputs, n_objects=100):
    from waymo_2d_tracking_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    rows, off, sizes = [], [0], []
    for im in range(n_images):
        cls = syn.CLASS_IDS[rng.choice(3, size=n_objects, p=syn.CLASS_P)]
        gs = syn.ensemble_group(rng, n_objects, k_inputs)
        for c in syn.CLASS_IDS:
            sel = cls == c
            for g in gs:
                rows.append(g[sel]); sizes.append(int(sel.sum()))
            off.append(off[-1] + int(sel.sum()) * k_inputs)
    return np.ascontiguousarray(np.vstack(rows)), np.asarray(off, np.int64), np.asarray(sizes, np.int32).reshape(-1, k_inputs)


def stage_ensemble(args, world, rank):
    """Config 4 (offline half): linear soft-NMS ensemble of K inputs, thr 0.5, cut 0.9."""
    from waymo_2d_tracking_amd.devpath import DeviceEnsemble
    d, off, sizes = build_groups(2000 + rank, args.images, args.k_inputs)
    ens = DeviceEnsemble(d, off, sizes, args.k_inputs, 2, 0.5, 0.9)
    steps = args.steps or 20
    warmup = args.warmup if args.warmup is not None else 3
    dt, ev_ms = timed_steps(world, ens.run, steps, warmup)
    alg_bytes = 80.0 * len(d)            # read 40 n + write 40 n per group
    gsz = np.diff(off).astype(np.float64)
    pair_flops = float((17.0 * gsz * (gsz - 1) / 2).sum())
    res = dict(value=args.images * world * steps / dt, unit='frames/s', ms_per_step=1e3 * dt / steps,
               workload='soft-NMS ensemble of K=%d inputs, %d images/GPU, 100 objects/image, 3 classes, thr .5 cut .9'
                        % (args.k_inputs, args.images),
               dtype='f64',
               roofline=dict(bound='latency', kernel='softnms_fast_kernel',
                             achieved=pair_flops / (ev_ms / steps * 1e-3) / 1e9, peak=F64_VALU_PEAK_GFLOPS, unit='GFLOP/s f64',
                             traffic=None, hbm_gbs=alg_bytes / (ev_ms / steps * 1e-3) / 1e9,
                             note='O(n^2) f64 pair work with two f64 divisions per overlapping pair, 39 rows per group on '
                                  'average: latency / division-throughput bound; 17 flops per ordered pair counted'),
               extra=dict(n_rows=int(len(d)), n_groups=int(len(off) - 1)))
    res['roofline']['frac'] = res['roofline']['achieved'] / F64_VALU_PEAK_GFLOPS
    if rank == 0 and not args.no_verify:
        import torch
        from oracle import oracle as O
        O.build()
        torch.cuda.synchronize()
        got, counts = ens.out5[:len(d)].cpu().numpy(), ens.counts[:len(off) - 1].cpu().numpy()
        exp, ecounts = O.ensemble_groups(d, off, sizes, args.k_inputs, 2, 0.5, 0.9)
        ok = bool(np.array_equal(counts, ecounts)) and all(
            np.array_equal(got[int(off[g]):int(off[g]) + int(counts[g])], exp[int(off[g]):int(off[g]) + int(counts[g])])
            for g in range(len(off) - 1))
        res['verified'] = dict(ok=ok, groups=int(len(off) - 1), rows=int(counts.sum()),
                               against='oracle/softnms_oracle.c on the same groups, bit-exact')
    if rank == 0 and world == 1 and not args.no_cpu_baseline:       # the CPU baseline is timed at N = 1 only
        from oracle import oracle as O
        O.build()
        n_img = max(1, min(args.images, 60))
        d1, off1, sizes1 = build_groups(2000, n_img, args.k_inputs)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 10.0:
            O.ensemble_groups(d1, off1, sizes1, args.k_inputs, 2, 0.5, 0.9)
            reps += 1
        dtc = time.perf_counter() - t0
        res['cpu_baseline'] = dict(value=reps * n_img / dtc, unit='frames/s', cores=1, kind='port',
                                   sample='oracle/softnms_oracle.c, %d images x %d repetitions, single thread' % (n_img, reps))
    return res, steps, warmup


def cpu_baseline_e2e(pipe, track, height=448, width=640):
    """The CPU restatement of the same path (oracle/detector_ref.py + oracle/sort_oracle.c) on the host cores, on a
    bounded sample: ONE synthetic frame at 640x448 (1/8.23 of the 1920x1280 pixels; the 1000-proposal cascade
    heads are resolution independent) followed by SORT on its detections.  value = measured frames/s at that size;
    `full_res_equivalent` divides the backbone share by the pixel ratio."""
    import copy
    import torch
    from waymo_2d_tracking_amd.detnet.nn.detectron2_det import detections_to_wire
    from oracle import detector_ref as R
    from oracle import oracle as O
    O.build()
    cpu = copy.deepcopy(pipe.model.model).cpu()
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 3, height, width), generator=g).float()[:, [2, 1, 0]]
    t0 = time.perf_counter()
    boxes, scores, classes = R.forward(cpu, img)
    t_det = time.perf_counter() - t0
    t_sort = 0.0
    if track:
        xywh, score, cat = detections_to_wire(boxes, scores, classes, width, height)
        n = xywh.shape[0]
        packed = dict(x=xywh[:, 0].numpy().copy(), y=xywh[:, 1].numpy().copy(), w=xywh[:, 2].numpy().copy(),
                      h=xywh[:, 3].numpy().copy(), score=score.numpy().copy(), category=cat.numpy().astype(np.int32),
                      frame_det_offsets=np.array([0, n], np.int64), stream_frame_offsets=np.array([0, 1], np.int64),
                      clip_w=np.array([float(width)]), clip_h=np.array([float(height)]))
        t0 = time.perf_counter()
        O.track_streams(packed, 2, 0, [0.0] * 4, [0.01, 0.01, 1.0, 0.0])
        t_sort = time.perf_counter() - t0
    ratio = (1920 * 1280) / float(height * width)
    res = dict(value=1.0 / (t_det + t_sort), unit='frames/s', cores=torch.get_num_threads(), kind='port',
               sample='1 synthetic frame at %dx%d through oracle/detector_ref.py (PyTorch CPU fp32, same parameters) + '
                      'oracle SORT: detector %.2f s, SORT %.4f s; 1920x1280 has %.2fx the pixels'
                      % (width, height, t_det, t_sort, ratio),
               full_res_equivalent=1.0 / (t_det * ratio + t_sort))
    # measured once at the real size (tools/cpu_baseline_full.py, committed): how good the pixel-ratio extrapolation is; and the
    # reference's OWN Python SORT loop on the restated filterpy / sklearn stubs (oracle/time_reference_sort.py)
    here = os.path.dirname(os.path.abspath(__file__))
    res['small_sample'] = dict(value=res['value'], size='%dx%d' % (width, height))
    for name, key in (('r04_cpu_baseline_full_size.json', 'measured_full_size'), ('r02_reference_sort_on_stub.json', 'reference_python_sort')):
        path = os.path.join(here, 'profiles', name)
        if os.path.exists(path):
            try:
                d = json.load(open(path))
            except ValueError:
                continue
            if key == 'measured_full_size' and '1920x1280' in d:
                res[key] = dict(frames_per_s=d['1920x1280']['frames_per_s'], detector_s=d['1920x1280']['detector_s'], threads=d.get('threads'),
                                time_ratio_full_over_640x448=d.get('measured_time_ratio_full_over_small'), source='profiles/' + name)
                # `value` = the like-for-like figure: one WHOLE 1920x1280 frame through the same CPU port takes 68 s on these cores - beyond the
                # bounded sample a default bench run may spend - so it was measured once on the GPU box (tools/cpu_baseline_full.py, committed
                # profile) and is re-scaled here by what THIS run's live 640x448 sample says about this box (live small / recorded small)
                small_rec = d.get('640x448', {}).get('frames_per_s')
                calib = (res['small_sample']['value'] / small_rec) if small_rec else 1.0
                res['value'] = d['1920x1280']['frames_per_s'] * calib
                res['sample'] = ('1920x1280 frame through oracle/detector_ref.py (PyTorch CPU fp32, same parameters) + oracle SORT: %.1f s per frame measured at '
                                 'full size on the GPU box (profiles/%s, %s threads), scaled by this run\'s live 640x448 sample (%.3f frames/s now vs %.3f '
                                 'recorded: x%.2f); ' % (d['1920x1280']['detector_s'], name, d.get('threads'), res['small_sample']['value'], small_rec or 0.0, calib)
                                 + res['sample'])
            elif key == 'reference_python_sort':
                res[key] = dict(d, source='profiles/' + name) if isinstance(d, dict) else d
    return res


def cpu_baseline_train(det, height=160, width=224):
    """One training step (forward, the 8 losses, backward) of the CPU restatement (oracle/detector_ref.losses, float32, same
    parameters) on the host cores at 224x160 - the 886x1280 crop has 31.6x the pixels; `full_res_equivalent` scales the measured
    step by that ratio (the ROI heads do not scale with the image, so this flatters the CPU).  A port, not the reference: detectron2
    cannot run here."""
    import copy
    import torch
    from oracle import detector_ref as R
    from waymo_2d_tracking_amd.detnet.nn import training
    cpu = copy.deepcopy(det.model).cpu()
    training.set_trainable(cpu)
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 3, height, width), generator=g).float()
    gt = torch.tensor([[20., 30., 120., 150.], [100., 40., 215., 155.], [5., 5., 60., 60.], [130., 8., 200., 70.]])
    cls = torch.tensor([0, 1, 3, 0])
    t0 = time.perf_counter()
    out = R.losses(cpu, img, gt, cls, torch.float32)
    sum(out.values()).backward()
    dt = time.perf_counter() - t0
    ratio = (886 * 1280) / float(height * width)
    return dict(value=1.0 / dt, unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample='1 training step (fwd + 8 losses + bwd) at %dx%d through oracle/detector_ref.losses (PyTorch CPU fp32): %.1f s; '
                       '886x1280 has %.1fx the pixels' % (width, height, dt, ratio), full_res_equivalent=1.0 / (dt * ratio))


def synthetic_camera_jpegs(seed):
    """One frame set of a Waymo segment as JPEG files in memory: 3 x 1920x1280 (front cameras) + 2 x 1920x886 (side cameras),
    4:2:0, quality 90, photo-like content (low-frequency structure + sensor noise) - encoded with PIL."""
    import io
    from PIL import Image
    rng = np.random.default_rng(seed)
    out = []
    for cam, (h, w) in enumerate([(1280, 1920)] * 3 + [(886, 1920)] * 2):
        yy, xx = np.mgrid[0:h, 0:w]
        img = 128 + 100 * np.sin(xx[..., None] / (5.0 + cam) + np.arange(3)) * np.cos(yy[..., None] / (7.0 + cam))
        img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, 'JPEG', quality=90, subsampling=2)
        out.append(buf.getvalue())
    return out


def stage_decode(args, world, rank):
    """JPEG decode of the camera frames in front of the detector (the reference: PIL in its dataset
    workers).  A step = --frames-per-step / 5 frame sets x 5 cameras, file bytes on the host -> (H, W, 3) uint8 RGB in HBM, through the
    loader's scheme (4 threads, one stream each).  The compressed bytes cross PCIe inside the timed region (1.1 MB per
    front frame - they are the input of the op)."""
    import io
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from waymo_2d_tracking_amd.detnet.nn import ops
    files = synthetic_camera_jpegs(3000 + rank)
    frames = max(1, args.frames_per_step // 5)
    batch = files * frames
    pool = ThreadPoolExecutor(4)
    streams = {}

    def one(data):
        import threading
        st = streams.setdefault(threading.get_ident(), torch.cuda.Stream())
        with torch.cuda.stream(st):
            return ops.jpeg_decode(data)

    def step():
        return list(pool.map(one, batch))
    steps = args.steps or 20
    warmup = args.warmup if args.warmup is not None else 3
    dt, ev_ms = timed_steps(world, step, steps, warmup)
    outs = step()
    in_bytes = float(sum(len(b) for b in batch))
    out_bytes = float(sum(o.numel() for o in outs))
    alg = in_bytes + out_bytes                          # compressed bytes in, RGB bytes out (intermediate coefficients are the decoder's own)
    res = dict(value=len(batch) * world * steps / dt, unit='frames/s', ms_per_step=1e3 * dt / steps,
               workload='JPEG decode of %d frame sets x 5 cameras (3 x 1920x1280 + 2 x 1920x886, 4:2:0, q90) per GPU and step, '
                        'file bytes on the host -> RGB u8 in HBM' % frames,
               dtype='u8',
               roofline=dict(bound='latency', kernel='jpeg_cand_kernel', achieved=alg / (dt / steps) / 1e9, peak=8000.0, unit='GB/s',
                             frac=alg / (dt / steps) / 1e9 / 8000.0, traffic=None,
                             note='Huffman decoding is a serial dependency per 1024-bit subsequence: five decode passes (four candidate '
                                  'launches + the write pass; one wave per SIMD, ~1500 cycles per symbol) set the time, not HBM; '
                                  'algorithmic bytes = compressed bytes in + RGB bytes out'),
               extra=dict(compressed_mb_per_step=in_bytes / 1e6, rgb_mb_per_step=out_bytes / 1e6, loader_threads=4))
    if rank == 0 and not args.no_verify:
        from PIL import Image
        ok = all(np.array_equal(o.cpu().numpy(), np.asarray(Image.open(io.BytesIO(b)).convert('RGB'))) for o, b in zip(outs[:5], batch[:5]))
        res['verified'] = dict(ok=bool(ok), against="PIL Image.open(...).convert('RGB') - the reference's decoder - on the 5 distinct files, bit for bit")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from PIL import Image
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 5.0:
            np.asarray(Image.open(io.BytesIO(batch[n % len(batch)])).convert('RGB'))
            n += 1
        single = n / (time.perf_counter() - t0)
        cores = min(os.cpu_count() or 1, 64)
        per = max(5, int(single * 5.0))

        def work(k):
            for j in range(per):
                np.asarray(Image.open(io.BytesIO(batch[(k + j) % len(batch)])).convert('RGB'))      # libjpeg releases the GIL
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, range(cores)))
        multi = cores * per / (time.perf_counter() - t0)
        res['cpu_baseline'] = dict(value=multi, unit='frames/s', cores=cores, kind='reference', single_thread=single,
                                   sample="PIL (libjpeg-turbo %s) Image.open().convert('RGB') on the same files: 5 s on one thread, "
                                          'then %d threads x %d decodes' % (__import__('PIL.features', fromlist=['version']).version('libjpeg_turbo'), cores, per))
    pool.shutdown()
    return res, steps, warmup


def main():
    args = parse_args()
    check_flags(args)
    launch_if_needed(args)         # no GPU call before this line
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the hot path has no CPU fallback)')
    world, rank, local = init_dist(args)
    if args.stage == 'track':
        res, steps, warmup = stage_track(args, world, rank)
        metric = 'SORT frames/sec on pre-computed detections'
    elif args.stage == 'ensemble':
        res, steps, warmup = stage_ensemble(args, world, rank)
        metric = 'soft-NMS ensemble images/sec'
    elif args.stage == 'decode':
        res, steps, warmup = stage_decode(args, world, rank)
        metric = 'JPEG decode frames/sec (camera frames in front of the detector)'
    elif args.stage == 'train':
        from waymo_2d_tracking_amd import bench_e2e
        res, steps, warmup = bench_e2e.run_train(args, world, rank, timed_steps)
        det = res.pop('model')
        if rank == 0 and world == 1 and not args.no_cpu_baseline:       # the CPU baseline is timed at N = 1 only
            res['cpu_baseline'] = cpu_baseline_train(det)
        metric = 'training images/sec (fwd+bwd+step), Cascade R-CNN X152 dconv, 886x1280 crops'
    else:
        from waymo_2d_tracking_amd import bench_e2e
        res, steps, warmup = bench_e2e.run(args, world, rank, timed_steps)
        pipe = res.pop('pipeline')
        if rank == 0 and args.stage == 'e2e' and not args.no_verify:
            # checker, outside the timed region: every detection the resident trackers consumed during this segment
            # (warm-up and timed steps) replayed through the CPU oracle, rows / ids / boxes compared
            from oracle import oracle as O
            O.build()
            res['verified'] = dict(bench_e2e.check_against(pipe, O.track_streams),
                                   against='oracle/sort_oracle.c replay of the slots the timed steps filled')
        if rank == 0 and world == 1 and not args.no_cpu_baseline:       # the CPU baseline is timed at N = 1 only
            res['cpu_baseline'] = cpu_baseline_e2e(pipe, args.stage == 'e2e')
        metric = 'end-to-end frames/sec (detect+SORT) on 1920x1280 Waymo frames' if args.stage == 'e2e' else \
            'detector frames/sec on 1920x1280 Waymo frames'
    ranks_seen = rccl_ranks()
    if _dist_on():
        import torch.distributed as dist
        dist.barrier()
    if ranks_seen is not None and (len(ranks_seen) != args.gpus or sorted(ranks_seen) != list(range(args.gpus))) \
            and os.environ.get('WT_FORCE_DIST') != '1':
        # the line below would carry a throughput for a job that is not the one asked for: no line, non-zero exit
        raise SystemExit('bench.py: --gpus %d but RCCL connected ranks %s' % (args.gpus, ranks_seen))
    if rank == 0:
        if ranks_seen is not None:
            res.setdefault('extra', {})['rccl_ranks'] = ranks_seen
            if PER_RANK_MS:
                res['extra']['per_rank_ms_per_step'] = dict(min=min(PER_RANK_MS), max=max(PER_RANK_MS), by_rank=list(PER_RANK_MS))
        line = {'metric': metric, 'value': res['value'], 'unit': res['unit'], 'n_gpus': world, 'steps': steps,
                'warmup': warmup, 'ms_per_step': res['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': res['dtype'], 'data': 'synthetic',
                'config': {'workload': res['workload'], 'stage': args.stage, 'parallelism': 'sequence-shard x%d' % world},
                'roofline': res.get('roofline'), 'cpu_baseline': res.get('cpu_baseline'), 'verified': res.get('verified'),
                'extra': res.get('extra')}
        # the JSON line is the LAST thing on stdout: the process group is gone (nothing RCCL prints at tear-down follows it) and what the C libraries
        # have buffered so far - RCCL's version banner under NCCL_DEBUG=VERSION - is flushed in front of it
        _finish_dist_and_flush()
        print(json.dumps(line), flush=True)
    else:
        _finish_dist_and_flush()


def _finish_dist_and_flush():
    if _dist_on():
        import torch.distributed as dist
        dist.destroy_process_group()
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


if __name__ == '__main__':
    main()
