"""The TIMED paths of bench.py against the CPU oracle: the device-resident (`*_dev`) entry points, the streaming tracker
(wt_track_state_* / wt_track_chunk_dev) and the whole DetectTrackPipeline (detector -> wire conversion -> frame-slotted
layout with category-0 holes -> SORT on a side stream -> per-segment history).  IDs / order / boxes bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _assert_rows_equal(got, ref, births=None):
    assert np.array_equal(got['object_id'], ref['object_id'])
    assert np.array_equal(got['frame'], ref['frame'])
    assert np.array_equal(got['category'], ref['category'])
    assert np.array_equal(got['bbox'], ref['bbox'])
    np.testing.assert_allclose(got['score'], ref['score'], rtol=4e-16, atol=0)
    if births is not None:
        assert births == ref['n_births']


def _with_holes(packed, rng, slots):
    """Re-lay a packed set of streams into fixed `slots`-wide frames with category-0 holes at random positions
    (the layout the end-to-end pipeline feeds to wt_track_streams_dev / wt_track_chunk_dev)."""
    nf = packed['frame_det_offsets'].size - 1
    out = {k: np.zeros(nf * slots, dtype=packed[k].dtype) for k in ('x', 'y', 'w', 'h', 'score', 'category')}
    fo = packed['frame_det_offsets']
    for f in range(nf):
        k = int(fo[f + 1] - fo[f])
        assert k <= slots
        pos = np.sort(rng.choice(slots, size=k, replace=False)) + f * slots
        for key in out:
            out[key][pos] = packed[key][fo[f]:fo[f + 1]]
    holes = dict(packed)
    holes.update(out)
    holes['frame_det_offsets'] = (np.arange(nf + 1, dtype=np.int64) * slots)
    return holes


def _synthetic_packed(seed, n_segments=2, n_frames=25, n_objects=30):
    from waymo_2d_tracking_amd import synthetic as syn
    from waymo_2d_tracking_amd.tracking import utils as T
    dets = syn.make_sequence_json(seed, n_segments=n_segments, n_frames=n_frames, n_objects=n_objects)
    predictions = {}
    for e in dets:
        seg, fr, cam = e['image_id'].split('/')
        predictions.setdefault(seg, {}).setdefault(cam, {}).setdefault(int(fr), []).append(
            {'bbox': e['bbox'], 'score': e['score'], 'category_id': e['category_id']})
    return T.pack_streams(predictions)


def test_track_streams_dev_with_category0_holes(oracle):
    """wt_track_streams_dev on the slotted layout (category 0 = empty slot) == oracle on the compacted detections."""
    from waymo_2d_tracking_amd.devpath import DeviceTracker
    sthr, ithr = [0.3, 0.2, 1.0, 0.1], [0.01, 0.01, 1.0, 0.0]
    packed = _synthetic_packed(7)
    ref = oracle.track_streams(packed, 2, 0, sthr, ithr)
    holes = _with_holes(packed, np.random.default_rng(0), 64)
    trk = DeviceTracker(holes, ithr, 2, 0, sthr)
    trk.run()
    out, births = trk.results()
    _assert_rows_equal(out, ref, births)
    assert len(out['frame']) > 500


@pytest.mark.parametrize('chunk_frames', [1, 3, 7])
def test_streaming_chunks_equal_one_shot(oracle, chunk_frames):
    """Feeding the frames of every stream chunk by chunk through the resident trackers == one pass over all frames
    (ids through wt_track_global_ids_dev), incl. streams that run out of frames before the others."""
    import torch
    from waymo_2d_tracking_amd.devpath import StreamingTracker
    sthr, ithr = [0.3, 0.2, 1.0, 0.1], [0.01, 0.01, 1.0, 0.0]
    packed = _synthetic_packed(11, n_segments=1, n_frames=20, n_objects=25)
    # ragged: drop the last frames of some streams
    so, fo = packed['stream_frame_offsets'], packed['frame_det_offsets']
    ns = so.size - 1
    lens = [int(so[s + 1] - so[s]) - (s % 3) * 2 for s in range(ns)]
    keep_frames = np.concatenate([np.arange(so[s], so[s] + lens[s]) for s in range(ns)])
    sel = np.concatenate([np.arange(fo[f], fo[f + 1]) for f in keep_frames]) if len(keep_frames) else np.zeros(0, int)
    rag = {k: packed[k][sel] for k in ('x', 'y', 'w', 'h', 'score', 'category')}
    cnt = np.diff(fo)[keep_frames]
    rag['frame_det_offsets'] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    rag['stream_frame_offsets'] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rag['clip_w'], rag['clip_h'] = packed['clip_w'], packed['clip_h']
    ref = oracle.track_streams(rag, 2, 0, sthr, ithr, id_base=100)
    max_frame = int(cnt.max())
    dev = torch.device('cuda')
    trk = StreamingTracker(ns, max_frame, ns * chunk_frames, ns * chunk_frames * max_frame, ithr, 2, 0, sthr)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    cw, ch = t(rag['clip_w'], torch.float64), t(rag['clip_h'], torch.float64)
    rows = {k: [] for k in ('frame', 'stream', 'category', 'bbox', 'score', 'lid')}
    births = 0
    rso, rfo = rag['stream_frame_offsets'], rag['frame_det_offsets']
    for c0 in range(0, max(lens), chunk_frames):
        fr_idx, s_off = [], [0]
        for s in range(ns):
            a = min(c0, lens[s]); b = min(c0 + chunk_frames, lens[s])
            fr_idx += list(range(rso[s] + a, rso[s] + b))
            s_off.append(len(fr_idx))
        dsel = np.concatenate([np.arange(rfo[f], rfo[f + 1]) for f in fr_idx]) if fr_idx else np.zeros(0, int)
        f_off = np.concatenate([[0], np.cumsum([rfo[f + 1] - rfo[f] for f in fr_idx])]).astype(np.int64)
        n = len(dsel)
        of = torch.zeros(n + 1, dtype=torch.int64, device=dev); oc = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        ob = torch.zeros((n + 1, 4), dtype=torch.float64, device=dev); osc = torch.zeros(n + 1, dtype=torch.float64, device=dev)
        oi = torch.zeros(n + 1, dtype=torch.int64, device=dev); cn = torch.zeros(2, dtype=torch.int64, device=dev)
        trk.feed(t(rag['x'][dsel], torch.float64), t(rag['y'][dsel], torch.float64), t(rag['w'][dsel], torch.float64),
                 t(rag['h'][dsel], torch.float64), t(rag['score'][dsel], torch.float64), t(rag['category'][dsel], torch.int32),
                 t(f_off, torch.int64), t(np.asarray(s_off), torch.int64), cw, ch, of, oc, ob, osc, oi, cn)
        k, nb = [int(v) for v in cn.cpu().tolist()]
        assert k >= 0
        births += nb
        lf = of[:k].cpu().numpy()
        gf = np.asarray(fr_idx, np.int64)[lf] if k else np.zeros(0, np.int64)
        rows['frame'].append(gf)
        rows['stream'].append(np.searchsorted(rso, gf, side='right') - 1)
        rows['category'].append(oc[:k].cpu().numpy()); rows['bbox'].append(ob[:k].cpu().numpy())
        rows['score'].append(osc[:k].cpu().numpy()); rows['lid'].append(oi[:k].cpu().numpy())
    rows = {k: np.concatenate(v) for k, v in rows.items()}
    gid = trk.global_ids(t(rows['stream'], torch.int32), t(rows['lid'], torch.int64), 100).cpu().numpy()
    order = np.argsort(rows['stream'], kind='stable')
    got = dict(frame=rows['frame'][order], category=rows['category'][order], bbox=rows['bbox'][order],
               score=rows['score'][order], object_id=gid[order])
    _assert_rows_equal(got, ref, births)


def test_streaming_golden_g4(golden_dir):
    """The reference-generated G4 fixture through the streaming tracker, 5 frames per chunk."""
    import torch
    from waymo_2d_tracking_amd.devpath import StreamingTracker
    from waymo_2d_tracking_amd.tracking import utils as T
    exp = json.load(open(os.path.join(golden_dir, 'sort_g4_expected_a.json')))
    p = exp['params']
    predictions = T.read_data_file(os.path.join(golden_dir, 'sort_g4_input.json'), p['score_threshold'])
    packed = T.pack_streams(predictions)
    so, fo = packed['stream_frame_offsets'], packed['frame_det_offsets']
    ns = so.size - 1
    lens = np.diff(so)
    max_frame = int(np.diff(fo).max())
    CH = 5
    dev = torch.device('cuda')
    trk = StreamingTracker(ns, max_frame, ns * CH, ns * CH * max_frame, p['iou_threshold'], p['max_age'], p['min_hits'],
                           p['score_threshold'])
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    cw, ch = t(packed['clip_w'], torch.float64), t(packed['clip_h'], torch.float64)
    acc = {k: [] for k in ('frame', 'stream', 'category', 'bbox', 'score', 'lid')}
    for c0 in range(0, int(lens.max()), CH):
        fr_idx, s_off = [], [0]
        for s in range(ns):
            fr_idx += list(range(so[s] + min(c0, lens[s]), so[s] + min(c0 + CH, lens[s])))
            s_off.append(len(fr_idx))
        dsel = np.concatenate([np.arange(fo[f], fo[f + 1]) for f in fr_idx]).astype(np.int64)
        f_off = np.concatenate([[0], np.cumsum([fo[f + 1] - fo[f] for f in fr_idx])]).astype(np.int64)
        n = len(dsel)
        of = torch.zeros(n + 1, dtype=torch.int64, device=dev); oc = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        ob = torch.zeros((n + 1, 4), dtype=torch.float64, device=dev); osc = torch.zeros(n + 1, dtype=torch.float64, device=dev)
        oi = torch.zeros(n + 1, dtype=torch.int64, device=dev); cn = torch.zeros(2, dtype=torch.int64, device=dev)
        trk.feed(*[t(packed[k][dsel], torch.float64) for k in ('x', 'y', 'w', 'h', 'score')], t(packed['category'][dsel], torch.int32),
                 t(f_off, torch.int64), t(np.asarray(s_off), torch.int64), cw, ch, of, oc, ob, osc, oi, cn)
        k = int(cn[0].item())
        gf = np.asarray(fr_idx, np.int64)[of[:k].cpu().numpy()]
        acc['frame'].append(gf); acc['stream'].append(np.searchsorted(so, gf, side='right') - 1)
        acc['category'].append(oc[:k].cpu().numpy()); acc['bbox'].append(ob[:k].cpu().numpy())
        acc['score'].append(osc[:k].cpu().numpy()); acc['lid'].append(oi[:k].cpu().numpy())
    acc = {k: np.concatenate(v) for k, v in acc.items()}
    gid = trk.global_ids(t(acc['stream'], torch.int32), t(acc['lid'], torch.int64), 0).cpu().numpy()
    order = np.argsort(acc['stream'], kind='stable')
    out = dict(frame=acc['frame'][order], category=acc['category'][order], bbox=acc['bbox'][order],
               score=acc['score'][order], object_id=gid[order])
    got = T.format_tracks(packed, out)
    rows = lambda tr: [(r['image_id'], r['category_id'], r['object_id']) for r in tr]
    assert rows(got) == rows(exp['tracks'])
    gb = np.array([r['bbox'] + [r['score']] for r in got]); eb = np.array([r['bbox'] + [r['score']] for r in exp['tracks']])
    np.testing.assert_allclose(gb, eb, rtol=0, atol=1e-6)


@pytest.mark.parametrize('method,k_inputs', [(2, 3), (2, 13), (1, 3), (0, 3)])
def test_ensemble_groups_dev_vs_oracle(oracle, method, k_inputs):
    """wt_ensemble_groups_dev (the entry point bench.py --stage ensemble times) bit-exact against the oracle."""
    import bench
    from waymo_2d_tracking_amd.devpath import DeviceEnsemble
    d, off, sizes = bench.build_groups(5, 12, k_inputs, n_objects=60)
    ens = DeviceEnsemble(d, off, sizes, k_inputs, method, 0.5, 0.9)
    ens.run()
    import torch
    torch.cuda.synchronize()
    got = ens.out5[:len(d)].cpu().numpy()
    counts = ens.counts[:len(off) - 1].cpu().numpy()
    exp, ecounts = oracle.ensemble_groups(d, off, sizes, k_inputs, method, 0.5, 0.9)
    assert np.array_equal(counts, ecounts)
    for g in range(len(off) - 1):
        a, n = int(off[g]), int(counts[g])
        assert np.array_equal(got[a:a + n], exp[a:a + n]), g


def test_detect_track_pipeline_vs_oracle(oracle):
    """bench.py's DetectTrackPipeline (small frames, random-init detector): >= 3 steps on the side stream, a segment
    wrap (trackers reset, slots reused), each time the rows of the whole segment equal the oracle's replay."""
    import torch
    from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline, check_against
    pipe = DetectTrackPipeline(n_cameras=5, frames_per_camera=2, height=256, width=384, seed=3, segment_frames=8,
                               distinct_times=6)
    for _ in range(3):
        pipe.step(True)
    rep = check_against(pipe, oracle.track_streams)
    assert rep['ok'], rep
    assert rep['chunks'] == 3 and rep['frames'] == 30 and rep['rows'] > 0
    pipe.step(True)                     # chunk 4 of 4
    rep = check_against(pipe, oracle.track_streams)
    assert rep['ok'] and rep['chunks'] == 4, rep
    pipe.step(True); pipe.step(True)    # wrap: new segment, chunks 1-2
    rep = check_against(pipe, oracle.track_streams)
    assert rep['ok'] and rep['chunks'] == 2 and pipe.segments_done == 1, rep


def test_graph_replay_equals_eager_launches():
    """The captured per-frame hipGraph and the same frame launched eagerly fill the same slots BIT FOR BIT (the timed path
    replays the graph; the oracle checks above see only what the slots hold), with TTA folded in as well.  The pipelines run with
    `deterministic=True`: MIOpen's deterministic solvers without the find-mode search (whose winners may accumulate split-K
    partial sums with atomics) and the library-default GEMM picks, so two executions of the same launches are bit-identical
    and the comparison can be exact (round 2 accepted 90 %)."""
    import torch
    from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline
    saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, torch.cuda.tunable.is_enabled())
    try:
        for tta in ('', 'x1.5,hflip'):
            graph = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=256, width=384, seed=5, segment_frames=8,
                                        distinct_times=4, tta=tta, deterministic=True)
            eager = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=256, width=384, seed=5, segment_frames=8,
                                        distinct_times=4, tta=tta, model=graph.model, use_graph=False, deterministic=True)
            for _ in range(2):
                graph.step(True)
                torch.cuda.synchronize()                                # (shared scratch buffers: one pipeline in flight at a time)
                eager.step(True)
                torch.cuda.synchronize()
            assert graph._graph is not None and eager._graph is None
            assert int((graph.category[:2] != 0).sum()) == int((eager.category[:2] != 0).sum()) > 0
            assert torch.equal(graph.category[:2], eager.category[:2])
            assert torch.equal(graph.xywhs[:2], eager.xywhs[:2])
            assert graph.n_dets_total == eager.n_dets_total
            for c in range(2):                                       # and the tracker rows built from them
                k = int(graph.chunk_counts[c, 0])
                assert k == int(eager.chunk_counts[c, 0]) and torch.equal(graph.out_bbox[c][:k], eager.out_bbox[c][:k])
                assert torch.equal(graph.out_id[c][:k], eager.out_id[c][:k])
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved[0], saved[1]
        torch.cuda.tunable.enable(saved[2])


def test_two_pipelines_in_flight_on_different_streams_equal_serial_runs():
    """Two detector + tracker instances of one process enqueued on two streams WITHOUT a synchronisation in between (advisor, round 3:
    the per-stream scratch of the deformable sampling table, the ROIAlign order / flag scratch, the NMS and hipBLASLt workspaces exist
    for exactly this and no test had both in flight).  Deterministic library kernels, eager launches: the slots and tracker rows must
    equal those of the same two pipelines run one after the other.  (Captured graphs are replayed on the lane they were captured on:
    their scratch buffers are baked in per stream.)"""
    import torch
    from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline
    saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, torch.cuda.tunable.is_enabled())
    kw = dict(n_cameras=2, frames_per_camera=2, height=256, width=384, segment_frames=8, distinct_times=4, use_graph=False, deterministic=True)

    def snapshot(p):
        out = [p.category[:2].clone(), p.xywhs[:2].clone()]
        for c in range(2):
            k = int(p.chunk_counts[c, 0])
            out += [p.out_bbox[c][:k].clone(), p.out_id[c][:k].clone()]
        return out
    try:
        a, b = DetectTrackPipeline(seed=5, **kw), DetectTrackPipeline(seed=6, **kw)
        for _ in range(2):
            a.step(True); torch.cuda.synchronize()
            b.step(True); torch.cuda.synchronize()
        ref = snapshot(a) + snapshot(b)
        a2, b2 = DetectTrackPipeline(seed=5, **kw), DetectTrackPipeline(seed=6, **kw)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(2):
            with torch.cuda.stream(sa):
                a2.step(True)
            with torch.cuda.stream(sb):
                b2.step(True)                                   # enqueued while a2's kernels are still running
        torch.cuda.synchronize()
        got = snapshot(a2) + snapshot(b2)
        assert int((ref[0] != 0).sum()) > 0 and len(ref) == len(got)
        for r, g in zip(ref, got):
            assert r.shape == g.shape and torch.equal(r, g)
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved[0], saved[1]
        torch.cuda.tunable.enable(saved[2])


@pytest.mark.parametrize('size,steps', [((256, 384), 3), ((1280, 1920), 2)])
def test_two_frames_in_flight_fill_the_same_slots_as_one(oracle, size, steps):
    """Round 6: two hipGraph lanes on separate streams (bench.py --inflight 2, the default) against one lane: the detector is deterministic in this mode
    (own kernels + deterministic library picks), so every detection slot and every tracker row must be IDENTICAL - at a small size and at 1920x1280,
    where two lanes of the round-2 (library) graph deadlocked; the tracker rows also equal the oracle's replay."""
    import torch
    from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline, check_against
    saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, torch.cuda.tunable.is_enabled())
    kw = dict(n_cameras=2, frames_per_camera=2, height=size[0], width=size[1], segment_frames=8, distinct_times=4, use_graph=True, deterministic=True,
              defer_tracking=True)
    try:
        snaps = []
        for lanes in (1, 2):
            p = DetectTrackPipeline(seed=5, n_inflight=lanes, **kw)
            for _ in range(steps):
                p.step(True)
            p.flush()
            torch.cuda.synchronize()
            rep = check_against(p, oracle.track_streams)
            assert rep['ok'], rep
            snap = [p.category[:steps].clone(), p.xywhs[:steps].clone()]
            for c in range(steps):
                k = int(p.chunk_counts[c, 0])
                snap += [p.out_bbox[c][:k].clone(), p.out_id[c][:k].clone()]
            snaps.append(snap)
            del p
        assert int((snaps[0][0] != 0).sum()) > 0
        for a, b in zip(*snaps):
            assert a.shape == b.shape and torch.equal(a, b)
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved[0], saved[1]
        torch.cuda.tunable.enable(saved[2])


def test_pipeline_with_frames_entering_as_jpeg(oracle):
    """SURVEY 8f rank 3 inside the timed pipeline: the frames of step s + 1 are decoded on the GPU by loader threads while the
    detector works on step s.  Every frame slot holds exactly PIL's decode of its file when its step reads it (checked after
    each step, i.e. the decode-ahead never overwrites a slot that is still being read), the detector saw those frames (same
    slots as a pipeline whose resident frames ARE the decoded files), and the tracker rows equal the oracle's replay."""
    import io
    import torch
    from PIL import Image
    from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline, check_against
    saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, torch.cuda.tunable.is_enabled())
    try:
        pipe = DetectTrackPipeline(n_cameras=3, frames_per_camera=2, height=256, width=384, seed=3, segment_frames=12,
                                   distinct_times=6, deterministic=True)
        pipe.enable_jpeg_input(quality=85, workers=3)
        ref = DetectTrackPipeline(n_cameras=3, frames_per_camera=2, height=256, width=384, seed=3, segment_frames=12,
                                  distinct_times=6, model=pipe.model, deterministic=True)
        for t in range(pipe.n_times):
            for cam in range(pipe.nc):
                dec = np.asarray(Image.open(io.BytesIO(pipe.jpeg[t][cam])).convert('RGB'))
                ref.frames[t, cam] = torch.from_numpy(dec.copy()).cuda()
        pipe._await_decoded(range(pipe.n_times))                        # the decodes enable_jpeg_input() started
        torch.cuda.synchronize()
        pipe.frames.zero_()                                             # from here on nothing but the decoder fills the slots
        torch.cuda.synchronize()
        for s in range(5):                                              # 5 steps x 2 times: wraps the 6 time slots
            times = [(pipe.time + j) % pipe.n_times for j in range(pipe.fpc)]
            pipe.step(True)
            torch.cuda.synchronize()                                    # (shared scratch buffers: one pipeline in flight at a time)
            ref.step(True)
            torch.cuda.synchronize()
            for t in times:
                assert torch.equal(pipe.frames[t], ref.frames[t]), (s, t)
        assert torch.equal(pipe.category[:5], ref.category[:5]) and int((pipe.category[:5] != 0).sum()) > 0
        assert torch.equal(pipe.xywhs[:5], ref.xywhs[:5])
        rep = check_against(pipe, oracle.track_streams)
        assert rep['ok'] and rep['chunks'] == 5, rep
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved[0], saved[1]
        torch.cuda.tunable.enable(saved[2])


def test_deferred_tracking_fills_the_same_rows(oracle):
    """defer_tracking (the bench's placement of the SORT call: behind the bottom-up pathway of the NEXT frame, frame = two graphs):
    every chunk is tracked exactly once and only after its slots are complete - the oracle replay of the consumed slots equals the
    tracker rows at every stage, across a segment wrap, like for the plain placement next to it."""
    import torch
    from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline, check_against
    saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, torch.cuda.tunable.is_enabled())
    try:
        a = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=256, width=384, seed=5, segment_frames=6,
                                distinct_times=4, deterministic=True, defer_tracking=True)
        b = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=256, width=384, seed=5, segment_frames=6,
                                distinct_times=4, model=a.model, deterministic=True)
        for s in range(5):                                              # 3 chunks per segment: wraps once
            a.step(True)
            torch.cuda.synchronize()                                    # two pipelines of one process share the library's scratch buffers
            b.step(True)                                                # (sampling table of the stride-2 layers ...): never in flight together
            torch.cuda.synchronize()
            if s == 2:
                ra, rb = check_against(a, oracle.track_streams), check_against(b, oracle.track_streams)
                assert ra['ok'] and rb['ok'] and ra['chunks'] == rb['chunks'] == 3 and ra['rows'] == rb['rows'] > 0
        assert a._pending_track is not None and 'graph_b' in a._lanes[0]
        a.flush()
        torch.cuda.synchronize()
        assert a._pending_track is None
        # same kernels, same frames, deterministic library picks, never in flight together: identical slots and tracker rows
        assert torch.equal(a.category[:2], b.category[:2]) and torch.equal(a.xywhs[:2], b.xywhs[:2])
        for c in range(2):
            k = int(a.chunk_counts[c, 0])
            assert k == int(b.chunk_counts[c, 0]) and k > 0
            assert torch.equal(a.out_id[c][:k], b.out_id[c][:k]) and torch.equal(a.out_bbox[c][:k], b.out_bbox[c][:k])
        ra, rb = check_against(a, oracle.track_streams), check_against(b, oracle.track_streams)
        assert ra['ok'] and ra['chunks'] == 2 and a.segments_done == 1, ra
        assert rb['ok'] and rb['chunks'] == 2, rb
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved[0], saved[1]
        torch.cuda.tunable.enable(saved[2])


def test_per_chunk_exchange_over_rccl_single_rank(tmp_path):
    """The N > 1 step (birth-count all_gather + block gather behind every chunk's SORT) exercised over RCCL with one rank
    (a fresh process: NCCL process group + hipGraph capture + side-stream collectives), then verified: the collated rows are
    the rank's own rows, the published counts match, the oracle replay still agrees."""
    import json
    import subprocess
    import sys
    from waymo_2d_tracking_amd.launcher import free_port
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(%d), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from waymo_2d_tracking_amd.bench_e2e import DetectTrackPipeline, check_against, collation_report
from oracle import oracle as O
O.build()
pipe = DetectTrackPipeline(n_cameras=5, frames_per_camera=2, height=256, width=384, seed=3, segment_frames=8, distinct_times=6)
pipe.collate = True
pipe._capture()
for _ in range(3):
    pipe.step(True)
rep = collation_report(pipe, 1, 0)
rep['oracle'] = check_against(pipe, O.track_streams)
dist.barrier(); dist.destroy_process_group()
json.dump(rep, open(%r, 'wt'))
''' % (ROOT, free_port(), str(tmp_path / 'rep.json'))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.load(open(tmp_path / 'rep.json'))
    assert rep['rccl_ranks'] == [0] and rep['backend'] == 'nccl' and rep['exchanges'] == 3
    assert rep['collated_ok'] and rep['collated_chunks'] == 3 and rep['collated_rows_by_rank'][0] == rep['oracle']['rows'] > 0
    assert rep['births_by_rank'][0] == rep['oracle']['births'] and rep['oracle']['ok']


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible(tmp_path):
    """The first multi-rank run should not be the driver's: `bench.py --gpus 2` through the package's own launcher when the box has
    two GPUs (skipped on the 1-GPU boxes of the pool).  n_gpus is what RCCL connected, the collated block of rank 0 equals its own
    rows, ids of rank 1 start behind the births of rank 0 (reference: the process-global counter, tracking/sort/sort.py:86), and the
    per-rank step times are reported."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{') and '"metric"' in l]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    ex = line['extra']
    assert line['n_gpus'] == 2 and ex['rccl_ranks'] == [0, 1] and ex['collated_ok']
    assert ex['id_offsets'][0] == 0 and ex['id_offsets'][1] == ex['births_by_rank'][0] > 0
    assert len(ex['per_rank_ms_per_step']['by_rank']) == 2 and ex['per_rank_ms_per_step']['max'] <= line['ms_per_step'] * 1.001
    assert line['verified']['ok']
