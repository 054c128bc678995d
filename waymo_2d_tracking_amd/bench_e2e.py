"""End-to-end detect -> SORT step used by bench.py (BASELINE.json metric) and by smoke().

One step = one 5-camera chunk: `frames_per_step` synthetic 1920x1280x3 uint8 frames already resident in HBM go
through the Cascade R-CNN X152-FPN detector one by one (batch 1, like the reference's --batch-size=1); their <= 100
detections per frame are converted on the device to the detection-JSON wire values (int box, 5-decimal score,
category) and written into the frame-slotted SoA layout of wt_track_chunk_dev; then every (camera, class) tracker -
resident in HBM for the whole segment, like the reference's one MultiClassTrackerSort per stream (tracking/utils.py:29)
- consumes the chunk as one wavefront of the persistent SORT kernel.  Nothing leaves the GPU inside the timed region.

Every chunk of the current segment keeps its own detection slots and output rows (a 198-frame segment is ~9 MB), so
after a run `history()` hands the exact detections the tracker saw and the rows it produced to a checker
(tests/test_gpu_e2e.py and bench.py replay them through the CPU oracle).
"""
import os

import numpy as np
import torch

from . import _lib
from .detnet.nn import ops
from .detnet.nn.detectron2_det import Detectron2Det
from .devpath import StreamingTracker
from .tuning import enable_gemm_tuning

SLOTS = 100          # detectron2 TEST.DETECTIONS_PER_IMAGE (top-100, detectron2_det via fast_rcnn_inference)
SEGMENT_FRAMES = 198  # frames of one Waymo segment per camera: the trackers are reset after that many


def moving_frames(n_cameras, n_times, height, width, seed, device):
    """Synthetic camera streams already resident in HBM: per camera a uint8 U{0..255} image that translates
    by (2, 3) px per frame, so consecutive frames of a stream show the same content moved - detections of consecutive
    frames overlap the way tracked objects do, instead of being independent draws.  Layout [time][camera], HWC."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    base = torch.randint(0, 256, (n_cameras, height, width, 3), generator=g, dtype=torch.uint8).to(device)
    out = torch.empty((n_times, n_cameras, height, width, 3), dtype=torch.uint8, device=device)
    for t in range(n_times):
        out[t] = torch.roll(base, shifts=(2 * t, 3 * t), dims=(1, 2))
    return out


class DetectTrackPipeline(object):
    def __init__(self, n_cameras=5, frames_per_camera=2, height=1280, width=1920, seed=0, device='cuda',
                 iou_threshold=(0.01, 0.01, 1.0, 0.0), score_threshold=(0.0, 0.0, 0.0, 0.0), max_age=2, min_hits=0, tta='',
                 segment_frames=SEGMENT_FRAMES, distinct_times=16, model=None, use_graph=True, n_inflight=1, deterministic=False, defer_tracking=False, auto_contrast=False):
        self.dev = torch.device(device)
        # --tta x1.5,hflip (nn/tta.py:228-267): one pass on the enlarged, flipped image, folded into the pre-processing kernel
        self.tta_scale, self.tta_hflip = 1.0, False
        for aug in [a for a in tta.split(',') if a]:
            if aug.startswith('x'):
                self.tta_scale *= float(aug[1:])
            elif aug == 'hflip':
                self.tta_hflip = True
            elif aug != 'orig':
                raise ValueError('bench supports --tta orig / xS / hflip, got %r' % aug)
        if deterministic:
            # run-to-run bit-identical library kernels (tests: graph replay == eager launches): MIOpen's deterministic solvers,
            # no find-mode search (its winner may accumulate split-K partial sums with atomics), library-default GEMM picks
            torch.backends.cudnn.benchmark = False
            torch.backends.cudnn.deterministic = True
            torch.cuda.tunable.enable(False)
        else:
            torch.backends.cudnn.benchmark = True  # the reference's --cudnn-benchmark: let MIOpen pick its fastest conv
            enable_gemm_tuning()                   # ... and TunableOp its fastest library GEMM per 1x1-conv shape
        self.nc, self.fpc, self.h, self.w = n_cameras, frames_per_camera, height, width
        self.model = model if model is not None else Detectron2Det(seed=seed).to(self.dev).eval()
        self.n_frames = n_cameras * frames_per_camera
        self.n_times = max(frames_per_camera, (distinct_times // frames_per_camera) * frames_per_camera)
        self.frames = moving_frames(n_cameras, self.n_times, height, width, seed, self.dev)
        self.max_chunks = max(1, segment_frames // frames_per_camera)
        n = self.n_frames * SLOTS
        self.chunk_dets = n
        R = self.max_chunks
        f64 = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=self.dev)
        # per-chunk detection slots (chunk-major; inside a chunk camera-major frames x 100 slots, category 0 = empty)
        self.xywhs = f64(R, 5, n)                    # x, y, w, h, score rows of every chunk (one strided copy per frame)
        self.x, self.y, self.wd, self.ht, self.score = (self.xywhs[:, q] for q in range(5))
        self.category = torch.zeros((R, n), dtype=torch.int32, device=self.dev)
        self.n_dets_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.use_graph, self._graph = use_graph, None
        self.n_inflight, self._lanes, self._frame_no = max(1, n_inflight), [], 0
        # per-chunk tracker output rows: ONE contiguous byte block per chunk (the columns are typed views of it), so the
        # collation of a chunk to rank 0 is a gather of that block as it lies in HBM (distributed.DeviceRowCollator)
        from . import distributed as D
        self.rows = D.DeviceRowCollator([('frame', torch.int64, ()), ('local_id', torch.int64, ()), ('bbox', torch.float64, (4,)),
                                         ('score', torch.float64, ()), ('category', torch.int32, ())], n + 1, R, self.dev)
        views = [self.rows.columns(c) for c in range(R)]
        self.out_frame = [v['frame'] for v in views]
        self.out_cat = [v['category'] for v in views]
        self.out_bbox = [v['bbox'] for v in views]
        self.out_score = [v['score'] for v in views]
        self.out_id = [v['local_id'] for v in views]
        self.chunk_counts = self.rows.counts
        self.collate = False           # exchange birth counts + gather the chunk's rows to rank 0 behind every track() call
        self.frame_off = (torch.arange(self.n_frames + 1, dtype=torch.int64) * SLOTS).to(self.dev)
        self.stream_off = (torch.arange(n_cameras + 1, dtype=torch.int64) * frames_per_camera).to(self.dev)
        self.clip_w = torch.full((n_cameras,), float(width), dtype=torch.float64, device=self.dev)
        self.clip_h = torch.full((n_cameras,), float(height), dtype=torch.float64, device=self.dev)
        self.track_params = dict(iou_threshold=list(iou_threshold), score_threshold=list(score_threshold),
                                 max_age=max_age, min_hits=min_hits)
        self.tracker = StreamingTracker(n_cameras, SLOTS, self.n_frames, n, list(iou_threshold), max_age, min_hits,
                                        list(score_threshold), device=self.dev)
        self.track_stream = torch.cuda.Stream(device=self.dev)
        self._slot_done = [None] * R      # per ring slot: event of the track() that last read it
        self.chunk = 0                 # chunks of the current segment processed so far
        self.time = 0                  # frame time index into self.frames
        self.jpeg = None               # enable_jpeg_input(): frames enter as JPEG bytes
        self.defer_tracking = bool(defer_tracking)
        self.auto_contrast = bool(auto_contrast)      # --auto-contrast=1 of the documented run (README.md:37): ImageOps.autocontrast per frame
        self._pending_track = None
        self.segments_done = 0

    @property
    def counts(self):
        """(rows, births) of the most recent chunk (device int64[2])."""
        return self.chunk_counts[max(self.chunk - 1, 0)]

    def _detect_core(self, img):
        """uint8 (1, H, W, 3) frame -> wire-format detections in 100 static slots: (xywhs (5, 100) float64, category (100) int32
        with 0 = empty slot).  Static shapes, no host synchronisation: capturable as ONE hipGraph."""
        if self.auto_contrast:
            img = ops.autocontrast_(img[0].clone()).unsqueeze(0)
        boxes, scores, classes, cnt = self.model.predict_padded(img, self.tta_scale, self.tta_hflip)
        ho, wo = self.model.last_input_size
        # HFlipTTA.post_process + Detectron2Det.predict + load_prediction in one launch; unused slots: category 0 = ignored
        xywhs, cat = ops.detections_to_wire(boxes, scores, classes, cnt, wo, ho, self.w, self.h, self.tta_hflip)
        return xywhs, cat, cnt

    def _detect_heads(self, feats):
        boxes, scores, classes, cnt = self.model.predict_padded_heads(feats)
        ho, wo = self.model.last_input_size
        xywhs, cat = ops.detections_to_wire(boxes, scores, classes, cnt, wo, ho, self.w, self.h, self.tta_hflip)
        return xywhs, cat, cnt

    def _capture(self):
        """Per lane: warm up eagerly on the lane's stream (MIOpen / TunableOp / hipBLASLt pick their kernels, per-stream scratch
        gets allocated), then capture the whole per-frame detector as one hipGraph on that stream.  Every lane owns its input /
        output / intermediate buffers, so the graphs of consecutive frames are in flight at the same time: the one-workgroup tails
        of one frame (NMS sweep, candidate sorts, top-k stages), the 16 CUs a 240-tile GEMM leaves idle and every kernel's ramp up /
        down run under the other frame's kernels.  Round 6: with 90 % of the frame on own kernels two lanes run at full size
        (1920x1280: 38.9 -> 43.4 frames/s on one box, 40.8 -> 44.6 detector alone; three lanes 44.4, four 43.6: tools/inflight_ab.sh,
        profiles/r06_inflight_ab.txt); in round 2, with 79 % library kernels, the same setting DEADLOCKED (library kernels that spin on
        partner workgroups need all of them resident) - the all-library exact-f32 graph (WD_SPLIT_GEMM=0) therefore keeps ONE lane.
        Every product kernel was checked as a co-resident victim of the split-operand kernel on another stream
        (profiles/r06_costream_victim_side.txt)."""
        saved, ops.EVENT_LOG = ops.EVENT_LOG, None          # no event records inside a capture
        torch.cuda.synchronize()
        self._lanes = []
        for k in range(self.n_inflight):
            lane = dict(stream=torch.cuda.Stream(device=self.dev), n_dets=torch.zeros(1, dtype=torch.int64, device=self.dev),
                        gin=torch.zeros((1, self.h, self.w, 3), dtype=torch.uint8, device=self.dev))
            with torch.cuda.stream(lane['stream']), torch.no_grad():
                for t in range(3):
                    lane['gin'].copy_(self.frames[t % self.n_times, 0].unsqueeze(0))
                    self._detect_core(lane['gin'])
            lane['stream'].synchronize()
            if self.defer_tracking:
                # two graphs per frame - bottom-up pathway | FPN + RPN + heads + tail - so that an event between them can release the
                # SORT kernel of the previous chunk (see step()); the second graph reads the first one's output tensors in place
                lane['graph'] = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(lane['graph'], stream=lane['stream'], capture_error_mode='thread_local'):
                    gin = ops.autocontrast_(lane['gin'][0].clone()).unsqueeze(0) if self.auto_contrast else lane['gin']
                    lane['feats'] = self.model.predict_padded_bottom_up(gin, self.tta_scale, self.tta_hflip)
                lane['graph_b'] = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(lane['graph_b'], stream=lane['stream'], pool=lane['graph'].pool(),
                                                       capture_error_mode='thread_local'):
                    lane['gout'] = self._detect_heads(lane['feats'])
            else:
                lane['graph'] = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(lane['graph'], stream=lane['stream'], capture_error_mode='thread_local'):
                    lane['gout'] = self._detect_core(lane['gin'])
            self._lanes.append(lane)
        torch.cuda.synchronize()
        self._graph = self._lanes[0]['graph']
        ops.EVENT_LOG = saved

    # ---- frames entering as JPEG files: decode on the GPU into the frame slots, one step ahead -------
    def enable_jpeg_input(self, quality=90, workers=4):
        """The camera frames exist as JPEG bytes on the host (photo-like synthetic content: white noise would be a 5 MB file per
        frame) and `self.frames` becomes the decode target: the frames of step s + 1 are decoded by `workers` loader threads
        (own streams, csrc/jpeg_decode.hip) while the detector works on step s."""
        import io
        from concurrent.futures import ThreadPoolExecutor
        from PIL import Image
        rng = np.random.default_rng(7)
        yy, xx = np.mgrid[0:self.h, 0:self.w]
        self.jpeg = [[None] * self.nc for _ in range(self.n_times)]
        for cam in range(self.nc):
            base = 128 + 100 * np.sin(xx[..., None] / (5.0 + cam) + np.arange(3)) * np.cos(yy[..., None] / (7.0 + cam))
            base = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
            for t in range(self.n_times):
                buf = io.BytesIO()
                Image.fromarray(np.roll(base, (2 * t, 3 * t), (0, 1))).save(buf, 'JPEG', quality=quality, subsampling=2)
                self.jpeg[t][cam] = buf.getvalue()
        self._dec_pool = ThreadPoolExecutor(workers)
        self._dec_streams = {}
        self._dec_pending = {}                                   # time slot -> futures of its cameras
        self._slot_readers = {}                                  # time slot -> events of the detector launches that read it
        self.jpeg_bytes = sum(len(b) for row in self.jpeg for b in row) / float(self.n_times * self.nc)
        self._request_decode(range(self.time, self.time + self.fpc))

    def _decode_one(self, t, cam):
        import threading
        from .detnet.nn import ops
        st = self._dec_streams.get(threading.get_ident())
        if st is None:
            st = self._dec_streams[threading.get_ident()] = torch.cuda.Stream()
        with torch.cuda.stream(st):
            ops.jpeg_decode(self.jpeg[t][cam], out=self.frames[t, cam])          # returns with the frame complete in HBM
        return True

    def _request_decode(self, times):
        for t in times:
            t %= self.n_times
            if t in self._dec_pending:
                continue
            for ev in self._slot_readers.pop(t, []):             # the detector pass that last read this slot must be done
                ev.synchronize()
            self._dec_pending[t] = [self._dec_pool.submit(self._decode_one, t, cam) for cam in range(self.nc)]

    def _await_decoded(self, times):
        for t in times:
            for f in self._dec_pending.pop(t % self.n_times, []):
                f.result()

    def flush(self):
        """defer_tracking: enqueue the SORT call that is still waiting for the next frame's bottom-up pass (end of a run)."""
        if self._pending_track is not None:
            fn, self._pending_track = self._pending_track, None
            fn(None)

    def detect_frame(self, c, cam, j, eager=False):
        """Frame j of camera cam of chunk c -> wire-format detections in that frame's 100 slots."""
        # decoded uint8 HWC RGB frame -> fused pre-processing kernel (ToTensor(scaling=False) + BGR + normalise + pad)
        img = self.frames[(self.time + j) % self.n_times, cam].unsqueeze(0)
        a = (cam * self.fpc + j) * SLOTS
        if not self.use_graph:
            xywhs, cat, cnt = self._detect_core(img)
            self.xywhs[c, :, a:a + SLOTS] = xywhs
            self.category[c, a:a + SLOTS] = cat
            self.n_dets_dev += cnt
            return
        if self._graph is None:
            self._capture()
        lane = self._lanes[self._frame_no % self.n_inflight]
        self._frame_no += 1
        others = [l for l in self._lanes if l is not lane]
        if eager and others:
            # the instrumented frame runs ALONE on the chip (its HIP events time kernels, not kernels sharing CUs with another frame's): it starts
            # behind everything the other lanes have queued and they resume behind it
            for l in others:
                ev = torch.cuda.Event()
                ev.record(l['stream'])
                lane['stream'].wait_event(ev)
        with torch.cuda.stream(lane['stream']):
            if eager:
                xywhs, cat, cnt = self._detect_core(img)
                if others:
                    ev = torch.cuda.Event()
                    ev.record(lane['stream'])
                    for l in others:
                        l['stream'].wait_event(ev)
                if self._pending_track is not None:              # (an instrumented frame is never the first of a step)
                    self.flush()
            else:
                lane['gin'].copy_(img)
                lane['graph'].replay()
                if self.defer_tracking:
                    if self._pending_track is not None:
                        # the previous chunk's SORT starts HERE: behind this frame's bottom-up pathway, i.e. under FPN / RPN / head
                        # kernels with thousands of workgroups - not under the persistent deformable-conv kernels, whose 256 workgroups
                        # (150 KB of LDS each) need every CU: one CU held by a tracker workgroup doubles such a launch
                        ev = torch.cuda.Event()
                        ev.record(lane['stream'])
                        fn, self._pending_track = self._pending_track, None
                        fn(ev)
                    lane['graph_b'].replay()
                xywhs, cat, cnt = lane['gout']
            self.xywhs[c, :, a:a + SLOTS] = xywhs
            self.category[c, a:a + SLOTS] = cat
            lane['n_dets'] += cnt

    @property
    def n_dets_total(self):
        return int(self.n_dets_dev.item()) + sum(int(l['n_dets'].item()) for l in getattr(self, '_lanes', []))

    def track(self, c):
        self.tracker.feed(self.x[c], self.y[c], self.wd[c], self.ht[c], self.score[c], self.category[c], self.frame_off,
                          self.stream_off, self.clip_w, self.clip_h, self.out_frame[c], self.out_cat[c], self.out_bbox[c],
                          self.out_score[c], self.out_id[c], self.chunk_counts[c])

    def step(self, with_tracking=True, instrument=False):
        main = torch.cuda.current_stream()
        if self.chunk == self.max_chunks:                        # segment complete: fresh trackers, slots reused
            if with_tracking:
                self.flush()                                     # the last chunk's SORT goes first
                with torch.cuda.stream(self.track_stream):
                    self.tracker.reset()
            self.chunk = 0
            self.segments_done += 1
        c = self.chunk
        streams = [l['stream'] for l in self._lanes] if (self.use_graph and self._lanes) else []
        if self._slot_done[c] is not None:
            # the slots of ring position c were last read by the track() of the previous segment's chunk c (a whole segment
            # ago): wait for THAT call only - waiting for the most recent track() would serialise SORT and the detector
            main.wait_event(self._slot_done[c])
            for st in streams:
                st.wait_event(self._slot_done[c])
        if self.use_graph and self._graph is None:
            self._capture()
            streams = [l['stream'] for l in self._lanes]
        if self.jpeg is not None:
            now = range(self.time, self.time + self.fpc)
            self._request_decode(now)
            self._await_decoded(now)                             # this step's frames are in HBM (decode calls are synchronous)
            self._request_decode(range(self.time + self.fpc, self.time + 2 * self.fpc))      # next step's: under this step's detector
        start = torch.cuda.Event()
        start.record(main)
        for st in streams:
            st.wait_event(start)                                 # lanes never run ahead of work queued on the caller's stream
        for cam in range(self.nc):
            for j in range(self.fpc):
                self.detect_frame(c, cam, j, eager=(instrument and cam == self.nc - 1 and j == self.fpc - 1))
        done = []
        for st in streams:                                       # no join on the caller's stream: the next step's frames may start
            ev = torch.cuda.Event()                              # while this step's last ones finish (callers synchronise the
            ev.record(st)                                        # device before reading results)
            done.append(ev)
        if with_tracking:
            filled = torch.cuda.Event()
            filled.record(main)

            def run_track(release, c=c, filled=filled, done=list(done)):
                with torch.cuda.stream(self.track_stream):
                    self.track_stream.wait_event(filled)         # slots of this chunk are complete (eager path)
                    for ev in done:
                        self.track_stream.wait_event(ev)         # ... on every lane
                    if release is not None:
                        self.track_stream.wait_event(release)    # defer_tracking: the next frame's bottom-up pathway is through
                    self.track(c)                                # SORT of chunk c runs under the detector pass of chunk c + 1
                    if self.collate:
                        self.rows.exchange(c)                    # (rows, births) all_gather + block gather, stream-ordered
                    self._slot_done[c] = torch.cuda.Event()
                    self._slot_done[c].record(self.track_stream)
            if self.defer_tracking and self.use_graph:
                self.flush()                                     # (never more than one call waiting)
                self._pending_track = run_track
            else:
                run_track(None)
        if self.jpeg is not None:                                # who read this step's frame slots (before they are decoded into again)
            evs = list(done)
            ev = torch.cuda.Event()
            ev.record(main)
            evs.append(ev)
            for t in range(self.time, self.time + self.fpc):
                self._slot_readers[t % self.n_times] = evs
        self.chunk += 1
        self.time = (self.time + self.fpc) % self.n_times
        return None

    def collated(self, c):
        """Rank 0, after the device is idle: what `exchange(c)` delivered - per rank the rows of chunk c and the
        (rows, births) table of all ranks."""
        torch.cuda.synchronize()
        return self.rows.decode(c)

    def history(self):
        """Synchronise and return what the trackers of the CURRENT segment consumed and produced so far:
        (packed, rows) - `packed` is the pack_streams() layout of the detections (streams = cameras, frames of all chunks
        in order, empty slots removed), `rows` the tracker output in the reference's order (stream-major) with the
        reference's global ids (wt_track_global_ids_dev), frame = index into packed's frames."""
        torch.cuda.synchronize()
        nch, F, nc = self.chunk, self.fpc, self.nc
        cat = self.category[:nch].cpu().numpy().reshape(nch, nc, F, SLOTS)
        arr = {k: getattr(self, a)[:nch].cpu().numpy().reshape(nch, nc, F, SLOTS)
               for k, a in (('x', 'x'), ('y', 'y'), ('w', 'wd'), ('h', 'ht'), ('score', 'score'))}
        cols = {k: [] for k in ('x', 'y', 'w', 'h', 'score', 'category')}
        frame_off, stream_off = [0], [0]
        n = 0
        for s in range(nc):
            for c in range(nch):
                for j in range(F):
                    m = cat[c, s, j] != 0
                    for k in arr:
                        cols[k].append(arr[k][c, s, j][m])
                    cols['category'].append(cat[c, s, j][m])
                    n += int(m.sum())
                    frame_off.append(n)
            stream_off.append(len(frame_off) - 1)
        packed = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in cols.items()}
        packed['category'] = packed['category'].astype(np.int32)
        packed.update(frame_det_offsets=np.asarray(frame_off, np.int64), stream_frame_offsets=np.asarray(stream_off, np.int64),
                      clip_w=self.clip_w.cpu().numpy(), clip_h=self.clip_h.cpu().numpy())
        counts = self.chunk_counts[:nch].cpu().numpy()
        if (counts[:, 0] < 0).any():
            raise _lib.WaymoTrackError('SORT kernel status %d' % -int(counts[:, 0].min()))
        fr, ct, bb, sc, lid, st = [], [], [], [], [], []
        for c in range(nch):
            k = int(counts[c, 0])
            f = self.out_frame[c][:k]
            s = torch.div(f, F, rounding_mode='floor')
            fr.append(s * (nch * F) + c * F + (f - s * F)); st.append(s)
            ct.append(self.out_cat[c][:k]); bb.append(self.out_bbox[c][:k]); sc.append(self.out_score[c][:k])
            lid.append(self.out_id[c][:k])
        fr, st, ct, bb, sc, lid = [torch.cat(v) for v in (fr, st, ct, bb, sc, lid)]
        gid = self.tracker.global_ids(st, lid, 0)
        order = torch.sort(st, stable=True)[1]                   # chunk-major -> stream-major (utils.py output order)
        rows = dict(frame=fr[order].cpu().numpy(), category=ct[order].cpu().numpy(), bbox=bb[order].cpu().numpy(),
                    score=sc[order].cpu().numpy(), object_id=gid[order].cpu().numpy())
        return packed, rows, int(counts[:, 1].sum())


def check_against(pipe, track_streams):
    """Replay the current segment's detections through `track_streams` (the CPU oracle's entry point, injected by the
    caller: tests / bench.py) and compare with what the GPU trackers produced.  Returns a small report dict."""
    pipe.flush()
    packed, rows, births = pipe.history()
    p = pipe.track_params
    ref = track_streams(packed, p['max_age'], p['min_hits'], p['score_threshold'], p['iou_threshold'])
    ok = (births == ref['n_births'] and np.array_equal(rows['object_id'], ref['object_id'])
          and np.array_equal(rows['frame'], ref['frame']) and np.array_equal(rows['category'], ref['category'])
          and np.array_equal(rows['bbox'], ref['bbox']) and np.allclose(rows['score'], ref['score'], rtol=4e-16, atol=0))
    return dict(ok=bool(ok), chunks=pipe.chunk, frames=int(packed['frame_det_offsets'].size - 1), dets=int(packed['x'].size),
                rows=int(len(rows['frame'])), rows_ref=int(len(ref['frame'])), births=births, births_ref=int(ref['n_births']))


def collation_report(pipe, world, rank):
    """After the timed region: what the per-step exchange delivered.  `rccl_ranks` comes from an actual all_gather of the
    rank numbers; on rank 0 the gathered block of rank 0 must equal its own rows and every rank's row count must equal the
    count that rank published."""
    import torch.distributed as dist
    pipe.flush()
    rep = dict(collation='all_gather(rows, births) + one block gather per chunk, device-resident', exchanges=pipe.rows.exchanges)
    if dist.is_available() and dist.is_initialized():
        mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=pipe.dev)
        seen = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=pipe.dev)
        dist.all_gather_into_tensor(seen, mine)
        rep['rccl_ranks'] = [int(v) for v in seen.cpu().tolist()]
        rep['backend'] = dist.get_backend()
    if rank == 0:
        nch = pipe.chunk
        rows_by_rank = np.zeros(world, np.int64)
        births_by_rank = np.zeros(world, np.int64)
        own_ok = True
        for c in range(nch):
            parts, counts = pipe.collated(c)
            rows_by_rank += counts[:, 0]
            births_by_rank += counts[:, 1]
            k = int(pipe.chunk_counts[c, 0].item())
            own = {n: v[:k].cpu().numpy() for n, v in pipe.rows.columns(c).items()}
            own_ok = own_ok and all(np.array_equal(parts[0][n], own[n]) for n in own) and \
                all(len(parts[r]['frame']) == int(counts[r, 0]) for r in range(world))
        rep.update(collated_chunks=nch, collated_rows_by_rank=rows_by_rank.tolist(), births_by_rank=births_by_rank.tolist(),
                   id_offsets=np.concatenate([[0], np.cumsum(births_by_rank)[:-1]]).tolist(), collated_ok=bool(own_ok))
    return rep


def deform_offset_sweep(stds=(0.0, 0.5, 1.0, 2.0), c=1024, h=80, w=120, launches=30):
    """The roofline kernel (res4 deformable conv through the pipeline's entry points: sampling table from the offset conv's gather
    launch, persistent kernel) at several offset spreads, iid N(0, std^2) px per pixel and tap - outside the timed region, HIP events
    on the launch stream.  The random-init bench model's own layers have a spread of 0.1 - 0.2 px; trained DCN offsets are
    routinely >= 1 px, which is why the bench line reports all of them instead of the best case only."""
    import ctypes as C
    from . import _lib
    from .detnet.nn import ops
    dev = torch.device('cuda')
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn((1, c, h, w), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    pw = ops.deform_pack_weight((torch.randn((c, c // 32, 3, 3), generator=g) * 0.05).to(dev), 32)
    sc, bi = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    lib = _lib.lib()
    lib.wd_deform_table_bytes.restype = C.c_size_t
    table = torch.empty(int(lib.wd_deform_table_bytes(C.c_int(1), C.c_int(h), C.c_int(w))), dtype=torch.uint8, device=dev)
    flops = 2.0 * c * (c // 32) * 9 * h * w
    out = {}
    for std in stds:
        off = (torch.randn((1, 18, h, w), generator=g) * std).to(dev)
        partial = torch.zeros((h * w, 176), device=dev)                     # the offsets as the centre-tap partial sums of the offset conv
        partial[:, 72:90] = off.permute(0, 2, 3, 1).reshape(-1, 18)
        offs = torch.empty((1, 18, h, w), device=dev).contiguous(memory_format=torch.channels_last)
        _lib.check(lib.wd_deform_offsets_table_f32(C.c_void_p(partial.data_ptr()), C.c_int(176), None, C.c_int(1), C.c_int(h), C.c_int(w),
                                                   C.c_void_p(offs.data_ptr()), C.c_void_p(table.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wd_deform_offsets_table_f32')
        run = lambda: ops.deform_conv3x3(x, offs, pw, 32, 1, 1, sc, bi, True, table=table)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / launches * 1e3
        share = float((off.abs() > 2.0).reshape(1, 9, 2, h, w).any(dim=2).float().mean().item())
        out['%.1f px' % std] = dict(avg_us=us, tflops=flops / us / 1e6, frac=flops / us / 1e6 / 157.3, samples_outside_patch=share)
    return out


BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)


def _pmc_traffic(tag):
    """HBM bytes per launch of the roofline kernel from the committed PMC pass (profiles/r0N_e2e_pmc_traffic.json,
    collected with tools/pmc_traffic.sh - counters cannot be read from inside the timed run); None if not recorded."""
    import json
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
    for fname in ('r06_e2e_pmc_traffic.json', 'r05_e2e_pmc_traffic.json', 'r04_e2e_pmc_traffic.json', 'r03_e2e_pmc_traffic.json', 'r02_e2e_pmc_traffic.json', 'r01_e2e_pmc_traffic.json'):
        try:
            rec = json.load(open(os.path.join(root, fname)))
        except (OSError, ValueError):
            continue
        for name, v in rec.items():
            # the profiler prints all template arguments ("..._pp_kernel<32, false>"), the bench tag only the first ("..._pp_kernel<32>: ...")
            base, targ = name.split('<')[0], name.split('<')[1].split(',')[0].rstrip('>') if '<' in name else ''
            if tag.startswith(name) or (tag.startswith(base + '<' + targ) and targ):
                return v.get('traffic_bytes_per_launch')
    return None


def exact_f32_line(args, rank, track, fps, steps, exact=True):
    """Secondary lines timed in the SAME invocation on a second pipeline.  exact=True: the all-exact-f32 configuration (every convolution on the f32
    library / f32-MFMA path, WD_SPLIT_GEMM=0; ONE frame in flight - its library kernels deadlock with two lanes), the line the round-4 review asked to
    carry next to the split-operand headline.  exact=False (round 6): the headline's own graph with ONE frame in flight, i.e. without the overlap of
    consecutive frames."""
    import time
    from .detnet.nn import cascade_rcnn
    cascade_rcnn.SPLIT_GEMM = not exact
    try:
        pipe = DetectTrackPipeline(5, fps, seed=rank, tta=getattr(args, 'tta', '') or '', use_graph=not getattr(args, 'no_graph', False),
                                   defer_tracking=track and not getattr(args, 'no_defer_track', False), auto_contrast=getattr(args, 'auto_contrast', False))
        if pipe.use_graph:
            pipe._capture()
        pipe.step(track)
        pipe.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step(track)
        pipe.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = dict(value=pipe.n_frames * steps / dt, unit='frames/s', ms_per_step=1e3 * dt / steps, steps=steps, warmup=1,
                   note=('same pipeline with WD_SPLIT_GEMM=0: 1x1 convolutions on hipBLASLt f32, dense 3x3 on MIOpen f32; one frame in flight' if exact else
                         'same graph, one frame in flight (no overlap of consecutive frames)'))
        del pipe
    finally:
        cascade_rcnn.SPLIT_GEMM = True
    torch.cuda.empty_cache()
    return out


def run(args, world, rank, timed_steps):
    from .detnet.nn import ops
    fps = max(1, args.frames_per_step // 5)
    track = args.stage == 'e2e'
    pipe = DetectTrackPipeline(5, fps, seed=rank, tta=getattr(args, 'tta', '') or '', use_graph=not getattr(args, 'no_graph', False), n_inflight=getattr(args, 'inflight', 1),
                               defer_tracking=track and not getattr(args, 'no_defer_track', False), auto_contrast=getattr(args, 'auto_contrast', False))
    steps = args.steps or 3
    warmup = args.warmup if args.warmup is not None else 1
    state = {'n': 0}
    # Timed steps whose LAST frame is instrumented (launched eagerly with HIP events around every launch).  With two frames in flight that frame runs alone on
    # the chip - the other lane drains before it and waits behind it - which costs the step ~3 % (same box: 41.96 / 42.09 frames/s with every step
    # instrumented, 43.20 / 43.22 with two of ten, 43.51 / 43.38 with none): a SAMPLE of the timed region carries the events - one step, two from 8 steps on
    # (71 launches of the roofline shape per frame; the kernel's time does not move by 1 % between frames).  WT_BENCH_INSTRUMENT_STEPS overrides.
    n_instr = int(os.environ.get('WT_BENCH_INSTRUMENT_STEPS', '2' if steps >= 8 else '1'))

    def step():
        # HIP-event instrumentation of the dominant hand-written kernel only inside the timed region
        if state['n'] == warmup:
            ops.EVENT_LOG = []
        state['n'] += 1
        # the LAST frame of every timed step runs eagerly so that its deform-conv launches carry HIP events (the first frames of a
        # step share the chip with the SORT kernel of the previous chunk); the other frames replay the captured hipGraph of the
        # same launches
        pipe.step(track, instrument=ops.EVENT_LOG is not None and state['n'] - warmup <= n_instr)
        if state['n'] in (warmup, warmup + steps):
            pipe.flush()           # the SORT call that waits for the next frame belongs to the region that produced its detections

    # N > 1 (or --collate / WT_FORCE_DIST=1): behind every chunk's SORT the (rows, births) pairs of all ranks are exchanged
    # (all_gather: the id offsets of the sharded sequences) and the chunk's rows travel to rank 0 in ONE gather of the block as
    # it lies in HBM - the single RCCL collation step north_star names, stream-ordered, no host synchronisation
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    pipe.collate = track and (dist_on or getattr(args, 'collate', False))
    if pipe.use_graph:
        pipe._capture()            # capture before any collective is in flight
    if getattr(args, 'from_jpeg', False):
        pipe.enable_jpeg_input()
    dt, ev_ms = timed_steps(world, step, steps, warmup)
    log, ops.EVENT_LOG = ops.EVENT_LOG or [], None
    torch.cuda.synchronize()
    n_out, births = [int(v) for v in pipe.counts.cpu().tolist()] if track else (0, 0)
    frames = pipe.n_frames
    # roofline of the dominant hand-written kernel: the deformable-conv implicit GEMM (47 launches per frame);
    # achieved = algorithmic flops per launch / mean launch duration of the most frequent shape (res4, 36 of 47)
    by_shape = {}
    for tag, flops, e0, e1 in log:
        d = by_shape.setdefault(tag, [0, 0.0, flops])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
    roofline = None
    split = {k: v for k, v in by_shape.items() if k.startswith('gemm_split_kernel')}
    deform_all = {k: v for k, v in by_shape.items() if not k.startswith('gemm_split_kernel')}
    deform_roof = None
    if deform_all:
        deform = {k: v for k, v in deform_all.items() if 'no offsets' not in k}
        tag = max(deform, key=lambda k: deform[k][1])
        cnt, ms, flops = deform[tag]
        ach = flops / (ms / cnt * 1e-3) / 1e12
        deform_roof = dict(bound='mfma', kernel=tag, achieved=ach, peak=157.3,
                           unit='TFLOP/s', frac=ach / 157.3, traffic=_pmc_traffic(tag), launches=cnt, avg_us=ms / cnt * 1e3,
                           flops_per_launch=flops,
                           all_shapes={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3, tflops=v[2] / (v[1] / v[0] * 1e-3) / 1e12)
                                       for k, v in deform_all.items()})
        tot_f = sum(v[2] * v[0] for v in deform.values())
        tot_s = sum(v[1] for v in deform.values()) * 1e-3
        deform_roof['all_deformable_launches'] = dict(launches=sum(v[0] for v in deform.values()), tflops=tot_f / tot_s / 1e12,
                                                     frac=tot_f / tot_s / 1e12 / 157.3)
        deform_roof['offset_spread_of_bench_model'] = 'std 0.10 - 0.21 px per layer (random-init offset convs, tools/offset_stats.py)'
        if rank == 0:
            deform_roof['by_offset_spread'] = deform_offset_sweep()
    split_roof = None
    if split:
        # the split-operand GEMM / convolution kernel: six bf16 MFMAs per f32 product tile -> flops = 6 x 2 M N K, priced against the DENSE bf16
        # MFMA peak (MI355X guide: 2.5 PFLOP/s); `f32_equivalent_tflops` = 2 M N K / time is what the f32 library GEMM is compared by
        tag = max(split, key=lambda k: split[k][1])
        cnt, ms, flops = split[tag]
        ach = flops / (ms / cnt * 1e-3) / 1e12
        tot_f = sum(v[2] * v[0] for v in split.values())
        tot_s = sum(v[1] for v in split.values()) * 1e-3
        split_roof = dict(bound='mfma', kernel=tag, achieved=ach, peak=BF16_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / BF16_MFMA_PEAK_TFLOPS,
                          traffic=_pmc_traffic(tag), launches=cnt, avg_us=ms / cnt * 1e3, flops_per_launch=flops,
                          f32_equivalent_tflops=ach / 6.0,
                          flops_note='bf16 matrix-core flops issued: six cross terms of the exact 3 x bf16 operand split = 6 x 2 M N K per launch',
                          all_shapes={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3, tflops_bf16=v[2] / (v[1] / v[0] * 1e-3) / 1e12,
                                              frac=v[2] / (v[1] / v[0] * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS) for k, v in split.items()},
                          all_split_launches=dict(launches=sum(v[0] for v in split.values()), tflops_bf16=tot_f / tot_s / 1e12,
                                                  frac=tot_f / tot_s / 1e12 / BF16_MFMA_PEAK_TFLOPS, ms_per_instrumented_frame=None))
    # the `roofline` object is the kernel family that takes most of the instrumented frame; the other family rides along
    t_split = sum(v[1] for v in split.values()) if split else 0.0
    t_deform = sum(v[1] for v in deform_all.values()) if deform_all else 0.0
    if split_roof is not None and t_split >= t_deform:
        roofline = split_roof
        roofline['deformable_conv'] = deform_roof
    elif deform_roof is not None:
        roofline = deform_roof
        roofline['split_gemm'] = split_roof
    if roofline is not None:
        roofline['timing_note'] = ('avg_us: HIP events on the launch stream around every launch of the LAST frame of the first timed step(s) (two from 8 steps on), which runs eagerly '
                                   'and ALONE on the chip (the other frames replay the captured hipGraph of the same launches, two frames in flight, and share '
                                   'the chip with each other and with the SORT kernel of the previous chunk: a kernel duration taken there would include time '
                                   'spent sharing CUs); rocprofv3 --kernel-trace --stats of the same command with --inflight 1 agrees '
                                   '(profiles/r06_e2e_kernel_stats_one_lane.csv; the two-lane trace is profiles/r06_e2e_kernel_stats.csv)')
    from .detnet.nn import cascade_rcnn
    split_on = cascade_rcnn.SPLIT_GEMM
    res = dict(value=frames * world * steps / dt, unit='frames/s', ms_per_step=1e3 * dt / steps, dtype='f32',
               workload='Cascade R-CNN X152-32x8d-FPN dconv (random-init, fp32, batch 1%s) on synthetic 1920x1280x3 frames'
                        ' -> top-100 detections/frame -> %s; %d cameras x %d frames per step per GPU'
                        % ((', --tta ' + args.tta if getattr(args, 'tta', '') else '') + (', --auto-contrast' if getattr(args, 'auto_contrast', False) else ''),
                           'SORT (max_age 2, min_hits 0, all boxes tracked; trackers resident for the whole segment)' if track else 'no tracking', 5, fps),
               roofline=roofline,
               extra=dict(frames_per_step=frames, dets_per_frame=pipe.n_dets_total / max(1, frames * state['n']), hip_graph=pipe._graph is not None, track_rows=n_out, births=births))
    if split_on:
        res['workload'] += ('; 1x1 and dense 3x3 convolutions: f32 operands as exact 3 x bf16 splits, six cross terms on the bf16 matrix cores, f32 accumulate '
                            '(csrc/det_gemm_split.hip; error vs float64 below the f32 library GEMM\'s, profiles/r05_split_gemm_error.txt), everything else fp32')
        res['extra']['split_gemm'] = True
        if world == 1 and os.environ.get('WT_BENCH_NO_EXACT') != '1':
            res['extra']['exact_f32'] = exact_f32_line(args, rank, track, fps, min(steps, 3))
            if pipe.n_inflight > 1:
                res['extra']['one_frame_in_flight'] = exact_f32_line(args, rank, track, fps, min(steps, 3), exact=False)
    res['extra']['frames_in_flight'] = pipe.n_inflight
    if pipe.n_inflight > 1:
        res['workload'] += ('; %d frames in flight (hipGraph lanes on separate streams; the instrumented frame of a step runs alone)' % pipe.n_inflight)
    if pipe.jpeg is not None:
        import io
        from PIL import Image
        pipe._await_decoded(range(pipe.n_times))
        torch.cuda.synchronize()
        t_chk = (pipe.time - 1) % pipe.n_times                  # a slot the last step decoded and read
        same = all(np.array_equal(pipe.frames[t_chk, cam].cpu().numpy(), np.asarray(Image.open(io.BytesIO(pipe.jpeg[t_chk][cam])).convert('RGB')))
                   for cam in range(pipe.nc))
        res['workload'] = res['workload'].replace('on synthetic 1920x1280x3 frames', 'on synthetic 1920x1280x3 frames entering as JPEG files '
                                                  '(%.2f MB each, 4:2:0 q90), decoded on the GPU one step ahead by 4 loader threads' % (pipe.jpeg_bytes / 1e6))
        res['extra'].update(input='jpeg', jpeg_mb_per_frame=pipe.jpeg_bytes / 1e6, decoded_frames_equal_pil=bool(same))
    if pipe.collate:
        res['extra'].update(collation_report(pipe, world, rank))
    res['pipeline'] = pipe         # bench.py times the CPU port (oracle) against the same parameters
    return res, steps, warmup


def run_train(args, world, rank, timed_steps):
    """Config 5: Cascade R-CNN X152 dconv training fwd+bwd+SGD step, 886x1280 crop (train.py:37-47), batch 1 per
    GPU, DDP over RCCL when world > 1 (bucketed gradient all-reduce overlapped with backward)."""
    import torch.nn as nn
    from .detnet.nn import training
    from .detnet.nn.detectron2_det import Detectron2Det
    torch.backends.cudnn.benchmark = True
    enable_gemm_tuning()
    dev = torch.device('cuda')
    det = Detectron2Det(seed=0).to(dev).train()
    params = training.set_trainable(det.model)

    class Wrapper(nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

        def forward(self, image_bgr, boxes, classes):
            return sum(training.losses(self.model, image_bgr, boxes, classes).values())

    net = Wrapper(det.model)
    if world > 1:
        net = nn.parallel.DistributedDataParallel(net, device_ids=[torch.cuda.current_device()])
    # detnet/configs/detectron2.cfg; fused=True: weight decay + momentum + update as ONE multi-tensor pass (same arithmetic per element as the
    # foreach form: three passes over 0.5 GB of parameters); WT_SGD_FUSED=0 for the A/B
    opt = torch.optim.SGD(params, lr=0.002, momentum=0.9, weight_decay=1e-4, fused=os.environ.get('WT_SGD_FUSED', '1') != '0')
    g = torch.Generator().manual_seed(rank)
    img = torch.randint(0, 256, (1, 3, 886, 1280), generator=g).float().to(dev)
    n = 30
    wh = torch.rand((n, 2), generator=g) * 280 + 20
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1280 - 300.0, 886 - 300.0])
    boxes = torch.cat((xy, xy + wh), 1).to(dev)
    classes = torch.randint(0, 4, (n,), generator=g).to(dev)
    last = {}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = net(img, boxes, classes)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()
        last['loss'] = loss.detach()

    steps = args.steps or 3
    warmup = args.warmup if args.warmup is not None else 1
    from .detnet.nn import ops
    state = {'n': 0}

    def timed():
        if state['n'] == warmup:
            ops.EVENT_LOG = []                                   # HIP events around the col2im launches of the timed steps only
        state['n'] += 1
        step()

    dt, ev_ms = timed_steps(world, timed, steps, warmup)
    log, ops.EVENT_LOG = ops.EVENT_LOG or [], None
    torch.cuda.synchronize()
    # roofline of the dominant hand-written BACKWARD kernel: deformable col2im (dx as a gather over the inverted sampling table +
    # doffset), HBM-bound: algorithmic bytes per launch / mean launch duration of the most expensive shape
    by = {}
    for tag, work, e0, e1 in log:
        if tag.startswith('deform_col2im') or tag.startswith('deform_dxoff'):
            d = by.setdefault(tag, [0, 0.0, work])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
    roofline = None
    if by:
        # round 4: the stride-1 layers of res3 / res4 run the fused backward (dcol = dY W on the f32 MFMAs inside the kernel, never in HBM): its
        # roof is the MFMA peak and the work logged is algorithmic flops; the remaining layers (stride 2, res5) keep the HBM-bound col2im form
        tag = max(by, key=lambda k: by[k][1])
        cnt, ms, work = by[tag]

        def rate(k, v):
            per_s = v[2] / (v[1] / v[0] * 1e-3)
            return dict(launches=v[0], avg_us=v[1] / v[0] * 1e3, **({'tflops': per_s / 1e12} if k.startswith('deform_dxoff') else {'gbs': per_s / 1e9}))
        if tag.startswith('deform_dxoff'):
            ach = work / (ms / cnt * 1e-3) / 1e12
            roofline = dict(bound='mfma', kernel=tag, achieved=ach, peak=157.3, unit='TFLOP/s', frac=ach / 157.3, traffic=None, launches=cnt,
                            avg_us=ms / cnt * 1e3, flops_per_launch=work,
                            note='avg_us covers the four launches of the fused dX / dOffset path (tables, main kernel, far samples); only the '
                                 'dcol GEMM flops are counted - the gather and the dOffset dot products are VALU work on the same SIMDs',
                            all_shapes={k: rate(k, v) for k, v in by.items()})
        else:
            ach = work / (ms / cnt * 1e-3) / 1e9
            roofline = dict(bound='hbm', kernel=tag, achieved=ach, peak=8000.0, unit='GB/s', frac=ach / 8000.0, traffic=None, launches=cnt,
                            avg_us=ms / cnt * 1e3, bytes_per_launch=work, all_shapes={k: rate(k, v) for k, v in by.items()})
    res = dict(value=world * steps / dt, unit='images/s', ms_per_step=1e3 * dt / steps, dtype='f32',
               workload='Cascade R-CNN X152-32x8d-FPN dconv training step (fwd+bwd+SGD), 886x1280 crop, batch 1/GPU, '
                        '30 synthetic gt boxes, FREEZE_AT 2, FrozenBN, %s' % ('DDP x%d over RCCL' % world if world > 1 else 'single GPU'),
               roofline=roofline,
               extra=dict(loss=float(last['loss']), trainable_params=sum(p.numel() for p in params)))
    res['model'] = det
    return res, steps, warmup
