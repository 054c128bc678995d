"""Multi-GPU sharding of the hot path: one process per GPU (torchrun), camera sequences / frames / ensemble groups
sharded with NO data-path collective; the only exchange is collation.

  * SORT: contiguous blocks of (segment, camera) streams per rank (like the reference's contiguous split_dataset,
    /root/reference/detnet/trainer/data/__init__.py:9-18, balanced by frame count).  Track IDs come from a
    process-global counter in the reference (tracking/sort/sort.py:86), so the IDs of rank r are offset by the number
    of tracks born on ranks < r: ONE all_gather of the per-rank birth counts, then a gather of the result rows to
    rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).
  * detection / ensemble: frames / groups are independent; results are gathered to rank 0 in rank order.
"""
import numpy as np


def world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except ImportError:
        pass
    return 1, 0


def contiguous_split(n_items, n_parts):
    """split_dataset(balanced=False) of the reference: equal contiguous slices, the last one takes the remainder."""
    size = n_items // n_parts
    bounds = [[i * size, i * size + size] for i in range(n_parts)]
    bounds[-1][-1] = n_items
    return [tuple(b) for b in bounds]


def balanced_stream_split(frame_counts, n_parts):
    """Contiguous blocks of streams with (nearly) equal frame totals; keeps the global stream order."""
    frame_counts = np.asarray(frame_counts, dtype=np.int64)
    total = int(frame_counts.sum())
    cum = np.concatenate([[0], np.cumsum(frame_counts)])
    bounds, start = [], 0
    for p in range(n_parts):
        if p == n_parts - 1:
            end = len(frame_counts)
        else:
            target = total * (p + 1) / n_parts
            end = int(np.searchsorted(cum, target, side='left'))
            end = max(start, min(end, len(frame_counts)))
        bounds.append((start, end))
        start = end
    return bounds


def gather_object_rank0(obj):
    """Variable-length gather to rank 0 (list in rank order on rank 0, None elsewhere)."""
    w, r = world()
    if w == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * w if r == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def _device():
    import torch.distributed as dist
    return 'cuda' if dist.get_backend() == 'nccl' else 'cpu'


def gather_columns_rank0(columns):
    """Variable-length gather of a dict of equally long 1-D / 2-D arrays (a columnar row set) to rank 0.

    One all_gather of the row counts + ONE padded tensor gather: every rank serialises its columns into a single byte
    buffer (the column names / dtypes / trailing shapes are the same on all ranks, only the row count differs), the buffers
    are padded to the longest one and gathered to rank 0 - over RCCL / xGMI when the backend is "nccl" (device tensors), gloo
    in the CPU tests.  Returns the concatenated columns (rank order) on rank 0, None elsewhere.  No pickling."""
    w, r = world()
    names = sorted(columns)
    cols = {k: np.ascontiguousarray(columns[k]) for k in names}
    n = len(cols[names[0]]) if names else 0
    assert all(len(cols[k]) == n for k in names), 'columns must have the same number of rows'
    if w == 1:
        return cols
    import torch
    import torch.distributed as dist
    counts = all_gather_int(n)
    row_bytes = [cols[k].dtype.itemsize * int(np.prod(cols[k].shape[1:], dtype=np.int64)) for k in names]
    max_rows = max(counts)
    dev = _device()
    buf = np.zeros(max_rows * sum(row_bytes), dtype=np.uint8)
    o = 0
    for k, rb in zip(names, row_bytes):                        # column-major inside the buffer: [col0 rows | col1 rows | ...]
        b = cols[k].reshape(-1).view(np.uint8)
        buf[o:o + b.size] = b
        o += max_rows * rb
    t = torch.from_numpy(buf).to(dev)
    if r == 0:
        parts = [torch.empty_like(t) for _ in range(w)]
        dist.gather(t, parts, dst=0)
    else:
        dist.gather(t, None, dst=0)
        return None
    out = {k: [] for k in names}
    for rank, part in enumerate(parts):
        raw = part.cpu().numpy()
        o = 0
        for k, rb in zip(names, row_bytes):
            seg = raw[o:o + counts[rank] * rb]
            out[k].append(seg.view(cols[k].dtype).reshape((counts[rank],) + cols[k].shape[1:]))
            o += max_rows * rb
    return {k: np.concatenate(v) for k, v in out.items()}


class DeviceRowCollator(object):
    """Collation of fixed-capacity row blocks that already live in HBM: the per-step exchange of the sharded detect+track path.

    Every rank owns, per chunk, ONE contiguous byte block holding its output columns (views of that block are what the SORT
    kernel writes into) and a device-side (rows, births) pair.  `exchange` enqueues on the CURRENT stream
      * an all_gather of the (rows, births) pairs - the birth-count exchange: ids of rank r start after the births of ranks < r
        (the reference's process-global counter, tracking/sort/sort.py:86), and
      * ONE gather of the byte blocks to rank 0 (RCCL over xGMI with backend "nccl", gloo in the CPU tests)
    without any host synchronisation or host staging: nothing is read back, the valid row counts travel next to the rows and
    rank 0 trims when it decodes (`decode`).  Column layout inside a block: [col0 capacity rows | col1 ... ], 8-byte aligned."""

    def __init__(self, spec, capacity, n_blocks, device):
        import torch
        self.w, self.r = world()
        self.spec, self.capacity = [], int(capacity)
        o = 0
        for name, dtype, trailing in spec:
            nbytes = torch.empty(0, dtype=dtype).element_size() * int(np.prod(trailing, dtype=np.int64)) * self.capacity
            self.spec.append((name, dtype, tuple(trailing), o, nbytes))
            o += (nbytes + 7) // 8 * 8
        self.block_bytes = o
        self.blocks = torch.zeros((n_blocks, o), dtype=torch.uint8, device=device)
        self.counts = torch.zeros((n_blocks, 2), dtype=torch.int64, device=device)
        self.all_counts = torch.zeros((n_blocks, self.w, 2), dtype=torch.int64, device=device)
        self.collated = torch.zeros((n_blocks, self.w, o), dtype=torch.uint8, device=device) if (self.r == 0 and self.w > 1) else None
        self.exchanges = 0

    def columns(self, b):
        """Typed views of block b (what the producer writes into)."""
        return self._views(self.blocks[b])

    def _views(self, raw):
        out = {}
        for name, dtype, trailing, o, nbytes in self.spec:
            out[name] = raw[o:o + nbytes].view(dtype).reshape((self.capacity,) + trailing)
        return out

    def exchange(self, b):
        self.exchanges += 1
        if self.w == 1 and not _initialized():
            return
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.all_counts[b].reshape(-1), self.counts[b])
        if self.w == 1:
            return
        if self.r == 0:
            dist.gather(self.blocks[b], [self.collated[b, k] for k in range(self.w)], dst=0)
        else:
            dist.gather(self.blocks[b], None, dst=0)

    def decode(self, b):
        """Rank 0, after a synchronisation: per rank the trimmed numpy columns of block b + the (rows, births) table."""
        if self.w == 1:
            k = int(self.counts[b, 0].item())
            return [{n: v[:k].cpu().numpy() for n, v in self.columns(b).items()}], self.counts[b:b + 1].cpu().numpy()
        counts = self.all_counts[b].cpu().numpy()
        return [{n: v[:int(counts[k, 0])].cpu().numpy() for n, v in self._views(self.collated[b, k]).items()}
                for k in range(self.w)], counts


def _initialized():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except ImportError:
        return False


def all_gather_int(value):
    w, r = world()
    if w == 1:
        return [int(value)]
    import torch
    import torch.distributed as dist
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    outs = [torch.zeros_like(t) for _ in range(w)]
    dist.all_gather(outs, t)
    return [int(o.item()) for o in outs]


def track_packed_sharded(packed, iou_thresholds, max_age, min_hits, score_threshold=None, track_fn=None, id_start=0):
    """Sharded tracking of a packed set of streams (tracking.utils.pack_streams / NativeDetFile.packed layout, identical on
    every rank): each rank tracks its contiguous block of streams (balanced by frame count), the per-rank birth counts are
    all_gathered (the one exchange step: ids of rank r start after the births of ranks < r, sort.py:86) and the result COLUMNS
    (frame index into the full packed set, category, bbox, score, object id) travel to rank 0 in one tensor gather.
    Returns (columns or None, total births)."""
    from .tracking import utils as T
    if track_fn is None:
        track_fn = T.track_packed
    w, r = world()
    so = np.asarray(packed['stream_frame_offsets'])
    lo, hi = balanced_stream_split(np.diff(so), w)[r]
    mine = T.slice_streams(packed, lo, hi)
    out, births = track_fn(mine, iou_thresholds, max_age, min_hits, score_threshold, 0)
    all_births = all_gather_int(births)                        # the one exchange step of the path
    offset = id_start + sum(all_births[:r])
    cols = dict(frame=np.asarray(out['frame'], np.int64) + int(so[lo]), category=np.asarray(out['category'], np.int32),
                bbox=np.asarray(out['bbox'], np.float64).reshape(-1, 4), score=np.asarray(out['score'], np.float64),
                object_id=np.asarray(out['object_id'], np.int64) + offset)
    return gather_columns_rank0(cols), sum(all_births)


def track_all_sharded(predictions, iou_thresholds, max_age, min_hits, segment_ids=None, track_fn=None, id_start=0):
    """Distributed twin of tracking.utils.track_all: rank 0 receives the full list of tracking-JSON rows in the reference's
    order with the reference's global IDs (None on other ranks).
    track_fn(packed, iou_thresholds, max_age, min_hits, score_threshold, id_base) -> (out dict, births); default =
    the HIP path (tracking.utils.track_packed)."""
    from .tracking import utils as T
    keys = [(s, c) for s in predictions if (segment_ids is None or s in segment_ids) for c in predictions[s]]
    packed = T.pack_streams(predictions, keys)
    cols, births = track_packed_sharded(packed, iou_thresholds, max_age, min_hits, None, track_fn, id_start)
    if cols is None:
        return None, births
    return T.format_tracks(packed, cols), births
