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


def track_all_sharded(predictions, iou_thresholds, max_age, min_hits, segment_ids=None, track_fn=None, id_start=0):
    """Distributed twin of tracking.utils.track_all: every rank tracks its block of streams, rank 0 receives the
    full list of tracking-JSON rows in the reference's order with the reference's global IDs (None on other ranks).
    track_fn(packed, iou_thresholds, max_age, min_hits, score_threshold, id_base) -> (out dict, births); default =
    the HIP path (tracking.utils.track_packed)."""
    from .tracking import utils as T
    if track_fn is None:
        track_fn = T.track_packed
    w, r = world()
    keys = [(s, c) for s in predictions if (segment_ids is None or s in segment_ids) for c in predictions[s]]
    counts = [len(predictions[s][c]) for s, c in keys]
    lo, hi = balanced_stream_split(counts, w)[r]
    packed = T.pack_streams(predictions, keys[lo:hi])
    out, births = track_fn(packed, iou_thresholds, max_age, min_hits, None, 0)
    all_births = all_gather_int(births)                        # the one exchange step of the path
    offset = id_start + sum(all_births[:r])
    out = dict(out)
    out['object_id'] = out['object_id'] + offset
    rows = T.format_tracks(packed, out)
    gathered = gather_object_rank0(rows)
    if r != 0:
        return None, sum(all_births)
    return [row for part in gathered for row in part], sum(all_births)
