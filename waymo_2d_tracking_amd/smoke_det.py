"""smoke(): one tiny detector forward + SORT on cuda:0 (called by __graft_entry__.smoke)."""
import torch


def run():
    from .bench_e2e import DetectTrackPipeline
    pipe = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=256, width=384, seed=0)
    n = pipe.step(True)
    torch.cuda.synchronize()
    n_out, births = [int(v) for v in pipe.counts.cpu().tolist()]
    assert n_out >= 0, 'SORT kernel status %d' % -n_out
    print('detector smoke ok: %d detections over %d frames, %d track rows, %d births' % (n, pipe.n_frames, n_out, births))
