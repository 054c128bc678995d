"""smoke(): a few small detector forwards + resident SORT on cuda:0 (called by __graft_entry__.smoke, which hands in
the CPU oracle's track_streams as the checker - this module never imports oracle/)."""
import torch


def run(track_streams=None):
    from .bench_e2e import DetectTrackPipeline, check_against
    pipe = DetectTrackPipeline(n_cameras=2, frames_per_camera=2, height=256, width=384, seed=0, distinct_times=6)
    for _ in range(3):
        pipe.step(True)
    torch.cuda.synchronize()
    n = pipe.n_dets_total
    n_out, births = [int(v) for v in pipe.counts.cpu().tolist()]
    assert n_out >= 0, 'SORT kernel status %d' % -n_out
    if track_streams is not None:
        rep = check_against(pipe, track_streams)
        assert rep['ok'], rep
        print('detector -> SORT smoke ok (rows identical to the oracle replay):', rep)
    else:
        print('detector smoke ok: %d detections, last chunk %d track rows, %d births' % (n, n_out, births))
