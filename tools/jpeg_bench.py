"""GPU JPEG decode throughput vs PIL on the host (tools; results in profiles/r03_jpeg_decode.txt).
    python tools/jpeg_bench.py [n_images]"""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import jpeg_cases as JC                                                                      # noqa: E402
from waymo_2d_tracking_amd.detnet.nn import ops                                              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from PIL import Image
    for (h, w, sub, q, kind, label) in [(1280, 1920, 2, 90, 2, 'front camera 1920x1280, 4:2:0, q90, photo-like'),
                                        (886, 1920, 2, 90, 2, 'side camera 1920x886, 4:2:0, q90, photo-like'),
                                        (1280, 1920, 2, 100, 1, '1920x1280, 4:2:0, q100, white noise (worst case)')]:
        data = JC.encode(JC.synth(h, w, kind, seed=3), quality=q, subsampling=sub)
        out = ops.jpeg_decode(data)
        assert np.array_equal(out.cpu().numpy(), JC.pil_rgb(data))
        import ctypes
        from waymo_2d_tracking_amd import _lib
        st = (ctypes.c_int32 * 4)()
        _lib.lib().wd_jpeg_last_stats(st)
        print('    sync launches %d, most iterations of a workgroup inside one launch %d, subsequence decodes %d for %d subsequences'
              % tuple(st), flush=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            ops.jpeg_decode(data)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t) / n * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.jpeg_decode(data)
        e1.record()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(max(3, n // 4)):
            np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        pil_ms = (time.perf_counter() - t) / max(3, n // 4) * 1e3
        print('%-52s %8d bytes  GPU decode %.2f ms/image (call, 1 thread; %.0f images/s)   PIL %.2f ms/image (1 thread)' %
              (label, len(data), gpu_ms, 1e3 / gpu_ms, pil_ms), flush=True)


if __name__ == '__main__':
    main()
