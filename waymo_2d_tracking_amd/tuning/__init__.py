"""Library-GEMM selection for the detector's 1x1 convolutions / FC layers.

The 1x1 convolutions of the NHWC backbone are plain library GEMMs (hipBLASLt / rocBLAS through torch).
PyTorch's TunableOp times the candidate solutions of both libraries per GEMM shape and keeps the fastest; on MI355X that is
worth ~6 % end-to-end over the default heuristic pick (bench.py: 24.1 -> 25.7 frames/s).  ``tunableop_gfx950.csv`` holds
the selections measured on an MI355X with this image's torch / hipBLASLt build (``tools/tune_gemms.sh``); it is loaded
when its validators (torch, ROCm, hipBLASLt, rocBLAS versions, gfx arch) match; shapes missing from it use the library's
default pick (or are tuned online with ``online=True`` / ``WT_GEMM_TUNING_ONLINE=1``).  This is plumbing around library calls - the same role as the reference's
``--cudnn-benchmark`` flag (detnet/inference.py:58).
"""
import os
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
PACKAGED = os.path.join(HERE, 'tunableop_gfx950.csv')
_done = False


def enable_gemm_tuning(online=None, max_tuning_ms=15, max_iterations=30):
    """Load the packaged selections.  online=True (or WT_GEMM_TUNING_ONLINE=1, used by tools/tune_gemms.sh) additionally
    tunes GEMM shapes that are not in the file when they first occur; it is off by default because the box-head GEMMs have
    a data-dependent row count (number of proposals) and would trigger a tuning run for every new value."""
    global _done
    import torch
    if online is None:
        online = os.environ.get('WT_GEMM_TUNING_ONLINE') == '1'
    if _done or not torch.cuda.is_available():
        return False
    t = torch.cuda.tunable
    t.enable(True)
    t.set_max_tuning_duration(int(os.environ.get('WT_TUNE_MS', max_tuning_ms)))           # tools/tune_gemms_long.sh: longer searches
    t.set_max_tuning_iterations(int(os.environ.get('WT_TUNE_ITERS', max_iterations)))
    # new selections of this process go to a scratch file (never into the package)
    t.set_filename(os.environ.get('WT_TUNABLEOP_OUT', os.path.join(tempfile.gettempdir(), 'wt_tunableop_%d.csv' % os.getpid())))
    loaded = False
    # WT_TUNABLEOP_IN: another selection file instead of the packaged one ('' = none, start from the library defaults) - A/B experiments
    # (tools/tune_gemms_long.sh) select their file here and never touch the packaged one
    source = os.environ.get('WT_TUNABLEOP_IN', PACKAGED)
    if source and os.path.exists(source):
        try:
            loaded = bool(t.read_file(source))
        except Exception:
            loaded = False
    t.tuning_enable(bool(online))
    _done = True
    return loaded
