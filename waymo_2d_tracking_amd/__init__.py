"""waymo_2d_tracking_amd - MI355X-native detect -> ensemble -> SORT hot path of xuyuan/waymo_2d_tracking.

Host side in Python (the reference is Python), mirroring the reference's module layout and call
signatures; all arithmetic of the hot path runs in hand-written HIP kernels (gfx950) behind the C ABI
of ``libwaymotrack.so`` (include/waymotrack.h, include/waymodet.h).  There is no CPU fallback: importing
the package works anywhere, but every operator raises if the library or a GPU is missing.
"""
__version__ = '0.1.0'
