"""Builds csrc/variants/lib_plabl<mask>.so for tools/split_planes_ablation.sh: timing-only ablations of gemm_split_planes_kernel on an EXPERIMENT COPY of
csrc/det_gemm_split.hip (the switches are not in the product source).  -DWD_PL_ABL=mask: 1 no W loads / waits, 2 no LDS-DMA, 4 no barrier, 16 no fragment reads."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'waymo_2d_tracking_amd', 'csrc')
VAR = os.path.join(CSRC, 'variants')
os.makedirs(os.path.join(VAR, 'src'), exist_ok=True)
s = open(os.path.join(CSRC, 'det_gemm_split.hip')).read()


def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new)


rep('''#define DMA(j_)                                                                                               \\
    if ((j_) < ND) { dma_piece(kt + 3, (unsigned)wr, (j_)); }''', '''#if WD_PL_ABL & 2
#define DMA(j_) if ((j_) < ND) { asm volatile("s_nop 0" ::: "memory"); }
#define ND_ 0
#else
#define DMA(j_)                                                                                               \\
    if ((j_) < ND) { dma_piece(kt + 3, (unsigned)wr, (j_)); }
#define ND_ ND
#endif''')
rep('''        WAITW(0, ND + 3);
        MF(2, 0, 0) RD(a1, 2) SB;''', '''#if !(WD_PL_ABL & 1)
        WAITW(0, ND_ + 3);
#endif
        MF(2, 0, 0) RD(a1, 2) SB;''')
rep('''        MF(0, 0, 0) RD(a1, 0) w_load(2 * kt + 2, 0); SB;
        WAITW(1, 3);''', '''#if WD_PL_ABL & 1
        MF(0, 0, 0) RD(a1, 0) SB;
#else
        MF(0, 0, 0) RD(a1, 0) w_load(2 * kt + 2, 0); SB;
        WAITW(1, 3);
#endif''')
rep('''        MF(0, 0, 1) RD(a0n, 0) w_load(2 * kt + 3, 1); SB;
        __builtin_amdgcn_s_barrier();
        SB;
        const int t = cur; cur = nxt; nxt = nx2; nx2 = wr; wr = t;''', '''#if WD_PL_ABL & 1
        MF(0, 0, 1) RD(a0n, 0) SB;
#else
        MF(0, 0, 1) RD(a0n, 0) w_load(2 * kt + 3, 1); SB;
#endif
#if !(WD_PL_ABL & 4)
        __builtin_amdgcn_s_barrier();
#endif
        SB;
        const int t = cur; cur = nxt; nxt = nx2; nx2 = wr; wr = t;''')
i = s.index('__global__ __launch_bounds__(NTHREADS, 2) void gemm_split_planes_kernel')
j = s.index('#define RD(addr, pl)', i)
k = s.index('#if WD_PL_ABL & 2', j)
s = s[:j] + '#if WD_PL_ABL & 16\n#define RD(addr, pl)\n#else\n' + s[j:k] + '#endif\n' + s[k:]
s = s.replace('#include "common.h"', '#include "../../common.h"').replace('"../../include/waymodet.h"', '"../../../../include/waymodet.h"')
src = os.path.join(VAR, 'src', 'det_gemm_split_abl.hip')
open(src, 'w').write(s)
objs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.o') and not f.endswith('.dbg.o') and f != 'det_gemm_split.o']
for m in (0, 1, 2, 3, 4, 7, 16):
    o = os.path.join(VAR, 'abl_%d.o' % m)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-DWD_PL_ABL=%d' % m, '-c', src, '-o', o])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(VAR, 'lib_plabl%d.so' % m)] + objs + [o])
    os.remove(o)
    print('variants/lib_plabl%d.so' % m)
