/* Laboratory entry points (csrc/debug/debug_kernels.hip).  Only in a library built with WD_DEBUG_BUILD=1; not part of the drop-in boundary. */
#ifndef WAYMODET_DEBUG_H
#define WAYMODET_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* canary workgroups (LDS / register / VALU / MFMA self-checks) for co-residency experiments; flags: 5 device uint32 counters */
int wd_debug_canary(int workgroups, int lds_bytes, int spins, unsigned* flags, void* stream);
/* workgroups that only occupy a CU slot (512 threads, lds_bytes of LDS) for ~ticks s_memtime ticks (negative: fill the LDS with NaN patterns first) */
int wd_debug_occupy(int workgroups, int lds_bytes, long long ticks, unsigned* sink, void* stream);
/* n_wg one-wave workgroups holding lds_bytes of LDS for `cycles` shader cycles */
int wd_debug_hold(int n_wg, int lds_bytes, long long cycles, int* sink, void* stream);
/* pure-register matrix-instruction burner: kind 1 bf16 32x32x16 random, 2 zeros, 3 f32 32x32x2 random, 4 bf16 16x16x32 random, 5 bf16 constant operands */
int wd_debug_mfma_burn(int workgroups, int kind, int iters, unsigned* sink, void* stream);
/* per-workgroup s_memtime stamps of the following split-operand launches (8 int64 per workgroup; NULL = off); csrc/det_gemm_split.hip under -DWD_DEBUG */
int wd_gemm_split_debug_stamps(long long* buf);
#ifdef __cplusplus
}
#endif
#endif
