// Small HBM-bound helpers of the detector graph (include/waymodet.h).
#include "common.h"
#include "../../include/waymodet.h"

namespace {

// y = act(y + bias[col]) in place; float4 per lane, grid-stride, 2048 workgroups max (G11/G13)
__global__ __launch_bounds__(256) void bias_relu_kernel(float4* __restrict__ y, const float4* __restrict__ bias, long n4,
                                                        int cols4, int relu) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = y[i];
        const float4 b = bias ? bias[i % cols4] : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        y[i] = v;
    }
}

// GroupNorm + ReLU in place on (R, HW, C) NHWC rows (box-head GN(32, 256) after each 3x3 conv, job.log:1150).
// One workgroup per ROI, one thread per channel: the HW (<= 64) values of a channel stay in registers, group
// statistics are reduced across the CPG lanes of the group with xor-shuffles (two-pass mean / variance).
template <int CPG>
__global__ __launch_bounds__(256) void groupnorm_relu_kernel(const float* x, float* y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int HW, int C, float eps,
                                                             int relu) {
    const float* base = x + (size_t)blockIdx.x * HW * C;
    float* obase = y + (size_t)blockIdx.x * HW * C;          // y == x: in place (a thread only rewrites what it read)
    for (int c = threadIdx.x; c < C; c += 256) {            // C is a multiple of CPG; 256 % CPG == 0
        float v[64];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            v[i] = (i < HW) ? base[(size_t)i * C + c] : 0.f;
            s += v[i];
        }
#pragma unroll
        for (int o = 1; o < CPG; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)(HW * CPG);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float d = (i < HW) ? v[i] - mean : 0.f;
            q += d * d;
        }
#pragma unroll
        for (int o = 1; o < CPG; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q / (float)(HW * CPG) + eps);
        const float g = gamma[c] * rstd, b = beta[c] - mean * gamma[c] * rstd;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i < HW) {
                float o = v[i] * g + b;
                if (relu) o = fmaxf(o, 0.f);
                obase[(size_t)i * C + c] = o;
            }
        }
    }
}

// Backward of GroupNorm + ReLU on the same layout (training graph, round 4: torch's group_norm copied every channels_last box-head map to
// NCHW and back - 42 x 26 MB per step).  Same mapping: the HW values of a channel (input x and incoming gradient) stay in registers, the
// statistics are recomputed, and with g = dy * (y > 0), xh = (x - mean) * rstd, m = HW * CPG:
//   dbeta[c] += sum_i g, dgamma[c] += sum_i g xh (one float atomic per channel and ROI)
//   dx = rstd * (gamma g - A / m - xh B / m),  A = sum over the group of gamma g, B = sum over the group of gamma g xh
template <int CPG>
__global__ __launch_bounds__(256) void groupnorm_relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int HW, int C, float eps, int relu,
                                                                 float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const size_t off = (size_t)blockIdx.x * HW * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float v[64], g[64];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            v[i] = (i < HW) ? x[off + (size_t)i * C + c] : 0.f;
            g[i] = (i < HW) ? dy[off + (size_t)i * C + c] : 0.f;
            s += v[i];
        }
#pragma unroll
        for (int o = 1; o < CPG; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)(HW * CPG);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float d = (i < HW) ? v[i] - mean : 0.f;
            q += d * d;
        }
#pragma unroll
        for (int o = 1; o < CPG; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q / (float)(HW * CPG) + eps);
        const float ga = gamma[c], be = beta[c];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float xh = (v[i] - mean) * rstd;
            if (relu && !(xh * ga + be > 0.f)) g[i] = 0.f;
            v[i] = xh;
            sg += g[i];
            sgx += g[i] * xh;
        }
        atomicAdd(dbeta + c, sg);
        atomicAdd(dgamma + c, sgx);
        float A = ga * sg, B = ga * sgx;
#pragma unroll
        for (int o = 1; o < CPG; o <<= 1) { A += __shfl_xor(A, o, 64); B += __shfl_xor(B, o, 64); }
        const float im = 1.f / (float)(HW * CPG);
        A *= im; B *= im;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (i < HW) dx[off + (size_t)i * C + c] = rstd * (ga * g[i] - A - v[i] * B);
    }
}

// 3x3 convolution with FEW output channels (the 18-channel offset conv of every deformable block, job.log:412) as
// "GEMM then shift-add": a library GEMM computes partial[p][tap*n_out + n] = sum_c x[p][c] * w[n][c][tap] for every
// INPUT pixel p (N = 9*n_out = 162 fills the MFMA tiles, where a direct implicit GEMM pads 18 -> 32), and this kernel
// gathers out[y][x][n] = bias[n] + sum_tap partial[(y*s + kh - 1, x*s + kw - 1)][tap*n_out + n] (zero padding).
// One thread per output element; the 9 reads of a wave are 9 x (64 / n_out) contiguous n_out-float runs.
__global__ __launch_bounds__(256) void tap_shift_add_kernel(const float* __restrict__ partial, int ld, int n_out,
                                                            const float* __restrict__ bias, int batch, int H, int W,
                                                            int Ho, int Wo, int stride, float* __restrict__ out) {
    const long total = (long)batch * Ho * Wo * n_out;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % n_out);
        long p = i / n_out;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho), b = (int)(p / Ho);
        float acc = bias ? bias[n] : 0.f;
        const float* base = partial + (size_t)b * H * W * ld + n;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int y = oy * stride + kh - 1;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int x = ox * stride + kw - 1;
                if (x < 0 || x >= W) continue;
                acc += base[((size_t)y * W + x) * ld + (kh * 3 + kw) * n_out];
            }
        }
        out[i] = acc;
    }
}

}  // namespace

extern "C" int wd_groupnorm_relu_out_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                                              float eps, int relu, void* stream);

extern "C" int wd_groupnorm_relu_nhwc_f32(float* x, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                                          float eps, int relu, void* stream) {
    return wd_groupnorm_relu_out_nhwc_f32(x, x, gamma, beta, n, hw, c, groups, eps, relu, stream);
}

static int gn_check(const char* who, int hw, int c, int groups) {
    const int cpg = groups > 0 ? c / groups : 0;
    if (groups <= 0 || c % groups || hw < 1 || hw > 64 || c % 64 || (cpg != 4 && cpg != 8 && cpg != 16 && cpg != 32)) {
        wt::set_error("%s: need hw <= 64, C %% 64 == 0, 4 / 8 / 16 / 32 channels per group (hw=%d C=%d groups=%d)", who, hw, c, groups);
        return WT_ERR_INVALID;
    }
    return WT_OK;
}

extern "C" int wd_groupnorm_relu_bwd_nhwc_f32(const float* x, const float* dy, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                                              float eps, int relu, float* dx, float* dgamma, float* dbeta, void* stream) {
    WT_TRY(wt::ensure_device());
    WT_TRY(gn_check("wd_groupnorm_relu_bwd_nhwc_f32", hw, c, groups));
    hipStream_t st = (hipStream_t)stream;
    WT_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * c, st));
    WT_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * c, st));
    if (n <= 0) return WT_OK;
    const int cpg = c / groups;
#define WD_GNB(CPG) hipLaunchKernelGGL(groupnorm_relu_bwd_kernel<CPG>, dim3((unsigned)n), dim3(256), 0, st, x, dy, gamma, beta, hw, c, eps, relu, dx, dgamma, dbeta)
    if (cpg == 8) WD_GNB(8);
    else if (cpg == 4) WD_GNB(4);
    else if (cpg == 16) WD_GNB(16);
    else WD_GNB(32);
#undef WD_GNB
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" int wd_groupnorm_relu_out_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta, int n, int hw, int c, int groups,
                                              float eps, int relu, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n <= 0) return WT_OK;
    const int cpg = groups > 0 ? c / groups : 0;
    if (groups <= 0 || c % groups || hw < 1 || hw > 64 || c % 64) {
        wt::set_error("wd_groupnorm_relu_nhwc_f32: need hw <= 64, C %% 64 == 0, C %% groups == 0 (hw=%d C=%d groups=%d)", hw, c, groups);
        return WT_ERR_INVALID;
    }
#define WD_GN(CPG) hipLaunchKernelGGL(groupnorm_relu_kernel<CPG>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta, hw, c, eps, relu)
    if (cpg == 8) WD_GN(8);
    else if (cpg == 4) WD_GN(4);
    else if (cpg == 16) WD_GN(16);
    else if (cpg == 32) WD_GN(32);
    else { wt::set_error("wd_groupnorm_relu_nhwc_f32: channels per group must be 4, 8, 16 or 32 (got %d)", cpg); return WT_ERR_INVALID; }
#undef WD_GN
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" int wd_bias_relu_f32(float* y, const float* bias, long m, int n, int relu, void* stream) {
    WT_TRY(wt::ensure_device());
    if (m <= 0 || n <= 0) return WT_OK;
    if ((n & 3) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) {
        wt::set_error("wd_bias_relu_f32: N must be a multiple of 4 and pointers 16-byte aligned");
        return WT_ERR_INVALID;
    }
    const long n4 = m * (long)(n / 4);
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(bias_relu_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream,
                       (float4*)y, (const float4*)bias, n4, n / 4, relu);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// Backward of y = act(z * scale[col] + bias[col]) with respect to z, given y (training: the FrozenBN affine + ReLU behind a deformable /
// 1x1 convolution, fused into that op's forward): g[m][n] = dy[m][n] * (y[m][n] > 0 if relu) * scale[n] - one pass instead of
// threshold_backward + the broadcast multiply of autograd.
__global__ __launch_bounds__(256) void act_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                      const float4* __restrict__ scale, long n4, int cols4, int relu, float4* __restrict__ g) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 d = dy[i];
        if (relu) {
            const float4 v = y[i];
            d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f; d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
        }
        if (scale) {
            const float4 sc = scale[i % cols4];
            d.x *= sc.x; d.y *= sc.y; d.z *= sc.z; d.w *= sc.w;
        }
        g[i] = d;
    }
}

extern "C" int wd_act_bwd_f32(const float* dy, const float* y, const float* scale, long m, int n, int relu, float* g, void* stream) {
    WT_TRY(wt::ensure_device());
    if (m <= 0 || n <= 0) return WT_OK;
    if ((n & 3) || !dy || !g || (relu && !y) || ((uintptr_t)dy & 15) || ((uintptr_t)g & 15) || (y && ((uintptr_t)y & 15)) ||
        (scale && ((uintptr_t)scale & 15))) {
        wt::set_error("wd_act_bwd_f32: N must be a multiple of 4 and pointers 16-byte aligned");
        return WT_ERR_INVALID;
    }
    const long n4 = m * (long)(n / 4);
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, (const float4*)y, (const float4*)scale, n4, n / 4, relu, (float4*)g);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" int wd_tap_shift_add_f32(const float* partial, int ld, int n_out, const float* bias, int batch, int h, int w,
                                    int stride, float* out, void* stream) {
    WT_TRY(wt::ensure_device());
    if (batch <= 0 || h <= 0 || w <= 0) return WT_OK;
    if (!partial || !out || n_out <= 0 || ld < 9 * n_out || stride < 1) {
        wt::set_error("wd_tap_shift_add_f32: invalid arguments (ld=%d n_out=%d stride=%d)", ld, n_out, stride);
        return WT_ERR_INVALID;
    }
    const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
    const long total = (long)batch * ho * wo * n_out;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tap_shift_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partial, ld, n_out, bias,
                       batch, h, w, ho, wo, stride, out);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// Nearest-neighbour x2 upsampling of an NHWC map (the FPN top-down pathway: F.interpolate(scale_factor=2, mode="nearest"),
// detectron2 fpn.py) - one source float4 -> the 2 x 2 output float4s; 64 lanes cover 256 consecutive channels.  HBM-bound: reads the
// source once, writes 4x that (the library kernel behind F.interpolate reaches 1.4 TB/s on the 157 MB p2 map; this one ~4 TB/s).
__global__ __launch_bounds__(256) void upsample2x_nhwc_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n_src4, int w,
                                                             int c4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_src4; i += (long)gridDim.x * 256) {
        const float4 v = src[i];
        const long pix = i / c4;
        const int ch = (int)(i - pix * c4);
        const long row = pix / w;                     // n * h + y
        const int x = (int)(pix - row * w);
        float4* o = dst + ((row * 2) * (2L * w) + 2 * x) * c4 + ch;
        o[0] = v; o[c4] = v;
        o += 2L * w * c4;
        o[0] = v; o[c4] = v;
    }
}

extern "C" int wd_upsample2x_nhwc_f32(const float* src, int batch, int h, int w, int c, float* dst, void* stream) {
    WT_TRY(wt::ensure_device());
    if (batch <= 0 || h <= 0 || w <= 0 || c <= 0) return WT_OK;
    if (!src || !dst || (c & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {
        wt::set_error("wd_upsample2x_nhwc_f32: C must be a multiple of 4 and pointers 16-byte aligned");
        return WT_ERR_INVALID;
    }
    const long n4 = (long)batch * h * w * (c / 4);
    long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                       n4, w, c / 4);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
