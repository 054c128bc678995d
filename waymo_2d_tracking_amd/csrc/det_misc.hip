// Small HBM-bound helpers of the detector graph (include/waymodet.h).
#include "common.h"
#include "../../include/waymodet.h"

namespace {

// y = act(y + bias[col]) in place; float4 per lane, grid-stride, 2048 workgroups max (G11/G13)
__global__ __launch_bounds__(256) void bias_relu_kernel(float4* __restrict__ y, const float4* __restrict__ bias, long n4,
                                                        int cols4, int relu) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = y[i];
        const float4 b = bias[i % cols4];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        y[i] = v;
    }
}

}  // namespace

extern "C" int wd_bias_relu_f32(float* y, const float* bias, long m, int n, int relu, void* stream) {
    WT_TRY(wt::ensure_device());
    if (m <= 0 || n <= 0) return WT_OK;
    if ((n & 3) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15)) {
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
