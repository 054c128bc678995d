// Fused image pre-processing in front of the detector (SURVEY rows a21 + a16, §8f rank 3): ONE pass over HBM replaces
//   TTA.pre_process            detnet/nn/tta.py:179-190 (ResizeTTA: F.interpolate bilinear, align_corners=False),
//                              detnet/nn/tta.py:147-156 (HFlipTTA / VFlipTTA: torch.flip)
//   Detectron2Det.forward      detnet/nn/detectron2_det/__init__.py:70-74 (RGB -> BGR)
//   detectron2 preprocess_image (x - PIXEL_MEAN) / PIXEL_STD, ImageList.from_tensors zero padding to a multiple of 32
// and writes the NHWC tensor the stem convolution reads.  HBM bound: reads H*W*3 source elements once (the 4 bilinear
// corners of neighbouring outputs hit L1/L2), writes Hp*Wp*3 floats once.
//
// Work item = 4 consecutive output pixels (12 floats, three float4 stores; a wavefront covers 3 KiB contiguous) so the
// bilinear coordinates are computed once per pixel and the planar / interleaved source reads stay coalesced.  Bilinear arithmetic follows ATen's
// upsample_bilinear2d (area_pixel_compute_source_index, float accumulation type).
#include "common.h"
#include "det_boxmath.h"
#include "../../include/waymodet.h"

namespace {

struct PreArgs {
    int n, h, w;            // source
    int ho, wo;             // resized (valid) extent
    int hp, wp;             // padded output extent
    float rh, rw;           // source step per output pixel (1/scale)
    int resize, hflip, vflip, swap_rb;
    float mean[3], inv_or_std[3];
};

template <typename T, bool HWC>
__device__ __forceinline__ float src_at(const T* __restrict__ src, const PreArgs& a, int n, int c, int y, int x) {
    if (HWC) return (float)src[(((size_t)n * a.h + y) * a.w + x) * 3 + c];
    return (float)src[(((size_t)n * 3 + c) * a.h + y) * a.w + x];
}

// ATen computes the source coordinate scale * (dst + 0.5) - 0.5 inside a kernel compiled with FMA contraction; the
// fused form is reproduced explicitly (at x ~ 1900 one ulp of the coordinate moves a 0..255 value by up to 0.03).
__device__ __forceinline__ void src_coord(float r, int dst, int size, int resize, int& i0, int& i1, float& l) {
    if (!resize) { i0 = i1 = dst; l = 0.f; return; }
    float s = fmaf(r, (float)dst + 0.5f, -0.5f);
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 > size - 1 ? size - 1 : i0;
    i1 = i0 + (i0 < size - 1 ? 1 : 0);
    l = s - (float)i0;
}

template <typename T, bool HWC>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ src, PreArgs a, float* __restrict__ out) {
    // work item = 4 consecutive output pixels of one row = 12 floats = three float4 stores (Wp is a multiple of 4)
    const int row_q = a.wp / 4;
    const long total = (long)a.n * a.hp * row_q;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % row_q);
        const long t = i / row_q;
        const int oy = (int)(t % a.hp), n = (int)(t / a.hp);
        float v[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) v[j] = 0.f;
        if (oy < a.ho && q * 4 < a.wo) {
            int y0, y1;
            float ly;
            src_coord(a.rh, a.vflip ? a.ho - 1 - oy : oy, a.h, a.resize, y0, y1, ly);
            const float hy = 1.f - ly;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int ox = q * 4 + p;
                if (ox >= a.wo) continue;
                int x0, x1;
                float lx;
                src_coord(a.rw, a.hflip ? a.wo - 1 - ox : ox, a.w, a.resize, x0, x1, lx);
                const float hx = 1.f - lx;
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const int ci = a.swap_rb ? 2 - co : co;
                    float val;
                    if (a.resize) {
                        const float p00 = src_at<T, HWC>(src, a, n, ci, y0, x0), p01 = src_at<T, HWC>(src, a, n, ci, y0, x1);
                        const float p10 = src_at<T, HWC>(src, a, n, ci, y1, x0), p11 = src_at<T, HWC>(src, a, n, ci, y1, x1);
                        val = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
                    } else {
                        val = src_at<T, HWC>(src, a, n, ci, y0, x0);
                    }
                    v[p * 3 + co] = (val - a.mean[co]) / a.inv_or_std[co];
                }
            }
        }
        float4* o = reinterpret_cast<float4*>(out) + i * 3;
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
        o[2] = make_float4(v[8], v[9], v[10], v[11]);
    }
}

// Box2BoxTransform.apply_deltas (detectron2) + Boxes.clip in one launch: replaces ~22 elementwise torch
// launches per call (8 calls per frame: 5 RPN levels + 3 cascade stages).  The arithmetic is the torch sequence
// operation by operation (this unit is built with -ffp-contract=off), so the result is bit-identical to it.
__global__ __launch_bounds__(256) void decode_boxes_kernel(const float4* __restrict__ deltas, const float4* __restrict__ boxes,
                                                           const long long* __restrict__ index, int n, float wx, float wy, float ww,
                                                           float wh, float scale_clamp, float clip_w, float clip_h,
                                                           float4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long j = index ? index[i] : i;
    out[i] = wd::decode_box(deltas[j], boxes[j], wx, wy, ww, wh, scale_clamp, clip_w, clip_h);
}

}  // namespace

extern "C" int wd_preprocess_out_shape(int h, int w, double scale, int divisor, int* ho, int* wo, int* hp, int* wp) {
    if (h <= 0 || w <= 0 || !(scale > 0.0) || divisor <= 0 || !ho || !wo || !hp || !wp) {
        wt::set_error("wd_preprocess_out_shape: invalid arguments (h=%d w=%d scale=%g divisor=%d)", h, w, scale, divisor);
        return WT_ERR_INVALID;
    }
    // F.interpolate(scale_factor=s): output size = floor(float(input_size) * s), computed in double
    *ho = scale == 1.0 ? h : (int)((double)h * scale);
    *wo = scale == 1.0 ? w : (int)((double)w * scale);
    if (*ho <= 0 || *wo <= 0) {
        wt::set_error("wd_preprocess_out_shape: scale %g collapses a %dx%d image", scale, h, w);
        return WT_ERR_INVALID;
    }
    *hp = (*ho + divisor - 1) / divisor * divisor;
    *wp = (*wo + divisor - 1) / divisor * divisor;
    return WT_OK;
}

// ImageOps.autocontrast(image) of PIL with cutoff 0 (the reference's AutoContrast transform, trainer/transforms/vision.py:1069-1075;
// README.md:37 runs the detector with --auto-contrast=1) on an (H, W, 3) uint8 image, in place.  Per channel: lo / hi = darkest /
// brightest value present; hi <= lo: unchanged; else lut[v] = clamp(int(v * scale + offset), 0, 255) with scale = 255.0 / (hi - lo),
// offset = -lo * scale in Python floats: four separately rounded float64 operations (this unit is compiled with -ffp-contract=off).
// Two launches: min / max by wave reduction + 6 atomics per wave, then every workgroup builds the 3 x 256 table in LDS and maps.
__global__ __launch_bounds__(256) void autocontrast_minmax_kernel(const uint8_t* __restrict__ img, long n_pix, unsigned* __restrict__ mm) {
    unsigned lo[3] = {255u, 255u, 255u}, hi[3] = {0u, 0u, 0u};
    const long n4 = n_pix / 4;                                  // 4 pixels = 12 bytes = 3 words
    const uint32_t* w = reinterpret_cast<const uint32_t*>(img);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint32_t a = w[3 * i], b = w[3 * i + 1], c = w[3 * i + 2];
        const unsigned px[12] = {a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, a >> 24, b & 255u, (b >> 8) & 255u, (b >> 16) & 255u, b >> 24,
                                 c & 255u, (c >> 8) & 255u, (c >> 16) & 255u, c >> 24};
#pragma unroll
        for (int k = 0; k < 12; ++k) { lo[k % 3] = min(lo[k % 3], px[k]); hi[k % 3] = max(hi[k % 3], px[k]); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long p = n4 * 4; p < n_pix; ++p)
            for (int c = 0; c < 3; ++c) { lo[c] = min(lo[c], (unsigned)img[3 * p + c]); hi[c] = max(hi[c], (unsigned)img[3 * p + c]); }
    __shared__ unsigned red[4][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = min(lo[c], (unsigned)__shfl_xor((int)lo[c], off));
            hi[c] = max(hi[c], (unsigned)__shfl_xor((int)hi[c], off));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][c] = lo[c]; red[threadIdx.x >> 6][3 + c] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {                                      // six atomics per workgroup (a few hundred per address and image)
        unsigned v = red[0][threadIdx.x];
        for (int k = 1; k < 4; ++k) v = threadIdx.x < 3 ? min(v, red[k][threadIdx.x]) : max(v, red[k][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&mm[threadIdx.x], v); else atomicMax(&mm[threadIdx.x], v);
    }
}

__global__ __launch_bounds__(256) void autocontrast_apply_kernel(uint8_t* __restrict__ img, long n_pix, const unsigned* __restrict__ mm) {
    __shared__ uint8_t lut[3][256];
    for (int t = threadIdx.x; t < 768; t += 256) {
        const int c = t >> 8, v = t & 255;
        const int lo = (int)mm[c], hi = (int)mm[3 + c];
        int o = v;
        if (hi > lo) {
            const double scale = 255.0 / (double)(hi - lo);
            const double offset = (double)(-lo) * scale;
            const double x = (double)v * scale + offset;        // two roundings (no FMA contraction in this unit)
            o = (int)x;
            o = o < 0 ? 0 : (o > 255 ? 255 : o);
        }
        lut[c][v] = (uint8_t)o;
    }
    __syncthreads();
    const long n4 = n_pix / 4;
    uint32_t* w = reinterpret_cast<uint32_t*>(img);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        uint32_t a = w[3 * i], b = w[3 * i + 1], c = w[3 * i + 2];
        a = lut[0][a & 255u] | (lut[1][(a >> 8) & 255u] << 8) | (lut[2][(a >> 16) & 255u] << 16) | ((uint32_t)lut[0][a >> 24] << 24);
        b = lut[1][b & 255u] | (lut[2][(b >> 8) & 255u] << 8) | (lut[0][(b >> 16) & 255u] << 16) | ((uint32_t)lut[1][b >> 24] << 24);
        c = lut[2][c & 255u] | (lut[0][(c >> 8) & 255u] << 8) | (lut[1][(c >> 16) & 255u] << 16) | ((uint32_t)lut[2][c >> 24] << 24);
        w[3 * i] = a; w[3 * i + 1] = b; w[3 * i + 2] = c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long p = n4 * 4; p < n_pix; ++p)
            for (int c = 0; c < 3; ++c) img[3 * p + c] = lut[c][img[3 * p + c]];
}

extern "C" int wd_autocontrast_u8(uint8_t* img, int h, int w, void* workspace24, void* stream) {
    WT_TRY(wt::ensure_device());
    if (!img || !workspace24 || h < 1 || w < 1 || ((uintptr_t)img & 3) || ((uintptr_t)workspace24 & 3)) {
        wt::set_error("wd_autocontrast_u8: (H, W, 3) uint8 image, 4-byte aligned, and a 24-byte device workspace");
        return WT_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    WT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(workspace24), 255, 3, st));                    // min of the channels
    WT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(static_cast<char*>(workspace24) + 12), 0, 3, st));   // max (memset nodes: capturable)
    const long n_pix = (long)h * w;
    long blocks = (n_pix / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(autocontrast_minmax_kernel, dim3((unsigned)(blocks > 512 ? 512 : blocks)), dim3(256), 0, st, img, n_pix,
                       (unsigned*)workspace24);
    hipLaunchKernelGGL(autocontrast_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, n_pix, (const unsigned*)workspace24);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" int wd_preprocess_f32(const void* src, int src_layout, int batch, int h, int w, double scale, int hflip, int vflip,
                                 int swap_rb, const float* mean3, const float* std3, int divisor, float* out, void* stream_) {
    WT_TRY(wt::ensure_device());
    if (batch <= 0) return WT_OK;
    PreArgs a;
    WT_TRY(wd_preprocess_out_shape(h, w, scale, divisor, &a.ho, &a.wo, &a.hp, &a.wp));
    if (!src || !out || (src_layout != WD_LAYOUT_NCHW_F32 && src_layout != WD_LAYOUT_NHWC_U8) || (divisor & 3) ||
        ((uintptr_t)out & 15)) {
        wt::set_error("wd_preprocess_f32: invalid arguments (layout=%d divisor=%d; divisor must be a multiple of 4, out 16-byte aligned)",
                      src_layout, divisor);
        return WT_ERR_INVALID;
    }
    a.n = batch; a.h = h; a.w = w;
    a.resize = scale != 1.0;
    // ATen: the user-provided scale_factor is used for the coordinate map (recompute_scale_factor unset): 1/scale
    a.rh = (float)(1.0 / scale);
    a.rw = (float)(1.0 / scale);
    a.hflip = hflip != 0; a.vflip = vflip != 0; a.swap_rb = swap_rb != 0;
    for (int c = 0; c < 3; ++c) {
        a.mean[c] = mean3 ? mean3[c] : 0.f;
        a.inv_or_std[c] = std3 ? std3[c] : 1.f;
    }
    const long total = (long)batch * a.hp * (a.wp / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t stream = (hipStream_t)stream_;
    if (src_layout == WD_LAYOUT_NCHW_F32)
        hipLaunchKernelGGL((preprocess_kernel<float, false>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)src, a, out);
    else
        hipLaunchKernelGGL((preprocess_kernel<uint8_t, true>), dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)src, a, out);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

extern "C" int wd_decode_boxes_f32(const float* deltas, const float* boxes, const int64_t* index, int n, float wx, float wy, float ww,
                                   float wh, float scale_clamp, float clip_w, float clip_h, float* out, void* stream) {
    WT_TRY(wt::ensure_device());
    if (n <= 0) return WT_OK;
    if (!deltas || !boxes || !out || ((uintptr_t)deltas & 15) || ((uintptr_t)boxes & 15) || ((uintptr_t)out & 15) ||
        wx == 0.f || wy == 0.f || ww == 0.f || wh == 0.f) {
        wt::set_error("wd_decode_boxes_f32: null / unaligned pointer or zero weight");
        return WT_ERR_INVALID;
    }
    hipLaunchKernelGGL(decode_boxes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)deltas, (const float4*)boxes, (const long long*)index, n, wx, wy, ww, wh, scale_clamp, clip_w,
                       clip_h, (float4*)out);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
