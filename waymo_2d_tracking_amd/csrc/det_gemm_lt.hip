// 1x1 convolution with residual as ONE library GEMM: out = relu?(A . W^T + residual + bias) through hipBLASLt's beta term
// and RELU_BIAS epilogue.  The plain GEMMs of the detector are library work (hipBLASLt sustains 130 TFLOP/s fp32 on these
// shapes); what this unit adds is the fused epilogue - the separate bias + ReLU pass over the block output (66 launches,
// 1.15 ms per 1920x1280 frame) disappears - and an explicit per-shape algorithm choice: the heuristic's candidates are
// timed once per shape (outside any stream capture) and the fastest is cached.
//
// hipBLASLt is resolved with dlopen at first use (libhipblaslt.so.1: PyTorch-ROCm's bundled copy when torch is loaded,
// /opt/rocm's otherwise), so the tracking / ensemble entry points of libwaymotrack.so do not depend on it.
#include "common.h"
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "../../include/waymodet.h"

namespace {

struct Api {
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    hipblasLtHandle_t handle = nullptr;
    bool ok = false;
};

Api g_api;
std::mutex g_mu;

int load_api() {
    if (g_api.ok) return WT_OK;
    void* h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { wt::set_error("hipBLASLt not found: %s", dlerror()); return WT_ERR_NO_DEVICE; }
#define WD_SYM(field, name)                                                              \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(h, name));               \
    if (!g_api.field) { wt::set_error("hipBLASLt symbol %s missing", name); return WT_ERR_NO_DEVICE; }
    WD_SYM(Create, "hipblasLtCreate")
    WD_SYM(DescCreate, "hipblasLtMatmulDescCreate")
    WD_SYM(DescSet, "hipblasLtMatmulDescSetAttribute")
    WD_SYM(LayoutCreate, "hipblasLtMatrixLayoutCreate")
    WD_SYM(PrefCreate, "hipblasLtMatmulPreferenceCreate")
    WD_SYM(PrefSet, "hipblasLtMatmulPreferenceSetAttribute")
    WD_SYM(PrefDestroy, "hipblasLtMatmulPreferenceDestroy")
    WD_SYM(Heuristic, "hipblasLtMatmulAlgoGetHeuristic")
    WD_SYM(Matmul, "hipblasLtMatmul")
#undef WD_SYM
    if (g_api.Create(&g_api.handle) != HIPBLAS_STATUS_SUCCESS) { wt::set_error("hipblasLtCreate failed"); return WT_ERR_HIP; }
    g_api.ok = true;
    return WT_OK;
}

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t lw = nullptr, la = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool tuned = false;
    float best_us = 0.f;
    int candidates = 0;
};

using Key = std::tuple<int, int, int, int, int, int>;     // m, n, k, relu, bias, residual
std::map<Key, Plan> g_plans;

#define WD_LT(call)                                                                       \
    do {                                                                                  \
        hipblasStatus_t s__ = (call);                                                     \
        if (s__ != HIPBLAS_STATUS_SUCCESS) {                                              \
            wt::set_error("hipBLASLt: %s -> status %d", #call, (int)s__);                 \
            return WT_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

// Row-major A (m, k), W (n, k), C / D (m, n)  ==  column-major D^T (n, m) = W^T' ... : op(W) = T on the (k, n) column-major
// view of W, op(A) = N on the (k, m) view of A; the bias runs along the rows of D^T = the output channels.
int make_plan(Plan& p, int m, int n, int k, int relu, bool bias) {
    WD_LT(g_api.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    WD_LT(g_api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)));
    WD_LT(g_api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)));
    hipblasLtEpilogue_t ep = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                                  : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
    WD_LT(g_api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
    WD_LT(g_api.LayoutCreate(&p.lw, HIP_R_32F, (uint64_t)k, (uint64_t)n, (int64_t)k));
    WD_LT(g_api.LayoutCreate(&p.la, HIP_R_32F, (uint64_t)k, (uint64_t)m, (int64_t)k));
    WD_LT(g_api.LayoutCreate(&p.lc, HIP_R_32F, (uint64_t)n, (uint64_t)m, (int64_t)n));
    return WT_OK;
}

int run(const Plan& p, const hipblasLtMatmulAlgo_t& algo, const float* a, const float* w, const float* bias, const float* c,
        float* d, float beta, void* ws, size_t ws_bytes, hipStream_t stream) {
    const float alpha = 1.0f;
    if (bias) WD_LT(g_api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    WD_LT(g_api.Matmul(g_api.handle, p.desc, &alpha, w, p.lw, a, p.la, &beta, c, p.lc, d, p.lc, &algo, ws, ws_bytes, stream));
    return WT_OK;
}

}  // namespace

extern "C" {

/* out (m, n) = relu?(a (m, k) . w (n, k)^T + residual (m, n) + bias (n)); residual may alias out (in-place block output).
 * The first call per shape outside a stream capture times the heuristic's candidates and caches the fastest. */
int wd_gemm_lt_f32(const float* a, const float* w, const float* bias, const float* residual, float* out, int m, int n, int k,
                   int relu, void* workspace, size_t workspace_bytes, void* stream_) {
    WT_TRY(wt::ensure_device());
    hipStream_t stream = (hipStream_t)stream_;
    if (m < 1 || n < 1 || k < 1 || !a || !w || !out) { wt::set_error("wd_gemm_lt_f32: bad argument"); return WT_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_mu);
    WT_TRY(load_api());
    const Key key(m, n, k, relu ? 1 : 0, bias ? 1 : 0, residual ? 1 : 0);
    Plan& p = g_plans[key];
    if (!p.desc) WT_TRY(make_plan(p, m, n, k, relu, bias != nullptr));
    const float beta = residual ? 1.0f : 0.0f;
    const float* c = residual ? residual : out;
    if (!p.tuned) {
        hipblasLtMatmulPreference_t pref = nullptr;
        WD_LT(g_api.PrefCreate(&pref));
        uint64_t max_ws = workspace ? (uint64_t)workspace_bytes : 0;
        WD_LT(g_api.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws)));
        if (bias) WD_LT(g_api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        std::vector<hipblasLtMatmulHeuristicResult_t> res(24);
        int found = 0;
        const hipblasStatus_t hs = g_api.Heuristic(g_api.handle, p.desc, p.lw, p.la, p.lc, p.lc, pref, (int)res.size(), res.data(), &found);
        (void)g_api.PrefDestroy(pref);
        if (hs != HIPBLAS_STATUS_SUCCESS || found < 1) {
            wt::set_error("wd_gemm_lt_f32: no hipBLASLt algorithm for %dx%dx%d (status %d)", m, n, k, (int)hs);
            return WT_ERR_HIP;
        }
        p.candidates = found;
        p.algo = res[0].algo;
        p.ws = res[0].workspaceSize;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap == hipStreamCaptureStatusNone && found > 1) {
            // time every candidate on a scratch output (the residual must not be accumulated into more than once)
            wt::DevBuf scratch;
            WT_TRY(scratch.alloc((size_t)m * n * sizeof(float)));
            hipEvent_t e0, e1;
            WT_HIP(hipEventCreate(&e0));
            WT_HIP(hipEventCreate(&e1));
            float best = 1e30f;
            for (int i = 0; i < found; ++i) {
                if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > max_ws) continue;
                if (run(p, res[i].algo, a, w, bias, c, scratch.as<float>(), beta, workspace, workspace_bytes, stream) != WT_OK) continue;
                (void)hipEventRecord(e0, stream);
                bool ok = true;
                for (int r = 0; r < 3 && ok; ++r)
                    ok = run(p, res[i].algo, a, w, bias, c, scratch.as<float>(), beta, workspace, workspace_bytes, stream) == WT_OK;
                (void)hipEventRecord(e1, stream);
                if (hipEventSynchronize(e1) != hipSuccess || !ok) continue;
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; p.algo = res[i].algo; p.ws = res[i].workspaceSize; }
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            WT_HIP(hipStreamSynchronize(stream));
            p.best_us = best * 1e3f / 3.f;
        }
        // the choice is final once it was made outside a capture (also when the heuristic offered a single candidate);
        // a first call inside a capture keeps the heuristic's pick for this call and decides at the next eager call
        if (cap == hipStreamCaptureStatusNone) p.tuned = true;
    }
    if (p.ws > (workspace ? workspace_bytes : 0)) {
        wt::set_error("wd_gemm_lt_f32: the cached algorithm for %dx%dx%d needs %zu workspace bytes, %zu passed", m, n, k,
                      (size_t)p.ws, workspace ? workspace_bytes : (size_t)0);
        return WT_ERR_INVALID;
    }
    return run(p, p.algo, a, w, bias, c, out, beta, workspace, workspace_bytes, stream);
}

/* The cached choice for a shape (after the first call): microseconds of the fastest candidate and how many were timed. */
int wd_gemm_lt_plan_info(int m, int n, int k, int relu, int has_bias, int has_residual, float* best_us, int* candidates) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(Key(m, n, k, relu ? 1 : 0, has_bias ? 1 : 0, has_residual ? 1 : 0));
    if (it == g_plans.end()) { wt::set_error("wd_gemm_lt_plan_info: shape not seen"); return WT_ERR_INVALID; }
    if (best_us) *best_us = it->second.best_us;
    if (candidates) *candidates = it->second.candidates;
    return WT_OK;
}

}  // extern "C"
