// libwaymotrack: error state, device probing, ID counter (include/waymotrack.h).
#include "common.h"

namespace wt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? WT_ERR_NO_DEVICE : WT_ERR_HIP;
}

int ensure_device() {
    static thread_local int ok_dev = -1;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s); libwaymotrack has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return WT_ERR_NO_DEVICE;
    }
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev == ok_dev) return WT_OK;
    hipDeviceProp_t prop;
    WT_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return WT_ERR_NO_DEVICE;
    }
    ok_dev = dev;
    return WT_OK;
}

int device_index() {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}

int device_cus() {
    static int cus[MAX_DEVICES] = {};
    const int dev = device_index();
    if (dev >= 0 && dev < MAX_DEVICES && cus[dev] > 0) return cus[dev];
    hipDeviceProp_t prop;
    if (dev < 0 || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (dev < MAX_DEVICES) cus[dev] = prop.multiProcessorCount;
    return prop.multiProcessorCount;
}

}  // namespace wt

extern "C" {

const char* wt_last_error(void) { return wt::g_err; }

int wt_abi_version(void) { return 4; }   // 4: static-shape detector tail (wd_rpn_topk_decode / wd_sort_candidates / wd_box_candidates / wd_gather_kept / wd_detections_to_wire), wd_gemm_lt, wt_detjson_*, wt_waymo_objects_write;   // 3: streaming tracker (wt_track_state_* / wt_track_chunk_dev / wt_track_global_ids_dev);   // 2: wd_deform_im2col/col2im take `groups` (group-major columns); new wd_preprocess / wd_decode_boxes / wd_tap_shift_add / wt_mct entry points

int wt_device_info(int* n_devices, char* arch, int arch_cap, int* n_cu) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (n_devices) *n_devices = n;
    if (arch && arch_cap > 0) arch[0] = 0;
    if (n_cu) *n_cu = 0;
    if (n <= 0) {
        wt::set_error("no HIP device visible");
        return WT_ERR_NO_DEVICE;
    }
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    WT_HIP(hipGetDeviceProperties(&prop, dev));
    if (arch && arch_cap > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_cap - 1);
        arch[arch_cap - 1] = 0;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    return WT_OK;
}

struct wt_idctr {
    int64_t next;
};

wt_idctr* wt_idctr_create(int64_t start) {
    wt_idctr* c = new wt_idctr;
    c->next = start;
    return c;
}
int64_t wt_idctr_get(const wt_idctr* c) { return c ? c->next : 0; }
void wt_idctr_set(wt_idctr* c, int64_t value) {
    if (c) c->next = value;
}
void wt_idctr_destroy(wt_idctr* c) { delete c; }

}  // extern "C"
