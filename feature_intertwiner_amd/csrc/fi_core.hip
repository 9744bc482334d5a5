// fi_core.hip -- version, error reporting and the in-library kernel timer.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "fi_common.h"

namespace fi {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiling -------------------------------------------------------------
struct Pending {
    int id;
    hipEvent_t start, stop;
};
static std::mutex g_mu;
static bool g_enabled = false;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static int g_launches[FI_K_COUNT];
static double g_ms[FI_K_COUNT];

static hipEvent_t take_event()
{
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int kernel_id, hipStream_t stream)
    : id_(kernel_id), stream_(stream), start_(nullptr), active_(false)
{
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lk(g_mu);
    start_ = take_event();
    if (start_ && hipEventRecord(start_, stream_) == hipSuccess) active_ = true;
}

ProfScope::~ProfScope()
{
    if (!active_) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t stop = take_event();
    if (stop && hipEventRecord(stop, stream_) == hipSuccess) {
        g_pending.push_back({id_, start_, stop});
    } else {
        if (stop) g_pool.push_back(stop);
        g_pool.push_back(start_);
    }
}

static void drain_locked()
{
    for (auto &p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess &&
            hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            g_launches[p.id] += 1;
            g_ms[p.id] += ms;
        }
        g_pool.push_back(p.start);
        g_pool.push_back(p.stop);
    }
    g_pending.clear();
}

}  // namespace fi

// Calibration kernel for the memory-side PMC counters (FETCH_SIZE / WRITE_SIZE): a plain streaming copy,
// 16 bytes per lane, whose HBM traffic is known exactly (n floats read, n floats written).
__global__ __launch_bounds__(256) void fi_calib_copy_kernel(const float4 *__restrict__ src,
                                                            float4 *__restrict__ dst, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" {

int fi_calib_copy(const float *src, float *dst, size_t n_floats, fi_stream_t stream)
{
    FI_REQUIRE(src && dst && n_floats % 4 == 0, "calibration copy needs 16-byte multiples");
    FI_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "calibration copy needs 16-byte aligned buffers");
    hipLaunchKernelGGL(fi_calib_copy_kernel, dim3(8192), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)src, (float4 *)dst, n_floats / 4);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

const char *fi_version(void) { return "fi_hip 0.1.0 gfx950"; }

const char *fi_last_error(void) { return fi::g_err; }

void fi_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(fi::g_mu);
    fi::g_enabled = on != 0;
}

void fi_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(fi::g_mu);
    fi::drain_locked();
    memset(fi::g_launches, 0, sizeof(fi::g_launches));
    for (int i = 0; i < FI_K_COUNT; ++i) fi::g_ms[i] = 0.0;
}

int fi_prof_get(int kernel_id, int *launches, float *total_ms)
{
    if (kernel_id < 0 || kernel_id >= FI_K_COUNT) {
        fi::set_error("fi_prof_get: bad kernel id %d", kernel_id);
        return FI_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(fi::g_mu);
    fi::drain_locked();
    if (launches) *launches = fi::g_launches[kernel_id];
    if (total_ms) *total_ms = (float)fi::g_ms[kernel_id];
    return FI_OK;
}

const char *fi_prof_kernel_name(int kernel_id)
{
    static const char *names[FI_K_COUNT] = {
        "crop_fwd_flat_kernel<7, 7, 8>", "crop_fwd_kernel<14, 14>", "crop_fwd_kernel<28, 28>",   /* slot 0: depth % 64 != 0 runs crop_fwd_kernel<7, 7> */
        "crop_fwd_kernel<0, 0>",   "crop_bwd_kernel<7, 7>",   "crop_bwd_kernel<14, 14>",
        "crop_bwd_kernel<28, 28>", "crop_bwd_kernel<0, 0>",   "roi_pool_fwd_kernel",
        "roi_pool_bwd_kernel",     "nms_mask_kernel",         "nms_scan_kernel",
        "sinkhorn_kernel",         "class_mean_fwd_kernel",
        "conv_fwd_kernel<64, 1, 1>",    "conv_fwd_kernel<64, 3, 3>",    "conv_fwd_kernel<64, 7, 7>",
        "conv_fwd_kernel<64, 0, 0>",    "conv_fwd_kernel<128, 1, 1>",   "conv_fwd_kernel<128, 3, 3>",
        "conv_fwd_kernel<128, 7, 7>",   "conv_fwd_kernel<128, 0, 0>",   "conv_wgrad_kernel<64, 1, 1>",
        "conv_wgrad_kernel<64, 3, 3>",  "conv_wgrad_kernel<64, 7, 7>",  "conv_wgrad_kernel<64, 0, 0>",
        "conv_wgrad_kernel<128, 1, 1>", "conv_wgrad_kernel<128, 3, 3>", "conv_wgrad_kernel<128, 7, 7>",
        "conv_wgrad_kernel<128, 0, 0>", "bn_act_bwd_kernel",
        "crop_fwd_cl_kernel<7, 7>", "crop_fwd_cl_kernel<14, 14>", "crop_fwd_cl_kernel<0, 0>",
        "crop_bwd_cl_kernel<7, 7>", "crop_bwd_cl_kernel<14, 14>", "crop_bwd_cl_kernel<0, 0>",
        "conv_bf16_fwd_kernel", "conv_bf16_wgrad_kernel",
        "conv3x3_patch_kernel<false>", "conv3x3_patch_kernel<true>", "conv1x1_reg_kernel",
        "proposal_select_kernel", "proposal_gather_kernel", "stride2_interleave_kernel", "gemm_slab_reduce_kernel"};
    if (kernel_id < 0 || kernel_id >= FI_K_COUNT) return "?";
    return names[kernel_id];
}

}  // extern "C"
