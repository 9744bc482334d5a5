// class_mean.hip -- per-class mean of RoI feature vectors (intertwiner statistics).
//
// Specification: Dev._assign_feat2cls, lib/sub_module.py:664-684 -- a Python loop
// over the classes present (unique1d + nonzero + index + mean per class, one host
// sync each).  Oracle: orc_class_mean.
//
// Forward: one pass over the [N, F] feature matrix.  A workgroup owns 64 feature
// columns (one per lane... per thread) and keeps a [num_classes][64] accumulator in
// LDS; rows are streamed in order (coalesced 256-byte segments, 8 rows in flight)
// and added to the row of their class, so no atomics and a deterministic sum order.
// Backward: grad_features[n, f] = grad_feat[f, gt[n]] / cnt[gt[n]].
#include "fi_common.h"

namespace {

constexpr int kCols = 64;
constexpr int kMaxClasses = 128;

__global__ __launch_bounds__(kCols) void class_mean_fwd_kernel(const float *__restrict__ features,
                                                               const int *__restrict__ gt, int N,
                                                               int F, int K, float *__restrict__ feat,
                                                               float *__restrict__ cnt)
{
    __shared__ float s_acc[kMaxClasses][kCols];
    __shared__ float s_cnt[kMaxClasses];
    const int tid = threadIdx.x;
    const int f = blockIdx.x * kCols + tid;
    for (int c = 0; c < K; ++c) s_acc[c][tid] = 0.0f;
    for (int c = tid; c < K; c += kCols) s_cnt[c] = 0.0f;
    __syncthreads();

    constexpr int U = 8;
    for (int n0 = 0; n0 < N; n0 += U) {
        float v[U];
        int cls[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u;
            cls[u] = (n < N) ? gt[n] : 0;
            v[u] = 0.0f;
            if (n < N && cls[u] > 0 && cls[u] < K && f < F) v[u] = features[(size_t)n * F + f];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (cls[u] > 0 && cls[u] < K) {  // uniform across the workgroup
                s_acc[cls[u]][tid] += v[u];
                if (tid == 0) s_cnt[cls[u]] += 1.0f;
            }
        }
    }
    __syncthreads();
    if (f < F) {
        for (int c = 0; c < K; ++c) {
            const float n = s_cnt[c];
            feat[(size_t)f * K + c] = (n > 0.0f) ? (s_acc[c][tid] / n) : 0.0f;
        }
    }
    if (blockIdx.x == 0)
        for (int c = tid; c < K; c += kCols) cnt[c] = s_cnt[c];
}

__global__ __launch_bounds__(256) void class_mean_bwd_kernel(const float *__restrict__ grad_feat,
                                                             const int *__restrict__ gt,
                                                             const float *__restrict__ cnt, int N,
                                                             int F, int K,
                                                             float *__restrict__ grad_features)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * F) return;
    const int n = (int)(i / F);
    const int f = (int)(i - (size_t)n * F);
    const int c = gt[n];
    float g = 0.0f;
    if (c > 0 && c < K) {
        const float m = cnt[c];
        if (m > 0.0f) g = grad_feat[(size_t)f * K + c] / m;
    }
    grad_features[i] = g;
}

}  // namespace

extern "C" {

int fi_class_mean_forward(const float *features, const int32_t *gt, int N, int F, int num_classes,
                          float *feat, float *cnt, fi_stream_t stream)
{
    FI_REQUIRE(N >= 0 && F >= 1, "N >= 0, F >= 1");
    if (num_classes < 1 || num_classes > kMaxClasses) {
        fi::set_error("fi_class_mean supports 1..%d classes (got %d)", kMaxClasses, num_classes);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE(feat && cnt && (N == 0 || (features && gt)), "null pointer");
    hipStream_t st = (hipStream_t)stream;
    fi::ProfScope prof(FI_K_CLASS_MEAN, st);
    hipLaunchKernelGGL(class_mean_fwd_kernel, dim3(fi::ceil_div(F, kCols)), dim3(kCols), 0, st,
                       features, gt, N, F, num_classes, feat, cnt);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_class_mean_backward(const float *grad_feat, const int32_t *gt, const float *cnt, int N, int F,
                           int num_classes, float *grad_features, fi_stream_t stream)
{
    FI_REQUIRE(N >= 0 && F >= 1 && num_classes >= 1, "bad sizes");
    if (N == 0) return FI_OK;
    FI_REQUIRE(grad_feat && gt && cnt && grad_features, "null pointer");
    const size_t total = (size_t)N * F;
    hipLaunchKernelGGL(class_mean_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, grad_feat, gt, cnt, N, F, num_classes, grad_features);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
