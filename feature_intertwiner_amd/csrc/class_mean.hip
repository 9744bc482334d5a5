// class_mean.hip -- per-class mean of RoI feature vectors (intertwiner statistics).
//
// Specification: Dev._assign_feat2cls, lib/sub_module.py:664-684 -- a Python loop
// over the classes present (unique1d + nonzero + index + mean per class, one host
// sync each).  Oracle: orc_class_mean.
//
// Forward, two launches, no atomics, deterministic summation order:
//   1. class_mean_partial_kernel: grid (F/64 column blocks) x (row chunks).  A workgroup streams its
//      row chunk (coalesced 256-byte segments, 8 rows in flight), adds each row into the
//      [num_classes][64] accumulator of its class in LDS and writes the partial sums to the
//      workspace [chunk][num_classes][F].  (The first version had only the F/64 = 16 workgroups and
//      took 200 us for an 8 MB matrix -- 1 % of the HBM roofline.)
//   2. class_mean_finish_kernel: sums the chunks in order, counts the rows per class and divides.
// Backward: grad_features[n, f] = grad_feat[f, gt[n]] / cnt[gt[n]].
#include "fi_common.h"

namespace {

constexpr int kCols = 64;
constexpr int kMaxClasses = 248;      // 3 levels x 81 classes in one launch (Dev._forward_static); 62 KB of LDS

constexpr int kRowsPerChunk = 64;

__global__ __launch_bounds__(kCols) void class_mean_partial_kernel(const float *__restrict__ features,
                                                                   const int *__restrict__ gt, int N,
                                                                   int F, int K, float *__restrict__ part)
{
    __shared__ float s_acc[kMaxClasses][kCols];
    const int tid = threadIdx.x;
    const int f = blockIdx.x * kCols + tid;
    const int n_begin = blockIdx.y * kRowsPerChunk;
    const int n_end = min(N, n_begin + kRowsPerChunk);
    for (int c = 0; c < K; ++c) s_acc[c][tid] = 0.0f;      // column `tid` is private to this thread

    constexpr int U = 8;
    for (int n0 = n_begin; n0 < n_end; n0 += U) {
        float v[U];
        int cls[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u;
            cls[u] = (n < n_end) ? gt[n] : 0;
            v[u] = 0.0f;
            if (n < n_end && cls[u] > 0 && cls[u] < K && f < F) v[u] = features[(size_t)n * F + f];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (cls[u] > 0 && cls[u] < K) s_acc[cls[u]][tid] += v[u];   // uniform across the workgroup
    }
    if (f < F) {
        float *__restrict__ dst = part + (size_t)blockIdx.y * K * F + f;
        for (int c = 0; c < K; ++c) dst[(size_t)c * F] = s_acc[c][tid];
    }
}

__global__ __launch_bounds__(256) void class_mean_finish_kernel(const float *__restrict__ part,
                                                                const int *__restrict__ gt, int N, int F,
                                                                int K, int chunks, float *__restrict__ feat,
                                                                float *__restrict__ cnt)
{
    __shared__ float s_cnt[kMaxClasses];
    const int tid = threadIdx.x;
    for (int c = tid; c < K; c += 256) s_cnt[c] = 0.0f;
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
        const int c = gt[n];
        if (c > 0 && c < K) atomicAdd(&s_cnt[c], 1.0f);     // integer-valued: exact in any order
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int c = tid; c < K; c += 256) cnt[c] = s_cnt[c];
    // 256 consecutive (c, f) entries of the [K][F] partial layout per workgroup (coalesced over f)
    const size_t i = (size_t)blockIdx.x * 256 + tid;
    if (i >= (size_t)K * F) return;
    const int c = (int)(i / F);
    const int f = (int)(i - (size_t)c * F);
    float sum = 0.0f;
    for (int k = 0; k < chunks; ++k) sum += part[(size_t)k * K * F + i];
    const float n = s_cnt[c];
    feat[(size_t)f * K + c] = (n > 0.0f) ? (sum / n) : 0.0f;
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

size_t fi_class_mean_workspace_bytes(int N, int F, int num_classes)
{
    if (N <= 0 || F <= 0 || num_classes <= 0) return 0;
    return sizeof(float) * (size_t)fi::ceil_div(N, kRowsPerChunk) * num_classes * F;
}

int fi_class_mean_forward(const float *features, const int32_t *gt, int N, int F, int num_classes,
                          float *feat, float *cnt, float *workspace, fi_stream_t stream)
{
    FI_REQUIRE(N >= 0 && F >= 1, "N >= 0, F >= 1");
    if (num_classes < 1 || num_classes > kMaxClasses) {
        fi::set_error("fi_class_mean supports 1..%d classes (got %d)", kMaxClasses, num_classes);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE(feat && cnt && (N == 0 || (features && gt && workspace)), "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = fi::ceil_div(N, kRowsPerChunk);
    fi::ProfScope prof(FI_K_CLASS_MEAN, st);
    if (chunks > 0) {
        hipLaunchKernelGGL(class_mean_partial_kernel, dim3(fi::ceil_div(F, kCols), chunks), dim3(kCols), 0, st,
                           features, gt, N, F, num_classes, workspace);
        FI_HIP_CHECK(hipGetLastError());
    }
    const size_t total = (size_t)num_classes * F;
    hipLaunchKernelGGL(class_mean_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       workspace, gt, N, F, num_classes, chunks, feat, cnt);
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
