// sgd.hip -- gradient-norm clipping + SGD(momentum, weight decay) for ALL parameters of the model in three
// launches (gfx950).  Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.SGD.step of the reference's
// training step (lib/workflow.py:226-230, tools/utils.py:474-501): those run ~8 multi-tensor passes over the
// 252 MB of parameters / gradients / momentum buffers (3.8 GB of traffic, 1.5 ms per step for ResNet-101-FPN);
// here one pass reads the gradients for the norm and one pass does clip + weight decay + momentum + update
// (1.5 GB).  Arithmetic per element, in torch's order:
//     g  = grad * clip_coef                      clip_coef = min(1, max_norm / (||grad||_2 + 1e-6))
//     g  = g + weight_decay * p                  (groups with weight decay)
//     b  = momentum * b + g                      (b starts at 0: the first step leaves b = g)
//     p  = p - lr * b
// The norm is reduced in a fixed order (per-chunk partial sums, then one workgroup over the partials), so
// data-parallel replicas that hold identical gradients compute identical clip factors.
#include "fi_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 8192;          // floats per workgroup: 8 x 16 bytes per lane

__device__ __forceinline__ int find_desc(const FiSgdDesc *__restrict__ descs, int n, long chunk)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {                              // last descriptor with chunk_base <= chunk
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].chunk_base <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float block_sum(float v, float *s_w)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(kThreads) void sgd_sumsq_kernel(const FiSgdDesc *__restrict__ descs, int n,
                                                             float *__restrict__ partial)
{
    __shared__ float s_w[4];
    const long chunk = blockIdx.x;
    const FiSgdDesc d = descs[find_desc(descs, n, chunk)];
    const long begin = (chunk - d.chunk_base) * kChunk;
    const long end = min(d.numel, begin + kChunk);
    const float *__restrict__ g = static_cast<const float *>(d.grad);
    float acc = 0.0f;
    const bool vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
    if (vec) {
        for (long i = begin + threadIdx.x * 4; i < end; i += kThreads * 4) {
            if (i + 3 < end) {
                const float4 v = *reinterpret_cast<const float4 *>(g + i);
                acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            } else {
                for (long j = i; j < end; ++j) acc += g[j] * g[j];
            }
        }
    } else {
        for (long i = begin + threadIdx.x; i < end; i += kThreads) acc += g[i] * g[i];
    }
    const float s = block_sum(acc, s_w);
    if (threadIdx.x == 0) partial[chunk] = s;
}

// out[0] = ||grad||_2, out[1] = clip factor; guard: out[2] = 1 when the norm is not finite (the update kernel then
// leaves parameters, momentum buffers and gradients untouched: the step is skipped), out[3] += 1 per skipped step
__global__ __launch_bounds__(kThreads) void sgd_norm_kernel(const float *__restrict__ partial, long chunks,
                                                            float max_norm, float *__restrict__ out, int guard)
{
    __shared__ double s_d[kThreads];
    double acc = 0.0;
    for (long i = threadIdx.x; i < chunks; i += kThreads) acc += (double)partial[i];
    s_d[threadIdx.x] = acc;
    __syncthreads();
    for (int off = kThreads / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) s_d[threadIdx.x] += s_d[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s_d[0]);
        out[0] = norm;
        float coef = 1.0f;
        if (max_norm > 0.0f) {
            coef = max_norm / (norm + 1e-6f);
            if (coef > 1.0f) coef = 1.0f;              // a NaN norm stays NaN, like torch.clamp(max=1)
        }
        out[1] = coef;
        if (guard) {
            const bool bad = !(norm <= 3.0e38f);        // inf or NaN (one fp16 overflow under the loss scale)
            out[2] = bad ? 1.0f : 0.0f;
            if (bad) out[3] += 1.0f;
        }
    }
}

__device__ __forceinline__ void update_one(float &p, float &g, float &b, float coef, float wd, float mom, float lr,
                                           bool has_buf)
{
    g = g * coef;
    float u = g;
    if (wd != 0.0f) u = u + wd * p;
    if (has_buf) {
        b = mom * b + u;
        u = b;
    }
    p = p - lr * u;
}

__global__ __launch_bounds__(kThreads) void sgd_update_kernel(const FiSgdDesc *__restrict__ descs, int n,
                                                              const float *__restrict__ norm_coef, int guard)
{
    if (guard && norm_coef[2] != 0.0f) return;          // non-finite gradient norm: skip the step (uniform branch)
    const long chunk = blockIdx.x;
    const FiSgdDesc d = descs[find_desc(descs, n, chunk)];
    const long begin = (chunk - d.chunk_base) * kChunk;
    const long end = min(d.numel, begin + kChunk);
    float *__restrict__ p = static_cast<float *>(d.param);
    float *__restrict__ g = static_cast<float *>(d.grad);
    float *__restrict__ b = static_cast<float *>(d.buf);
    const float coef = norm_coef[1];
    const bool wb = coef != 1.0f;                       // clip_grad_norm_ leaves the scaled gradient behind
    const bool has_buf = b != nullptr;
    const float wd = d.weight_decay, mom = d.momentum, lr = d.lr;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                       reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    if (vec) {
        for (long i = begin + threadIdx.x * 4; i < end; i += kThreads * 4) {
            if (i + 3 < end) {
                float4 pv = *reinterpret_cast<float4 *>(p + i);
                float4 gv = *reinterpret_cast<float4 *>(g + i);
                float4 bv = has_buf ? *reinterpret_cast<float4 *>(b + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                update_one(pv.x, gv.x, bv.x, coef, wd, mom, lr, has_buf);
                update_one(pv.y, gv.y, bv.y, coef, wd, mom, lr, has_buf);
                update_one(pv.z, gv.z, bv.z, coef, wd, mom, lr, has_buf);
                update_one(pv.w, gv.w, bv.w, coef, wd, mom, lr, has_buf);
                *reinterpret_cast<float4 *>(p + i) = pv;
                if (has_buf) *reinterpret_cast<float4 *>(b + i) = bv;
                if (wb) *reinterpret_cast<float4 *>(g + i) = gv;
            } else {
                for (long j = i; j < end; ++j) {
                    float pv = p[j], gv = g[j], bv = has_buf ? b[j] : 0.0f;
                    update_one(pv, gv, bv, coef, wd, mom, lr, has_buf);
                    p[j] = pv;
                    if (has_buf) b[j] = bv;
                    if (wb) g[j] = gv;
                }
            }
        }
    } else {
        for (long j = begin + threadIdx.x; j < end; j += kThreads) {
            float pv = p[j], gv = g[j], bv = has_buf ? b[j] : 0.0f;
            update_one(pv, gv, bv, coef, wd, mom, lr, has_buf);
            p[j] = pv;
            if (has_buf) b[j] = bv;
            if (wb) g[j] = gv;
        }
    }
}

}  // namespace

extern "C" {

long fi_sgd_chunks(long numel) { return numel <= 0 ? 0 : (numel + kChunk - 1) / kChunk; }

static int sgd_clip_step(const FiSgdDesc *descs_dev, int n, long total_chunks, float max_norm, float *partial_ws,
                         float *norm_coef, fi_stream_t stream, int guard)
{
    FI_REQUIRE(n >= 0 && total_chunks >= 0, "negative count");
    FI_REQUIRE(total_chunks < 2147483647L, "too many chunks");
    if (n == 0 || total_chunks == 0) return FI_OK;
    FI_REQUIRE(descs_dev && partial_ws && norm_coef, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sgd_sumsq_kernel, dim3((unsigned)total_chunks), dim3(kThreads), 0, st, descs_dev, n, partial_ws);
    hipLaunchKernelGGL(sgd_norm_kernel, dim3(1), dim3(kThreads), 0, st, partial_ws, total_chunks, max_norm, norm_coef,
                       guard);
    hipLaunchKernelGGL(sgd_update_kernel, dim3((unsigned)total_chunks), dim3(kThreads), 0, st, descs_dev, n, norm_coef,
                       guard);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_sgd_clip_step(const FiSgdDesc *descs_dev, int n, long total_chunks, float max_norm, float *partial_ws,
                     float *norm_coef, fi_stream_t stream)
{
    return sgd_clip_step(descs_dev, n, total_chunks, max_norm, partial_ws, norm_coef, stream, 0);
}

int fi_sgd_clip_step_guarded(const FiSgdDesc *descs_dev, int n, long total_chunks, float max_norm, float *partial_ws,
                             float *norm_coef4, fi_stream_t stream)
{
    return sgd_clip_step(descs_dev, n, total_chunks, max_norm, partial_ws, norm_coef4, stream, 1);
}

}  // extern "C"
