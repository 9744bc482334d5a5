// roi_pool.hip -- Caffe-style max RoIPool forward (+argmax) and backward, gfx950.
//
// Specification: ROIPoolForward / ROIPoolBackward,
// lib/roi_pooling/src/roi_pooling_kernel.cu:24-93 / :128-203 (the CUDA kernel is the
// only usable definition of the operator -- SURVEY Q4).  Oracle: orc_roi_pool_*.
//
// Forward: one workgroup per (RoI, channel chunk); the RoI's bin boundaries
// (hstart/hend per pooled row, wstart/wend per pooled column, after rounding,
// offsetting and clipping) are computed once into LDS instead of per output.
// Backward: the reference gathers -- every INPUT element loops over ALL RoIs
// (O(B*C*H*W*N), 67 M threads x N at P2).  Here each pooled cell scatters its
// gradient to its argmax with a hardware fp32 atomic, after re-checking the
// reference's feasibility conditions (same image, pixel inside the rounded RoI,
// pooled cell inside the pixel's candidate window) so that the set of summed terms
// is identical; only the summation order differs.
#include <float.h>

#include "fi_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPool = 64;

struct Roi {
    int img;
    int start_w, start_h, end_w, end_h;
    float bin_h, bin_w;
};

// roi_pooling_kernel.cu:44-54.  roundf = round half away from zero, as CUDA round().
__device__ __forceinline__ Roi decode_roi(const float *__restrict__ r, float scale, int ph, int pw)
{
    Roi o;
    o.img = (int)r[0];
    o.start_w = (int)roundf(r[1] * scale);
    o.start_h = (int)roundf(r[2] * scale);
    o.end_w = (int)roundf(r[3] * scale);
    o.end_h = (int)roundf(r[4] * scale);
    const int roi_w = max(o.end_w - o.start_w + 1, 1);
    const int roi_h = max(o.end_h - o.start_h + 1, 1);
    o.bin_h = (float)roi_h / (float)ph;
    o.bin_w = (float)roi_w / (float)pw;
    return o;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(kThreads) void roi_pool_fwd_kernel(
    const float *__restrict__ features, const float *__restrict__ rois, int num_rois, int batch,
    int channels, int height, int width, int ph, int pw, float scale, int chan_per_block,
    int chunks, float *__restrict__ output, int *__restrict__ argmax)
{
    __shared__ int s_h0[kMaxPool], s_h1[kMaxPool], s_w0[kMaxPool], s_w1[kMaxPool];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / chunks;
    const int chunk = blockIdx.x - n * chunks;
    const int c_begin = chunk * chan_per_block;
    const int c_count = min(chan_per_block, channels - c_begin);
    const int bins = ph * pw;
    const int total = c_count * bins;

    const Roi r = decode_roi(rois + 5 * (size_t)n, scale, ph, pw);
    if (tid < ph) {
        const int hs = (int)floorf((float)tid * r.bin_h);
        const int he = (int)ceilf((float)(tid + 1) * r.bin_h);
        s_h0[tid] = clampi(hs + r.start_h, 0, height);
        s_h1[tid] = clampi(he + r.start_h, 0, height);
    }
    if (tid >= 64 && tid < 64 + pw) {
        const int q = tid - 64;
        const int ws = (int)floorf((float)q * r.bin_w);
        const int we = (int)ceilf((float)(q + 1) * r.bin_w);
        s_w0[q] = clampi(ws + r.start_w, 0, width);
        s_w1[q] = clampi(we + r.start_w, 0, width);
    }
    __syncthreads();

    const size_t o_base = ((size_t)n * channels + c_begin) * bins;
    const bool img_ok = r.img >= 0 && r.img < batch;
    for (int idx = tid; idx < total; idx += kThreads) {
        const int c = idx / bins;
        const int bin = idx - c * bins;
        const int p = bin / pw;
        const int q = bin - p * pw;
        const int hs = s_h0[p], he = s_h1[p], ws = s_w0[q], we = s_w1[q];
        const bool empty = (he <= hs) || (we <= ws) || !img_ok;
        float best = empty ? 0.0f : -FLT_MAX;
        int best_i = -1;
        if (!empty) {
            const int plane_off = ((r.img * channels) + c_begin + c) * height * width;
            const float *__restrict__ src = features + plane_off;
            for (int h = hs; h < he; ++h) {
                const float *__restrict__ row = src + h * width;
                for (int w = ws; w < we; ++w) {
                    const float v = row[w];
                    if (v > best) {  // strict: the first maximum in (h, w) order wins
                        best = v;
                        best_i = plane_off + h * width + w;
                    }
                }
            }
        }
        output[o_base + idx] = best;
        if (argmax) argmax[o_base + idx] = best_i;
    }
}

__global__ __launch_bounds__(kThreads) void roi_pool_bwd_kernel(
    const float *__restrict__ top_grad, const float *__restrict__ rois,
    const int *__restrict__ argmax, int num_rois, int batch, int channels, int height, int width,
    int ph, int pw, float scale, int per_roi, int blocks_per_roi, float *__restrict__ bottom_grad)
{
    // blockIdx.x = roi * blocks_per_roi + j; j covers channels*ph*pw of that roi
    const int n = blockIdx.x / blocks_per_roi;
    const int idx = (blockIdx.x - n * blocks_per_roi) * kThreads + threadIdx.x;
    if (idx >= per_roi) return;
    const size_t o = (size_t)n * per_roi + idx;
    const int a = argmax[o];
    if (a < 0) return;
    const Roi r = decode_roi(rois + 5 * (size_t)n, scale, ph, pw);
    const int q = idx % pw;
    const int p = (idx / pw) % ph;
    const int c = idx / (pw * ph);
    // decode the argmax position (flat NCHW index)
    const int w = a % width;
    const int h = (a / width) % height;
    const int ca = (a / (width * height)) % channels;
    const int img = a / (width * height * channels);
    if (img != r.img || ca != c) return;  // kernel.cu:147 (+ the index identity)
    if (!(w >= r.start_w && w <= r.end_w && h >= r.start_h && h <= r.end_h)) return;  // :155-160
    int p0 = (int)floorf((float)(h - r.start_h) / r.bin_h);       // :173-176
    int p1 = (int)ceilf((float)(h - r.start_h + 1) / r.bin_h);
    int q0 = (int)floorf((float)(w - r.start_w) / r.bin_w);
    int q1 = (int)ceilf((float)(w - r.start_w + 1) / r.bin_w);
    p0 = clampi(p0, 0, ph);
    p1 = clampi(p1, 0, ph);
    q0 = clampi(q0, 0, pw);
    q1 = clampi(q1, 0, pw);
    if (p < p0 || p >= p1 || q < q0 || q >= q1) return;
    atomicAdd(bottom_grad + a, top_grad[o]);
}

}  // namespace

extern "C" {

int fi_roi_pool_forward(const float *features, const float *rois, int num_rois, int batch,
                        int channels, int height, int width, int pooled_h, int pooled_w,
                        float spatial_scale, float *output, int32_t *argmax, fi_stream_t stream)
{
    FI_REQUIRE(num_rois >= 0 && batch > 0 && channels > 0 && height > 0 && width > 0, "bad sizes");
    FI_REQUIRE(pooled_h >= 1 && pooled_w >= 1, "pooled size must be >= 1");
    if (pooled_h > kMaxPool || pooled_w > kMaxPool) {
        fi::set_error("pooled size %dx%d exceeds the supported maximum %d", pooled_h, pooled_w,
                      kMaxPool);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE((long)batch * channels * height * width < 2147483647L,
               "features tensor too large for int32 argmax (reference limitation)");
    if (num_rois == 0) return FI_OK;
    FI_REQUIRE(features && rois && output, "null pointer");
    int cpb = fi::ceil_div(channels, 8);
    while (cpb > 8 && (long)num_rois * fi::ceil_div(channels, cpb) < 2048) cpb = fi::ceil_div(cpb, 2);
    const int chunks = fi::ceil_div(channels, cpb);
    hipStream_t st = (hipStream_t)stream;
    fi::ProfScope prof(FI_K_ROIPOOL_FWD, st);
    hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3((unsigned)((long)num_rois * chunks)), dim3(kThreads),
                       0, st, features, rois, num_rois, batch, channels, height, width, pooled_h,
                       pooled_w, spatial_scale, cpb, chunks, output, argmax);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_roi_pool_backward(const float *top_grad, const float *rois, const int32_t *argmax,
                         int num_rois, int batch, int channels, int height, int width, int pooled_h,
                         int pooled_w, float spatial_scale, float *bottom_grad, fi_stream_t stream)
{
    FI_REQUIRE(num_rois >= 0 && batch > 0 && channels > 0 && height > 0 && width > 0, "bad sizes");
    FI_REQUIRE(pooled_h >= 1 && pooled_w >= 1, "pooled size must be >= 1");
    FI_REQUIRE(bottom_grad != nullptr, "null bottom_grad");
    hipStream_t st = (hipStream_t)stream;
    FI_HIP_CHECK(hipMemsetAsync(bottom_grad, 0,
                                sizeof(float) * (size_t)batch * channels * height * width, st));
    if (num_rois == 0) return FI_OK;
    FI_REQUIRE(top_grad && rois && argmax, "null pointer");
    const int per_roi = channels * pooled_h * pooled_w;
    fi::ProfScope prof(FI_K_ROIPOOL_BWD, st);
    const int bpr = fi::ceil_div(per_roi, kThreads);
    FI_REQUIRE((long)bpr * num_rois < 2147483647L, "grid too large");
    hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3((unsigned)((long)bpr * num_rois)), dim3(kThreads), 0,
                       st, top_grad, rois, argmax, num_rois, batch, channels, height, width, pooled_h,
                       pooled_w, spatial_scale, per_roi, bpr, bottom_grad);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
