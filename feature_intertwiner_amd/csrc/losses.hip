// losses.hip -- the five detector losses of one training step and their gradients in ONE pass.
//
// Reference: lib/layers.py:808-934 -- compute_rpn_class_loss (cross entropy over the sampled anchors), compute_rpn_bbox_loss
// (smooth L1 over the positive anchors), compute_mrcnn_class_loss (cross entropy over all RoIs, zero when the batch has no
// foreground), compute_mrcnn_bbox_loss (smooth L1 on the positive RoIs' target-class box), compute_mrcnn_mask_loss (binary
// cross entropy on the positive RoIs' target-class mask).  The framework formulation (feature_intertwiner_amd/layers.py)
// costs ~35 small launches forward and ~45 backward, all of them on the critical path between the forward and the
// backward pass.  Here: one kernel writes, per element, the loss term's contribution to per-row partial sums and the
// UNNORMALISED gradient of its loss with respect to the network output; a second, single-workgroup kernel adds the
// partials in a fixed order (deterministic), divides by the counts and leaves the five losses and the five factors
// 1 / count the backward pass multiplies the stored gradients with.
#include <stdint.h>

#include "fi_common.h"

namespace {

__device__ __forceinline__ float smooth_l1(float d, float *g)
{
    const float a = fabsf(d);
    *g = a < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
    return a < 1.0f ? 0.5f * d * d : a - 0.5f;
}

// workgroup sum of one float per thread (256 threads), result in every thread
__device__ __forceinline__ float block_sum(float v, float *s)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}

struct LossArgs {
    // RPN rows
    const float *rpn_match;      // [b][A]
    const float *rpn_deltas;     // [b][A][4]
    const long *row_image;       // [Rr]   (-1: no row)
    const long *row_anchor;      // [Rr]
    const float *row_logits;     // [Rr][2]
    const float *row_bbox;       // [Rr][4]
    int Rr, A;
    // RoI heads
    const int *roi_cls;          // [N] target class ids
    const float *cls_logits;     // [N][K]
    const float *roi_deltas;     // [N][4]
    const float *roi_bbox;       // [N][K][4]
    int N, K;
    // masks: logits of the target class, un-shuffled [Nm][2][2][h][w]; targets [Nm][2h][2w]; class ids [Nm]
    const int *mask_cls;
    const float *mask_logits;
    const float *mask_target;
    int Nm, h, w;
    // outputs
    float *g_row_logits, *g_row_bbox, *g_cls_logits, *g_roi_bbox, *g_mask_logits;
    float *partial;              // [blocks][10]: sums of the five losses, counts (rpn rows with match != 0, rpn positives,
                                 //               roi positives, mask-row positives), spare
};

// grid: blocks 0 .. nb_r-1 -> RPN rows (256 per block); next nb_n -> RoI rows (one wavefront per RoI: 4 per block);
// next nb_m -> mask rows (one block per RoI)
__global__ __launch_bounds__(256) void losses_kernel(LossArgs a, int nb_r, int nb_n)
{
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int blk = blockIdx.x;
    if (blk < nb_r) {
        const int r = blk * 256 + tid;
        if (r < a.Rr) {
            const long im = a.row_image[r], an = a.row_anchor[r];
            const bool valid = im >= 0;
            const float m = valid ? a.rpn_match[im * a.A + an] : 0.0f;
            // cross entropy over (background, foreground) on the rows with match != 0 (lib/layers.py:808-829)
            const float l0 = a.row_logits[r * 2], l1 = a.row_logits[r * 2 + 1];
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            const float lse = mx + logf(e0 + e1);
            const int t = m == 1.0f ? 1 : 0;
            const float on = m != 0.0f ? 1.0f : 0.0f;
            acc[0] = on * (lse - (t ? l1 : l0));
            acc[5] = on;
            const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
            a.g_row_logits[r * 2] = on * (p0 - (t ? 0.0f : 1.0f));
            a.g_row_logits[r * 2 + 1] = on * (p1 - (t ? 1.0f : 0.0f));
            // smooth L1 on the positive rows (:832-861)
            const float pos = m == 1.0f ? 1.0f : 0.0f;
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) {
                const float tgt = valid ? a.rpn_deltas[(im * a.A + an) * 4 + k] : 0.0f;
                float g;
                s += smooth_l1(a.row_bbox[r * 4 + k] - tgt, &g);
                a.g_row_bbox[r * 4 + k] = pos * g;
            }
            acc[1] = pos * s;
            acc[6] = pos;
        }
    } else if (blk < nb_r + nb_n) {
        // one wavefront per RoI: soft-max cross entropy over K classes, smooth L1 on the target class's box
        const int n = (blk - nb_r) * 4 + (tid >> 6);
        const int lane = tid & 63;
        if (n < a.N) {
            const int t = a.roi_cls[n];
            const float *__restrict__ lg = a.cls_logits + (size_t)n * a.K;
            float mx = -3.4e38f;
            for (int k = lane; k < a.K; k += 64) mx = fmaxf(mx, lg[k]);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float se = 0.0f;
            for (int k = lane; k < a.K; k += 64) se += expf(lg[k] - mx);
            for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
            const float lse = mx + logf(se);
            for (int k = lane; k < a.K; k += 64)
                a.g_cls_logits[(size_t)n * a.K + k] = expf(lg[k] - mx) / se - (k == t ? 1.0f : 0.0f);
            const float pos = t > 0 ? 1.0f : 0.0f;
            float *__restrict__ gb = a.g_roi_bbox + (size_t)n * a.K * 4;
            float lb = 0.0f;
            for (int i = lane; i < a.K * 4; i += 64) {             // every class row is written: zeros off the target class
                float g = 0.0f;
                if (t > 0 && (i >> 2) == t)
                    lb += smooth_l1(a.roi_bbox[(size_t)n * a.K * 4 + i] - a.roi_deltas[n * 4 + (i & 3)], &g);
                gb[i] = g;
            }
            acc[3] = lb;
            if (lane == 0) {
                acc[2] = lse - lg[t];
                acc[7] = pos;
            }
        }
    } else {
        // one workgroup per mask row: sigmoid + binary cross entropy against the pixel-shuffled target (:905-934)
        const int n = blk - nb_r - nb_n;
        const int hw = a.h * a.w, W2 = 2 * a.w;
        const bool pos = a.mask_cls[n] > 0;
        const float *__restrict__ lg = a.mask_logits + (size_t)n * 4 * hw;
        const float *__restrict__ tg = a.mask_target + (size_t)n * 4 * hw;
        float *__restrict__ gl = a.g_mask_logits + (size_t)n * 4 * hw;
        float s = 0.0f;
        for (int i = tid; i < 4 * hw; i += 256) {
            float g = 0.0f;
            if (pos) {
                const int ab = i / hw, p = i - ab * hw;
                const int y = p / a.w, x = p - y * a.w;
                const float t = tg[(2 * y + (ab >> 1)) * W2 + 2 * x + (ab & 1)];
                const float pr = 1.0f / (1.0f + expf(-lg[i]));
                // F.binary_cross_entropy: the logarithms are clamped at -100
                const float lp = fmaxf(logf(pr), -100.0f), lq = fmaxf(logf(1.0f - pr), -100.0f);
                s -= t * lp + (1.0f - t) * lq;
                // its backward: (p - t) / max((1 - p) p, 1e-12), times the sigmoid's p (1 - p)
                g = (pr - t) / fmaxf((1.0f - pr) * pr, 1e-12f) * (pr * (1.0f - pr));
            }
            gl[i] = g;
        }
        acc[4] = s;
        if (tid == 0) acc[8] = pos ? 1.0f : 0.0f;
    }
    // per-block partial sums (one row of 8 per block), reduced in block order by losses_finish_kernel
    for (int q = 0; q < 10; ++q) {
        const float v = block_sum(acc[q], s_red);
        if (tid == 0) a.partial[(size_t)blockIdx.x * 10 + q] = v;
    }
}

// out[0..4] = the five losses, out[5..9] = the factors d loss / d (stored gradient) = 1 / count (0.. when the loss is
// switched off): rpn_class, rpn_bbox, mrcnn_class, mrcnn_bbox, mrcnn_mask
__global__ __launch_bounds__(256) void losses_finish_kernel(const float *__restrict__ partial, int blocks, int N, int mask_px,
                                                            float *__restrict__ out)
{
    __shared__ float s_red[4];
    float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < blocks; b += 256)
        for (int q = 0; q < 10; ++q) acc[q] += partial[(size_t)b * 10 + q];
    float tot[10];
    for (int q = 0; q < 10; ++q) tot[q] = block_sum(acc[q], s_red);
    if (threadIdx.x == 0) {
        const float n_rpn = fmaxf(tot[5], 1.0f), n_rpn_pos4 = fmaxf(tot[6] * 4.0f, 1.0f);
        const float has_fg = tot[7] > 0.0f ? 1.0f : 0.0f;
        const float n_pos4 = fmaxf(tot[7] * 4.0f, 1.0f), n_mask = fmaxf(tot[8] * (float)mask_px, 1.0f);
        out[0] = tot[0] / n_rpn;            out[5] = 1.0f / n_rpn;
        out[1] = tot[1] / n_rpn_pos4;       out[6] = 1.0f / n_rpn_pos4;
        out[2] = tot[2] / (float)N * has_fg; out[7] = has_fg / (float)N;
        out[3] = tot[3] / n_pos4;           out[8] = 1.0f / n_pos4;
        out[4] = tot[4] / n_mask;           out[9] = 1.0f / n_mask;
    }
}

}   // namespace

extern "C" {

size_t fi_detector_losses_workspace_bytes(int rpn_rows, int rois, int mask_rows)
{
    const size_t blocks = (size_t)fi::ceil_div(rpn_rows, 256) + (size_t)fi::ceil_div(rois, 4) + (size_t)mask_rows;
    return blocks * 10 * sizeof(float);
}

int fi_detector_losses(const float *rpn_match, const float *rpn_deltas, const int64_t *row_image, const int64_t *row_anchor,
                       const float *row_logits, const float *row_bbox, int rpn_rows, int anchors,
                       const int32_t *roi_class_ids, const float *class_logits, const float *roi_deltas,
                       const float *roi_bbox, int rois, int num_classes, const int32_t *mask_class_ids,
                       const float *mask_logits, const float *mask_targets, int mask_rows, int mask_h, int mask_w,
                       float *grad_row_logits, float *grad_row_bbox, float *grad_class_logits, float *grad_roi_bbox,
                       float *grad_mask_logits, float *losses_and_factors, void *workspace, fi_stream_t stream)
{
    FI_REQUIRE(rpn_rows >= 1 && anchors >= 1 && rois >= 1 && num_classes >= 2 && mask_rows >= 1 && mask_h >= 1 && mask_w >= 1,
               "sizes must be positive");
    FI_REQUIRE(rpn_match && rpn_deltas && row_image && row_anchor && row_logits && row_bbox && roi_class_ids && class_logits &&
               roi_deltas && roi_bbox && mask_class_ids && mask_logits && mask_targets && grad_row_logits && grad_row_bbox &&
               grad_class_logits && grad_roi_bbox && grad_mask_logits && losses_and_factors && workspace, "null pointer");
    LossArgs a;
    a.rpn_match = rpn_match; a.rpn_deltas = rpn_deltas;
    a.row_image = reinterpret_cast<const long *>(row_image); a.row_anchor = reinterpret_cast<const long *>(row_anchor);
    a.row_logits = row_logits; a.row_bbox = row_bbox; a.Rr = rpn_rows; a.A = anchors;
    a.roi_cls = roi_class_ids; a.cls_logits = class_logits; a.roi_deltas = roi_deltas; a.roi_bbox = roi_bbox;
    a.N = rois; a.K = num_classes;
    a.mask_cls = mask_class_ids; a.mask_logits = mask_logits; a.mask_target = mask_targets;
    a.Nm = mask_rows; a.h = mask_h; a.w = mask_w;
    a.g_row_logits = grad_row_logits; a.g_row_bbox = grad_row_bbox; a.g_cls_logits = grad_class_logits;
    a.g_roi_bbox = grad_roi_bbox; a.g_mask_logits = grad_mask_logits;
    a.partial = reinterpret_cast<float *>(workspace);
    const int nb_r = fi::ceil_div(rpn_rows, 256), nb_n = fi::ceil_div(rois, 4);
    const int blocks = nb_r + nb_n + mask_rows;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(losses_kernel, dim3(blocks), dim3(256), 0, st, a, nb_r, nb_n);
    hipLaunchKernelGGL(losses_finish_kernel, dim3(1), dim3(256), 0, st, a.partial, blocks, rois, 4 * mask_h * mask_w,
                       losses_and_factors);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}   // extern "C"
