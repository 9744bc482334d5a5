// targets.hip -- RPN anchor targets and detection (RoI) targets of one training step, each as ONE pair / ONE kernel
// instead of ~140 / ~130 framework launches.
//
// Reference: lib/layers.py:439-604 (generate_target: IoU of 261 888 anchors x G ground-truth boxes, negative / positive
// thresholds, every GT claims its best anchor, at most TRAIN_ANCHORS_PER_IMAGE/2 positives and negatives up to the
// budget, both sub-sampled at random, refinements of the positives) and :224-376 (generate_roi: IoU of the proposals
// x GT, positives IoU >= 0.5, negatives < 0.5 and not on a crowd box, random sub-sample with ROI_POSITIVE_RATIO,
// positives first, class ids, box refinements, mask-target crop boxes in mini-mask space).
//
// Random sub-sampling: the caller passes one uniform key in [1, 2) per candidate (torch.rand(...) + 1 on the device
// generator); "keep k at random" = keep the k largest keys, ties broken towards the lower index.  The arithmetic
// (IoU, refinements) is the framework formulation's, operation by operation (this file is built with -ffp-contract=off),
// so the kernels and feature_intertwiner_amd/layers.py agree bit for bit given the same keys.
#include <stdint.h>

#include "fi_common.h"

namespace {

constexpr int kMaxGT = 256;
constexpr float kEpsIoU = 10e-20f;          // tools/box_utils.py:4

// order-preserving integer image of a sampling key in [1, 2] (rand + 1 may round up to exactly 2): 0 .. 0x800000
__device__ __forceinline__ unsigned key_of(float x)
{
    const unsigned b = __float_as_uint(x);
    const unsigned k = b >= 0x3F800000u ? b - 0x3F800000u : 0u;
    return k > 0x800000u ? 0x800000u : k;
}

__device__ __forceinline__ float iou_of(float y1a, float x1a, float y2a, float x2a, float y1b, float x1b, float y2b,
                                        float x2b)
{
    // layers.bbox_overlaps: inter / (a1 + a2 - inter + eps), no "+1" convention
    const float y1 = fmaxf(y1a, y1b), x1 = fmaxf(x1a, x1b);
    const float y2 = fminf(y2a, y2b), x2 = fminf(x2a, x2b);
    const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
    const float a1 = (y2a - y1a) * (x2a - x1a);
    const float a2 = (y2b - y1b) * (x2b - x1b);
    return inter / (a1 + a2 - inter + kEpsIoU);
}

// layers.box_refinement(box, gt) / std, one component at a time
__device__ __forceinline__ void refine(const float *bx, const float *gt, const float *std4, float *out)
{
    const float h = bx[2] - bx[0], w = bx[3] - bx[1];
    const float cy = bx[0] + 0.5f * h, cx = bx[1] + 0.5f * w;
    const float gh = gt[2] - gt[0], gw = gt[3] - gt[1];
    const float gcy = gt[0] + 0.5f * gh, gcx = gt[1] + 0.5f * gw;
    out[0] = ((gcy - cy) / h) / std4[0];
    out[1] = ((gcx - cx) / w) / std4[1];
    out[2] = logf(gh / h) / std4[2];
    out[3] = logf(gw / w) / std4[3];
}

struct Std4 {
    float v[4];
};

// ------------------------------------------------------------------------------------------------------------------
// RPN targets, phase 1: per anchor the best valid GT (first maximum), the crowd test, and per GT its best anchor
// ------------------------------------------------------------------------------------------------------------------
// iou_max [b][A]; arg [b][A]: best GT index, bit 31 set when the anchor lies on a crowd box (IoU >= 0.001);
// gt_best [b][G] packed (IoU bits << 32) | (0xFFFFFFFF - anchor): atomicMax = highest IoU, then lowest anchor index
__global__ __launch_bounds__(256) void rpn_iou_kernel(const float *__restrict__ anchors, const long *__restrict__ ids,
                                                      const float *__restrict__ gts, int A, int G,
                                                      float *__restrict__ iou_max, unsigned *__restrict__ arg,
                                                      unsigned long long *__restrict__ gt_best)
{
    __shared__ float s_gt[kMaxGT][4];
    __shared__ int s_kind[kMaxGT];                       // 1 valid, -1 crowd, 0 padding
    __shared__ unsigned long long s_best[kMaxGT];
    const int img = blockIdx.y;
    for (int g = threadIdx.x; g < G; g += 256) {
        const long c = ids[(size_t)img * G + g];
        s_kind[g] = c > 0 ? 1 : (c < 0 ? -1 : 0);
        for (int k = 0; k < 4; ++k) s_gt[g][k] = gts[((size_t)img * G + g) * 4 + k];
        s_best[g] = 0ull;
    }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool ok = a < A;
    const int ac = ok ? a : A - 1;
    const float4 bx = *reinterpret_cast<const float4 *>(anchors + (size_t)ac * 4);
    float best = 0.0f, crowd = 0.0f;
    int bi = 0;
    for (int g = 0; g < G; ++g) {
        const float v = iou_of(s_gt[g][0], s_gt[g][1], s_gt[g][2], s_gt[g][3], bx.x, bx.y, bx.z, bx.w);
        const int kind = s_kind[g];
        const float vv = kind == 1 ? v : 0.0f;
        if (g == 0 || vv > best) {
            best = vv;
            bi = g;
        }
        if (kind == -1) crowd = fmaxf(crowd, v);
        if (ok) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(vv) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)a);
            if (key > s_best[g]) atomicMax(&s_best[g], key);       // (the plain read only filters; the atomic decides)
        }
    }
    if (ok) {
        iou_max[(size_t)img * A + a] = best;
        arg[(size_t)img * A + a] = (unsigned)bi | (crowd < 0.001f ? 0u : 0x80000000u);
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256)
        if (s_kind[g] == 1) atomicMax(&gt_best[(size_t)img * G + g], s_best[g]);
}

// block-wide exclusive prefix of a flag in thread order (1024 threads = 16 wavefronts); returns the prefix and, in
// *total, the block's count.  s_w: 17 ints of LDS.  Two barriers.
__device__ __forceinline__ int block_prefix(bool flag, int *s_w, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int pre = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                                   // s_w may still be read from the previous call
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int c = s_w[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    *total = tot;
    return base + pre;
}

// ------------------------------------------------------------------------------------------------------------------
// RPN targets, phase 2 (one 1024-thread workgroup per image): candidate classes, the two random sub-samples (exact
// selection of the k largest 23-bit keys: two histogram passes + an ordered tie pass), match / deltas / compact rows
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rpn_sample_kernel(const float *__restrict__ anchors, const long *__restrict__ ids,
                                                          const float *__restrict__ gts,
                                                          const float *__restrict__ key_pos,
                                                          const float *__restrict__ key_neg, int A, int G, float neg_thres,
                                                          float pos_thres, int n_total, Std4 std4,
                                                          const float *__restrict__ iou_max,
                                                          const unsigned *__restrict__ arg,
                                                          const unsigned long long *__restrict__ gt_best,
                                                          float *__restrict__ match, float *__restrict__ deltas,
                                                          long *__restrict__ row_image, long *__restrict__ row_anchor)
{
    __shared__ unsigned s_hist[2][4096];
    __shared__ int s_claim[kMaxGT];
    __shared__ int s_nclaim;
    __shared__ int s_cnt[2];
    __shared__ unsigned s_sel[2][3];                   // per class: high bucket, quota inside it; then threshold key, ties to keep
    const int img = blockIdx.x;
    const int tid = threadIdx.x;
    const float *__restrict__ im = iou_max + (size_t)img * A;
    const unsigned *__restrict__ ar = arg + (size_t)img * A;
    const float *__restrict__ kp = key_pos + (size_t)img * A;
    const float *__restrict__ kn = key_neg + (size_t)img * A;
    float *__restrict__ mt = match + (size_t)img * A;

    for (int i = tid; i < 2 * 4096; i += 1024) (&s_hist[0][0])[i] = 0u;
    if (tid == 0) {
        int n = 0;
        for (int g = 0; g < G; ++g)
            if (ids[(size_t)img * G + g] > 0) s_claim[n++] = (int)(0xFFFFFFFFu - (unsigned)(gt_best[(size_t)img * G + g] & 0xFFFFFFFFull));
        s_nclaim = n;
        s_cnt[0] = s_cnt[1] = 0;
    }
    __syncthreads();
    const int nclaim = s_nclaim;

    // pass 1: candidate class of every anchor (written to `match`), counts, histogram of the keys' high 12 bits
    int c_pos = 0, c_neg = 0;
    // (4 anchors per thread and iteration, their loads issued together: the kernel is ONE workgroup per image walking
    // 261 888 anchors five times -- its time is memory latency per iteration, round 4: 1.6-4.0 ms)
    for (int a0 = tid; a0 < A; a0 += 4096) {
        float v4[4], kp4[4], kn4[4];
        unsigned ar4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 1024 * u;
            const bool in = a < A;
            v4[u] = in ? im[a] : 0.0f;
            ar4[u] = in ? ar[a] : 0u;
            kp4[u] = in ? kp[a] : 1.0f;
            kn4[u] = in ? kn[a] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 1024 * u;
            if (a >= A) break;
            const float v = v4[u];
            float m = (v < neg_thres && !(ar4[u] & 0x80000000u)) ? -1.0f : 0.0f;
            bool claimed = false;
            for (int g = 0; g < nclaim; ++g) claimed = claimed || (s_claim[g] == a);
            if (claimed || v >= pos_thres) m = 1.0f;
            mt[a] = m;
            if (m > 0.0f) {
                ++c_pos;
                atomicAdd(&s_hist[0][key_of(kp4[u]) >> 12], 1u);
            } else if (m < 0.0f) {
                ++c_neg;
                atomicAdd(&s_hist[1][key_of(kn4[u]) >> 12], 1u);
            }
        }
    }
    atomicAdd(&s_cnt[0], c_pos);
    atomicAdd(&s_cnt[1], c_neg);
    __syncthreads();
    const int n_pc = s_cnt[0], n_nc = s_cnt[1];
    const int keep_pos = min(n_pc, n_total / 2);
    const int keep_neg = min(n_nc, max(n_total - keep_pos, 0));
    // the high bucket that holds the keep-th largest key (scanning from the top), per class
    if (tid < 2) {
        const int keep = tid == 0 ? keep_pos : keep_neg;
        const int have = tid == 0 ? n_pc : n_nc;
        unsigned bucket = 0xFFFFFFFFu, quota = 0;          // bucket 0xFFFFFFFF: everything is kept / nothing is
        if (keep > 0 && keep < have) {
            int acc = 0;
            for (int h = 4095; h >= 0; --h) {
                const int c = (int)s_hist[tid][h];
                if (acc + c >= keep) {
                    bucket = (unsigned)h;
                    quota = (unsigned)(keep - acc);        // 1 .. c of this bucket's candidates are kept
                    break;
                }
                acc += c;
            }
        }
        s_sel[tid][0] = bucket;
        s_sel[tid][1] = quota;
    }
    __syncthreads();
    const unsigned bk_p = s_sel[0][0], bk_n = s_sel[1][0];
    // pass 2: histogram of the low 12 bits inside the threshold buckets
    for (int i = tid; i < 2 * 4096; i += 1024) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    if (bk_p != 0xFFFFFFFFu || bk_n != 0xFFFFFFFFu) {
        for (int a0 = tid; a0 < A; a0 += 4096) {
            float m4[4], kp4[4], kn4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + 1024 * u;
                const bool in = a < A;
                m4[u] = in ? mt[a] : 0.0f;
                kp4[u] = in ? kp[a] : 1.0f;
                kn4[u] = in ? kn[a] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float m = m4[u];
                if (m > 0.0f && bk_p != 0xFFFFFFFFu) {
                    const unsigned k = key_of(kp4[u]);
                    if ((k >> 12) == bk_p) atomicAdd(&s_hist[0][k & 4095u], 1u);
                } else if (m < 0.0f && bk_n != 0xFFFFFFFFu) {
                    const unsigned k = key_of(kn4[u]);
                    if ((k >> 12) == bk_n) atomicAdd(&s_hist[1][k & 4095u], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (tid < 2) {
        const unsigned bucket = s_sel[tid][0];
        unsigned thr = 0, ties = 0;                        // keep keys > thr, and the first `ties` keys == thr
        if (bucket != 0xFFFFFFFFu) {
            const int quota = (int)s_sel[tid][1];
            int acc = 0;
            for (int l = 4095; l >= 0; --l) {
                const int c = (int)s_hist[tid][l];
                if (acc + c >= quota) {
                    thr = (bucket << 12) | (unsigned)l;
                    ties = (unsigned)(quota - acc);
                    break;
                }
                acc += c;
            }
        }
        s_sel[tid][1] = thr;
        s_sel[tid][2] = ties;
    }
    __syncthreads();
    const bool all_p = bk_p == 0xFFFFFFFFu, all_n = bk_n == 0xFFFFFFFFu;
    const unsigned thr_p = s_sel[0][1], thr_n = s_sel[1][1];
    const int ties_p = (int)s_sel[0][2], ties_n = (int)s_sel[1][2];
    // pass 3, in anchor order: final match, refinements of the kept positives, compact (image, anchor) rows.  Ties at a
    // selection boundary and the row positions need ORDERED counts: every wavefront owns a contiguous range of anchors
    // (walked 64 at a time, coalesced), counts its ties / kept anchors in a first sweep, the 16 counts are scanned once,
    // and a second sweep assigns ranks with wavefront ballots only -- no barrier inside the loops.
    __shared__ int s_wcnt[3][16];
    const int lane = tid & 63, wave = tid >> 6;
    const int per_wave = ((A + 15) / 16 + 63) / 64 * 64;
    const int a_begin = wave * per_wave, a_end = min(A, a_begin + per_wave);
    struct Loaded { float m, kp, kn; };
    auto load4 = [&](int a0, Loaded (&v)[4]) {               // the next four 64-anchor chunks of this wavefront: 12 loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 64 * u + lane;
            const bool in = a < a_end;
            v[u].m = in ? mt[a] : 0.0f;
            v[u].kp = in ? kp[a] : 1.0f;
            v[u].kn = in ? kn[a] : 1.0f;
        }
    };
    auto classify = [&](const Loaded &v, bool in, bool &cand_p, bool &cand_n, bool &tie_p, bool &tie_n, bool &sure_p, bool &sure_n) {
        const float m = in ? v.m : 0.0f;
        const unsigned k_p = in ? key_of(v.kp) : 0u;
        const unsigned k_n = in ? key_of(v.kn) : 0u;
        cand_p = m > 0.0f;
        cand_n = m < 0.0f;
        tie_p = cand_p && !all_p && k_p == thr_p;
        tie_n = cand_n && !all_n && k_n == thr_n;
        sure_p = cand_p && keep_pos > 0 && (all_p ? (keep_pos >= n_pc) : (k_p > thr_p));
        sure_n = cand_n && keep_neg > 0 && (all_n ? (keep_neg >= n_nc) : (k_n > thr_n));
    };
    int c_tp = 0, c_tn = 0;
    for (int a4 = a_begin; a4 < a_end; a4 += 256) {
        Loaded ld[4];
        load4(a4, ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a4 + 64 * u + lane;
            bool cp, cn, tp, tn, sp, sn;
            classify(ld[u], a < a_end, cp, cn, tp, tn, sp, sn);
            c_tp += __popcll(__ballot(tp));
            c_tn += __popcll(__ballot(tn));
        }
    }
    if (lane == 0) {
        s_wcnt[0][wave] = c_tp;
        s_wcnt[1][wave] = c_tn;
    }
    __syncthreads();
    int seen_tp = 0, seen_tn = 0;
    for (int w = 0; w < wave; ++w) {
        seen_tp += s_wcnt[0][w];
        seen_tn += s_wcnt[1][w];
    }
    // second sweep: the kept flags are now decidable; count the rows per wavefront first (third counter), then write
    int c_rows = 0;
    {
        int tp_run = seen_tp, tn_run = seen_tn;
        for (int a4 = a_begin; a4 < a_end; a4 += 256) {
            Loaded ld[4];
            load4(a4, ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a4 + 64 * u + lane;
                bool cp, cn, tp, tn, sp, sn;
                classify(ld[u], a < a_end, cp, cn, tp, tn, sp, sn);
                const unsigned long long mtp = __ballot(tp), mtn = __ballot(tn);
                const unsigned long long below = (1ull << lane) - 1ull;
                const bool keep_p = sp || (tp && keep_pos > 0 && tp_run + __popcll(mtp & below) < ties_p);
                const bool keep_n = sn || (tn && keep_neg > 0 && tn_run + __popcll(mtn & below) < ties_n);
                tp_run += __popcll(mtp);
                tn_run += __popcll(mtn);
                c_rows += __popcll(__ballot(keep_p || keep_n));
            }
        }
    }
    if (lane == 0) s_wcnt[2][wave] = c_rows;
    __syncthreads();
    int rows = 0, row_base = 0;
    for (int w = 0; w < 16; ++w) {
        row_base += w < wave ? s_wcnt[2][w] : 0;
        rows += s_wcnt[2][w];
    }
    {
        int tp_run = seen_tp, tn_run = seen_tn, r_run = row_base;
        for (int a4 = a_begin; a4 < a_end; a4 += 256) {
            Loaded ld[4];
            load4(a4, ld);               // (before any of the four chunks' match values is replaced below)
            unsigned ar4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a4 + 64 * u + lane;
                ar4[u] = a < a_end ? ar[a] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a4 + 64 * u + lane;
                const bool in = a < a_end;
                bool cp, cn, tp, tn, sp, sn;
                classify(ld[u], in, cp, cn, tp, tn, sp, sn);
                const unsigned long long mtp = __ballot(tp), mtn = __ballot(tn);
                const unsigned long long below = (1ull << lane) - 1ull;
                const bool keep_p = sp || (tp && keep_pos > 0 && tp_run + __popcll(mtp & below) < ties_p);
                const bool keep_n = sn || (tn && keep_neg > 0 && tn_run + __popcll(mtn & below) < ties_n);
                tp_run += __popcll(mtp);
                tn_run += __popcll(mtn);
                const unsigned long long mk = __ballot(keep_p || keep_n);
                if (in) {
                    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (keep_p) {
                        const int g = (int)(ar4[u] & 0x7FFFFFFFu);
                        const float4 bx = *reinterpret_cast<const float4 *>(anchors + (size_t)a * 4);
                        const float b4[4] = {bx.x, bx.y, bx.z, bx.w};
                        const float *gt = gts + ((size_t)img * G + g) * 4;
                        const float g4[4] = {gt[0], gt[1], gt[2], gt[3]};
                        refine(b4, g4, std4.v, d);
                    }
                    *reinterpret_cast<float4 *>(deltas + ((size_t)img * A + a) * 4) = make_float4(d[0], d[1], d[2], d[3]);
                    const int pos = r_run + __popcll(mk & below);
                    if ((keep_p || keep_n) && row_image && pos < n_total) {
                        row_image[(size_t)img * n_total + pos] = img;
                        row_anchor[(size_t)img * n_total + pos] = a;
                    }
                }
                r_run += __popcll(mk);
                // (the final match replaces the candidate class LAST: it was read into ld[] for this and the following
                // three chunks already, and no other wavefront reads this element)
                if (in) mt[a] = keep_p ? 1.0f : (keep_n ? -1.0f : 0.0f);
            }
        }
    }
    if (row_image)
        for (int r = rows + tid; r < n_total; r += 1024) {
            row_image[(size_t)img * n_total + r] = -1;
            row_anchor[(size_t)img * n_total + r] = -1;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// detection targets: one 1024-thread workgroup per image, P <= 2048 proposals
// ------------------------------------------------------------------------------------------------------------------
// LDS bitonic sort of 2048 64-bit keys, descending
__device__ __forceinline__ void bitonic_desc_2048(unsigned long long *s)
{
    for (int k = 2; k <= 2048; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < 1024; t += 1024) {
                const int i = ((t / j) * 2 * j) + (t % j);          // lower index of the pair
                const int p = i + j;
                const bool desc = ((i & k) == 0);
                const unsigned long long x = s[i], y = s[p];
                if ((x < y) == desc) {
                    s[i] = y;
                    s[p] = x;
                }
            }
        }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void det_targets_kernel(const float *__restrict__ proposals, const long *__restrict__ num_prop,
                                                           const long *__restrict__ ids, const float *__restrict__ gts,
                                                           const float *__restrict__ key_pos,
                                                           const float *__restrict__ key_neg, int P, int G, int R,
                                                           int pos_cap, double ratio, int mini_mask, Std4 std4,
                                                           float *__restrict__ rois, int *__restrict__ cls_out,
                                                           float *__restrict__ deltas, float *__restrict__ mask_boxes,
                                                           int *__restrict__ mask_ids, float *__restrict__ is_pos_out)
{
    __shared__ float s_gt[kMaxGT][4];
    __shared__ int s_kind[kMaxGT];
    __shared__ long s_cls[kMaxGT];
    __shared__ unsigned long long s_pos[2048], s_neg[2048];
    __shared__ unsigned short s_assign[2048];
    __shared__ int s_cnt[2];
    const int img = blockIdx.x, tid = threadIdx.x;
    for (int g = tid; g < G; g += 1024) {
        const long c = ids[(size_t)img * G + g];
        s_cls[g] = c;
        s_kind[g] = c > 0 ? 1 : (c < 0 ? -1 : 0);
        for (int k = 0; k < 4; ++k) s_gt[g][k] = gts[((size_t)img * G + g) * 4 + k];
    }
    if (tid == 0) s_cnt[0] = s_cnt[1] = 0;
    __syncthreads();
    const long nvalid = num_prop[img];
    for (int p = tid; p < 2048; p += 1024) {
        unsigned long long kpos = 0ull, kneg = 0ull;
        if (p < P) {
            const float4 bx = *reinterpret_cast<const float4 *>(proposals + ((size_t)img * P + p) * 4);
            float best = 0.0f, crowd = 0.0f;
            int bi = 0;
            for (int g = 0; g < G; ++g) {
                const float v = iou_of(bx.x, bx.y, bx.z, bx.w, s_gt[g][0], s_gt[g][1], s_gt[g][2], s_gt[g][3]);
                const float vv = s_kind[g] == 1 ? v : 0.0f;
                if (g == 0 || vv > best) {
                    best = vv;
                    bi = g;
                }
                if (s_kind[g] == -1) crowd = fmaxf(crowd, v);
            }
            s_assign[p] = (unsigned short)bi;
            const bool valid = p < nvalid;
            const bool is_p = best >= 0.5f && valid;
            const bool is_n = best < 0.5f && crowd < 0.001f && valid;
            const unsigned low = 0xFFFFFFFFu - (unsigned)p;                 // ties: lower index first
            if (is_p) {
                kpos = ((unsigned long long)__float_as_uint(key_pos[(size_t)img * P + p]) << 32) | low;
                atomicAdd(&s_cnt[0], 1);
            }
            if (is_n) {
                kneg = ((unsigned long long)__float_as_uint(key_neg[(size_t)img * P + p]) << 32) | low;
                atomicAdd(&s_cnt[1], 1);
            }
        }
        s_pos[p] = kpos;
        s_neg[p] = kneg;
    }
    bitonic_desc_2048(s_pos);
    bitonic_desc_2048(s_neg);
    const int n_pos_avail = s_cnt[0], n_neg_avail = s_cnt[1];
    const int pos_cnt = min(n_pos_avail, pos_cap);
    const long neg_want = (long)floor(ratio * (double)pos_cnt - (double)pos_cnt);
    long neg_cnt = neg_want < (long)n_neg_avail ? neg_want : (long)n_neg_avail;
    neg_cnt = neg_cnt < R ? neg_cnt : R;
    neg_cnt = neg_cnt < (long)(R - pos_cnt) ? neg_cnt : (long)(R - pos_cnt);
    if (neg_cnt < 0) neg_cnt = 0;
    for (int r = tid; r < R; r += 1024) {
        const bool isp = r < pos_cnt;
        const bool isn = !isp && r < pos_cnt + (int)neg_cnt;
        // layers.prepare_det_target: sel = gather(pos_idx / neg_idx) with clamped slots; unused slots are zeroed
        int sel = 0;
        if (isp)
            sel = (int)(0xFFFFFFFFu - (unsigned)(s_pos[r] & 0xFFFFFFFFull));
        else if (isn)
            sel = (int)(0xFFFFFFFFu - (unsigned)(s_neg[r - pos_cnt] & 0xFFFFFFFFull));
        const bool used = isp || isn;
        float bx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (used)
            for (int k = 0; k < 4; ++k) bx[k] = proposals[((size_t)img * P + sel) * 4 + k];
        const int g = used ? (int)s_assign[sel] : 0;
        float d[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (isp) {
            refine(bx, s_gt[g], std4.v, d);
            if (mini_mask) {
                const float gh = s_gt[g][2] - s_gt[g][0], gw = s_gt[g][3] - s_gt[g][1];
                mb[0] = (bx[0] - s_gt[g][0]) / gh;
                mb[1] = (bx[1] - s_gt[g][1]) / gw;
                mb[2] = (bx[2] - s_gt[g][0]) / gh;
                mb[3] = (bx[3] - s_gt[g][1]) / gw;
            } else {
                for (int k = 0; k < 4; ++k) mb[k] = bx[k];
            }
        }
        const size_t o = (size_t)img * R + r;
        for (int k = 0; k < 4; ++k) {
            rois[o * 4 + k] = bx[k];
            deltas[o * 4 + k] = d[k];
            mask_boxes[o * 4 + k] = mb[k];
        }
        cls_out[o] = isp ? (int)s_cls[g] : 0;
        mask_ids[o] = g + img * G;
        is_pos_out[o] = isp ? 1.0f : 0.0f;
    }
}

}   // namespace

extern "C" {

size_t fi_rpn_targets_workspace_bytes(int batch, int anchors, int max_gt)
{
    return (size_t)batch * anchors * 8 + (size_t)batch * max_gt * 8 + 64;
}

int fi_rpn_targets(const float *anchors, const int64_t *gt_class_ids, const float *gt_boxes, const float *key_pos,
                   const float *key_neg, int batch, int n_anchors, int max_gt, float neg_thres, float pos_thres,
                   int n_total, const float *bbox_std_dev, float *match, float *deltas, int64_t *row_image,
                   int64_t *row_anchor, void *workspace, fi_stream_t stream)
{
    FI_REQUIRE(batch >= 1 && n_anchors >= 1 && max_gt >= 1 && max_gt <= kMaxGT, "1 <= max_gt <= 256, batch, anchors >= 1");
    FI_REQUIRE(n_total >= 2 && n_total <= 4096, "2 <= anchors per image <= 4096");
    FI_REQUIRE(anchors && gt_class_ids && gt_boxes && key_pos && key_neg && bbox_std_dev && match && deltas && workspace,
               "null pointer");
    FI_REQUIRE((row_image == nullptr) == (row_anchor == nullptr), "row_image and row_anchor come together");
    FI_REQUIRE((uintptr_t)anchors % 16 == 0 && (uintptr_t)deltas % 16 == 0 && (uintptr_t)workspace % 16 == 0,
               "anchors, deltas and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float *iou_max = reinterpret_cast<float *>(workspace);
    unsigned *arg = reinterpret_cast<unsigned *>(iou_max + (size_t)batch * n_anchors);
    unsigned long long *gt_best = reinterpret_cast<unsigned long long *>(
        reinterpret_cast<char *>(workspace) + (((size_t)batch * n_anchors * 8 + 15) / 16) * 16);
    FI_HIP_CHECK(hipMemsetAsync(gt_best, 0, (size_t)batch * max_gt * 8, st));
    Std4 s4;
    for (int k = 0; k < 4; ++k) s4.v[k] = bbox_std_dev[k];
    hipLaunchKernelGGL(rpn_iou_kernel, dim3(fi::ceil_div(n_anchors, 256), batch), dim3(256), 0, st, anchors,
                       reinterpret_cast<const long *>(gt_class_ids), gt_boxes, n_anchors, max_gt, iou_max, arg, gt_best);
    hipLaunchKernelGGL(rpn_sample_kernel, dim3(batch), dim3(1024), 0, st, anchors,
                       reinterpret_cast<const long *>(gt_class_ids), gt_boxes, key_pos, key_neg, n_anchors, max_gt, neg_thres,
                       pos_thres, n_total, s4, iou_max, arg, gt_best, match, deltas, reinterpret_cast<long *>(row_image),
                       reinterpret_cast<long *>(row_anchor));
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_detection_targets(const float *proposals, const int64_t *num_proposals, const int64_t *gt_class_ids,
                         const float *gt_boxes, const float *key_pos, const float *key_neg, int batch, int n_proposals,
                         int max_gt, int rois_per_image, int positive_cap, double negatives_per_positive, int use_mini_mask,
                         const float *bbox_std_dev, float *rois, int32_t *target_class_ids, float *target_deltas,
                         float *mask_boxes, int32_t *mask_box_ids, float *is_positive, fi_stream_t stream)
{
    FI_REQUIRE(batch >= 1 && n_proposals >= 1 && n_proposals <= 2048 && max_gt >= 1 && max_gt <= kMaxGT,
               "1 <= proposals <= 2048, 1 <= max_gt <= 256");
    FI_REQUIRE(rois_per_image >= 1 && positive_cap >= 0 && positive_cap <= rois_per_image && negatives_per_positive >= 1.0,
               "rois_per_image >= 1, 0 <= positive_cap <= rois_per_image, negatives_per_positive = 1 / ROI_POSITIVE_RATIO >= 1");
    FI_REQUIRE(proposals && num_proposals && gt_class_ids && gt_boxes && key_pos && key_neg && bbox_std_dev && rois &&
               target_class_ids && target_deltas && mask_boxes && mask_box_ids && is_positive, "null pointer");
    FI_REQUIRE((uintptr_t)proposals % 16 == 0, "proposals must be 16-byte aligned");
    Std4 s4;
    for (int k = 0; k < 4; ++k) s4.v[k] = bbox_std_dev[k];
    const int pos_cap = positive_cap;                    // int(R * ROI_POSITIVE_RATIO), evaluated by the caller (:270)
    const double ratio = negatives_per_positive;         // 1 / ROI_POSITIVE_RATIO in double, as the caller computes it
    hipLaunchKernelGGL(det_targets_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, proposals,
                       reinterpret_cast<const long *>(num_proposals), reinterpret_cast<const long *>(gt_class_ids), gt_boxes,
                       key_pos, key_neg, n_proposals, max_gt, rois_per_image, pos_cap, ratio, use_mini_mask, s4, rois,
                       target_class_ids, target_deltas, mask_boxes, mask_box_ids, is_positive);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}   // extern "C"
