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
        // seeded with the key the workgroup's FIRST anchor has at IoU +0 (its real key is at least that): the anchors
        // without overlap -- nearly all -- are then filtered by the plain read below instead of all 256 threads
        // queueing a same-address 64-bit LDS atomic per GT (that queue was most of this kernel: 235 -> see profiles)
        s_best[g] = (unsigned long long)(0xFFFFFFFFu - (unsigned)min(blockIdx.x * 256, (unsigned)(A - 1)));
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
    // one global atomic per (workgroup, GT) whose best key can still win: a key with zero IoU bits only competes on the
    // anchor index, and the lowest anchor of the image sits in workgroup 0 (1023 workgroups x 20 GTs of 64-bit
    // atomicMax on 20 addresses per image were most of this kernel's 236 us)
    for (int g = threadIdx.x; g < G; g += 256)
        if (s_kind[g] == 1 && ((s_best[g] >> 32) != 0ull || blockIdx.x == 0))
            atomicMax(&gt_best[(size_t)img * G + g], s_best[g]);
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
// RPN targets, phase 2, as SEVEN small launches of kSampleWGs workgroups per image (round 5; rounds 2-4: one
// 1024-thread workgroup per image walked the 261 888 anchors five times -- 1.4 ms, issue bound on ONE compute unit):
// candidate classes, the two random sub-samples (exact selection of the k largest 23-bit keys: two histogram passes
// + an ordered tie pass), match / deltas / compact rows.  The state between the launches lives in the workspace
// (SampleState per image, cleared by the entry point): the histograms are accumulated in LDS per workgroup and added
// to the image's with integer atomics, the ordered passes keep the per-WAVEFRONT counts (every wavefront of every
// workgroup owns a contiguous range of anchors, in anchor order), so the results -- which anchors are kept, the row
// order -- are what the one-workgroup kernel produced, bit for bit.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSampleWGs = 16;                       // workgroups per image
constexpr int kSampleWaves = kSampleWGs * 16;        // wavefronts per image: the units of the ordered passes

struct SampleState {
    unsigned hist_hi[2][4096];
    unsigned hist_lo[2][4096];
    int cnt[2];                                      // candidates per class (positive, negative)
    int keep[2];                                     // how many of them are kept
    unsigned sel[2][3];                              // high bucket (0xFFFFFFFF: all / none), threshold key, ties to keep
    int wcnt[3][kSampleWaves];                       // per wavefront: ties (pos), ties (neg), kept rows
};

struct SampleArgs {
    const float *anchors;
    const long *ids;
    const float *gts, *key_pos, *key_neg;
    int A, G;
    float neg_thres, pos_thres;
    int n_total;
    Std4 std4;
    const float *iou_max;
    const unsigned *arg;
    const unsigned long long *gt_best;
    float *match, *deltas;
    long *row_image, *row_anchor;
    SampleState *state;
};

__device__ __forceinline__ int sample_per_wave(int A) { return ((A + kSampleWaves - 1) / kSampleWaves + 63) / 64 * 64; }

// launch 1: candidate class of every anchor (written to `match`), counts, histogram of the keys' high 12 bits
__global__ __launch_bounds__(1024) void rpn_cand_kernel(SampleArgs p)
{
    __shared__ unsigned s_hist[2][4096];
    __shared__ int s_claim[kMaxGT];
    __shared__ int s_nclaim;
    __shared__ int s_cnt[2];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int A = p.A;
    const float *__restrict__ im = p.iou_max + (size_t)img * A;
    const unsigned *__restrict__ ar = p.arg + (size_t)img * A;
    const float *__restrict__ kp = p.key_pos + (size_t)img * A;
    const float *__restrict__ kn = p.key_neg + (size_t)img * A;
    float *__restrict__ mt = p.match + (size_t)img * A;
    SampleState *__restrict__ st = p.state + img;
    for (int i = tid; i < 2 * 4096; i += 1024) (&s_hist[0][0])[i] = 0u;
    if (tid == 0) {
        int n = 0;
        for (int g = 0; g < p.G; ++g)
            if (p.ids[(size_t)img * p.G + g] > 0)
                s_claim[n++] = (int)(0xFFFFFFFFu - (unsigned)(p.gt_best[(size_t)img * p.G + g] & 0xFFFFFFFFull));
        s_nclaim = n;
        s_cnt[0] = s_cnt[1] = 0;
    }
    __syncthreads();
    const int nclaim = s_nclaim;
    const int per_wg = sample_per_wave(A) * 16;
    const int a_lo = blockIdx.x * per_wg, a_hi = min(A, a_lo + per_wg);
    int c_pos = 0, c_neg = 0;
    for (int a0 = a_lo + tid; a0 < a_hi; a0 += 4096) {
        float v4[4], kp4[4], kn4[4];
        unsigned ar4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 1024 * u;
            const bool in = a < a_hi;
            v4[u] = in ? im[a] : 0.0f;
            ar4[u] = in ? ar[a] : 0u;
            kp4[u] = in ? kp[a] : 1.0f;
            kn4[u] = in ? kn[a] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 1024 * u;
            if (a >= a_hi) break;
            const float v = v4[u];
            float m = (v < p.neg_thres && !(ar4[u] & 0x80000000u)) ? -1.0f : 0.0f;
            bool claimed = false;
            for (int g = 0; g < nclaim; ++g) claimed = claimed || (s_claim[g] == a);
            if (claimed || v >= p.pos_thres) m = 1.0f;
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
    if (c_pos) atomicAdd(&s_cnt[0], c_pos);
    if (c_neg) atomicAdd(&s_cnt[1], c_neg);
    __syncthreads();
    for (int i = tid; i < 2 * 4096; i += 1024) {
        const unsigned c = (&s_hist[0][0])[i];
        if (c) atomicAdd(&st->hist_hi[0][0] + i, c);
    }
    if (tid < 2 && s_cnt[tid]) atomicAdd(&st->cnt[tid], s_cnt[tid]);
}

// The bin (scanning from the top) that holds the quota-th largest entry of a 4096-bin histogram, and how many entries of
// that bin are wanted: 64 threads sum 64 bins each, one thread walks the 64 sums and then the bin's 64 counts.
__device__ __forceinline__ void pick_bin(const unsigned *__restrict__ hist, int quota, unsigned *s_part, int tid64,
                                         unsigned *bin, unsigned *rest)
{
    unsigned sum = 0;
    for (int i = 0; i < 64; ++i) sum += hist[4095 - (tid64 * 64 + i)];       // chunk tid64: bins 4095 - 64 t .. down
    s_part[tid64] = sum;
    __syncthreads();
    if (tid64 == 0) {
        int acc = 0, c = 0;
        for (; c < 64; ++c) {
            if (acc + (int)s_part[c] >= quota) break;
            acc += (int)s_part[c];
        }
        unsigned b = 0, r = 0;
        if (c < 64) {
            for (int i = 0; i < 64; ++i) {
                const int h = 4095 - (c * 64 + i);
                const int n = (int)hist[h];
                if (acc + n >= quota) {
                    b = (unsigned)h;
                    r = (unsigned)(quota - acc);
                    break;
                }
                acc += n;
            }
        }
        *bin = b;
        *rest = r;
    }
    __syncthreads();
}

// launches 2 and 4 (one 128-thread workgroup per image; threads 0..63: positives, 64..127: negatives)
template <bool LOW>
__global__ __launch_bounds__(128) void rpn_pick_kernel(SampleState *__restrict__ state, int n_total)
{
    __shared__ unsigned s_part[2][64];
    __shared__ unsigned s_out[2][2];
    SampleState *__restrict__ st = state + blockIdx.x;
    const int cls = threadIdx.x >> 6, t64 = threadIdx.x & 63;
    const int n_pc = st->cnt[0], n_nc = st->cnt[1];
    const int keep_pos = min(n_pc, n_total / 2);
    const int keep_neg = min(n_nc, max(n_total - keep_pos, 0));
    const int keep = cls == 0 ? keep_pos : keep_neg, have = cls == 0 ? n_pc : n_nc;
    if (!LOW) {
        const bool scan = keep > 0 && keep < have;                       // otherwise everything is kept / nothing is
        unsigned b = 0, r = 0;
        // (both halves of the workgroup reach the barriers inside pick_bin: a quota of 1 << 30 finds no bin)
        pick_bin(st->hist_hi[cls], scan ? keep : (1 << 30), s_part[cls], t64, &s_out[cls][0], &s_out[cls][1]);
        if (t64 == 0) {
            b = scan ? s_out[cls][0] : 0xFFFFFFFFu;
            r = scan ? s_out[cls][1] : 0u;
            st->sel[cls][0] = b;
            st->sel[cls][1] = r;                                         // quota inside the bucket (launch 4 reads it)
            st->keep[cls] = keep;
        }
    } else {
        const unsigned bucket = st->sel[cls][0];
        const int quota = (int)st->sel[cls][1];
        pick_bin(st->hist_lo[cls], bucket != 0xFFFFFFFFu ? quota : (1 << 30), s_part[cls], t64, &s_out[cls][0], &s_out[cls][1]);
        if (t64 == 0) {
            st->sel[cls][1] = bucket != 0xFFFFFFFFu ? ((bucket << 12) | s_out[cls][0]) : 0u;      // threshold key
            st->sel[cls][2] = bucket != 0xFFFFFFFFu ? s_out[cls][1] : 0u;                           // ties to keep
        }
    }
}

// launch 3: histogram of the low 12 bits inside the threshold buckets
__global__ __launch_bounds__(1024) void rpn_hist_lo_kernel(SampleArgs p)
{
    __shared__ unsigned s_hist[2][4096];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int A = p.A;
    SampleState *__restrict__ st = p.state + img;
    const unsigned bk_p = st->sel[0][0], bk_n = st->sel[1][0];
    if (bk_p == 0xFFFFFFFFu && bk_n == 0xFFFFFFFFu) return;
    const float *__restrict__ kp = p.key_pos + (size_t)img * A;
    const float *__restrict__ kn = p.key_neg + (size_t)img * A;
    const float *__restrict__ mt = p.match + (size_t)img * A;
    for (int i = tid; i < 2 * 4096; i += 1024) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    const int per_wg = sample_per_wave(A) * 16;
    const int a_lo = blockIdx.x * per_wg, a_hi = min(A, a_lo + per_wg);
    for (int a0 = a_lo + tid; a0 < a_hi; a0 += 4096) {
        float m4[4], kp4[4], kn4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 1024 * u;
            const bool in = a < a_hi;
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
    __syncthreads();
    for (int i = tid; i < 2 * 4096; i += 1024) {
        const unsigned c = (&s_hist[0][0])[i];
        if (c) atomicAdd(&st->hist_lo[0][0] + i, c);
    }
}

// launches 5-7, in anchor order: ties at a selection boundary and the row positions need ORDERED counts.  PASS 0 counts
// every wavefront's ties, PASS 1 (knowing the ties in front of it) its kept rows, PASS 2 (knowing both) writes the final
// match, the refinements of the kept positives and the compact (image, anchor) rows.
template <int PASS>
__global__ __launch_bounds__(1024) void rpn_order_kernel(SampleArgs p)
{
    const int img = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int gw = blockIdx.x * 16 + wave;                     // this wavefront among the image's kSampleWaves
    const int A = p.A, n_total = p.n_total;
    SampleState *__restrict__ st = p.state + img;
    const float *__restrict__ kp = p.key_pos + (size_t)img * A;
    const float *__restrict__ kn = p.key_neg + (size_t)img * A;
    const unsigned *__restrict__ ar = p.arg + (size_t)img * A;
    float *__restrict__ mt = p.match + (size_t)img * A;
    const int n_pc = st->cnt[0], n_nc = st->cnt[1];
    const int keep_pos = st->keep[0], keep_neg = st->keep[1];
    const bool all_p = st->sel[0][0] == 0xFFFFFFFFu, all_n = st->sel[1][0] == 0xFFFFFFFFu;
    const unsigned thr_p = st->sel[0][1], thr_n = st->sel[1][1];
    const int ties_p = (int)st->sel[0][2], ties_n = (int)st->sel[1][2];
    const int per_wave = sample_per_wave(A);
    const int a_begin = min(A, gw * per_wave), a_end = min(A, a_begin + per_wave);

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
    auto classify = [&](const Loaded &v, bool in, bool &tie_p, bool &tie_n, bool &sure_p, bool &sure_n) {
        const float m = in ? v.m : 0.0f;
        const unsigned k_p = in ? key_of(v.kp) : 0u;
        const unsigned k_n = in ? key_of(v.kn) : 0u;
        const bool cand_p = m > 0.0f, cand_n = m < 0.0f;
        tie_p = cand_p && !all_p && k_p == thr_p;
        tie_n = cand_n && !all_n && k_n == thr_n;
        sure_p = cand_p && keep_pos > 0 && (all_p ? (keep_pos >= n_pc) : (k_p > thr_p));
        sure_n = cand_n && keep_neg > 0 && (all_n ? (keep_neg >= n_nc) : (k_n > thr_n));
    };
    if (PASS == 0) {
        int c_tp = 0, c_tn = 0;
        for (int a4 = a_begin; a4 < a_end; a4 += 256) {
            Loaded ld[4];
            load4(a4, ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bool tp, tn, sp, sn;
                classify(ld[u], a4 + 64 * u + lane < a_end, tp, tn, sp, sn);
                c_tp += __popcll(__ballot(tp));
                c_tn += __popcll(__ballot(tn));
            }
        }
        if (lane == 0) {
            st->wcnt[0][gw] = c_tp;
            st->wcnt[1][gw] = c_tn;
        }
        return;
    }
    // ties in front of this wavefront (and, PASS 2, rows in front of it / in total): lanes sum the per-wavefront counts
    int seen_tp = 0, seen_tn = 0, row_base = 0, rows = 0;
    for (int w = lane; w < kSampleWaves; w += 64) {
        if (w < gw) {
            seen_tp += st->wcnt[0][w];
            seen_tn += st->wcnt[1][w];
        }
        if (PASS == 2) {
            const int r = st->wcnt[2][w];
            rows += r;
            if (w < gw) row_base += r;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        seen_tp += __shfl_xor(seen_tp, o);
        seen_tn += __shfl_xor(seen_tn, o);
        row_base += __shfl_xor(row_base, o);
        rows += __shfl_xor(rows, o);
    }
    int tp_run = seen_tp, tn_run = seen_tn, r_run = row_base, c_rows = 0;
    for (int a4 = a_begin; a4 < a_end; a4 += 256) {
        Loaded ld[4];
        load4(a4, ld);               // (before any of the four chunks' match values is replaced below)
        unsigned ar4[4];
        if (PASS == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a4 + 64 * u + lane;
                ar4[u] = a < a_end ? ar[a] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a4 + 64 * u + lane;
            const bool in = a < a_end;
            bool tp, tn, sp, sn;
            classify(ld[u], in, tp, tn, sp, sn);
            const unsigned long long mtp = __ballot(tp), mtn = __ballot(tn);
            const unsigned long long below = (1ull << lane) - 1ull;
            const bool keep_p = sp || (tp && keep_pos > 0 && tp_run + __popcll(mtp & below) < ties_p);
            const bool keep_n = sn || (tn && keep_neg > 0 && tn_run + __popcll(mtn & below) < ties_n);
            tp_run += __popcll(mtp);
            tn_run += __popcll(mtn);
            const unsigned long long mk = __ballot(keep_p || keep_n);
            if (PASS == 1) {
                c_rows += __popcll(mk);
                continue;
            }
            if (in) {
                float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (keep_p) {
                    const int g = (int)(ar4[u] & 0x7FFFFFFFu);
                    const float4 bx = *reinterpret_cast<const float4 *>(p.anchors + (size_t)a * 4);
                    const float b4[4] = {bx.x, bx.y, bx.z, bx.w};
                    const float *gt = p.gts + ((size_t)img * p.G + g) * 4;
                    const float g4[4] = {gt[0], gt[1], gt[2], gt[3]};
                    refine(b4, g4, p.std4.v, d);
                }
                *reinterpret_cast<float4 *>(p.deltas + ((size_t)img * A + a) * 4) = make_float4(d[0], d[1], d[2], d[3]);
                const int pos = r_run + __popcll(mk & below);
                if ((keep_p || keep_n) && p.row_image && pos < n_total) {
                    p.row_image[(size_t)img * n_total + pos] = img;
                    p.row_anchor[(size_t)img * n_total + pos] = a;
                }
                // (the final match replaces the candidate class LAST: it was read into ld[] for this and the following
                // three chunks already, and no other wavefront reads this element)
                mt[a] = keep_p ? 1.0f : (keep_n ? -1.0f : 0.0f);
            }
            r_run += __popcll(mk);
        }
    }
    if (PASS == 1) {
        if (lane == 0) st->wcnt[2][gw] = c_rows;
        return;
    }
    if (p.row_image && blockIdx.x == 0)
        for (int r = rows + tid; r < n_total; r += 1024) {
            p.row_image[(size_t)img * n_total + r] = -1;
            p.row_anchor[(size_t)img * n_total + r] = -1;
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
    // iou_max + arg per anchor, gt_best per GT, the sampling state per image (SampleState)
    return (((size_t)batch * anchors * 8 + 15) / 16) * 16 + (((size_t)batch * max_gt * 8 + 15) / 16) * 16 +
           (size_t)batch * sizeof(SampleState) + 64;
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
    SampleState *state = reinterpret_cast<SampleState *>(reinterpret_cast<char *>(gt_best) +
                                                         (((size_t)batch * max_gt * 8 + 15) / 16) * 16);
    FI_HIP_CHECK(hipMemsetAsync(state, 0, (size_t)batch * sizeof(SampleState), st));
    SampleArgs sa = {anchors, reinterpret_cast<const long *>(gt_class_ids), gt_boxes, key_pos, key_neg, n_anchors, max_gt,
                     neg_thres, pos_thres, n_total, s4, iou_max, arg, gt_best, match, deltas,
                     reinterpret_cast<long *>(row_image), reinterpret_cast<long *>(row_anchor), state};
    const dim3 grid(kSampleWGs, batch);
    hipLaunchKernelGGL(rpn_cand_kernel, grid, dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(rpn_pick_kernel<false>, dim3(batch), dim3(128), 0, st, state, n_total);
    hipLaunchKernelGGL(rpn_hist_lo_kernel, grid, dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(rpn_pick_kernel<true>, dim3(batch), dim3(128), 0, st, state, n_total);
    hipLaunchKernelGGL(rpn_order_kernel<0>, grid, dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(rpn_order_kernel<1>, grid, dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(rpn_order_kernel<2>, grid, dim3(1024), 0, st, sa);
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
