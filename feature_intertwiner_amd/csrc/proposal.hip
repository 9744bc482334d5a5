// proposal.hip -- the pre-NMS stage of the proposal layer as ONE kernel, and the post-NMS gather as another.
//
// Specification: proposal_layer, lib/layers.py:71-139 of the reference: foreground scores -> the
// PRE_NMS_LIMIT best anchors in descending score order (lib/layers.py:99-106: a full sort of all 261 888
// anchors, then a slice) -> deltas * BBOX_STD_DEV -> apply_box_deltas (tools/box_utils.py:7-33) ->
// clip_boxes to the image window (:36-60) -> [y1, x1, y2, x2, score] rows for NMS.  Ties between equal
// scores are broken by the lower anchor index (a stable sort); the reference's torch.sort leaves them
// unspecified.
//
// One 1024-thread workgroup per image:
//   1. exact radix SELECT of the K-th largest score (3 passes of 11/11/10 bits over an order-preserving
//      integer image of the float, histogram in LDS) -- no sort of the 261 888 scores;
//   2. compaction of the K winners into LDS as 64-bit keys (~score bits : index); ties at the threshold are
//      taken in index order;
//   3. bitonic sort of the (<= 8192) keys in LDS: descending score, ascending index;
//   4. decode + clip of the K boxes and the dets rows written coalesced.
// Optional EXTERNAL candidates (`extra` [batch, E, 5] = box + score, e.g. precomputed proposals) take part
// in the same selection; they rank before anchors at equal score.
#include <stdint.h>

#include <atomic>

#include "fi_common.h"

namespace {

typedef unsigned long long u64;
constexpr int kSelThreads = 1024;
constexpr int kSortCap = 8192;            // keys sorted in LDS (64 KB)
constexpr int kBins = 2048;

// order-preserving map float -> uint32 (larger float <=> larger integer; -0 < +0; NaN above +inf)
__device__ __forceinline__ unsigned sortable(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SelectArgs {
    const float *probs;      // [batch, A, prob_stride], foreground score at +prob_off
    const float *deltas;     // [batch, A, 4]
    const float *anchors;    // [A, 4] pixels (y1, x1, y2, x2)
    const float *extra;      // [batch, E, 5] or null
    float *dets;             // [batch, K, 5]
    int A, E, K, prob_stride, prob_off;
    float std0, std1, std2, std3, win_h, win_w;
    int lds_sort;            // A/B (FI_PROPOSAL_LDS_SORT): the all-LDS bitonic sort in proposal_sort_kernel
};

__device__ __forceinline__ float score_of(const SelectArgs &a, int img, int i)
{
    if (i < a.E) return a.extra[((size_t)img * a.E + i) * 5 + 4];
    return a.probs[((size_t)img * a.A + (i - a.E)) * a.prob_stride + a.prob_off];
}

// Block-wide: given hist[kBins] (counts per digit), find the digit d with
//   count(digits > d) < want <= count(digits >= d); returns d and count(digits > d) through LDS.
__device__ __forceinline__ void find_digit(int *hist, int nbins, int want, int *s_wave, int *s_out)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // thread t owns bins [2t, 2t+1] counted from the TOP (descending digits)
    const int hi = nbins - 1 - 2 * tid, lo = hi - 1;
    const int c_hi = (hi >= 0) ? hist[hi] : 0;
    const int c_lo = (lo >= 0) ? hist[lo] : 0;
    int incl = c_hi + c_lo;                       // inclusive prefix over threads (descending digits)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    const int before = base + incl - (c_hi + c_lo);     // count of digits above this thread's pair
    if (before < want && want <= before + c_hi + c_lo) {
        if (want <= before + c_hi) { s_out[0] = hi; s_out[1] = before; }
        else { s_out[0] = lo; s_out[1] = before + c_hi; }
    }
    __syncthreads();
}

// decode + clip of the K sorted winners (tools/box_utils.py:7-60, each operation rounded separately)
__device__ __forceinline__ void decode_rows(const SelectArgs &a, int img, const u64 *s_keys, int first = -1, int step = 0)
{
    const int K = a.K;
    float *out = a.dets + (size_t)img * K * 5;
    if (first < 0) { first = threadIdx.x; step = kSelThreads; }
    for (int r = first; r < K; r += step) {
        const int i = (int)(unsigned)(s_keys[r] & 0xffffffffull);
        float y1, x1, y2, x2, sc;
        if (i < a.E) {
            const float *e = a.extra + ((size_t)img * a.E + i) * 5;
            y1 = e[0]; x1 = e[1]; y2 = e[2]; x2 = e[3]; sc = e[4];
        } else {
            const int an = i - a.E;
            const float4 b = *reinterpret_cast<const float4 *>(a.anchors + 4 * (size_t)an);
            const float4 d = *reinterpret_cast<const float4 *>(a.deltas + ((size_t)img * a.A + an) * 4);
            sc = a.probs[((size_t)img * a.A + an) * a.prob_stride + a.prob_off];
            const float d0 = d.x * a.std0, d1 = d.y * a.std1, d2 = d.z * a.std2, d3 = d.w * a.std3;
            float height = b.z - b.x;
            float width = b.w - b.y;
            float cy = b.x + 0.5f * height;
            float cx = b.y + 0.5f * width;
            cy = cy + d0 * height;
            cx = cx + d1 * width;
            height = height * expf(d2);
            width = width * expf(d3);
            y1 = cy - 0.5f * height;
            x1 = cx - 0.5f * width;
            y2 = y1 + height;
            x2 = x1 + width;
        }
        // clamp(min, max) = min(max(v, lo), hi), NaN propagates as in torch.clamp
        y1 = fminf(fmaxf(y1, 0.0f), a.win_h);
        x1 = fminf(fmaxf(x1, 0.0f), a.win_w);
        y2 = fminf(fmaxf(y2, 0.0f), a.win_h);
        x2 = fminf(fmaxf(x2, 0.0f), a.win_w);
        out[r * 5 + 0] = y1;
        out[r * 5 + 1] = x1;
        out[r * 5 + 2] = y2;
        out[r * 5 + 3] = x2;
        out[r * 5 + 4] = sc;
    }
}

__global__ __launch_bounds__(kSelThreads) void proposal_select_kernel(SelectArgs a)
{
    extern __shared__ u64 s_keys[];                      // kSortCap keys; the histogram aliases its start
    int *hist = reinterpret_cast<int *>(s_keys);
    __shared__ int s_wave[kSelThreads / 64];
    __shared__ int s_out[2];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const int total = a.A + a.E;
    const int K = a.K;                                   // host guarantees 1 <= K <= min(total, kSortCap)

    // ---- 1. radix select of the K-th largest key ---------------------------------------------------------
    unsigned prefix = 0, prefix_mask = 0;
    int want = K, count_eq = 0;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int p = 0; p < 3; ++p) {
        const int nb = 1 << widths[p];
        for (int i = tid; i < kBins; i += kSelThreads) hist[i] = 0;
        __syncthreads();
        // (4 scores per thread and iteration, their loads in flight together: one workgroup walks 261 888 scores four
        // times, its time is memory latency per iteration)
        for (int i0 = tid; i0 < total; i0 += 4 * kSelThreads) {
            float sc4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kSelThreads;
                sc4[u] = i < total ? score_of(a, img, i) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * kSelThreads >= total) break;
                const unsigned k = sortable(sc4[u]);
                if ((k & prefix_mask) == prefix) atomicAdd(&hist[(k >> shifts[p]) & (nb - 1)], 1);
            }
        }
        __syncthreads();
        find_digit(hist, nb, want, s_wave, s_out);
        const int d = s_out[0];
        want -= s_out[1];
        count_eq = hist[d];              // after the last pass: how many keys equal the threshold key
        prefix |= (unsigned)d << shifts[p];
        prefix_mask |= (unsigned)(nb - 1) << shifts[p];
        __syncthreads();
    }
    const unsigned T = prefix;           // the K-th largest key; `want` of the keys equal to T are still needed
    const int need_eq = want;

    // ---- 2. compaction -----------------------------------------------------------------------------------
    __shared__ int s_cnt_eq;
    if (tid == 0) { s_cnt = 0; s_cnt_eq = 0; }
    __syncthreads();
    const bool all_ties = (count_eq == need_eq);         // every key equal to T is a winner: no order needed
    for (int i0 = tid; i0 < total; i0 += 4 * kSelThreads) {
        float sc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kSelThreads;
            sc4[u] = i < total ? score_of(a, img, i) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kSelThreads;
            if (i >= total) break;
            const unsigned k = sortable(sc4[u]);
            if (k > T) {
                const int pos = atomicAdd(&s_cnt, 1);
                s_keys[pos] = ((u64)(~k) << 32) | (unsigned)i;
            } else if (k == T && all_ties) {
                const int pos = atomicAdd(&s_cnt_eq, 1);
                s_keys[(K - need_eq) + pos] = ((u64)(~k) << 32) | (unsigned)i;
            }
        }
    }
    __syncthreads();
    // more keys equal to T than places left: the lowest indices win -- walk the indices in order, 1024 at a time,
    // ranks from wavefront ballots, stop as soon as the places are filled
    if (!all_ties) {
        int taken = 0;                                   // uniform across the block
        for (int base = 0; base < total && taken < need_eq; base += kSelThreads) {
            const int i = base + tid;
            const bool is_eq = i < total && sortable(score_of(a, img, i)) == T;
            const u64 ballot = __ballot(is_eq);
            const int lane = tid & 63, wave = tid >> 6;
            const int in_wave = __popcll(ballot & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[wave] = __popcll(ballot);
            __syncthreads();
            int before = 0, chunk = 0;
            for (int w = 0; w < kSelThreads / 64; ++w) {
                const int c = s_wave[w];
                if (w < wave) before += c;
                chunk += c;
            }
            const int rank = taken + before + in_wave;
            if (is_eq && rank < need_eq) s_keys[(K - need_eq) + rank] = ((u64)(~T) << 32) | (unsigned)i;
            taken += chunk;
            __syncthreads();
        }
    }
    // pad to the next power of two with maximal keys
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    for (int i = K + tid; i < n2; i += kSelThreads) s_keys[i] = ~0ull;
    __syncthreads();

    // ---- 3. bitonic sort, ascending u64 = descending score, ascending index ---------------------------------
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += kSelThreads) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const u64 x = s_keys[lo], y = s_keys[hi];
                if ((x > y) == up) { s_keys[lo] = y; s_keys[hi] = x; }
            }
            __syncthreads();
        }
    }

    decode_rows(a, img, s_keys);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same selection as proposal_select_kernel in NINE small launches (round 6, fi_proposal_candidates_ws): the single
// kernel is one workgroup per image that walks the 261 888 scores four times -- 4 workgroups on 256 CUs, ~310 us, memory
// latency per iteration on ONE CU each.  Here the three radix passes and the compaction run on kHistWgs workgroups per
// image with the state in a caller-provided workspace; only the sort + decode stays one workgroup per image.  The result
// is the same: the final order comes from the sort of (score, index) keys, not from the arrival order of the atomics.
//   per image:  hist[2048] | state {prefix, mask, want, count_eq, n_cand, n_tie, T, need_eq} | cand[8192] | tie[8192]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kHistWgs = 32;              // workgroups per image of the multi-workgroup passes
constexpr int kHistThreads = 256;
constexpr int kCompactWgs = 128;          // workgroups per image of the compaction (8 iterations of 256 scores each at 261 888)
struct SelState {
    unsigned prefix, mask;
    int want, count_eq, n_cand, n_tie;
    unsigned T;
    int need_eq;
};
constexpr size_t kWsHist = sizeof(int) * kBins;
constexpr size_t kWsState = 64;           // >= sizeof(SelState), keeps the key arrays 16-byte aligned
constexpr size_t kWsKeys = sizeof(u64) * kSortCap;
constexpr size_t kWsPerImage = kWsHist + kWsState + 2 * kWsKeys;

__device__ __forceinline__ int *ws_hist(char *ws, int img) { return reinterpret_cast<int *>(ws + (size_t)img * kWsPerImage); }
__device__ __forceinline__ SelState *ws_state(char *ws, int img)
{
    return reinterpret_cast<SelState *>(ws + (size_t)img * kWsPerImage + kWsHist);
}
__device__ __forceinline__ u64 *ws_cand(char *ws, int img)
{
    return reinterpret_cast<u64 *>(ws + (size_t)img * kWsPerImage + kWsHist + kWsState);
}
__device__ __forceinline__ u64 *ws_tie(char *ws, int img) { return ws_cand(ws, img) + kSortCap; }

// pass p of the radix select: histogram of digit p of the keys that match the prefix so far
__global__ __launch_bounds__(kHistThreads) void proposal_hist_kernel(SelectArgs a, char *__restrict__ ws, int pass)
{
    __shared__ int s_hist[kBins];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int total = a.A + a.E;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    const int nb = 1 << widths[pass], sh = shifts[pass];
    const SelState st = *ws_state(ws, img);
    const unsigned prefix = pass ? st.prefix : 0u, pmask = pass ? st.mask : 0u;
    for (int i = tid; i < nb; i += kHistThreads) s_hist[i] = 0;
    __syncthreads();
    const int per = (total + kHistWgs - 1) / kHistWgs;
    const int lo = blockIdx.x * per, hi = min(total, lo + per);
    for (int i0 = lo + tid; i0 < hi; i0 += 4 * kHistThreads) {
        float sc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kHistThreads;
            sc4[u] = i < hi ? score_of(a, img, i) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * kHistThreads >= hi) break;
            const unsigned k = sortable(sc4[u]);
            if ((k & pmask) == prefix) atomicAdd(&s_hist[(k >> sh) & (nb - 1)], 1);
        }
    }
    __syncthreads();
    int *g = ws_hist(ws, img);
    for (int i = tid; i < nb; i += kHistThreads) {
        const int c = s_hist[i];
        if (c) atomicAdd(&g[i], c);
    }
}

// after pass p: the digit that holds the K-th largest key; the state moves on; the histogram is cleared for the next pass
__global__ __launch_bounds__(kSelThreads) void proposal_digit_kernel(SelectArgs a, char *__restrict__ ws, int pass)
{
    __shared__ int s_hist[kBins];
    __shared__ int s_wave[kSelThreads / 64];
    __shared__ int s_out[2];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    const int nb = 1 << widths[pass];
    int *g = ws_hist(ws, img);
    SelState *st = ws_state(ws, img);
    for (int i = tid; i < kBins; i += kSelThreads) {
        s_hist[i] = i < nb ? g[i] : 0;
        g[i] = 0;
    }
    __syncthreads();
    const int want = pass ? st->want : a.K;
    find_digit(s_hist, nb, want, s_wave, s_out);
    if (tid == 0) {
        const int d = s_out[0];
        const unsigned prefix = (pass ? st->prefix : 0u) | ((unsigned)d << shifts[pass]);
        st->prefix = prefix;
        st->mask = (pass ? st->mask : 0u) | ((unsigned)(nb - 1) << shifts[pass]);
        st->want = want - s_out[1];
        st->count_eq = s_hist[d];
        if (pass == 2) {
            st->T = prefix;
            st->need_eq = want - s_out[1];
        }
    }
}

// keys above the threshold go to cand[], keys equal to it to tie[] (used when every tie is a winner).  A workgroup reserves
// its two ranges with ONE pair of global atomics (one returning atomic per wavefront and iteration on two words per image
// serialised at the memory side: 40 us of the selection); the places inside the range come from LDS counters.
constexpr int kCompactPerThread = 16;     // scores per thread held in registers between the two phases
__global__ __launch_bounds__(kHistThreads) void proposal_compact_kernel(SelectArgs a, char *__restrict__ ws)
{
    __shared__ int s_n[2], s_base[2];
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int total = a.A + a.E;
    SelState *st = ws_state(ws, img);
    const unsigned T = st->T;
    u64 *cand = ws_cand(ws, img), *tie = ws_tie(ws, img);
    const int per = (total + kCompactWgs - 1) / kCompactWgs;
    const int lo = blockIdx.x * per, hi = min(total, lo + per);
    if (tid < 2) s_n[tid] = 0;
    __syncthreads();
    for (int base0 = lo; base0 < hi; base0 += kCompactPerThread * kHistThreads) {       // (one round unless per > 4096)
        unsigned key[kCompactPerThread];
        int pos[kCompactPerThread];                   // place inside the workgroup's range; bit 30: a tie; -1: neither
#pragma unroll
        for (int u = 0; u < kCompactPerThread; ++u) {                                  // all scores in flight together
            const int i = base0 + u * kHistThreads + tid;
            key[u] = i < hi ? sortable(score_of(a, img, i)) : 0u;
        }
#pragma unroll
        for (int u = 0; u < kCompactPerThread; ++u) {
            const int i = base0 + u * kHistThreads + tid;
            const bool in = i < hi;
            const bool above = in && key[u] > T, eq = in && key[u] == T;
            const u64 b_above = __ballot(above), b_eq = __ballot(eq);
            pos[u] = -1;
            if (b_above) {
                int p0 = 0;
                if (lane == 0) p0 = atomicAdd(&s_n[0], __popcll(b_above));
                p0 = __shfl(p0, 0, 64);
                if (above) pos[u] = p0 + __popcll(b_above & ((1ull << lane) - 1ull));
            }
            if (b_eq) {
                int p0 = 0;
                if (lane == 0) p0 = atomicAdd(&s_n[1], __popcll(b_eq));
                p0 = __shfl(p0, 0, 64);
                if (eq) pos[u] = (p0 + __popcll(b_eq & ((1ull << lane) - 1ull))) | (1 << 30);
            }
        }
        __syncthreads();
        if (tid < 2) {
            const int n = s_n[tid];
            s_base[tid] = n ? atomicAdd(tid == 0 ? &st->n_cand : &st->n_tie, n) : 0;
            s_n[tid] = 0;
        }
        __syncthreads();
        const int b0 = s_base[0], b1 = s_base[1];
#pragma unroll
        for (int u = 0; u < kCompactPerThread; ++u) {
            if (pos[u] < 0) continue;
            const int i = base0 + u * kHistThreads + tid;
            const u64 k64 = ((u64)(~key[u]) << 32) | (unsigned)i;
            if (pos[u] & (1 << 30)) {
                const int p = b1 + (pos[u] & ~(1 << 30));
                if (p < kSortCap) tie[p] = k64;
            } else {
                cand[b0 + pos[u]] = k64;
            }
        }
        __syncthreads();                              // (s_base is rewritten by a further round)
    }
}

// Bitonic sort of KPT * 1024 u64 keys, ascending, with the keys in REGISTERS (thread t holds keys t*KPT .. t*KPT+KPT-1):
// compare-exchange distances below KPT stay inside a thread, distances up to 32 threads are lane exchanges inside the
// wavefront (ds_bpermute, no barrier), and only the distances of 64 threads and more go through LDS -- 10 of the 91
// stages of the 8192-key sort (the all-LDS form: one barrier and four LDS round trips per thread in every stage).
template <int KPT>
__device__ __forceinline__ void sort_keys_in_registers(u64 *s_keys, u64 *__restrict__ sorted_out, int n_out)
{
    constexpr int N2 = KPT * kSelThreads;
    const int tid = threadIdx.x;
    u64 key[KPT];
    // (the input order is arbitrary: a conflict-free read)
#pragma unroll
    for (int e = 0; e < KPT; ++e) key[e] = s_keys[e * kSelThreads + tid];
#pragma unroll
    for (int size = 2; size <= N2; size <<= 1) {
        const bool asc_thread = ((tid * KPT) & size) == 0;      // (for size >= 2 * KPT: the same for all keys of a thread)
#pragma unroll
        for (int stride = size / 2; stride >= 64 * KPT; stride >>= 1) {
            // partner thread tid ^ (stride / KPT) is the same lane of another wavefront: keys through LDS, key-major
            // ([e][thread]: consecutive lanes, consecutive words -- no bank conflicts), compared in registers
            const int partner = tid ^ (stride / KPT);
            const bool keep_min = (((tid * KPT) & stride) == 0) == asc_thread;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < KPT; ++e) s_keys[e * kSelThreads + tid] = key[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < KPT; ++e) {
                const u64 other = s_keys[e * kSelThreads + partner];
                const bool take = keep_min ? (other < key[e]) : (other > key[e]);
                key[e] = take ? other : key[e];
            }
        }
#pragma unroll
        for (int stride = (size / 2 < 64 * KPT ? size / 2 : 32 * KPT); stride >= KPT; stride >>= 1) {
            // lane exchange: the partner thread is tid ^ (stride / KPT), inside the wavefront
            const int ld = stride / KPT;
            const bool lower = ((tid * KPT) & stride) == 0;
            const bool keep_min = lower == asc_thread;
#pragma unroll
            for (int e = 0; e < KPT; ++e) {
                const unsigned olo = __shfl_xor((unsigned)(key[e] & 0xffffffffull), ld, 64);
                const unsigned ohi = __shfl_xor((unsigned)(key[e] >> 32), ld, 64);
                const u64 other = ((u64)ohi << 32) | olo;
                const bool take = keep_min ? (other < key[e]) : (other > key[e]);
                key[e] = take ? other : key[e];
            }
        }
#pragma unroll
        for (int stride = (size / 2 < KPT ? size / 2 : KPT / 2); stride >= 1; stride >>= 1) {
#pragma unroll
            for (int e = 0; e < KPT; ++e) {
                if (e & stride) continue;
                const bool up = (((tid * KPT + e) & size) == 0);
                const u64 x = key[e], y = key[e + stride];
                const bool sw = (x > y) == up;
                key[e] = sw ? y : x;
                key[e + stride] = sw ? x : y;
            }
        }
    }
    // thread t holds the sorted keys t*KPT .. t*KPT+KPT-1: straight to memory
#pragma unroll
    for (int e = 0; e < KPT; ++e)
        if (tid * KPT + e < n_out) sorted_out[tid * KPT + e] = key[e];
}

// one workgroup per image: winners into LDS, ties, bitonic sort, decode + clip (as proposal_select_kernel's steps 2b-4)
__global__ __launch_bounds__(kSelThreads) void proposal_sort_kernel(SelectArgs a, char *__restrict__ ws)
{
    extern __shared__ u64 s_keys[];
    __shared__ int s_wave[kSelThreads / 64];
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const int total = a.A + a.E;
    const int K = a.K;
    const SelState st = *ws_state(ws, img);
    const unsigned T = st.T;
    const int need_eq = st.need_eq, n_above = K - need_eq;
    u64 *cand = ws_cand(ws, img);
    const u64 *tie = ws_tie(ws, img);
    {
        u64 v[kSortCap / kSelThreads];                   // (all loads in flight together)
#pragma unroll
        for (int u = 0; u < kSortCap / kSelThreads; ++u) {
            const int i = tid + u * kSelThreads;
            v[u] = i < n_above ? cand[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kSortCap / kSelThreads; ++u) {
            const int i = tid + u * kSelThreads;
            if (i < n_above) s_keys[i] = v[u];
        }
    }
    const bool all_ties = (st.count_eq == need_eq);
    if (all_ties) {
        for (int i = tid; i < need_eq; i += kSelThreads) s_keys[n_above + i] = tie[i];
    }
    __syncthreads();
    if (!all_ties) {
        // more keys equal to T than places left: the lowest indices win (walk the indices in order, stop when filled)
        int taken = 0;
        for (int base = 0; base < total && taken < need_eq; base += kSelThreads) {
            const int i = base + tid;
            const bool is_eq = i < total && sortable(score_of(a, img, i)) == T;
            const u64 ballot = __ballot(is_eq);
            const int lane = tid & 63, wave = tid >> 6;
            const int in_wave = __popcll(ballot & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[wave] = __popcll(ballot);
            __syncthreads();
            int before = 0, chunk = 0;
            for (int w = 0; w < kSelThreads / 64; ++w) {
                const int c = s_wave[w];
                if (w < wave) before += c;
                chunk += c;
            }
            const int rank = taken + before + in_wave;
            if (is_eq && rank < need_eq) s_keys[n_above + rank] = ((u64)(~T) << 32) | (unsigned)i;
            taken += chunk;
            __syncthreads();
        }
    }
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    for (int i = K + tid; i < n2; i += kSelThreads) s_keys[i] = ~0ull;
    __syncthreads();
    // (sorted keys go back to the workspace -- the candidate array is done with: proposal_decode_kernel reads them)
    if (n2 == 8 * kSelThreads && !a.lds_sort) { sort_keys_in_registers<8>(s_keys, cand, K); return; }
    if (n2 == 4 * kSelThreads && !a.lds_sort) { sort_keys_in_registers<4>(s_keys, cand, K); return; }
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += kSelThreads) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const u64 x = s_keys[lo], y = s_keys[hi];
                if ((x > y) == up) { s_keys[lo] = y; s_keys[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < K; i += kSelThreads) cand[i] = s_keys[i];
}

// decode + clip of the sorted winners on kDecodeWgs workgroups per image (inside the sort kernel: one workgroup per image,
// six rows per thread one after the other, each behind three dependent loads)
constexpr int kDecodeThreads = 256;
__global__ __launch_bounds__(kDecodeThreads) void proposal_decode_kernel(SelectArgs a, char *__restrict__ ws)
{
    const int img = blockIdx.y;
    decode_rows(a, img, ws_cand(ws, img), blockIdx.x * kDecodeThreads + threadIdx.x, gridDim.x * kDecodeThreads);
}

// proposals[b][j] = j < num[b] ? dets[b][keep[b][j]][0:4] / (h, w, h, w) : 0   (lib/layers.py:131-137)
__global__ __launch_bounds__(256) void proposal_gather_kernel(const float *__restrict__ dets, int K, int det_stride,
                                                              const long long *__restrict__ keep, int keep_stride,
                                                              const int *__restrict__ num, int P, float nh, float nw,
                                                              float *__restrict__ out)
{
    const int img = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < num[img] && j < keep_stride) {
        const long long k = keep[(size_t)img * keep_stride + j];
        if (k >= 0 && k < K) {
            const float *d = dets + ((size_t)img * K + k) * det_stride;
            r = make_float4(d[0] / nh, d[1] / nw, d[2] / nh, d[3] / nw);
        }
    }
    *reinterpret_cast<float4 *>(out + ((size_t)img * P + j) * 4) = r;
}

}  // namespace

extern "C" {

int fi_proposal_candidates(const float *probs, int prob_stride, int prob_offset, const float *deltas,
                           const float *anchors, const float *extra, int batch, int num_anchors, int num_extra,
                           int pre_nms, const float *bbox_std_host, float window_h, float window_w, float *dets,
                           fi_stream_t stream)
{
    FI_REQUIRE(batch >= 0 && num_anchors >= 0 && num_extra >= 0, "sizes must be non-negative");
    FI_REQUIRE(pre_nms >= 1 && pre_nms <= kSortCap, "1 <= pre_nms <= 8192");
    FI_REQUIRE((long)num_anchors + num_extra >= pre_nms, "fewer candidates than pre_nms");
    FI_REQUIRE(prob_stride >= 1 && prob_offset >= 0 && prob_offset < prob_stride, "bad score stride / offset");
    FI_REQUIRE(bbox_std_host && dets && (num_anchors == 0 || (probs && deltas && anchors)), "null pointer");
    FI_REQUIRE(num_extra == 0 || extra, "null extra candidates");
    FI_REQUIRE(((uintptr_t)anchors | (uintptr_t)deltas) % 16 == 0, "anchors / deltas must be 16-byte aligned");
    if (batch == 0) return FI_OK;
    SelectArgs a;
    a.probs = probs; a.deltas = deltas; a.anchors = anchors; a.extra = extra; a.dets = dets;
    a.A = num_anchors; a.E = num_extra; a.K = pre_nms; a.prob_stride = prob_stride; a.prob_off = prob_offset;
    a.std0 = bbox_std_host[0]; a.std1 = bbox_std_host[1]; a.std2 = bbox_std_host[2]; a.std3 = bbox_std_host[3];
    a.win_h = window_h; a.win_w = window_w;
    static const int lds_sort = getenv("FI_PROPOSAL_LDS_SORT") != nullptr;
    a.lds_sort = lds_sort;
    const size_t lds = sizeof(u64) * kSortCap;
    // the attribute belongs to the (function, device) pair: once per device of the process, from any thread
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    FI_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        FI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(proposal_select_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    fi::ProfScope prof(FI_K_PROPOSAL_SELECT, (hipStream_t)stream);
    hipLaunchKernelGGL(proposal_select_kernel, dim3(batch), dim3(kSelThreads), lds, (hipStream_t)stream, a);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

size_t fi_proposal_workspace_bytes(int batch) { return batch <= 0 ? 0 : (size_t)batch * kWsPerImage; }

int fi_proposal_candidates_ws(const float *probs, int prob_stride, int prob_offset, const float *deltas,
                              const float *anchors, const float *extra, int batch, int num_anchors, int num_extra,
                              int pre_nms, const float *bbox_std_host, float window_h, float window_w, float *dets,
                              void *workspace, size_t workspace_bytes, fi_stream_t stream)
{
    FI_REQUIRE(batch >= 0 && num_anchors >= 0 && num_extra >= 0, "sizes must be non-negative");
    FI_REQUIRE(pre_nms >= 1 && pre_nms <= kSortCap, "1 <= pre_nms <= 8192");
    FI_REQUIRE((long)num_anchors + num_extra >= pre_nms, "fewer candidates than pre_nms");
    FI_REQUIRE(prob_stride >= 1 && prob_offset >= 0 && prob_offset < prob_stride, "bad score stride / offset");
    FI_REQUIRE(bbox_std_host && dets && (num_anchors == 0 || (probs && deltas && anchors)), "null pointer");
    FI_REQUIRE(num_extra == 0 || extra, "null extra candidates");
    FI_REQUIRE(((uintptr_t)anchors | (uintptr_t)deltas) % 16 == 0, "anchors / deltas must be 16-byte aligned");
    if (batch == 0) return FI_OK;
    FI_REQUIRE(workspace && (uintptr_t)workspace % 16 == 0 && workspace_bytes >= fi_proposal_workspace_bytes(batch),
               "workspace: fi_proposal_workspace_bytes(batch) bytes, 16-byte aligned");
    SelectArgs a;
    a.probs = probs; a.deltas = deltas; a.anchors = anchors; a.extra = extra; a.dets = dets;
    a.A = num_anchors; a.E = num_extra; a.K = pre_nms; a.prob_stride = prob_stride; a.prob_off = prob_offset;
    a.std0 = bbox_std_host[0]; a.std1 = bbox_std_host[1]; a.std2 = bbox_std_host[2]; a.std3 = bbox_std_host[3];
    a.win_h = window_h; a.win_w = window_w;
    static const int lds_sort = getenv("FI_PROPOSAL_LDS_SORT") != nullptr;
    a.lds_sort = lds_sort;
    hipStream_t st = (hipStream_t)stream;
    char *ws = static_cast<char *>(workspace);
    const size_t lds = sizeof(u64) * kSortCap;
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    FI_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        FI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(proposal_sort_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    fi::ProfScope prof(FI_K_PROPOSAL_SELECT, st);
    // histogram + state of every image: one strided clear (the key arrays need none)
    FI_HIP_CHECK(hipMemset2DAsync(ws, kWsPerImage, 0, kWsHist + kWsState, (size_t)batch, st));
    const dim3 wide(kHistWgs, batch);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(proposal_hist_kernel, wide, dim3(kHistThreads), 0, st, a, ws, pass);
        hipLaunchKernelGGL(proposal_digit_kernel, dim3(batch), dim3(kSelThreads), 0, st, a, ws, pass);
    }
    hipLaunchKernelGGL(proposal_compact_kernel, dim3(kCompactWgs, batch), dim3(kHistThreads), 0, st, a, ws);
    hipLaunchKernelGGL(proposal_sort_kernel, dim3(batch), dim3(kSelThreads), lds, st, a, ws);
    hipLaunchKernelGGL(proposal_decode_kernel, dim3((pre_nms + kDecodeThreads - 1) / kDecodeThreads, batch),
                       dim3(kDecodeThreads), 0, st, a, ws);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_proposal_gather(const float *dets, int pre_nms, int det_stride, const int64_t *keep, int keep_stride,
                       const int32_t *num, int batch, int proposal_count, float norm_h, float norm_w,
                       float *proposals, fi_stream_t stream)
{
    FI_REQUIRE(batch >= 0 && pre_nms >= 0 && proposal_count >= 0 && det_stride >= 4 && keep_stride >= 0,
               "bad sizes");
    FI_REQUIRE(batch == 0 || proposal_count == 0 || (dets && keep && num && proposals), "null pointer");
    FI_REQUIRE((uintptr_t)proposals % 16 == 0, "proposals must be 16-byte aligned");
    if (batch == 0 || proposal_count == 0) return FI_OK;
    fi::ProfScope prof(FI_K_PROPOSAL_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(proposal_gather_kernel, dim3((proposal_count + 255) / 256, batch), dim3(256), 0,
                       (hipStream_t)stream, dets, pre_nms, det_stride, (const long long *)keep, keep_stride, num,
                       proposal_count, norm_h, norm_w, proposals);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
