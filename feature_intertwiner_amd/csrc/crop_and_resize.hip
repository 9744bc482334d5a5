// crop_and_resize.hip -- RoIAlign (single bilinear tap per bin), forward/backward,
// single-level and pyramid (all FPN levels in one launch) forms, for gfx950.
//
// Arithmetic specification: lib/roi_align/src/crop_and_resize.c:31-110 (forward)
// and :190-251 (backward) of the reference -- see oracle/fi_oracle.c, which this
// file is tested against bit for bit (bin assignment) / exactly (forward values).
//
// Design (not a translation of the reference's one-thread-per-output kernel):
//   * one 256-thread workgroup = one RoI x one channel chunk; the RoI's sampling
//     table (tap rows/cols, lerp weights, in-range flags) is computed ONCE per
//     workgroup by crop_h + crop_w lanes and parked in LDS, instead of being
//     re-derived (2 divisions, 10 flops) for each of the C*ch*cw outputs;
//   * channel chunks are the fast grid index and their count is a multiple of 8
//     where possible, so that (with the observed block -> XCD = block % 8
//     round-robin) each XCD's private L2 only ever sees one slice of the channel
//     planes, and all RoIs that overlap in space share those lines in one L2;
//   * output addresses are contiguous in the flat (channel, y, x) index, so every
//     wavefront store is a fully coalesced 256-byte write; the 4 gathers per output
//     are issued for 4 outputs at a time (16 loads in flight per lane);
//   * crop sizes 7, 14, 28 (every size the model uses) are compile-time so the
//     flat-index -> (c, y, x) split is multiply-shift, not integer division;
//   * every output element is written (zeros / extrapolation value included), so
//     the reference's separate 25..100 MB zero-fill pass disappears.
#include <stdint.h>
#include <stdlib.h>

#include "fi_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCrop = 64;   // LDS table entries per axis
constexpr int kMaxLevels = 8;
constexpr int kTileFloats = 8192;   // 32 KB LDS accumulation tile of the backward kernel

// <2 x float> that may sit at any 4-byte boundary: lowers to one global_load_dwordx2
typedef float pair_f32 __attribute__((ext_vector_type(2)));
typedef pair_f32 pair_f32_a4 __attribute__((aligned(4)));

struct Tap {
    int i0;      // floorf(coord)
    int i1;      // ceilf(coord)
    float frac;  // coord - i0
    int valid;   // 0 when coord is outside [0, extent-1]
};

struct LevelSet {
    const float *img[kMaxLevels];
    int H[kMaxLevels];
    int W[kMaxLevels];
    int n;
};

struct LevelSetMut {
    float *img[kMaxLevels];
    int H[kMaxLevels];
    int W[kMaxLevels];
    int n;
};

// Sampling coordinate k of one axis.  Mirrors the operation order of the
// reference exactly (each * / + rounded to fp32; the crop == 1 case goes through
// double because of the 0.5 literal) -- oracle: orc_axis_taps().
// c(k) = base + (float)k * step: the two numbers the coordinate of every bin of one axis is made of.  crop == 1: the
// reference's double-precision midpoint as base and a zero step (base + 0 * 0 = base up to the sign of zero, which no
// result depends on: the validity tests, floorf / ceilf and 1 - frac are the same for +0 and -0).
__device__ __forceinline__ void axis_coeff(float lo, float hi, int extent, int crop, float *base, float *step)
{
    const float span = (float)(extent - 1);
    if (crop > 1) {
        const float d = hi - lo;
        const float m = d * span;
        *step = m / (float)(crop - 1);
        *base = lo * span;
    } else {
        const float s = lo + hi;
        const double dc = 0.5 * (double)s * (double)(extent - 1);
        *base = (float)dc;
        *step = 0.0f;
    }
}

__device__ __forceinline__ Tap tap_at(float c, int extent)
{
    const float span = (float)(extent - 1);
    Tap t;
    if (c < 0.0f || c > span) {
        t.valid = 0;
        t.i0 = 0;
        t.i1 = 0;
        t.frac = 0.0f;
    } else {
        t.valid = 1;
        t.i0 = (int)floorf(c);
        t.i1 = (int)ceilf(c);
        t.frac = c - (float)t.i0;  // via the int, as the reference: keeps -0.0 - 0 == -0.0
    }
    return t;
}

__device__ __forceinline__ Tap make_tap(float lo, float hi, int extent, int crop, int k)
{
    float base, step;
    axis_coeff(lo, hi, extent, crop, &base, &step);
    if (crop > 1) {
        const float off = (float)k * step;
        return tap_at(base + off, extent);
    }
    return tap_at(base, extent);
}

// Resolve the (uniform) per-workgroup box header.  Returns false when the box has
// no valid source (bad image index or level): the caller writes zeros.
struct BoxHeader {
    int lvl, H, W, img;
    float y1, x1, y2, x2;
};

template <typename LS>
__device__ __forceinline__ bool load_box(const LS &ls, const float *__restrict__ boxes,
                                         const int *__restrict__ box_ind,
                                         const int *__restrict__ level, int box, int batch,
                                         BoxHeader &h)
{
    h.lvl = level ? (level[box] - 2) : 0;
    h.img = box_ind[box];
    const bool ok = (h.lvl >= 0) && (h.lvl < ls.n) && (h.img >= 0) && (h.img < batch);
    if (!ok) return false;
    h.H = ls.H[h.lvl];
    h.W = ls.W[h.lvl];
    const float *b = boxes + 4 * (size_t)box;  // scalar loads: no alignment demand on callers
    h.y1 = b[0];
    h.x1 = b[1];
    h.y2 = b[2];
    h.x2 = b[3];
    return true;
}

template <int CH, int CW>
__global__ __launch_bounds__(kThreads) void crop_fwd_kernel(
    LevelSet ls, const float *__restrict__ boxes, const int *__restrict__ box_ind,
    const int *__restrict__ level, int num_boxes, int batch, int depth, int crop_h_rt,
    int crop_w_rt, float extrap, int chan_per_block, int chunks, float *__restrict__ crops,
    int *__restrict__ status)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    __shared__ Tap s_ty[kMaxCrop];
    __shared__ Tap s_tx[kMaxCrop];

    const int tid = threadIdx.x;
    // XCD-aware, chunk-major order (see pick_fwd_chunks): block b runs on XCD b % 8; each XCD works through
    // its channel chunks one after the other, each for ALL boxes, so that the few planes of one chunk stay
    // resident in that XCD's 4 MB L2 while every RoI that overlaps them is gathered.
    const int xcd = blockIdx.x & 7;
    const int seq = blockIdx.x >> 3;
    const int chunk = (seq / num_boxes) * 8 + xcd;
    const int box = seq % num_boxes;
    if (chunk >= chunks) return;
    const int c_begin = chunk * chan_per_block;
    const int c_count = min(chan_per_block, depth - c_begin);
    const int total = c_count * bins;
    float *__restrict__ out = crops + ((size_t)box * depth + c_begin) * bins;

    BoxHeader h;
    if (!load_box(ls, boxes, box_ind, level, box, batch, h)) {
        if (level && level[box] == -1) return;      // level -1: a row of static capacity that nobody reads -- not even written
        for (int i = tid; i < total; i += kThreads) out[i] = 0.0f;
        if (status && tid == 0 && chunk == 0) atomicOr(status, 1);
        return;
    }
    if (tid < crop_h) s_ty[tid] = make_tap(h.y1, h.y2, h.H, crop_h, tid);
    if (tid >= 64 && tid < 64 + crop_w) s_tx[tid - 64] = make_tap(h.x1, h.x2, h.W, crop_w, tid - 64);
    __syncthreads();

    const size_t plane = (size_t)h.H * (size_t)h.W;
    const float *__restrict__ src = ls.img[h.lvl] + ((size_t)h.img * depth + c_begin) * plane;
    const int W = h.W;

    // The left/right taps of a bin are adjacent floats (x1 == x0 or x0 + 1), so each row is
    // fetched with ONE 8-byte load at column min(x0, W-2) instead of two 4-byte gathers: the
    // kernel is bound by L1 line look-ups per wavefront instruction, not by bytes, and this
    // halves them.  (W == 1 maps use the scalar path.)
    const bool pair_ok = (W >= 2);
    constexpr int UNROLL = 4;
    for (int base = tid; base < total; base += kThreads * UNROLL) {
        float2 top2[UNROLL], bot2[UNROLL];
        float fx[UNROLL], fy[UNROLL];
        int sel[UNROLL];   // bit0: left tap is .y, bit1: right tap is .y
        int ok[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int idx = base + u * kThreads;
            ok[u] = 0;
            sel[u] = 0;
            top2[u] = make_float2(0.0f, 0.0f);
            bot2[u] = make_float2(0.0f, 0.0f);
            fx[u] = fy[u] = 0.0f;
            if (idx < total) {
                const int c = idx / bins;
                const int bin = idx - c * bins;
                const int y = bin / crop_w;
                const int x = bin - y * crop_w;
                const Tap ty = s_ty[y];
                const Tap tx = s_tx[x];
                ok[u] = (ty.valid & tx.valid) ? 1 : 2;
                if (ok[u] == 1) {
                    const float *__restrict__ p = src + (size_t)c * plane;
                    const int r0 = ty.i0 * W, r1 = ty.i1 * W;
                    if (pair_ok) {
                        const int xb = min(tx.i0, W - 2);
                        sel[u] = (tx.i0 != xb ? 1 : 0) | (tx.i1 != xb ? 2 : 0);
                        const float *pt = p + r0 + xb;
                        const float *pb = p + r1 + xb;
                        // 4-byte aligned 8-byte loads (global memory allows unaligned dwordx2)
                        const pair_f32 vt = *reinterpret_cast<const pair_f32_a4 *>(pt);
                        const pair_f32 vb = *reinterpret_cast<const pair_f32_a4 *>(pb);
                        top2[u] = make_float2(vt.x, vt.y);
                        bot2[u] = make_float2(vb.x, vb.y);
                    } else {
                        top2[u].x = p[r0 + tx.i0];
                        top2[u].y = p[r0 + tx.i1];
                        bot2[u].x = p[r1 + tx.i0];
                        bot2[u].y = p[r1 + tx.i1];
                        sel[u] = 2;
                    }
                    fx[u] = tx.frac;
                    fy[u] = ty.frac;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int idx = base + u * kThreads;
            if (ok[u] == 1) {
                const float tl = (sel[u] & 1) ? top2[u].y : top2[u].x;
                const float tr = (sel[u] & 2) ? top2[u].y : top2[u].x;
                const float bl = (sel[u] & 1) ? bot2[u].y : bot2[u].x;
                const float br = (sel[u] & 2) ? bot2[u].y : bot2[u].x;
                const float dt = tr - tl;
                const float top = tl + dt * fx[u];
                const float db = br - bl;
                const float bot = bl + db * fx[u];
                const float dv = bot - top;
                out[idx] = top + dv * fy[u];
            } else if (ok[u] == 2) {
                out[idx] = extrap;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// Single-channel maps (the mask-target crop of lib/layers.py:301-322: every positive RoI's 28 x 28 window of its GT
// mini-mask, depth 1): a thread owns ONE bin of one RoI -- no table, no barrier, no per-workgroup header chain; the
// RoI x channel-chunk workgroups of crop_fwd_kernel spend their time on the header and the barrier when there is one
// plane (2048 workgroups of 784 outputs: 94 us alone on the chip, round 5).  Same make_tap, same lerp order:
// bit-identical to crop_fwd_kernel.
// -------------------------------------------------------------------------------------
template <int CH, int CW>
__global__ __launch_bounds__(kThreads) void crop_fwd_c1_kernel(LevelSet ls, const float *__restrict__ boxes,
                                                               const int *__restrict__ box_ind, const int *__restrict__ level,
                                                               int num_boxes, int batch, int crop_h_rt, int crop_w_rt,
                                                               float extrap, float *__restrict__ crops, int *__restrict__ status)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= (long)num_boxes * bins) return;
    const int box = (int)(idx / bins);
    const int bin = (int)(idx - (long)box * bins);
    const int y = bin / crop_w, x = bin - y * crop_w;
    BoxHeader h;
    if (!load_box(ls, boxes, box_ind, level, box, batch, h)) {
        if (level && level[box] == -1) return;
        crops[idx] = 0.0f;
        if (status && bin == 0) atomicOr(status, 1);
        return;
    }
    const Tap ty = make_tap(h.y1, h.y2, h.H, crop_h, y);
    const Tap tx = make_tap(h.x1, h.x2, h.W, crop_w, x);
    if (!(ty.valid & tx.valid)) {
        crops[idx] = extrap;
        return;
    }
    const float *__restrict__ p = ls.img[h.lvl] + (size_t)h.img * h.H * h.W;
    const int r0 = ty.i0 * h.W, r1 = ty.i1 * h.W;
    const float tl = p[r0 + tx.i0], tr = p[r0 + tx.i1], bl = p[r1 + tx.i0], br = p[r1 + tx.i1];
    const float dt = tr - tl;
    const float top = tl + dt * tx.frac;
    const float db = br - bl;
    const float bot = bl + db * tx.frac;
    const float dv = bot - top;
    crops[idx] = top + dv * ty.frac;
}

// -------------------------------------------------------------------------------------
// Table-free forward for small crops (7 x 7): a thread owns ONE bin of one RoI for CG consecutive channels.
// No LDS, no barrier, no per-workgroup header chain: the flat thread index runs over (channel group, RoI, bin),
// the two sampling coordinates are computed per thread (two make_tap() -- ~70 VALU amortised over CG outputs), the
// 2 x CG gathers of the bin are all in flight before the first lerp, and the stores of one channel are the 49
// consecutive floats of a (RoI, channel) crop.  XCD-aware like crop_fwd_kernel: workgroup b runs on XCD b % 8 and
// XCD x owns the channel groups g with g % 8 == x, each for ALL RoIs, so that a group's planes stay in one L2
// (without this order the same kernel takes 63-69 us instead of 29.5 at 512 x 256 x 7 x 7: scripts/micro/
// crop_nchw_variants.hip).  Bit-identical to crop_fwd_kernel (same make_tap, same lerp order).
// -------------------------------------------------------------------------------------
template <int CH, int CW, int CG>
__global__ __launch_bounds__(kThreads) void crop_fwd_flat_kernel(
    LevelSet ls, const float *__restrict__ boxes, const int *__restrict__ box_ind,
    const int *__restrict__ level, int num_boxes, int batch, int depth, float extrap, int groups_per_xcd,
    float *__restrict__ crops, int *__restrict__ status)
{
    constexpr int bins = CH * CW;
    const int xcd = blockIdx.x & 7;
    const long seq = (long)(blockIdx.x >> 3) * kThreads + threadIdx.x;      // index within this XCD's items
    const long per_group = (long)num_boxes * bins;
    const int gi = (int)(seq / per_group);
    if (gi >= groups_per_xcd) return;
    const long rem = seq - (long)gi * per_group;
    const int box = (int)(rem / bins);
    const int bin = (int)(rem - (long)box * bins);
    const int g = gi * 8 + xcd;
    const int y = bin / CW, x = bin - y * CW;
    float *__restrict__ out = crops + ((size_t)box * depth + (size_t)g * CG) * bins + bin;

    const int lvl = level ? (level[box] - 2) : 0;
    const int img = box_ind[box];
    // per-lane level record (lanes of a wavefront may sit on different RoIs): a select chain over the (few) levels
    const float *__restrict__ base = nullptr;
    int H = 1, W = 1;
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l)
        if (l < ls.n && l == lvl) {
            base = ls.img[l];
            H = ls.H[l];
            W = ls.W[l];
        }
    if (base == nullptr || img < 0 || img >= batch) {
#pragma unroll
        for (int k = 0; k < CG; ++k) out[(size_t)k * bins] = 0.0f;
        if (status && bin == 0 && g == 0) atomicOr(status, 1);
        return;
    }
    const float *b = boxes + 4 * (size_t)box;          // scalar-width loads: no alignment demand on callers
    const Tap ty = make_tap(b[0], b[2], H, CH, y);
    const Tap tx = make_tap(b[1], b[3], W, CW, x);
    if (!(ty.valid & tx.valid)) {
#pragma unroll
        for (int k = 0; k < CG; ++k) out[(size_t)k * bins] = extrap;
        return;
    }
    const size_t plane = (size_t)H * W;
    const float *__restrict__ p0 = base + ((size_t)img * depth + (size_t)g * CG) * plane;
    if (W >= 2) {
        // left / right taps are adjacent floats: ONE (4-byte aligned) 8-byte load per row at column min(x0, W - 2)
        const int xb = min(tx.i0, W - 2);
        const bool sl = tx.i0 != xb, sr = tx.i1 != xb;
        const float *__restrict__ pt = p0 + (size_t)ty.i0 * W + xb;
        const float *__restrict__ pb = p0 + (size_t)ty.i1 * W + xb;
        pair_f32 vt[CG], vb[CG];
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            vt[k] = *reinterpret_cast<const pair_f32_a4 *>(pt + (size_t)k * plane);
            vb[k] = *reinterpret_cast<const pair_f32_a4 *>(pb + (size_t)k * plane);
        }
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            const float tl = sl ? vt[k].y : vt[k].x;
            const float tr = sr ? vt[k].y : vt[k].x;
            const float bl = sl ? vb[k].y : vb[k].x;
            const float br = sr ? vb[k].y : vb[k].x;
            const float dt = tr - tl;
            const float top = tl + dt * tx.frac;
            const float db = br - bl;
            const float bot = bl + db * tx.frac;
            const float dv = bot - top;
            out[(size_t)k * bins] = top + dv * ty.frac;
        }
    } else {
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            const float *__restrict__ p = p0 + (size_t)k * plane;
            const float tl = p[(size_t)ty.i0 * W + tx.i0], tr = p[(size_t)ty.i0 * W + tx.i1];
            const float bl = p[(size_t)ty.i1 * W + tx.i0], br = p[(size_t)ty.i1 * W + tx.i1];
            const float dt = tr - tl;
            const float top = tl + dt * tx.frac;
            const float db = br - bl;
            const float bot = bl + db * tx.frac;
            const float dv = bot - top;
            out[(size_t)k * bins] = top + dv * ty.frac;
        }
    }
}

template <int CH, int CW>
__global__ __launch_bounds__(kThreads) void crop_bwd_kernel(
    LevelSetMut ls, const float *__restrict__ grads, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, const int *__restrict__ level, int num_boxes, int batch,
    int depth, int crop_h_rt, int crop_w_rt, int chan_per_block, int chunks)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    __shared__ Tap s_ty[kMaxCrop];
    __shared__ Tap s_tx[kMaxCrop];

    const int tid = threadIdx.x;
    const int box = blockIdx.x / chunks;
    const int chunk = blockIdx.x - box * chunks;
    const int c_begin = chunk * chan_per_block;
    const int c_count = min(chan_per_block, depth - c_begin);
    const int total = c_count * bins;

    BoxHeader h;
    if (!load_box(ls, boxes, box_ind, level, box, batch, h)) return;
    if (tid < crop_h) s_ty[tid] = make_tap(h.y1, h.y2, h.H, crop_h, tid);
    if (tid >= 64 && tid < 64 + crop_w) s_tx[tid - 64] = make_tap(h.x1, h.x2, h.W, crop_w, tid - 64);
    __syncthreads();

    const size_t plane = (size_t)h.H * (size_t)h.W;
    float *__restrict__ dst = ls.img[h.lvl] + ((size_t)h.img * depth + c_begin) * plane;
    const float *__restrict__ g = grads + ((size_t)box * depth + c_begin) * bins;
    const int W = h.W;

    // ---- LDS path: when the RoI's footprint on the map is small, the 4*bins scattered adds of a
    // channel are first combined in an LDS tile (ds_add_f32) and each touched cell is then sent to
    // memory ONCE, as row-contiguous atomics (a few L2 line operations per row instead of one per
    // tap).  Falls through to direct global atomics for large footprints.
    __shared__ float s_tile[kTileFloats];
    int ymin = 1 << 30, ymax = -1, xmin = 1 << 30, xmax = -1;
    for (int i = 0; i < crop_h; ++i) {
        const Tap t = s_ty[i];
        if (t.valid) { ymin = min(ymin, t.i0); ymax = max(ymax, t.i1); }
    }
    for (int i = 0; i < crop_w; ++i) {
        const Tap t = s_tx[i];
        if (t.valid) { xmin = min(xmin, t.i0); xmax = max(xmax, t.i1); }
    }
    if (ymax < 0 || xmax < 0) return;                 // nothing of this RoI is inside the map
    const int fh = ymax - ymin + 1, fw = xmax - xmin + 1;
    const int cells = fh * fw;
    if (cells <= kTileFloats && cells * 2 < 4 * bins) {
        const int G = min(c_count, kTileFloats / cells);   // channels per LDS pass
        for (int cg = 0; cg < c_count; cg += G) {
            const int gc = min(G, c_count - cg);
            for (int i = tid; i < gc * cells; i += kThreads) s_tile[i] = 0.0f;
            __syncthreads();
            for (int idx = tid; idx < gc * bins; idx += kThreads) {
                const int c = idx / bins;
                const int bin = idx - c * bins;
                const int y = bin / crop_w;
                const int x = bin - y * crop_w;
                const Tap ty = s_ty[y];
                const Tap tx = s_tx[x];
                if (!(ty.valid & tx.valid)) continue;
                const float gv = g[(size_t)(cg + c) * bins + bin];
                float *t = s_tile + c * cells;
                const float wy0 = 1.0f - ty.frac;
                const float wx0 = 1.0f - tx.frac;
                const float gtop = wy0 * gv;
                const float gbot = ty.frac * gv;
                const int r0 = (ty.i0 - ymin) * fw, r1 = (ty.i1 - ymin) * fw;
                const int c0 = tx.i0 - xmin, c1 = tx.i1 - xmin;
                atomicAdd(t + r0 + c0, wx0 * gtop);
                atomicAdd(t + r0 + c1, tx.frac * gtop);
                atomicAdd(t + r1 + c0, wx0 * gbot);
                atomicAdd(t + r1 + c1, tx.frac * gbot);
            }
            __syncthreads();
            for (int i = tid; i < gc * cells; i += kThreads) {
                const float v = s_tile[i];
                if (v != 0.0f) {
                    const int c = i / cells;
                    const int rem = i - c * cells;
                    const int r = rem / fw;
                    const int col = rem - r * fw;
                    atomicAdd(dst + (size_t)(cg + c) * plane + (size_t)(ymin + r) * W + (xmin + col), v);
                }
            }
            __syncthreads();
        }
        return;
    }

    for (int idx = tid; idx < total; idx += kThreads) {
        const int c = idx / bins;
        const int bin = idx - c * bins;
        const int y = bin / crop_w;
        const int x = bin - y * crop_w;
        const Tap ty = s_ty[y];
        const Tap tx = s_tx[x];
        if (!(ty.valid & tx.valid)) continue;
        const float gv = g[idx];
        float *__restrict__ p = dst + (size_t)c * plane;
        // reference order: dtop = (1-ly)*g; TL += (1-lx)*dtop; TR += lx*dtop;
        //                  dbot = ly*g;     BL += (1-lx)*dbot; BR += lx*dbot
        const float wy0 = 1.0f - ty.frac;
        const float wx0 = 1.0f - tx.frac;
        const float gtop = wy0 * gv;
        const float gbot = ty.frac * gv;
        const int r0 = ty.i0 * W, r1 = ty.i1 * W;
        atomicAdd(p + r0 + tx.i0, wx0 * gtop);
        atomicAdd(p + r0 + tx.i1, tx.frac * gtop);
        atomicAdd(p + r1 + tx.i0, wx0 * gbot);
        atomicAdd(p + r1 + tx.i1, tx.frac * gbot);
    }
}

// -------------------------------------------------------------------------------------
// NCHW backward without global atomics and without a memset pass: TILE-OWNER form.
// A workgroup owns one tile (kTileCells cells: 16 x 64, 32 x 32 or 64 x 16 by map width) of CG channel planes of one
// image.  It walks the boxes in chunks of 256 -- every thread tests one box against the tile and works out which range
// of its bins per axis can reach it (the sampling coordinate is monotone in the bin index; conservative, every listed
// bin is filtered again tap by tap) -- and the listed boxes are accumulated into an LDS copy of the tile with
// ds_add_f32: a wavefront takes one box, a lane one bin of its range, computes the two taps once and applies them to
// the CG channels (CG loads in flight; the bins of a (box, channel) are contiguous).  At the end the tile is written with plain
// row-contiguous stores -- which is also the zero fill of the cells nobody touched (the accumulate form reads the tile
// first instead of clearing it).  Against the scatter kernel above: no global_atomic_add_f32 (25.7 M of them at
// 512 x 256 x 7 x 7, executed memory side at ~54 G/s), no 134..268 MB memset, every output byte written once.
// The LDS accumulators are fp64 (see the kernel): every contribution is the reference's fp32 product, their sum is
// rounded once; the reference adds them serially in fp32, so the two differ by that rounding (tests: 2e-5).
// Workgroup order: the channel group is the fast index, so that (block -> XCD = block % 8) the tiles of one channel
// group -- which read the same gradient rows -- share an L2.
// -------------------------------------------------------------------------------------
constexpr int kTileCells = 1024;

struct TileGrid {
    int base[kMaxLevels + 1];   // first (image, tile) index of level l; base[n] = total
    int th[kMaxLevels], tw[kMaxLevels];
    int nty[kMaxLevels], ntx[kMaxLevels];
};

// One axis of one box against the tile span [t0, t0 + tn): the coefficients of c(k) = base + k * step and the range of
// bins [k_lo, k_lo + k_n) outside of which no tap (floor / ceil of c) can be a tile cell.  Conservative: c is monotone
// in k, the bounds carry a margin of one bin against the rounding of the division, and every listed bin is tested
// again tap by tap.  false: no bin of the axis reaches the tile (or the map).
__device__ __forceinline__ bool axis_range(float a, float b, int extent, int crop, int t0, int tn, float *base,
                                           float *step, int *k_lo, int *k_n)
{
    axis_coeff(a, b, extent, crop, base, step);
    const float span = (float)(extent - 1);
    const float c0 = *base;
    const float c1 = (crop > 1) ? *base + (float)(crop - 1) * *step : *base;
    const float mn = fminf(c0, c1), mx = fmaxf(c0, c1);
    if (!(mx >= 0.0f) || !(mn <= span)) return false;                      // (also rejects NaN coordinates)
    const float lo_cell = (float)(t0 - 1), hi_cell = (float)(t0 + tn);    // taps of c reach cells floor(c), ceil(c)
    if (!(mx >= lo_cell) || !(mn <= hi_cell)) return false;
    int lo = 0, hi = crop - 1;
    if (*step != 0.0f) {
        const float t1 = (lo_cell - *base) / *step, t2 = (hi_cell - *base) / *step;
        const float fl = floorf(fminf(t1, t2)) - 1.0f, fh = ceilf(fmaxf(t1, t2)) + 1.0f;
        if (fl > (float)(crop - 1) || fh < 0.0f) return false;
        if (fl > 0.0f) lo = (int)fl;                                       // NaN / -inf: stays 0
        if (fh < (float)(crop - 1)) hi = (int)fh;                          // NaN / +inf: stays crop - 1
    }
    *k_lo = lo;
    *k_n = hi - lo + 1;
    return true;
}

template <int CH, int CW, int CG, bool ACC>
__global__ __launch_bounds__(kThreads) void crop_bwd_tiles_kernel(
    LevelSetMut ls, TileGrid tg, const float *__restrict__ grads, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, const int *__restrict__ level, int num_boxes, int batch,
    int depth, int crop_h_rt, int crop_w_rt, int ncg)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    // fp64 accumulators: ds_add_f64 runs at 7.7 lane-operations per clock per CU on gfx950, ds_add_f32 at 0.33 (it costs
    // ~3 clocks per ACTIVE LANE; scripts/micro/lds_atomic_rate.hip, profiles/r05_lds_atomic_rate.txt) -- and the sum of
    // the fp32 contributions is rounded once, at the end
    __shared__ double s_tile[CG * kTileCells];
    __shared__ float s_box[kThreads][4];        // base_y, step_y, base_x, step_x
    __shared__ int s_rng[kThreads];             // k_lo(y) | k_n(y) << 8 | k_lo(x) << 16 | k_n(x) << 24
    __shared__ int s_id[kThreads];
    __shared__ int s_wave_n[kThreads / 64];

    const int tid = threadIdx.x;
    const int cg = blockIdx.x % ncg;
    int t = blockIdx.x / ncg;
    int lvl = 0;
    while (lvl + 1 < ls.n && t >= tg.base[lvl + 1]) ++lvl;
    t -= tg.base[lvl];
    const int H = ls.H[lvl], W = ls.W[lvl];
    const int th = tg.th[lvl], tw = tg.tw[lvl];
    const int tiles = tg.nty[lvl] * tg.ntx[lvl];
    const int img = t / tiles;
    const int tile = t - img * tiles;
    const int row0 = (tile / tg.ntx[lvl]) * th;
    const int col0 = (tile % tg.ntx[lvl]) * tw;
    const int c_begin = cg * CG;
    const int c_count = min(CG, depth - c_begin);
    const size_t plane = (size_t)H * (size_t)W;
    float *__restrict__ dst = ls.img[lvl] + ((size_t)img * depth + c_begin) * plane;

    // ---- the tile starts as zeros (or as what the map holds: accumulate form)
    if (!ACC) {
        for (int i = tid * 2; i < CG * kTileCells; i += kThreads * 2)
            *reinterpret_cast<double2 *>(s_tile + i) = make_double2(0.0, 0.0);
    }
    for (int i = tid; ACC && i < CG * kTileCells; i += kThreads) {
        float v = 0.0f;
        if (ACC) {
            const int c = i / kTileCells;
            const int rem = i - c * kTileCells;
            const int r = rem / tw, col = rem - r * tw;
            if (c < c_count && row0 + r < H && col0 + col < W)
                v = dst[(size_t)c * plane + (size_t)(row0 + r) * W + (col0 + col)];
        }
        s_tile[i] = (double)v;
    }

    const int lane = tid & 63, wave = tid >> 6;
    for (int chunk = 0; chunk < num_boxes; chunk += kThreads) {
        // ---- which boxes of this chunk reach the tile, and with which bins (list in box order: ballot compaction)
        const int box = chunk + tid;
        bool hit = false;
        float by = 0, sy = 0, bx = 0, sx = 0;
        int ky = 0, ny = 0, kx = 0, nx = 0;
        if (box < num_boxes && box_ind[box] == img && (level ? (level[box] - 2) : 0) == lvl) {
            const float *b = boxes + 4 * (size_t)box;
            hit = axis_range(b[0], b[2], H, crop_h, row0, th, &by, &sy, &ky, &ny) &&
                  axis_range(b[1], b[3], W, crop_w, col0, tw, &bx, &sx, &kx, &nx);
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_wave_n[wave] = __popcll(m);
        __syncthreads();                      // (also: the tile is initialised / the previous chunk's list is consumed)
        int off = 0, n = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int k = s_wave_n[w];
            if (w < wave) off += k;
            n += k;
        }
        if (hit) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_box[slot][0] = by; s_box[slot][1] = sy; s_box[slot][2] = bx; s_box[slot][3] = sx;
            s_rng[slot] = ky | (ny << 8) | (kx << 16) | (nx << 24);
            s_id[slot] = box;
        }
        __syncthreads();
        // ---- a wavefront = one listed box, a lane = one of its bins in range: two taps, then CG channels
        for (int j = wave; j < n; j += kThreads / 64) {
            const int rng = s_rng[j];
            const int ky0 = rng & 255, nyj = (rng >> 8) & 255, kx0 = (rng >> 16) & 255, nxj = (rng >> 24) & 255;
            const float byj = s_box[j][0], syj = s_box[j][1], bxj = s_box[j][2], sxj = s_box[j][3];
            const float *__restrict__ gbox = grads + ((size_t)s_id[j] * depth + c_begin) * bins;
            for (int e = lane; e < nyj * nxj; e += 64) {
                const int yy = e / nxj;
                const int y = ky0 + yy;
                const int x = kx0 + (e - yy * nxj);
                const Tap ty = tap_at(crop_h > 1 ? byj + (float)y * syj : byj, H);
                const Tap tx = tap_at(crop_w > 1 ? bxj + (float)x * sxj : bxj, W);
                if (!(ty.valid & tx.valid)) continue;
                const unsigned r0 = (unsigned)(ty.i0 - row0), r1 = (unsigned)(ty.i1 - row0);
                const unsigned c0 = (unsigned)(tx.i0 - col0), c1 = (unsigned)(tx.i1 - col0);
                const bool in_r0 = r0 < (unsigned)th, in_r1 = r1 < (unsigned)th;
                const bool in_c0 = c0 < (unsigned)tw, in_c1 = c1 < (unsigned)tw;
                if (!((in_r0 | in_r1) & (in_c0 | in_c1))) continue;
                const float *__restrict__ g = gbox + (y * crop_w + x);
                float gv[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) gv[c] = (c < c_count) ? g[(size_t)c * bins] : 0.0f;
                // reference order: dtop = (1-ly)*g; TL += (1-lx)*dtop; TR += lx*dtop; dbot = ly*g; BL, BR likewise
                const float wy0 = 1.0f - ty.frac, wx0 = 1.0f - tx.frac;
                const int a00 = (int)(r0 * tw + c0), a01 = (int)(r0 * tw + c1);
                const int a10 = (int)(r1 * tw + c0), a11 = (int)(r1 * tw + c1);
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    double *tl = s_tile + c * kTileCells;
                    const float gtop = wy0 * gv[c];
                    const float gbot = ty.frac * gv[c];
                    if (in_r0 & in_c0) atomicAdd(tl + a00, (double)(wx0 * gtop));
                    if (in_r0 & in_c1) atomicAdd(tl + a01, (double)(tx.frac * gtop));
                    if (in_r1 & in_c0) atomicAdd(tl + a10, (double)(wx0 * gbot));
                    if (in_r1 & in_c1) atomicAdd(tl + a11, (double)(tx.frac * gbot));
                }
            }
        }
    }
    __syncthreads();
    // ---- the tile leaves with plain stores (rows of tw consecutive floats; 16 bytes per lane where the rows allow)
    const int rows = min(th, H - row0), cols = min(tw, W - col0);
    if ((W & 3) == 0 && (cols & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        for (int i = tid * 4; i < c_count * kTileCells; i += kThreads * 4) {
            const int c = i / kTileCells;
            const int rem = i - c * kTileCells;
            const int r = rem / tw, col = rem - r * tw;
            if (r < rows && col < cols) {
                const double2 a = *reinterpret_cast<const double2 *>(s_tile + i);
                const double2 b = *reinterpret_cast<const double2 *>(s_tile + i + 2);
                *reinterpret_cast<float4 *>(dst + (size_t)c * plane + (size_t)(row0 + r) * W + (col0 + col)) =
                    make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
            }
        }
        return;
    }
    for (int i = tid; i < c_count * kTileCells; i += kThreads) {
        const int c = i / kTileCells;
        const int rem = i - c * kTileCells;
        const int r = rem / tw, col = rem - r * tw;
        if (r < rows && col < cols) dst[(size_t)c * plane + (size_t)(row0 + r) * W + (col0 + col)] = (float)s_tile[i];
    }
}

// -------------------------------------------------------------------------------------
// Channels-last maps ([batch][H][W][depth]).  In NCHW every (bin, channel) sample lives in its own
// cache line, so the gather is bound by L1 line look-ups and outstanding misses (PMC: 64-byte
// sectors fetched for 8 useful bytes).  With the channel axis innermost, the `depth` values of one
// tap are contiguous: a wavefront instruction reads 64 channels of one tap = 256 contiguous bytes,
// every fetched line is fully used, and the backward pass's atomics become line-wide as well.
// The crop output stays [num_boxes][depth][ch][cw] (what the heads consume): lanes run over
// channels, results are transposed through an LDS tile [channels][bins] (odd pitch: conflict
// free) and written back as one contiguous block.  Arithmetic is identical to the NCHW kernels.
// -------------------------------------------------------------------------------------
template <int V> struct VecF;
template <> struct VecF<1> { typedef float type; };
template <> struct VecF<4> { typedef float4 type; };
__device__ __forceinline__ float vget(const float &v, int) { return v; }
__device__ __forceinline__ float vget(const float4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

// V = channels per lane: 4 (16-byte loads; depth % 4 == 0 and 16-byte aligned maps) or 1.  A lane
// keeps UNROLL bins x 4 taps in flight; with V = 4 that is 256 bytes per lane, which is what the
// gather needs to cover the L2/HBM latency at the 3..6 workgroups per CU the LDS tile allows.
template <int CH, int CW, int V>
__global__ __launch_bounds__(kThreads) void crop_fwd_cl_kernel(
    LevelSet ls, const float *__restrict__ boxes, const int *__restrict__ box_ind,
    const int *__restrict__ level, int num_boxes, int batch, int depth, int crop_h_rt,
    int crop_w_rt, float extrap, int cpb, int chunks, float *__restrict__ crops)
{
    typedef typename VecF<V>::type vec_t;
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    const int pitch = bins | 1;
    extern __shared__ float s_dyn[];          // [cpb][pitch]
    __shared__ Tap s_ty[kMaxCrop];
    __shared__ Tap s_tx[kMaxCrop];

    const int tid = threadIdx.x;
    const int box = blockIdx.x / chunks;
    const int chunk = blockIdx.x - box * chunks;
    const int c_begin = chunk * cpb;
    const int c_count = min(cpb, depth - c_begin);
    const int total = c_count * bins;
    float *__restrict__ out = crops + ((size_t)box * depth + c_begin) * bins;

    BoxHeader h;
    if (!load_box(ls, boxes, box_ind, level, box, batch, h)) {
        if (level && level[box] == -1) return;      // level -1: a row of static capacity that nobody reads -- not even written
        for (int i = tid; i < total; i += kThreads) out[i] = 0.0f;
        return;
    }
    if (tid < crop_h) s_ty[tid] = make_tap(h.y1, h.y2, h.H, crop_h, tid);
    if (tid >= 64 && tid < 64 + crop_w) s_tx[tid - 64] = make_tap(h.x1, h.x2, h.W, crop_w, tid - 64);
    __syncthreads();

    const int lanes = cpb / V;                 // lanes per bin (consecutive lanes: consecutive channels)
    const int c = (tid % lanes) * V;           // first channel of this lane
    const int g = tid / lanes;                 // bin group
    const int ng = kThreads / lanes;
    const float *__restrict__ src =
        ls.img[h.lvl] + (size_t)h.img * h.H * h.W * depth + c_begin + c;
    const int W = h.W;
    if (c < c_count) {
        constexpr int UNROLL = 4;
        for (int b0 = g; b0 < bins; b0 += ng * UNROLL) {
            vec_t tl[UNROLL], tr[UNROLL], bl[UNROLL], br[UNROLL];
            float fx[UNROLL], fy[UNROLL];
            int ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int b = b0 + u * ng;
                ok[u] = 0;
                fx[u] = fy[u] = 0.0f;
                if (b < bins) {
                    const int y = b / crop_w;
                    const int x = b - y * crop_w;
                    const Tap ty = s_ty[y];
                    const Tap tx = s_tx[x];
                    ok[u] = (ty.valid & tx.valid) ? 1 : 2;
                    if (ok[u] == 1) {
                        const size_t r0 = (size_t)ty.i0 * W, r1 = (size_t)ty.i1 * W;
                        tl[u] = *reinterpret_cast<const vec_t *>(src + (r0 + tx.i0) * depth);
                        tr[u] = *reinterpret_cast<const vec_t *>(src + (r0 + tx.i1) * depth);
                        bl[u] = *reinterpret_cast<const vec_t *>(src + (r1 + tx.i0) * depth);
                        br[u] = *reinterpret_cast<const vec_t *>(src + (r1 + tx.i1) * depth);
                        fx[u] = tx.frac;
                        fy[u] = ty.frac;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int b = b0 + u * ng;
                if (ok[u] == 1) {
#pragma unroll
                    for (int q = 0; q < V; ++q) {
                        const float vtl = vget(tl[u], q), vtr = vget(tr[u], q);
                        const float vbl = vget(bl[u], q), vbr = vget(br[u], q);
                        const float dt = vtr - vtl;
                        const float top = vtl + dt * fx[u];
                        const float db = vbr - vbl;
                        const float bot = vbl + db * fx[u];
                        const float dv = bot - top;
                        s_dyn[(c + q) * pitch + b] = top + dv * fy[u];
                    }
                } else if (ok[u] == 2) {
#pragma unroll
                    for (int q = 0; q < V; ++q) s_dyn[(c + q) * pitch + b] = extrap;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += kThreads) {
        const int cc = i / bins;
        out[i] = s_dyn[cc * pitch + (i - cc * bins)];
    }
}

template <int CH, int CW>
__global__ __launch_bounds__(kThreads) void crop_bwd_cl_kernel(
    LevelSetMut ls, const float *__restrict__ grads, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, const int *__restrict__ level, int num_boxes, int batch,
    int depth, int crop_h_rt, int crop_w_rt, int cpb, int chunks)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    const int pitch = bins | 1;
    extern __shared__ float s_dyn[];          // [cpb][pitch]: the box's gradient block, transposed on read
    __shared__ Tap s_ty[kMaxCrop];
    __shared__ Tap s_tx[kMaxCrop];

    const int tid = threadIdx.x;
    const int box = blockIdx.x / chunks;
    const int chunk = blockIdx.x - box * chunks;
    const int c_begin = chunk * cpb;
    const int c_count = min(cpb, depth - c_begin);
    const int total = c_count * bins;

    BoxHeader h;
    if (!load_box(ls, boxes, box_ind, level, box, batch, h)) return;
    if (tid < crop_h) s_ty[tid] = make_tap(h.y1, h.y2, h.H, crop_h, tid);
    if (tid >= 64 && tid < 64 + crop_w) s_tx[tid - 64] = make_tap(h.x1, h.x2, h.W, crop_w, tid - 64);
    const float *__restrict__ gsrc = grads + ((size_t)box * depth + c_begin) * bins;
    for (int i = tid; i < total; i += kThreads) {
        const int cc = i / bins;
        s_dyn[cc * pitch + (i - cc * bins)] = gsrc[i];
    }
    __syncthreads();

    const int c = tid % cpb;
    const int g = tid / cpb;
    const int ng = kThreads / cpb;
    if (c >= c_count) return;
    float *__restrict__ dst = ls.img[h.lvl] + (size_t)h.img * h.H * h.W * depth + c_begin + c;
    const int W = h.W;
    for (int b = g; b < bins; b += ng) {
        const int y = b / crop_w;
        const int x = b - y * crop_w;
        const Tap ty = s_ty[y];
        const Tap tx = s_tx[x];
        if (!(ty.valid & tx.valid)) continue;
        const float gv = s_dyn[c * pitch + b];
        const float wy0 = 1.0f - ty.frac;
        const float wx0 = 1.0f - tx.frac;
        const float gtop = wy0 * gv;
        const float gbot = ty.frac * gv;
        const size_t r0 = (size_t)ty.i0 * W, r1 = (size_t)ty.i1 * W;
        atomicAdd(dst + (r0 + tx.i0) * depth, wx0 * gtop);
        atomicAdd(dst + (r0 + tx.i1) * depth, tx.frac * gtop);
        atomicAdd(dst + (r1 + tx.i0) * depth, wx0 * gbot);
        atomicAdd(dst + (r1 + tx.i1) * depth, tx.frac * gbot);
    }
}

// -------------------------------------------------------------------------------------
// Channels-last backward, TILE-OWNER form: no atomics of any kind, no memset, and the reference's summation order.
// A workgroup owns a tile of kClTH x kClTW cells of one image's map for 256 channels: thread t owns channel t of every
// cell (its private column of an LDS tile [cell][256]: plain read-add-write, no conflicts, nothing shared).  The boxes
// that reach the tile are listed in box order (as crop_bwd_tiles_kernel does) and every thread walks the listed boxes
// and their bins in (y, x) order, adding TL, TR, BL, BR -- exactly the order in which the reference's serial loop
// (lib/roi_align/src/crop_and_resize.c:190-251) reaches a given (cell, channel), with the same fp32 products: the
// result is BIT-IDENTICAL to the reference / the oracle, and the same from run to run (the atomic forms are not).
// Opt-in (FI_CROP_BWD_CL_TILES=1, see backward_cl_impl): slower than the atomic kernels on crowded coarse levels.
// The tile then leaves with 1 KB-contiguous stores (256 channels of a cell), which is also the zero fill; the
// accumulate form skips tiles no box reaches and reads the others first.
// (Zero-weight duplicates are skipped: when a coordinate is an integer, floor == ceil and the reference adds
// frac * g = +-0 to the same cell a second time, which never changes a sum.)
// -------------------------------------------------------------------------------------
constexpr int kClTH = 4, kClTW = 8, kClCells = kClTH * kClTW, kClChan = 256;

template <int CH, int CW, bool ACC>
__global__ __launch_bounds__(kThreads) void crop_bwd_cl_tiles_kernel(
    LevelSetMut ls, TileGrid tg, const float *__restrict__ grads, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, const int *__restrict__ level, int num_boxes, int batch,
    int depth, int crop_h_rt, int crop_w_rt, int nchunk)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    __shared__ float s_tile[kClCells * kClChan];
    __shared__ float s_box[kThreads][4];        // base_y, step_y, base_x, step_x
    __shared__ int s_rng[kThreads];             // k_lo(y) | k_n(y) << 8 | k_lo(x) << 16 | k_n(x) << 24
    __shared__ int s_id[kThreads];
    __shared__ int s_wave_n[kThreads / 64];

    const int tid = threadIdx.x;
    const int chunk = blockIdx.x % nchunk;
    int t = blockIdx.x / nchunk;
    int lvl = 0;
    while (lvl + 1 < ls.n && t >= tg.base[lvl + 1]) ++lvl;
    t -= tg.base[lvl];
    const int H = ls.H[lvl], W = ls.W[lvl];
    const int tiles = tg.nty[lvl] * tg.ntx[lvl];
    const int img = t / tiles;
    const int tile = t - img * tiles;
    const int row0 = (tile / tg.ntx[lvl]) * kClTH;
    const int col0 = (tile % tg.ntx[lvl]) * kClTW;
    const int c = chunk * kClChan + tid;
    const bool c_ok = c < depth;
    float *__restrict__ dst = ls.img[lvl] + (size_t)img * H * W * depth + (c_ok ? c : 0);
    float *mine = s_tile + tid;                  // [cell] at stride kClChan

    bool started = false;                        // the tile holds its starting values
    auto start = [&]() {
#pragma unroll
        for (int cell = 0; cell < kClCells; ++cell) {
            float v = 0.0f;
            if (ACC) {
                const int r = row0 + cell / kClTW, col = col0 + cell % kClTW;
                if (c_ok && r < H && col < W) v = dst[((size_t)r * W + col) * depth];
            }
            mine[cell * kClChan] = v;
        }
        started = true;
    };
    if (!ACC) start();

    const int lane = tid & 63, wave = tid >> 6;
    for (int chunk0 = 0; chunk0 < num_boxes; chunk0 += kThreads) {
        const int box = chunk0 + tid;
        bool hit = false;
        float by = 0, sy = 0, bx = 0, sx = 0;
        int ky = 0, ny = 0, kx = 0, nx = 0;
        if (box < num_boxes && box_ind[box] == img && (level ? (level[box] - 2) : 0) == lvl) {
            const float *b = boxes + 4 * (size_t)box;
            hit = axis_range(b[0], b[2], H, crop_h, row0, kClTH, &by, &sy, &ky, &ny) &&
                  axis_range(b[1], b[3], W, crop_w, col0, kClTW, &bx, &sx, &kx, &nx);
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_wave_n[wave] = __popcll(m);
        __syncthreads();
        int off = 0, n = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int k = s_wave_n[w];
            if (w < wave) off += k;
            n += k;
        }
        if (hit) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_box[slot][0] = by; s_box[slot][1] = sy; s_box[slot][2] = bx; s_box[slot][3] = sx;
            s_rng[slot] = ky | (ny << 8) | (kx << 16) | (nx << 24);
            s_id[slot] = box;
        }
        __syncthreads();
        if (n > 0 && !started) start();           // (uniform over the workgroup)
        for (int j = 0; j < n; ++j) {             // box order; every thread: its own channel
            const int rng = s_rng[j];
            const int ky0 = rng & 255, nyj = (rng >> 8) & 255, kx0 = (rng >> 16) & 255, nxj = (rng >> 24) & 255;
            const float byj = s_box[j][0], syj = s_box[j][1], bxj = s_box[j][2], sxj = s_box[j][3];
            const float *__restrict__ gch = grads + ((size_t)s_id[j] * depth + (c_ok ? c : 0)) * bins;
            for (int y = ky0; y < ky0 + nyj; ++y) {
                const Tap ty = tap_at(crop_h > 1 ? byj + (float)y * syj : byj, H);
                if (!ty.valid) continue;
                const unsigned r0 = (unsigned)(ty.i0 - row0), r1 = (unsigned)(ty.i1 - row0);
                const bool in_r0 = r0 < (unsigned)kClTH, in_r1 = (r1 < (unsigned)kClTH) && (ty.i1 != ty.i0);
                if (!(in_r0 | in_r1)) continue;
                const float wy0 = 1.0f - ty.frac;
                const float *__restrict__ grow = gch + y * crop_w;
                for (int xb = kx0; xb < kx0 + nxj; xb += 4) {
                    float gv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) gv[u] = (xb + u < kx0 + nxj) ? grow[xb + u] : 0.0f;   // 4 loads in flight
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int x = xb + u;
                        if (x >= kx0 + nxj) break;
                        const Tap tx = tap_at(crop_w > 1 ? bxj + (float)x * sxj : bxj, W);
                        if (!tx.valid) continue;
                        const unsigned c0 = (unsigned)(tx.i0 - col0), c1 = (unsigned)(tx.i1 - col0);
                        const bool in_c0 = c0 < (unsigned)kClTW, in_c1 = (c1 < (unsigned)kClTW) && (tx.i1 != tx.i0);
                        if (!(in_c0 | in_c1)) continue;
                        // reference order: dtop = (1-ly)*g; TL += (1-lx)*dtop; TR += lx*dtop; dbot = ly*g; BL, BR
                        const float wx0 = 1.0f - tx.frac;
                        const float gtop = wy0 * gv[u], gbot = ty.frac * gv[u];
                        if (in_r0 & in_c0) mine[(r0 * kClTW + c0) * kClChan] += wx0 * gtop;
                        if (in_r0 & in_c1) mine[(r0 * kClTW + c1) * kClChan] += tx.frac * gtop;
                        if (in_r1 & in_c0) mine[(r1 * kClTW + c0) * kClChan] += wx0 * gbot;
                        if (in_r1 & in_c1) mine[(r1 * kClTW + c1) * kClChan] += tx.frac * gbot;
                    }
                }
            }
        }
        __syncthreads();                          // the list is rebuilt by the next chunk
    }
    if (!started || !c_ok) return;                // accumulate form: nothing reached this tile
#pragma unroll
    for (int cell = 0; cell < kClCells; ++cell) {
        const int r = row0 + cell / kClTW, col = col0 + cell % kClTW;
        if (r < H && col < W) dst[((size_t)r * W + col) * depth] = mine[cell * kClChan];
    }
}

// Channels-last backward for crops with more bins than source cells (14 x 14 crops of the mask branch: a box
// of a pyramid level covers about 8..21 source rows/columns, so its 4 * 196 bilinear contributions per channel
// land on 64..441 distinct cells).  GATHER form: the sample coordinates of an axis are monotone in the bin
// index, so the bins whose lower (upper) tap is source row r form ONE contiguous range; a thread owns a
// (source cell, channel) pair, sums its <= 2 x 2 ranges of bins from the LDS copy of the box's gradient block
// and issues ONE global atomic -- 2..10x fewer atomics than the scatter form (crop_bwd_cl_kernel), no LDS
// atomics, no tile to clear.  Boxes with a footprint wider than 64 rows or columns (not produced by the level
// assignment; the tests do) take the scatter loop.
constexpr int kGatherCpb = 32;    // channels per workgroup: 128-byte runs in the channels-last maps
constexpr int kGatherSpan = 64;   // footprint rows / columns with a range table entry

// ranges of bins per source row: byte 0 = first bin with i0 == row, byte 1 = their count, byte 2 = first bin
// with i1 == row, byte 3 = their count
__device__ __forceinline__ unsigned tap_ranges(const Tap *__restrict__ taps, int n, int row)
{
    int a0 = 0, n0 = 0, a1 = 0, n1 = 0;
    for (int k = 0; k < n; ++k) {
        const Tap t = taps[k];
        if (!t.valid) continue;
        if (t.i0 == row) {
            if (n0 == 0) a0 = k;
            ++n0;
        }
        if (t.i1 == row) {
            if (n1 == 0) a1 = k;
            ++n1;
        }
    }
    return (unsigned)a0 | ((unsigned)n0 << 8) | ((unsigned)a1 << 16) | ((unsigned)n1 << 24);
}

template <int CH, int CW>
__global__ __launch_bounds__(kThreads) void crop_bwd_cl_gather_kernel(
    LevelSetMut ls, const float *__restrict__ grads, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, const int *__restrict__ level, int num_boxes, int batch,
    int depth, int crop_h_rt, int crop_w_rt, int chunks)
{
    const int crop_h = CH ? CH : crop_h_rt;
    const int crop_w = CW ? CW : crop_w_rt;
    const int bins = crop_h * crop_w;
    const int pitch = bins | 1;
    extern __shared__ float s_dyn[];          // [kGatherCpb][pitch]: the box's gradient block
    __shared__ Tap s_ty[kMaxCrop];
    __shared__ Tap s_tx[kMaxCrop];
    __shared__ int s_lim[4];                  // ymin, ymax, xmin, xmax over the valid taps
    __shared__ unsigned s_rows[kGatherSpan], s_cols[kGatherSpan];

    const int tid = threadIdx.x;
    const int box = blockIdx.x / chunks;
    const int chunk = blockIdx.x - box * chunks;
    const int c_begin = chunk * kGatherCpb;
    const int c_count = min(kGatherCpb, depth - c_begin);
    const int total = c_count * bins;

    BoxHeader h;
    if (!load_box(ls, boxes, box_ind, level, box, batch, h)) return;
    if (tid == 0) {
        s_lim[0] = 0x7fffffff; s_lim[1] = -1; s_lim[2] = 0x7fffffff; s_lim[3] = -1;
    }
    __syncthreads();
    if (tid < crop_h) {
        const Tap t = make_tap(h.y1, h.y2, h.H, crop_h, tid);
        s_ty[tid] = t;
        if (t.valid) {
            atomicMin(&s_lim[0], t.i0);
            atomicMax(&s_lim[1], t.i1);
        }
    }
    if (tid >= 64 && tid < 64 + crop_w) {
        const Tap t = make_tap(h.x1, h.x2, h.W, crop_w, tid - 64);
        s_tx[tid - 64] = t;
        if (t.valid) {
            atomicMin(&s_lim[2], t.i0);
            atomicMax(&s_lim[3], t.i1);
        }
    }
    const float *__restrict__ gsrc = grads + ((size_t)box * depth + c_begin) * bins;
    for (int i = tid; i < total; i += kThreads) {
        const int cc = i / bins;
        s_dyn[cc * pitch + (i - cc * bins)] = gsrc[i];
    }
    __syncthreads();
    const int ymin = s_lim[0], xmin = s_lim[2];
    const int fh = s_lim[1] - ymin + 1, fw = s_lim[3] - xmin + 1;
    if (fh <= 0 || fw <= 0) return;           // every sample of an axis lies outside the map
    float *__restrict__ dst = ls.img[h.lvl] + (size_t)h.img * h.H * h.W * depth + c_begin;
    const int W = h.W;

    if (fh > kGatherSpan || fw > kGatherSpan) {          // scatter form
        const int c = tid % kGatherCpb;
        const int g = tid / kGatherCpb;
        constexpr int ng = kThreads / kGatherCpb;
        if (c >= c_count) return;
        for (int b = g; b < bins; b += ng) {
            const int y = b / crop_w;
            const int x = b - y * crop_w;
            const Tap ty = s_ty[y];
            const Tap tx = s_tx[x];
            if (!(ty.valid & tx.valid)) continue;
            const float gv = s_dyn[c * pitch + b];
            const float gtop = (1.0f - ty.frac) * gv;
            const float gbot = ty.frac * gv;
            const float wx0 = 1.0f - tx.frac;
            const size_t r0 = (size_t)ty.i0 * W, r1 = (size_t)ty.i1 * W;
            atomicAdd(dst + (r0 + tx.i0) * depth + c, wx0 * gtop);
            atomicAdd(dst + (r0 + tx.i1) * depth + c, tx.frac * gtop);
            atomicAdd(dst + (r1 + tx.i0) * depth + c, wx0 * gbot);
            atomicAdd(dst + (r1 + tx.i1) * depth + c, tx.frac * gbot);
        }
        return;
    }

    if (tid < fh) s_rows[tid] = tap_ranges(s_ty, crop_h, ymin + tid);
    if (tid >= 64 && tid < 64 + fw) s_cols[tid - 64] = tap_ranges(s_tx, crop_w, xmin + tid - 64);
    __syncthreads();

    // a wavefront covers 2 cells x 32 channels: two 128-byte runs per atomic instruction
    const int c = tid % kGatherCpb;
    if (c >= c_count) return;
    const float *__restrict__ gc = s_dyn + c * pitch;
    const int cells = fh * fw;
    int py = 0, px = tid / kGatherCpb;          // cell = py * fw + px, advanced incrementally
    while (px >= fw) {
        px -= fw;
        ++py;
    }
    constexpr int step = kThreads / kGatherCpb;
    for (int cell = tid / kGatherCpb; cell < cells; cell += step) {
        const unsigned ri = s_rows[py], ci = s_cols[px];
        if ((ri & 0xff00ff00u) && (ci & 0xff00ff00u)) {
            float sum = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
                const int ya = (ri >> (16 * ky)) & 0xff, yn = (ri >> (16 * ky + 8)) & 0xff;
                for (int y = ya; y < ya + yn; ++y) {
                    const float fy = s_ty[y].frac;
                    const float wy = ky ? fy : 1.0f - fy;
                    float rowsum = 0.0f;
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) {
                        const int xa = (ci >> (16 * kx)) & 0xff, xn = (ci >> (16 * kx + 8)) & 0xff;
                        for (int x = xa; x < xa + xn; ++x) {
                            const float fx = s_tx[x].frac;
                            rowsum += (kx ? fx : 1.0f - fx) * gc[y * crop_w + x];
                        }
                    }
                    sum += wy * rowsum;
                }
            }
            atomicAdd(dst + ((size_t)(ymin + py) * W + (xmin + px)) * depth + c, sum);
        }
        px += step;
        while (px >= fw) {
            px -= fw;
            ++py;
        }
    }
}

__global__ void crop_taps_kernel(const float *__restrict__ boxes, int num_boxes, int H, int W,
                                 int crop_h, int crop_w, int *y_valid, int *y0, int *y1,
                                 float *y_frac, int *x_valid, int *x0, int *x1, float *x_frac)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_box = crop_h + crop_w;
    if (i >= num_boxes * per_box) return;
    const int box = i / per_box;
    const int k = i - box * per_box;
    const float *b = boxes + 4 * (size_t)box;
    if (k < crop_h) {
        const Tap t = make_tap(b[0], b[2], H, crop_h, k);
        y_valid[box * crop_h + k] = t.valid;
        y0[box * crop_h + k] = t.i0;
        y1[box * crop_h + k] = t.i1;
        y_frac[box * crop_h + k] = t.frac;
    } else {
        const int kx = k - crop_h;
        const Tap t = make_tap(b[1], b[3], W, crop_w, kx);
        x_valid[box * crop_w + kx] = t.valid;
        x0[box * crop_w + kx] = t.i0;
        x1[box * crop_w + kx] = t.i1;
        x_frac[box * crop_w + kx] = t.frac;
    }
}

// Channel chunking: prefer 8 chunks (one per XCD) of >= 8 channels; shrink the
// chunk while the grid would not fill the chip (256 CUs x 8 workgroups).
void pick_chunks(int num_boxes, int depth, int *chan_per_block, int *chunks)
{
    int cpb = fi::ceil_div(depth, 8);
    if (cpb < 1) cpb = 1;
    while (cpb > 8 && (long)num_boxes * fi::ceil_div(depth, cpb) < 2048) cpb = fi::ceil_div(cpb, 2);
    *chan_per_block = cpb;
    *chunks = fi::ceil_div(depth, cpb);
}

// forward kernel: 16 channels per workgroup, chunk-major per XCD (measured at 512 x 256 x 7 x 7 on one
// 256^2 map: 4 / 8 / 16 / 32 channels -> 50.6 / 33.2 / 30.8 / 34.3 us; the box-major order of round 1: 34.4)
void pick_fwd_chunks(int depth, int *chan_per_block, int *chunks)
{
    const int cpb = 16;
    *chan_per_block = cpb;
    *chunks = fi::ceil_div(depth, cpb);
}

template <typename K, typename... Args>
int launch_sized(int crop_h, int crop_w, K k77, K k1414, K k2828, K kgen, dim3 grid,
                 hipStream_t st, Args... args)
{
    K k = kgen;
    if (crop_h == 7 && crop_w == 7) k = k77;
    else if (crop_h == 14 && crop_w == 14) k = k1414;
    else if (crop_h == 28 && crop_w == 28) k = k2828;
    hipLaunchKernelGGL(k, grid, dim3(kThreads), 0, st, args...);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int size_class(int crop_h, int crop_w)
{
    if (crop_h == 7 && crop_w == 7) return 0;
    if (crop_h == 14 && crop_w == 14) return 1;
    if (crop_h == 28 && crop_w == 28) return 2;
    return 3;
}

int check_common(int num_boxes, int batch, int depth, int crop_h, int crop_w)
{
    FI_REQUIRE(num_boxes >= 0 && batch > 0 && depth > 0, "sizes must be positive");
    FI_REQUIRE(crop_h >= 1 && crop_w >= 1, "crop size must be >= 1");
    if (crop_h > kMaxCrop || crop_w > kMaxCrop) {
        fi::set_error("crop size %dx%d exceeds the supported maximum %d", crop_h, crop_w, kMaxCrop);
        return FI_ERR_UNSUPPORTED;
    }
    return FI_OK;
}

int forward_impl(const LevelSet &ls, const float *boxes, const int32_t *box_ind,
                 const int32_t *level, int num_boxes, int batch, int depth, int crop_h, int crop_w,
                 float extrap, float *crops, int32_t *status, hipStream_t st)
{
    if (num_boxes == 0) return FI_OK;
    constexpr int kFlatCG = 8;
    if (crop_h == 7 && crop_w == 7 && depth % (8 * kFlatCG) == 0 && !getenv("FI_CROP_NO_FLAT")) {
        // 7 x 7: table-free kernel, 8 channels per thread (-9 % against the table kernel at the north-star shape)
        const int groups_per_xcd = depth / (8 * kFlatCG);
        const long per_xcd = ((long)num_boxes * groups_per_xcd * 49 + kThreads - 1) / kThreads;
        FI_REQUIRE(per_xcd * 8 < 2147483647L, "grid too large");
        fi::ProfScope prof(FI_K_CROP_FWD_7X7, st);
        hipLaunchKernelGGL((crop_fwd_flat_kernel<7, 7, kFlatCG>), dim3((unsigned)(per_xcd * 8)), dim3(kThreads), 0, st, ls,
                           boxes, box_ind, level, num_boxes, batch, depth, extrap, groups_per_xcd, crops, status);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    if (depth == 1 && !getenv("FI_CROP_NO_C1")) {
        // one plane per map (the GT-mask crop): a thread per bin
        const long nb = ((long)num_boxes * crop_h * crop_w + kThreads - 1) / kThreads;
        FI_REQUIRE(nb < 2147483647L, "grid too large");
        fi::ProfScope prof(FI_K_CROP_FWD_7X7 + size_class(crop_h, crop_w), st);
        return launch_sized(crop_h, crop_w, crop_fwd_c1_kernel<7, 7>, crop_fwd_c1_kernel<14, 14>, crop_fwd_c1_kernel<28, 28>,
                            crop_fwd_c1_kernel<0, 0>, dim3((unsigned)nb), st, ls, boxes, box_ind, level, num_boxes, batch, crop_h,
                            crop_w, extrap, crops, status);
    }
    int cpb, chunks;
    pick_fwd_chunks(depth, &cpb, &chunks);
    const long nblk = (long)num_boxes * fi::ceil_div(chunks, 8) * 8;
    FI_REQUIRE(nblk < 2147483647L, "grid too large");
    fi::ProfScope prof(FI_K_CROP_FWD_7X7 + size_class(crop_h, crop_w), st);
    return launch_sized(crop_h, crop_w, crop_fwd_kernel<7, 7>, crop_fwd_kernel<14, 14>,
                        crop_fwd_kernel<28, 28>, crop_fwd_kernel<0, 0>, dim3((unsigned)nblk), st, ls,
                        boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w, extrap, cpb,
                        chunks, crops, status);
}

template <int CG, bool ACC>
int launch_bwd_tiles(int crop_h, int crop_w, dim3 grid, hipStream_t st, const LevelSetMut &ls, const TileGrid &tg,
                     const float *grads, const float *boxes, const int32_t *box_ind, const int32_t *level,
                     int num_boxes, int batch, int depth, int ncg)
{
    return launch_sized(crop_h, crop_w, crop_bwd_tiles_kernel<7, 7, CG, ACC>, crop_bwd_tiles_kernel<14, 14, CG, ACC>,
                        crop_bwd_tiles_kernel<28, 28, CG, ACC>, crop_bwd_tiles_kernel<0, 0, CG, ACC>, grid, st, ls, tg,
                        grads, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w, ncg);
}

int backward_impl(const LevelSetMut &ls, const float *grads, const float *boxes,
                  const int32_t *box_ind, const int32_t *level, int num_boxes, int batch, int depth,
                  int crop_h, int crop_w, hipStream_t st, bool clear = true)
{
    static const bool scatter = getenv("FI_CROP_BWD_SCATTER") != nullptr;      // A/B: the round-1..4 kernel
    if (!scatter) {
        // tile-owner form: every cell of every map is written by exactly one workgroup (no memset, no global atomics)
        // channels per workgroup: 4 (32 KB of fp64 accumulators, 4 workgroups per CU) measured against 8 (64 KB, 2 per CU)
        // at 512 x 256 on one 256^2 map: 7 x 7 95 vs 114 us, 14 x 14 160 vs 196 us (profiles/r05_crop_bwd_tiles.txt)
        static const int CG = getenv("FI_CROP_TILE_CG") ? atoi(getenv("FI_CROP_TILE_CG")) : 4;
        FI_REQUIRE(CG == 4 || CG == 8, "FI_CROP_TILE_CG must be 4 or 8 (the two instantiated channel groups)");
        TileGrid tg = {};
        long total = 0;
        for (int l = 0; l < ls.n; ++l) {
            const int W = ls.W[l];
            tg.tw[l] = W > 32 ? 64 : (W > 16 ? 32 : 16);
            tg.th[l] = kTileCells / tg.tw[l];
            tg.nty[l] = fi::ceil_div(ls.H[l], tg.th[l]);
            tg.ntx[l] = fi::ceil_div(W, tg.tw[l]);
            tg.base[l] = (int)total;
            total += (long)batch * tg.nty[l] * tg.ntx[l];
        }
        tg.base[ls.n] = (int)total;
        const int ncg = fi::ceil_div(depth, CG);
        FI_REQUIRE(total * ncg < 2147483647L, "grid too large");
        if (num_boxes == 0 && !clear) return FI_OK;
        fi::ProfScope prof(FI_K_CROP_BWD_7X7 + size_class(crop_h, crop_w), st);
        const dim3 grid((unsigned)(total * ncg));
        if (CG == 4)
            return clear ? launch_bwd_tiles<4, false>(crop_h, crop_w, grid, st, ls, tg, grads, boxes, box_ind, level,
                                                      num_boxes, batch, depth, ncg)
                         : launch_bwd_tiles<4, true>(crop_h, crop_w, grid, st, ls, tg, grads, boxes, box_ind, level,
                                                     num_boxes, batch, depth, ncg);
        return clear ? launch_bwd_tiles<8, false>(crop_h, crop_w, grid, st, ls, tg, grads, boxes, box_ind, level,
                                                  num_boxes, batch, depth, ncg)
                     : launch_bwd_tiles<8, true>(crop_h, crop_w, grid, st, ls, tg, grads, boxes, box_ind, level,
                                                 num_boxes, batch, depth, ncg);
    }
    for (int l = 0; clear && l < ls.n; ++l) {
        const size_t bytes = sizeof(float) * (size_t)batch * depth * ls.H[l] * ls.W[l];
        FI_HIP_CHECK(hipMemsetAsync(ls.img[l], 0, bytes, st));
    }
    if (num_boxes == 0) return FI_OK;
    int cpb, chunks;
    pick_chunks(num_boxes, depth, &cpb, &chunks);
    const long nblk = (long)num_boxes * chunks;
    FI_REQUIRE(nblk < 2147483647L, "grid too large");
    fi::ProfScope prof(FI_K_CROP_BWD_7X7 + size_class(crop_h, crop_w), st);
    return launch_sized(crop_h, crop_w, crop_bwd_kernel<7, 7>, crop_bwd_kernel<14, 14>,
                        crop_bwd_kernel<28, 28>, crop_bwd_kernel<0, 0>, dim3((unsigned)nblk), st, ls,
                        grads, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w, cpb,
                        chunks);
}

// channels-last launch geometry: channels per workgroup (multiple of 64 so that a wavefront shares
// one bin) sized to keep the LDS tile <= 50 KB; returns 0 when the crop is too large for the tile
int pick_cl_chunk(int depth, int bins)
{
    const int pitch = bins | 1;
    int cpb = 256;
    while (cpb > 64 && ((long)cpb * pitch * 4 > 28 * 1024 || cpb / 2 >= depth)) cpb /= 2;
    if ((long)cpb * pitch * 4 > 56 * 1024) return 0;
    return cpb;
}

int cl_size_class(int crop_h, int crop_w)
{
    if (crop_h == 7 && crop_w == 7) return 0;
    if (crop_h == 14 && crop_w == 14) return 1;
    return 2;
}

int forward_cl_impl(const LevelSet &ls, const float *boxes, const int32_t *box_ind,
                    const int32_t *level, int num_boxes, int batch, int depth, int crop_h, int crop_w,
                    float extrap, float *crops, hipStream_t st)
{
    if (num_boxes == 0) return FI_OK;
    const int bins = crop_h * crop_w;
    const int cpb = pick_cl_chunk(depth, bins);
    if (cpb == 0) {
        fi::set_error("channels-last crop supports crop_h*crop_w <= 220 (got %dx%d)", crop_h, crop_w);
        return FI_ERR_UNSUPPORTED;
    }
    const int chunks = fi::ceil_div(depth, cpb);
    const long nblk = (long)num_boxes * chunks;
    FI_REQUIRE(nblk < 2147483647L, "grid too large");
    const size_t lds = sizeof(float) * (size_t)cpb * (bins | 1);
    const int cls = cl_size_class(crop_h, crop_w);
    fi::ProfScope prof(FI_K_CROP_FWD_NHWC_7X7 + cls, st);
    bool vec4 = (depth % 4 == 0);
    for (int l = 0; l < ls.n; ++l) vec4 = vec4 && ((uintptr_t)ls.img[l] % 16 == 0);
    auto k = vec4 ? (cls == 0 ? crop_fwd_cl_kernel<7, 7, 4> : cls == 1 ? crop_fwd_cl_kernel<14, 14, 4> : crop_fwd_cl_kernel<0, 0, 4>)
                  : (cls == 0 ? crop_fwd_cl_kernel<7, 7, 1> : cls == 1 ? crop_fwd_cl_kernel<14, 14, 1> : crop_fwd_cl_kernel<0, 0, 1>);
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(kThreads), lds, st, ls, boxes, box_ind, level, num_boxes,
                       batch, depth, crop_h, crop_w, extrap, cpb, chunks, crops);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int backward_cl_impl(const LevelSetMut &ls, const float *grads, const float *boxes,
                     const int32_t *box_ind, const int32_t *level, int num_boxes, int batch, int depth,
                     int crop_h, int crop_w, hipStream_t st, bool clear = true)
{
    // FI_CROP_BWD_CL_TILES=1 (read at every call): the deterministic tile-owner form -- bit-identical to the reference's
    // serial loop, no atomics, no memset -- instead of the atomic kernels below.  Not the default: a thread walks its
    // tile's boxes one after the other and waits for every box's gradient values (one HBM round trip each), so the
    // crowded tiles of the coarse levels set the time: pyramid 2048 x 256, 7 x 7 492 vs 346 (+130 memset) us,
    // 14 x 14 1567 vs 410 (+90) us (profiles/r05_crop_bwd_tiles.txt).
    const char *tiles_env = getenv("FI_CROP_BWD_CL_TILES");
    if (tiles_env && tiles_env[0] == '1') {
        // every cell of every map is written by exactly one workgroup
        TileGrid tg = {};
        long total = 0;
        for (int l = 0; l < ls.n; ++l) {
            tg.th[l] = kClTH;
            tg.tw[l] = kClTW;
            tg.nty[l] = fi::ceil_div(ls.H[l], kClTH);
            tg.ntx[l] = fi::ceil_div(ls.W[l], kClTW);
            tg.base[l] = (int)total;
            total += (long)batch * tg.nty[l] * tg.ntx[l];
        }
        tg.base[ls.n] = (int)total;
        const int nchunk = fi::ceil_div(depth, kClChan);
        FI_REQUIRE(total * nchunk < 2147483647L, "grid too large");
        if (num_boxes == 0 && !clear) return FI_OK;
        fi::ProfScope prof(FI_K_CROP_BWD_NHWC_7X7 + cl_size_class(crop_h, crop_w), st);
        const dim3 grid((unsigned)(total * nchunk));
        if (clear)
            return launch_sized(crop_h, crop_w, crop_bwd_cl_tiles_kernel<7, 7, false>, crop_bwd_cl_tiles_kernel<14, 14, false>,
                                crop_bwd_cl_tiles_kernel<28, 28, false>, crop_bwd_cl_tiles_kernel<0, 0, false>, grid, st, ls,
                                tg, grads, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w, nchunk);
        return launch_sized(crop_h, crop_w, crop_bwd_cl_tiles_kernel<7, 7, true>, crop_bwd_cl_tiles_kernel<14, 14, true>,
                            crop_bwd_cl_tiles_kernel<28, 28, true>, crop_bwd_cl_tiles_kernel<0, 0, true>, grid, st, ls, tg,
                            grads, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w, nchunk);
    }
    for (int l = 0; clear && l < ls.n; ++l) {
        const size_t bytes = sizeof(float) * (size_t)batch * depth * ls.H[l] * ls.W[l];
        FI_HIP_CHECK(hipMemsetAsync(ls.img[l], 0, bytes, st));
    }
    if (num_boxes == 0) return FI_OK;
    const int bins = crop_h * crop_w;
    const int cls = cl_size_class(crop_h, crop_w);
    // 7 x 7 crops: as many atomics as touched cells when a box is 14 cells wide, up to 3x more when it is 7 -- and the
    // step's pyramid call (2048 jittered-GT boxes crowding the coarse levels) is bound by same-cell atomics: the gather form
    // is 345 -> 282 us there, but 94 -> 115 us for 512 boxes on one map (profiles/r06_crop_bwd_gather_small.txt).
    // FI_CROP_BWD_GATHER_SMALL=0 / 1 forces the choice.
    static const char *gs_env = getenv("FI_CROP_BWD_GATHER_SMALL");
    const bool gather_small = gs_env ? atoi(gs_env) != 0 : (ls.n > 1 && num_boxes >= 1024);
    if ((bins >= 100 || gather_small) && bins <= 220) {
        // more bins than source cells per box: gather form, one atomic per touched cell
        const int chunks = fi::ceil_div(depth, kGatherCpb);
        const long nblk = (long)num_boxes * chunks;
        FI_REQUIRE(nblk < 2147483647L, "grid too large");
        const size_t lds = sizeof(float) * (size_t)kGatherCpb * (bins | 1);
        fi::ProfScope prof(FI_K_CROP_BWD_NHWC_7X7 + cls, st);
        auto k = cls == 1 ? crop_bwd_cl_gather_kernel<14, 14> : crop_bwd_cl_gather_kernel<0, 0>;
        hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(kThreads), lds, st, ls, grads, boxes, box_ind, level,
                           num_boxes, batch, depth, crop_h, crop_w, chunks);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    const int cpb = pick_cl_chunk(depth, bins);
    if (cpb == 0) {
        fi::set_error("channels-last crop supports crop_h*crop_w <= 220 (got %dx%d)", crop_h, crop_w);
        return FI_ERR_UNSUPPORTED;
    }
    const int chunks = fi::ceil_div(depth, cpb);
    const long nblk = (long)num_boxes * chunks;
    FI_REQUIRE(nblk < 2147483647L, "grid too large");
    const size_t lds = sizeof(float) * (size_t)cpb * (bins | 1);
    fi::ProfScope prof(FI_K_CROP_BWD_NHWC_7X7 + cls, st);
    auto k = cls == 0 ? crop_bwd_cl_kernel<7, 7> : cls == 1 ? crop_bwd_cl_kernel<14, 14> : crop_bwd_cl_kernel<0, 0>;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(kThreads), lds, st, ls, grads, boxes, box_ind, level,
                       num_boxes, batch, depth, crop_h, crop_w, cpb, chunks);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fill_levels(LevelSet &ls, const float *const *img, const int *hh, const int *ww, int n)
{
    FI_REQUIRE(n >= 1 && n <= kMaxLevels, "1 <= num_levels <= 8");
    FI_REQUIRE(img && hh && ww, "null level arrays");
    ls.n = n;
    for (int l = 0; l < n; ++l) {
        ls.img[l] = img[l];
        ls.H[l] = hh[l];
        ls.W[l] = ww[l];
        FI_REQUIRE(ls.img[l] && ls.H[l] >= 1 && ls.W[l] >= 1, "bad level entry");
    }
    return FI_OK;
}

}  // namespace

extern "C" {

int fi_crop_and_resize_forward(const float *image, const float *boxes, const int32_t *box_ind,
                               int num_boxes, int batch, int depth, int image_h, int image_w,
                               int crop_h, int crop_w, float extrapolation_value, float *crops,
                               int32_t *dev_status, fi_stream_t stream)
{
    int rc = check_common(num_boxes, batch, depth, crop_h, crop_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(image_h >= 1 && image_w >= 1, "image size must be positive");
    FI_REQUIRE(num_boxes == 0 || (image && boxes && box_ind && crops), "null pointer");
    LevelSet ls = {};
    ls.img[0] = image;
    ls.H[0] = image_h;
    ls.W[0] = image_w;
    ls.n = 1;
    return forward_impl(ls, boxes, box_ind, nullptr, num_boxes, batch, depth, crop_h, crop_w,
                        extrapolation_value, crops, dev_status, (hipStream_t)stream);
}

int fi_crop_and_resize_backward(const float *grads, const float *boxes, const int32_t *box_ind,
                                int num_boxes, int batch, int depth, int image_h, int image_w,
                                int crop_h, int crop_w, float *grads_image, fi_stream_t stream)
{
    int rc = check_common(num_boxes, batch, depth, crop_h, crop_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(image_h >= 1 && image_w >= 1, "image size must be positive");
    FI_REQUIRE(grads_image != nullptr, "null grads_image");
    FI_REQUIRE(num_boxes == 0 || (grads && boxes && box_ind), "null pointer");
    LevelSetMut ls = {};
    ls.img[0] = grads_image;
    ls.H[0] = image_h;
    ls.W[0] = image_w;
    ls.n = 1;
    return backward_impl(ls, grads, boxes, box_ind, nullptr, num_boxes, batch, depth, crop_h, crop_w,
                         (hipStream_t)stream);
}

int fi_crop_and_resize_taps(const float *boxes, int num_boxes, int image_h, int image_w, int crop_h,
                            int crop_w, int32_t *y_valid, int32_t *y0, int32_t *y1, float *y_frac,
                            int32_t *x_valid, int32_t *x0, int32_t *x1, float *x_frac,
                            fi_stream_t stream)
{
    FI_REQUIRE(num_boxes >= 0 && crop_h >= 1 && crop_w >= 1, "bad sizes");
    if (num_boxes == 0) return FI_OK;
    const int total = num_boxes * (crop_h + crop_w);
    hipLaunchKernelGGL(crop_taps_kernel, dim3(fi::ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, boxes, num_boxes, image_h, image_w, crop_h, crop_w,
                       y_valid, y0, y1, y_frac, x_valid, x0, x1, x_frac);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_pyramid_crop_forward(const float *const *level_images_host, const int *level_h_host,
                            const int *level_w_host, int num_levels, const float *boxes,
                            const int32_t *box_ind, const int32_t *level, int num_boxes, int batch,
                            int depth, int crop_h, int crop_w, float extrapolation_value,
                            float *crops, fi_stream_t stream)
{
    int rc = check_common(num_boxes, batch, depth, crop_h, crop_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(num_levels >= 1 && num_levels <= kMaxLevels, "1 <= num_levels <= 8");
    FI_REQUIRE(level_images_host && level_h_host && level_w_host, "null level arrays");
    FI_REQUIRE(num_boxes == 0 || (boxes && box_ind && level && crops), "null pointer");
    LevelSet ls = {};
    ls.n = num_levels;
    for (int l = 0; l < num_levels; ++l) {
        ls.img[l] = level_images_host[l];
        ls.H[l] = level_h_host[l];
        ls.W[l] = level_w_host[l];
        FI_REQUIRE(ls.img[l] && ls.H[l] >= 1 && ls.W[l] >= 1, "bad level entry");
    }
    return forward_impl(ls, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w,
                        extrapolation_value, crops, nullptr, (hipStream_t)stream);
}

static int pyramid_backward_entry(const float *grads, float *const *level_grads_host,
                                  const int *level_h_host, const int *level_w_host, int num_levels,
                                  const float *boxes, const int32_t *box_ind, const int32_t *level,
                                  int num_boxes, int batch, int depth, int crop_h, int crop_w,
                                  fi_stream_t stream, bool clear)
{
    int rc = check_common(num_boxes, batch, depth, crop_h, crop_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(num_levels >= 1 && num_levels <= kMaxLevels, "1 <= num_levels <= 8");
    FI_REQUIRE(level_grads_host && level_h_host && level_w_host, "null level arrays");
    FI_REQUIRE(num_boxes == 0 || (grads && boxes && box_ind && level), "null pointer");
    LevelSetMut ls = {};
    ls.n = num_levels;
    for (int l = 0; l < num_levels; ++l) {
        ls.img[l] = level_grads_host[l];
        ls.H[l] = level_h_host[l];
        ls.W[l] = level_w_host[l];
        FI_REQUIRE(ls.img[l] && ls.H[l] >= 1 && ls.W[l] >= 1, "bad level entry");
    }
    return backward_impl(ls, grads, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w,
                         (hipStream_t)stream, clear);
}

int fi_pyramid_crop_backward(const float *grads, float *const *level_grads_host,
                             const int *level_h_host, const int *level_w_host, int num_levels,
                             const float *boxes, const int32_t *box_ind, const int32_t *level,
                             int num_boxes, int batch, int depth, int crop_h, int crop_w,
                             fi_stream_t stream)
{
    return pyramid_backward_entry(grads, level_grads_host, level_h_host, level_w_host, num_levels, boxes, box_ind,
                                  level, num_boxes, batch, depth, crop_h, crop_w, stream, true);
}

int fi_pyramid_crop_backward_accumulate(const float *grads, float *const *level_grads_host,
                                        const int *level_h_host, const int *level_w_host, int num_levels,
                                        const float *boxes, const int32_t *box_ind, const int32_t *level,
                                        int num_boxes, int batch, int depth, int crop_h, int crop_w,
                                        fi_stream_t stream)
{
    return pyramid_backward_entry(grads, level_grads_host, level_h_host, level_w_host, num_levels, boxes, box_ind,
                                  level, num_boxes, batch, depth, crop_h, crop_w, stream, false);
}

int fi_pyramid_crop_forward_nhwc(const float *const *level_images_host, const int *level_h_host,
                                 const int *level_w_host, int num_levels, const float *boxes,
                                 const int32_t *box_ind, const int32_t *level, int num_boxes,
                                 int batch, int depth, int crop_h, int crop_w,
                                 float extrapolation_value, float *crops, fi_stream_t stream)
{
    int rc = check_common(num_boxes, batch, depth, crop_h, crop_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(num_boxes == 0 || (boxes && box_ind && crops), "null pointer");
    LevelSet ls = {};
    rc = fill_levels(ls, level_images_host, level_h_host, level_w_host, num_levels);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(level != nullptr || num_levels == 1, "level[] is required with more than one map");
    return forward_cl_impl(ls, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w,
                           extrapolation_value, crops, (hipStream_t)stream);
}

static int pyramid_backward_nhwc_entry(const float *grads, float *const *level_grads_host,
                                       const int *level_h_host, const int *level_w_host, int num_levels,
                                       const float *boxes, const int32_t *box_ind, const int32_t *level,
                                       int num_boxes, int batch, int depth, int crop_h, int crop_w,
                                       fi_stream_t stream, bool clear)
{
    int rc = check_common(num_boxes, batch, depth, crop_h, crop_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(num_boxes == 0 || (grads && boxes && box_ind), "null pointer");
    LevelSet lsc = {};
    rc = fill_levels(lsc, level_grads_host, level_h_host, level_w_host, num_levels);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(level != nullptr || num_levels == 1, "level[] is required with more than one map");
    LevelSetMut ls = {};
    ls.n = lsc.n;
    for (int l = 0; l < lsc.n; ++l) {
        ls.img[l] = level_grads_host[l];
        ls.H[l] = lsc.H[l];
        ls.W[l] = lsc.W[l];
    }
    return backward_cl_impl(ls, grads, boxes, box_ind, level, num_boxes, batch, depth, crop_h, crop_w,
                            (hipStream_t)stream, clear);
}

int fi_pyramid_crop_backward_nhwc(const float *grads, float *const *level_grads_host,
                                  const int *level_h_host, const int *level_w_host, int num_levels,
                                  const float *boxes, const int32_t *box_ind, const int32_t *level,
                                  int num_boxes, int batch, int depth, int crop_h, int crop_w,
                                  fi_stream_t stream)
{
    return pyramid_backward_nhwc_entry(grads, level_grads_host, level_h_host, level_w_host, num_levels, boxes,
                                       box_ind, level, num_boxes, batch, depth, crop_h, crop_w, stream, true);
}

int fi_pyramid_crop_backward_nhwc_accumulate(const float *grads, float *const *level_grads_host,
                                             const int *level_h_host, const int *level_w_host, int num_levels,
                                             const float *boxes, const int32_t *box_ind, const int32_t *level,
                                             int num_boxes, int batch, int depth, int crop_h, int crop_w,
                                             fi_stream_t stream)
{
    return pyramid_backward_nhwc_entry(grads, level_grads_host, level_h_host, level_w_host, num_levels, boxes,
                                       box_ind, level, num_boxes, batch, depth, crop_h, crop_w, stream, false);
}

}  // extern "C"
