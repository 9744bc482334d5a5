// conv_bf16.hip -- bf16-input, fp32-accumulate implicit-GEMM convolution on the CDNA4 matrix cores
// (v_mfma_f32_32x32x16_bf16): the reduced-precision conv path of BASELINE configs[4] ("fp16 MFMA conv
// path"), selected by configuration and never used for the fp32 headline (conv_igemm.hip is exact fp32).
//
// Tensors stay fp32 in memory (NCHW activations, tap-major [Cout][R][S][Cin] weights, fp32 outputs and
// gradients), so every other kernel of the step is untouched; operands are rounded to bf16 (RNE) on their
// way into LDS and products are accumulated in fp32 by the MFMA.  Against the exact kernel the result
// differs by the rounding of the two operands (2^-9 relative each), not by accumulation.
//
// Forward / data gradient (one kernel, as in conv_igemm.hip):   Y[m][p] = sum_k A[m][k] B[k][p]
//   m = output channel, p = (image, oh, ow), k = (tap, ci) tap-major; tile 128 x 128 x 32 (BM = 64 for
//   narrow layers) per 256-thread workgroup, 4 wavefronts as 2 x 2, each (BM/2) x 64 of 32x32 MFMA tiles.
//   The MFMA wants both operands K-contiguous per lane (8 bf16 = one ds_read_b128):
//     A (weights) is K-contiguous in memory: float4 loads, 16 values -> two ds_write_b128;
//     B (im2col, never materialised) is PIXEL-contiguous in memory: a thread loads 4 consecutive pixels
//     of two adjacent channels (two unaligned float4 loads; pixels that straddle a row end or the halo
//     take a guarded scalar path) and writes four (c, c+1) bf16 pairs -> LDS rows [pixel][k].
//   Global loads of K-tile t+1 are in flight during the MFMAs of tile t (register staging, two LDS
//   buffers, one barrier per K-tile).
// Weight gradient:   dW[m][(tap, ci)] = sum_p dY[m][p] X_tap[ci][p]: both operands are pixel-contiguous
//   in memory = K-contiguous for the MFMA, so both tiles are float4 loads -> ds_write_b64; the pixel range
//   is split across workgroups and accumulated with fp32 atomics (as the fp32 kernel does).
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "fi_common.h"

#ifdef FI_E16_HALF
#define E16 _Float16
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define FI16(stem, tail) stem##_f16##tail
#else
#define E16 __bf16
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define FI16(stem, tail) stem##_bf16##tail
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
// The 16-bit operand type of this translation unit.  conv_f16.hip includes this file with FI_E16_HALF defined and
// gets the same kernels on IEEE half (v_mfma_f32_32x32x16_f16) under the *_f16 entry-point names -- BASELINE
// configs[4] names an fp16 path; bf16 has fp32's exponent range and needs no loss scaling, so it is the default.
typedef E16 bf16x8 __attribute__((ext_vector_type(8)));
typedef E16 bf16x4 __attribute__((ext_vector_type(4)));
typedef E16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int TN = 128;      // pixel tile (forward) / (tap, ci) tile (weight gradient)
constexpr int TK = 32;       // K per tile: two MFMA k-steps of 16
constexpr int LP = 40;       // LDS row pitch in bf16 (80 B: 16-byte aligned rows, 20-bank stride)

struct Geom {
    int N, Cin, H, W, Cout, R, S, sh, sw, ph, pw, OH, OW;
    int P;          // N*OH*OW
    int flip;       // taps of the weight applied in reverse order (data gradient)
    int out_nhwc;
    const int *n_live;   // optional DEVICE count of real images (forward) / rows (weight gradient used as a GEMM): tiles that
                         // lie completely past it are skipped -- see fi_conv2d_forward_live in csrc/conv_igemm.hip
};

struct Epi {
    const float *bias, *scale, *residual;
    int relu;
    const float *gate;      // shaped like y, or null: y *= (gate > 0) (conv_igemm.hip's Epilogue::gate)
};

__device__ __forceinline__ bf16x2 pack2(float a, float b)
{
    bf16x2 r;
    r.x = (E16)a;
    r.y = (E16)b;
    return r;
}

// ------------------------------------------------------------------------------------------------
// forward / data gradient
// ------------------------------------------------------------------------------------------------
// The GEMM's pixel axis runs over a VIRTUAL pixel space in which every output row is padded to a multiple
// of 4 pixels: pv = ((n*OH + oh)*OWQ + qx)*4 + j, OWQ = ceil(OW/4).  A quad of 4 consecutive virtual pixels
// then never straddles an output row, so (stride 1) its 4 taps are 4 consecutive floats of ONE input row:
// one 16-byte load at a column clamped into the row, plus a register shift where the quad hangs over the
// left / right halo -- no divergent per-pixel path (on 14x14 maps half of all quads touch a row end).
// Padding pixels are computed and dropped (14 -> 16: 14 % more MFMA work on the RoI heads, 0 % on maps whose
// width is a multiple of 4).
__device__ __forceinline__ float pick4(const f32x4 &v, int i)
{
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// NCHW epilogue of a tile that lies completely inside the output (all 32 rows of the lane's wavefront block
// below Cout, every quad complete): no predicate, one basic block, so the scale / bias / shortcut loads of a
// group of 4 rows are issued together instead of one dependent round trip each (the general path's ISA is a
// chain of `global_load; s_waitcnt vmcnt(0)`: +40 % on a fused 1x1 layer).  Absent scale / bias are dropped by
// a select so the arithmetic stays `acc [*scale] [+bias] [+shortcut] [relu]`.
template <int NT, bool HAS_RES, bool HAS_GATE>
__device__ __forceinline__ void epilogue_full_nchw(const f32x16 (&acc)[NT], const Epi &ep, float *__restrict__ y,
                                                   size_t obase, int OHW, int mb)
{
    typedef float f32xN __attribute__((ext_vector_type(NT)));
    typedef f32xN f32xN_a4 __attribute__((aligned(4)));
    const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr, relu = ep.relu != 0;
    const float *__restrict__ sp = has_sc ? ep.scale + mb : y;        // any readable address when absent
    const float *__restrict__ bp = has_bi ? ep.bias + mb : y;
    const int smul = has_sc ? 1 : 0, bmul = has_bi ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float sc[4], bi[4];
        f32xN rr[4], gg[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            sc[e4] = sp[(8 * q + e4) * smul];
            bi[e4] = bp[(8 * q + e4) * bmul];
            if (HAS_RES) rr[e4] = *reinterpret_cast<const f32xN_a4 *>(ep.residual + obase + (size_t)(8 * q + e4) * OHW);
            if (HAS_GATE) gg[e4] = *reinterpret_cast<const f32xN_a4 *>(ep.gate + obase + (size_t)(8 * q + e4) * OHW);
        }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            f32xN v;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float t = acc[j][4 * q + e4];
                t = has_sc ? t * sc[e4] : t;
                t = has_bi ? t + bi[e4] : t;
                if (HAS_RES) t += rr[e4][j];
                t = relu ? fmaxf(t, 0.0f) : t;
                if (HAS_GATE) t = gg[e4][j] > 0.0f ? t : 0.0f;
                v[j] = t;
            }
            *reinterpret_cast<f32xN_a4 *>(y + obase + (size_t)(8 * q + e4) * OHW) = v;
        }
    }
}

// SWT = column stride of the fast gather (1 or 2).  Stride 2: the 4 outputs of a quad tap input columns
// iw0, iw0 + 2, iw0 + 4, iw0 + 6 -- two 16-byte loads (8 consecutive floats), every second element.
template <int BM, int SWT>
__global__ __launch_bounds__(kThreads, 2) void conv_bf16_fwd_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ w, Epi ep,
                                                                 float *__restrict__ y, Geom g, int mtiles,
                                                                 int ptiles)
{
    // wavefront layout: WM x WN wavefronts, each a 32 x (128 / WN) block of the tile.  BM = 128: 4 x 1 -- a
    // lane then owns all 4 pixels of its quad (see the row permutation of the B tile), so the epilogue
    // stores 16 bytes per lane; BM = 64: 2 x 2 (two adjacent pixels per lane).
    constexpr int WM = BM / 32, WN = 4 / WM, NT = (TN / WN) / 32;
    __shared__ __align__(16) E16 As[2][BM][LP];
    __shared__ __align__(16) E16 Bs[2][TN][LP];

    // XCD-aware tile order: block b runs on XCD b % 8; the Cout tiles of one pixel tile follow each other on
    // the same XCD, so the activation tile is fetched into ONE L2 and re-read there
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int pt = (seq / mtiles) * 8 + xcd;
    if (pt >= ptiles) return;
    const int m0 = (seq % mtiles) * BM;
    const int p0 = pt * TN;
    if (g.n_live && p0 >= *g.n_live * g.OH * ((g.OW + 3) >> 2) * 4) return;      // (virtual pixels: rows padded to quads)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int RS = g.R * g.S;
    const int OHW = g.OH * g.OW;
    const int OWQ = (g.OW + 3) >> 2;
    const int PV = g.N * g.OH * OWQ * 4;                     // virtual pixels
    const size_t HW = (size_t)g.H * g.W;

    // ---- A loader: thread -> (row, 16-float half of the 32-wide K slice) --------------------------
    const int a_row = tid >> 1, a_half = tid & 1;
    const bool a_on = a_row < BM;
    const int a_m = min(m0 + a_row, g.Cout - 1);             // rows past Cout re-read the last row
    const float *__restrict__ a_src = w + (size_t)a_m * RS * g.Cin + a_half * 16;

    // ---- B loader: thread -> (channel pair kp, quads q0 and q0 + 16 of the tile) ------------------
    const int kp = tid >> 4;                                 // 0..15 -> channels 2kp, 2kp+1 of the slice
    const int q0 = tid & 15;
    int b_oh[2], b_ow[2], b_ok[2];
    const float *b_img[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pv = p0 + 4 * (q0 + 16 * u);
        b_ok[u] = pv < PV;
        const int quad = min(pv, PV - 4) >> 2;               // (n*OH + oh)*OWQ + qx
        const int row = quad / OWQ;
        b_ow[u] = (quad - row * OWQ) * 4;
        const int n = row / g.OH;
        b_oh[u] = row - n * g.OH;
        b_img[u] = x + (size_t)n * g.Cin * HW + (size_t)(2 * kp) * HW;
    }
    constexpr int NL = SWT;                                  // 16-byte loads per channel and quad
    const bool fast = (g.sw == SWT) && (g.W >= 4 * SWT);

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    const int cblocks = g.Cin / TK;
    const int ktiles = RS * cblocks;

    struct Regs {
        f32x4 a[4];            // A: 16 consecutive k of one row
        f32x4 b[2][2][NL];     // B: [quad][channel of the pair][16-byte load] raw input columns
        int d[2];              // B: column shift of the quad's load (0 inside the row) / -99: all zero
    };

    // Load cursor: K-tiles are visited tap-major, channel blocks inside a tap.  Everything that depends on
    // the tap (input row, clamped column, halo shift, weight tap) is computed once per tap; inside a tap a
    // tile is three pointer increments -- no index arithmetic in the steady state.
    int cur_tap = 0, cur_cb = 0;
    const float *pa = a_src;
    const float *pb[2] = {b_img[0], b_img[1]};
    int dcur[2] = {0, 0};
    const size_t cstep = (size_t)TK * HW;
    auto begin_tap = [&](int tap) {
        const int r = tap / g.S, s_ = tap - r * g.S;
        pa = a_src + (size_t)(g.flip ? (RS - 1 - tap) : tap) * g.Cin;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ih = b_oh[u] * g.sh - g.ph + r;
            const int iw0 = b_ow[u] * SWT - g.pw + s_;
            const int iwc = min(max(iw0, 0), g.W - 4 * SWT);
            const bool row_ok = b_ok[u] && ih >= 0 && ih < g.H && iw0 > -4 * SWT && iw0 < g.W;
            pb[u] = b_img[u] + (size_t)min(max(ih, 0), g.H - 1) * g.W + iwc;
            dcur[u] = row_ok ? (iw0 - iwc) : -99;
        }
    };
    if (fast) begin_tap(0);

    auto load_tile = [&](Regs &R) {
        if (fast) {
            if (a_on) {
#pragma unroll
                for (int v = 0; v < 4; ++v) R.a[v] = *reinterpret_cast<const f32x4 *>(pa + 4 * v);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    R.b[u][0][l] = *reinterpret_cast<const f32x4_a4 *>(pb[u] + 4 * l);
                    R.b[u][1][l] = *reinterpret_cast<const f32x4_a4 *>(pb[u] + HW + 4 * l);
                }
                R.d[u] = dcur[u];
                pb[u] += cstep;
            }
            pa += TK;
            if (++cur_cb == cblocks) {
                cur_cb = 0;
                if (++cur_tap < RS) begin_tap(cur_tap);
            }
        } else {
            // strided layers / maps narrower than 4: guarded scalar gather
            const int tap = cur_tap, c0 = cur_cb * TK;
            const int r = tap / g.S, s_ = tap - r * g.S;
            if (a_on) {
                const float *p = a_src + (size_t)(g.flip ? (RS - 1 - tap) : tap) * g.Cin + c0;
#pragma unroll
                for (int v = 0; v < 4; ++v) R.a[v] = *reinterpret_cast<const f32x4 *>(p + 4 * v);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float *px = b_img[u] + (size_t)c0 * HW;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v0 = 0.0f, v1 = 0.0f;
                    const int ih = b_oh[u] * g.sh - g.ph + r, iw = (b_ow[u] + j) * g.sw - g.pw + s_;
                    if (b_ok[u] && b_ow[u] + j < g.OW && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) {
                        v0 = px[(size_t)ih * g.W + iw];
                        v1 = px[HW + (size_t)ih * g.W + iw];
                    }
                    R.b[u][0][0][j] = v0;
                    R.b[u][1][0][j] = v1;
                }
                R.d[u] = -77;                                // values are already in place
            }
            if (++cur_cb == cblocks) {
                cur_cb = 0;
                ++cur_tap;
            }
        }
    };

    // LDS rows of the B tile are PERMUTED: pixel j of quad q lives in row j*32 + q.  The four stores of a
    // thread (one per pixel of its quad) then go, across the 64 lanes, to 16 consecutive rows x 4 channel
    // pairs = 64 different banks (row pitch 20 dwords); the natural order 4q + j put 16 lanes on 4 banks.
    // The MFMA column index n therefore means pixel 4*(n & 31) + (n >> 5) -- see the epilogue.
    auto store_tile = [&](int buf, const Regs &R) {
        if (a_on) {
            bf16x8 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = (E16)R.a[0][e];
                lo[4 + e] = (E16)R.a[1][e];
                hi[e] = (E16)R.a[2][e];
                hi[4 + e] = (E16)R.a[3][e];
            }
            *reinterpret_cast<bf16x8 *>(&As[buf][a_row][a_half * 16]) = lo;
            *reinterpret_cast<bf16x8 *>(&As[buf][a_row][a_half * 16 + 8]) = hi;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = q0 + 16 * u;
            f32x4 c0v = R.b[u][0][0], c1v = R.b[u][1][0];
            const int d = R.d[u];
            if (d != -77 && (SWT == 2 || d != 0)) {          // quad over a row end / outside, or strided gather
                f32x4 t0, t1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = SWT * j + d;              // source element; outside 0..4*SWT-1 <=> outside the row
                    const bool ok = (unsigned)k < (unsigned)(4 * SWT);
                    float e0 = pick4(R.b[u][0][0], k & 3), e1 = pick4(R.b[u][1][0], k & 3);
                    if (SWT == 2 && k >= 4) {
                        e0 = pick4(R.b[u][0][NL - 1], k & 3);
                        e1 = pick4(R.b[u][1][NL - 1], k & 3);
                    }
                    t0[j] = ok ? e0 : 0.0f;
                    t1[j] = ok ? e1 : 0.0f;
                }
                c0v = t0;
                c1v = t1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<bf16x2 *>(&Bs[buf][j * 32 + q][2 * kp]) = pack2(c0v[j], c1v[j]);
        }
    };

    auto mma = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[NT];
            const bf16x8 af = *reinterpret_cast<const bf16x8 *>(&As[cur][wm * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][wn * (TN / WN) + j * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = MFMA16(af, bfr[j], acc[j], 0, 0, 0);
        }
    };

    // Two K-tiles of global loads in flight (register sets R0 / R1), two LDS buffers: while tile t is in the
    // MFMAs, tile t+1 sits in registers waiting to be converted and tile t+2's loads are being issued.
    Regs R0, R1;
    load_tile(R0);
    if (ktiles > 1) load_tile(R1);
    store_tile(0, R0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; kt += 2) {
        // even tile: LDS buffer 0; R1 holds tile kt+1; R0 is free for tile kt+2
        if (kt + 2 < ktiles) load_tile(R0);
        mma(0);
        if (kt + 1 < ktiles) store_tile(1, R1);
        __syncthreads();
        // odd tile: LDS buffer 1; R0 holds tile kt+2; R1 is free for tile kt+3
        if (kt + 1 < ktiles) {
            if (kt + 3 < ktiles) load_tile(R1);
            mma(1);
            if (kt + 2 < ktiles) store_tile(0, R0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) ---------
    // column n = wn*(128/WN) + j*32 + l31 is pixel 4*l31 + (NT*wn + j) of the tile (row permutation above): a
    // lane holds NT ADJACENT pixels of quad l31 -> one 16-byte (NT = 4) or 8-byte (NT = 2) NCHW store
    {
        const int jx0 = NT * wn;
        const int pv = p0 + 4 * l31 + jx0;
        const int quad = pv >> 2;
        const int row = quad / OWQ;
        const int ow = (quad - row * OWQ) * 4 + jx0;
        const int n = row / g.OH;
        const int rem = (row - n * g.OH) * g.OW + ow;
        const int nvalid = pv < PV ? min(NT, g.OW - ow) : 0;            // pixels of this lane inside the row
        const size_t p = (size_t)n * OHW + rem;
        const int mb = m0 + wm * 32 + 4 * lh;
        // workgroup-uniform: the whole tile is inside the output and every quad is complete
        const bool full = !g.out_nhwc && (g.OW & 3) == 0 && p0 + TN <= PV && m0 + BM <= g.Cout;
        if (full) {
            const size_t obase = ((size_t)n * g.Cout + mb) * OHW + rem;
            if (ep.gate) {
                if (ep.residual)
                    epilogue_full_nchw<NT, true, true>(acc, ep, y, obase, OHW, mb);
                else
                    epilogue_full_nchw<NT, false, true>(acc, ep, y, obase, OHW, mb);
            } else if (ep.residual)
                epilogue_full_nchw<NT, true, false>(acc, ep, y, obase, OHW, mb);
            else
                epilogue_full_nchw<NT, false, false>(acc, ep, y, obase, OHW, mb);
            return;
        }
        if (nvalid > 0) {
            if (g.out_nhwc) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (j >= nvalid) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = mb + 8 * q;
                        if (m >= g.Cout) continue;                       // Cout % 4 == 0
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = acc[j][4 * q + e];
                            if (ep.scale) t *= ep.scale[m + e];
                            if (ep.bias) t += ep.bias[m + e];
                            if (ep.relu) t = fmaxf(t, 0.0f);
                            v[e] = t;
                        }
                        *reinterpret_cast<f32x4 *>(y + (p + j) * g.Cout + m) = v;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    if (m >= g.Cout) continue;
                    const float sc = ep.scale ? ep.scale[m] : 1.0f;
                    const float bi = ep.bias ? ep.bias[m] : 0.0f;
                    const size_t o = ((size_t)n * g.Cout + m) * OHW + rem;
                    float t[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) t[j] = acc[j][e] * sc + bi;
                    if (nvalid == NT) {
                        if constexpr (NT == 4) {
                            if (ep.residual) {
                                const f32x4 rr = *reinterpret_cast<const f32x4_a4 *>(ep.residual + o);
#pragma unroll
                                for (int j = 0; j < 4; ++j) t[j] += rr[j];
                            }
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = ep.relu ? fmaxf(t[j], 0.0f) : t[j];
                            if (ep.gate) {
                                const f32x4 gg = *reinterpret_cast<const f32x4_a4 *>(ep.gate + o);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = gg[j] > 0.0f ? v[j] : 0.0f;
                            }
                            *reinterpret_cast<f32x4_a4 *>(y + o) = v;
                        } else {
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            typedef f32x2 f32x2_a4 __attribute__((aligned(4)));
                            if (ep.residual) {
                                const f32x2 rr = *reinterpret_cast<const f32x2_a4 *>(ep.residual + o);
                                t[0] += rr.x;
                                t[1] += rr.y;
                            }
                            f32x2 v;
                            v.x = ep.relu ? fmaxf(t[0], 0.0f) : t[0];
                            v.y = ep.relu ? fmaxf(t[1], 0.0f) : t[1];
                            if (ep.gate) {
                                const f32x2 gg = *reinterpret_cast<const f32x2_a4 *>(ep.gate + o);
                                v.x = gg.x > 0.0f ? v.x : 0.0f;
                                v.y = gg.y > 0.0f ? v.y : 0.0f;
                            }
                            *reinterpret_cast<f32x2_a4 *>(y + o) = v;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            if (j >= nvalid) continue;
                            float u = t[j];
                            if (ep.residual) u += ep.residual[o + j];
                            u = ep.relu ? fmaxf(u, 0.0f) : u;
                            if (ep.gate) u = ep.gate[o + j] > 0.0f ? u : 0.0f;
                            y[o + j] = u;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 forward + data gradient: input patch in LDS (the bf16 twin of conv3x3_patch_kernel)
// ------------------------------------------------------------------------------------------------
// conv_bf16_fwd_kernel is ISSUE bound (~200 VALU + 190 SALU instructions per 8 MFMAs: address arithmetic, halo
// selects, fp32->bf16 packs for every tap).  Here a workgroup owns 8 rows x 16 columns of output pixels x 128
// output channels and stages, per block of 32 input channels, the (8+2) x 24 input patch ONCE, already
// converted: a thread loads 4 pixels of 8 channels (eight 16-byte loads), packs them and writes four 16-byte
// LDS slots, one per pixel, each holding the pixel's 8 channels = one MFMA B-fragment.  The nine taps are nine
// slot offsets.  Slots of a patch row are ordered by (column mod 4, column div 4) and rows are 28 slots apart, so
// that the 16 lanes of a quarter-wavefront -- 4 rows x 4 quads, each lane reading pixel 4q + j + s -- hit all 32
// banks exactly twice: the minimum for 256 bytes.  Weights skip LDS: a lane loads the 8 channels of its
// (output channel, tap, k-step) from the tap-major fp32 weight two taps ahead and packs them.
// Per 32 channels: 72 MFMAs (32x32x16) per wavefront, one barrier.
constexpr int PB_TH = 8, PB_TW = 16;
constexpr int PB_RP = 28;                       // slots per patch row (24 used)
constexpr int PB_ROWS = PB_TH + 2;
constexpr int PB_NSLOT = (PB_ROWS + 1) * PB_RP; // + one row of zeros
// FLAT tiles (maps of 12..16 columns, e.g. the 14 x 14 RoI maps -- see conv3x3_patch_kernel<true> in conv_igemm.hip):
// a tile is 128 CONSECUTIVE pixels of the flattened (stacked row, column) space; up to 11 rows + halo
constexpr int PB_ROWS_FLAT = 13;
constexpr int PB_NSLOT_FLAT = (PB_ROWS_FLAT + 1) * PB_RP;
constexpr int PB_CB = 32;                       // channels per stage = 4 groups of 8

// staging loads of threads whose patch group lies outside the image are redirected here (no branch around the loads)
__device__ __attribute__((aligned(16))) float g_zero_quad[4];

struct PatchGeomB {
    int N, Cin, H, W, Cout;
    int flip, tiles_x, ptiles, mtiles;
};

__device__ __forceinline__ bf16x8 pack8(const f32x4 &lo, const f32x4 &hi)
{
    bf16x8 r;
    r[0] = (E16)lo.x; r[1] = (E16)lo.y; r[2] = (E16)lo.z; r[3] = (E16)lo.w;
    r[4] = (E16)hi.x; r[5] = (E16)hi.y; r[6] = (E16)hi.z; r[7] = (E16)hi.w;
    return r;
}

// WB16: the weights are already bf16 (pre-converted once per step by the caller): a lane's A-operand is ONE 16-byte
// load per (tap, k-step), no packing.  With fp32 weights every wavefront streams 37 KB of weights per 32 channels
// from L2 -- 8 wavefronts of a CU ask for more than the L2->L1 path delivers at the bf16 MFMA rate.
template <bool WB16, bool FLAT = false>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_patch_bf16_kernel(const float *__restrict__ x,
                                                                        const void *__restrict__ wv, Epi ep,
                                                                        float *__restrict__ y, PatchGeomB g)
{
    constexpr int ROWS = FLAT ? PB_ROWS_FLAT : PB_ROWS;
    constexpr int NSLOT = FLAT ? PB_NSLOT_FLAT : PB_NSLOT;
    constexpr int SGRP = FLAT ? 4 : 6;            // staged 4-column groups per patch row (flat: columns 0..15 only)
    const float *__restrict__ w = static_cast<const float *>(wv);
    const E16 *__restrict__ wb = static_cast<const E16 *>(wv);
    __shared__ __align__(16) bf16x8 Ps[2][PB_CB / 8][NSLOT];

    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_xcd = (g.ptiles + 7) >> 3;
    const int mt = local % g.mtiles;
    const int pt = xcd * per_xcd + local / g.mtiles;
    if (pt >= g.ptiles) return;
    const int tile_y = FLAT ? 0 : pt / g.tiles_x, tile_x = FLAT ? 0 : pt - tile_y * g.tiles_x;
    // 2-D: first stacked row / column of the tile.  Flat: stacked row of the tile's first pixel; the patch always
    // starts at image column -4
    const int Y0 = FLAT ? (pt * 128) / g.W : tile_y * PB_TH, X0 = FLAT ? 0 : tile_x * PB_TW, m0 = mt * 128;
    const int NH = g.N * g.H;
    const size_t HW = (size_t)g.H * g.W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;

    // ---- A operand ------------------------------------------------------------------------------------
    const int am = min(m0 + wave * 32 + l31, g.Cout - 1);
    const float *__restrict__ a_base = w + (size_t)am * 9 * g.Cin + khalf * 8;
    const E16 *__restrict__ ab_base = wb + (size_t)am * 9 * g.Cin + khalf * 8;
    auto load_a_raw = [&](f32x4 (&raw)[4], int tap, int cb) {          // both k-steps of a tap
        const float *__restrict__ p = a_base + (size_t)(g.flip ? 8 - tap : tap) * g.Cin + cb * PB_CB;
        raw[0] = *reinterpret_cast<const f32x4 *>(p);
        raw[1] = *reinterpret_cast<const f32x4 *>(p + 4);
        raw[2] = *reinterpret_cast<const f32x4 *>(p + 16);
        raw[3] = *reinterpret_cast<const f32x4 *>(p + 20);
    };
    auto load_a_b16 = [&](bf16x8 (&pk)[2], int tap, int cb) {
        const E16 *__restrict__ p = ab_base + (size_t)(g.flip ? 8 - tap : tap) * g.Cin + cb * PB_CB;
        pk[0] = *reinterpret_cast<const bf16x8 *>(p);
        pk[1] = *reinterpret_cast<const bf16x8 *>(p + 16);
    };

    // ---- B operand: slot of (patch row of tap row r, quad) ---------------------------------------------
    const int ty = l31 >> 2, q = l31 & 3;
    const int Yo = Y0 + ty;
    const int yo = Yo - (Yo / g.H) * g.H;
    int rowslot[3];
    rowslot[0] = (yo != 0 ? ty : PB_ROWS) * PB_RP + q;
    rowslot[1] = (ty + 1) * PB_RP + q;
    rowslot[2] = (yo != g.H - 1 ? ty + 2 : PB_ROWS) * PB_RP + q;
    // flat: the lane's 4 pixels may sit on two rows -> one slot per (kernel row, kernel column, pixel)
    int fslot[FLAT ? 3 : 1][FLAT ? 3 : 1][FLAT ? 4 : 1];
    int fY[4], fx[4];
    if constexpr (FLAT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pt * 128 + 4 * l31 + j;
            fY[j] = p / g.W;
            fx[j] = p - fY[j] * g.W;
            const int yj = fY[j] - (fY[j] / g.H) * g.H;
            const int pr = fY[j] - Y0 + 1;                         // patch row of the pixel's own row
#pragma unroll
            for (int r_ = 0; r_ < 3; ++r_) {
                const int prow = r_ == 0 ? (yj != 0 ? pr - 1 : ROWS) : (r_ == 1 ? pr : (yj != g.H - 1 ? pr + 1 : ROWS));
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    const int c = fx[j] + s_ + 3;                  // patch column (column 0 = image column -4)
                    fslot[r_][s_][j] = prow * PB_RP + (c & 3) * 6 + (c >> 2);
                }
            }
        }
    }

    // ---- staging: thread = (channel group of 8, patch row, 4 columns) -----------------------------------
    const bool s_item = tid < (PB_CB / 8) * ROWS * SGRP;
    const int s_kg = tid / (ROWS * SGRP);
    const int s_rem = tid - s_kg * (ROWS * SGRP);
    const int s_prow = s_rem / SGRP, s_grp = s_rem - s_prow * SGRP + (FLAT ? 1 : 0);
    const int s_Ys = Y0 - 1 + s_prow, s_xx = X0 - 4 + s_grp * 4;
    // 2-D (W % 16 == 0): whole groups.  Flat (W even): a group that hangs over the row end by 2 columns is loaded 2
    // columns to the left (inside the row) and moved into place when it is stored
    const int s_sh = FLAT && s_xx < g.W && s_xx + 3 >= g.W ? s_xx + 4 - g.W : 0;
    const bool s_ok = s_item && s_Ys >= 0 && s_Ys < NH && s_xx >= 0 && s_xx < g.W;
    const int s_n = s_ok ? s_Ys / g.H : 0;
    const size_t s_off = ((size_t)s_n * g.Cin + s_kg * 8) * HW + (s_ok ? (size_t)(s_Ys - s_n * g.H) * g.W + s_xx - s_sh : 0);
    f32x4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7;
    // Branch-free: a group outside the image reads the 16 zero bytes of g_zero_quad eight times (channel stride 0).
    // With a branch around the loads the compiler kept the eight registers in scratch and waited for the loads at
    // the join, i.e. at the top of every channel block.
    const float *__restrict__ s_base = s_ok ? x + s_off : g_zero_quad;
    const size_t s_cs = s_ok ? HW : 0;                     // channel stride
    auto stage_load = [&](int cb) {
        const float *__restrict__ p = s_base + (size_t)cb * PB_CB * s_cs;
        typedef typename std::conditional<FLAT, f32x4_a4, f32x4>::type ld_t;      // flat rows start 8-byte aligned
        sr0 = *reinterpret_cast<const ld_t *>(p);
        sr1 = *reinterpret_cast<const ld_t *>(p + s_cs);
        sr2 = *reinterpret_cast<const ld_t *>(p + 2 * s_cs);
        sr3 = *reinterpret_cast<const ld_t *>(p + 3 * s_cs);
        sr4 = *reinterpret_cast<const ld_t *>(p + 4 * s_cs);
        sr5 = *reinterpret_cast<const ld_t *>(p + 5 * s_cs);
        sr6 = *reinterpret_cast<const ld_t *>(p + 6 * s_cs);
        sr7 = *reinterpret_cast<const ld_t *>(p + 7 * s_cs);
    };
    auto stage_store = [&](int buf) {
        if (!s_item) return;
        if (FLAT && s_sh != 0) {                               // s_sh == 2: elements 2, 3 are columns W-2, W-1; then halo
            auto mv = [](f32x4 &v) { v = f32x4{v.z, v.w, 0.0f, 0.0f}; };
            mv(sr0); mv(sr1); mv(sr2); mv(sr3); mv(sr4); mv(sr5); mv(sr6); mv(sr7);
        }
        bf16x8 *__restrict__ d = &Ps[buf][s_kg][s_prow * PB_RP + s_grp];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                          // pixel i of the group: column c = 4*grp + i -> slot i*6 + grp
            bf16x8 v;
            v[0] = (E16)sr0[i]; v[1] = (E16)sr1[i]; v[2] = (E16)sr2[i]; v[3] = (E16)sr3[i];
            v[4] = (E16)sr4[i]; v[5] = (E16)sr5[i]; v[6] = (E16)sr6[i]; v[7] = (E16)sr7[i];
            d[i * 6] = v;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // the row of zeros (both buffers, all channel groups); flat: also the never-staged outer column groups
    if constexpr (FLAT) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (E16)0.0f;
        bf16x8 *__restrict__ pz = &Ps[0][0][0];
        for (int i = tid; i < 2 * (PB_CB / 8) * NSLOT; i += kThreads) pz[i] = z;
        __syncthreads();
    } else {
        for (int i = tid; i < 2 * (PB_CB / 8) * PB_RP; i += kThreads) {
            const int b = i / ((PB_CB / 8) * PB_RP), r_ = i - b * ((PB_CB / 8) * PB_RP);
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (E16)0.0f;
            Ps[b][r_ / PB_RP][PB_ROWS * PB_RP + (r_ % PB_RP)] = z;
        }
    }
    const int ncb = g.Cin / PB_CB;
    stage_load(0);
    stage_store(0);
    __syncthreads();

    // 18 sub-steps (9 taps x 2 k-steps) of 4 MFMAs per stage.  Weights: the raw fp32 values of tap T+2 are loaded
    // while tap T is in the MFMAs and packed at the end of tap T+1; three register sets rotate (9 % 3 == 0, so the
    // pattern is the same in every stage and runs across stage boundaries).
    f32x4 araw[3][4];
    bf16x8 apk[3][2];
    bf16x8 bfr[2][4];                // [sub-step parity][pixel j]
    if (WB16) {
        load_a_b16(apk[0], 0, 0);
        load_a_b16(apk[1], 1, 0);
    } else {
        load_a_raw(araw[0], 0, 0);
        load_a_raw(araw[1], 1, 0);
        apk[0][0] = pack8(araw[0][0], araw[0][1]);
        apk[0][1] = pack8(araw[0][2], araw[0][3]);
    }
    for (int cb = 0; cb < ncb; ++cb) {
        const bf16x8 *__restrict__ pbuf = &Ps[cb & 1][khalf][0];           // k-step ks adds 2 channel groups
        const bool more = cb + 1 < ncb;
        auto bload = [&](bf16x8 (&bv)[4], int r_, int s_, int ks) {
            if constexpr (FLAT) {
                const bf16x8 *__restrict__ bp = pbuf + ks * 2 * NSLOT;
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = bp[fslot[r_][s_][j]];
            } else {
                const bf16x8 *__restrict__ bp = pbuf + ks * 2 * NSLOT + rowslot[r_];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = bp[((j + s_ + 3) & 3) * 6 + ((j + s_ + 3) >> 2)];
            }
        };
        bload(bfr[0], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                 // LDS reads of sub-step 0
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const bool ld = (t + 2 < 9) || more;
            const int tap2 = t + 2 < 9 ? t + 2 : t + 2 - 9, cb2 = t + 2 < 9 ? cb : cb + 1;
            if (ld) {
                if (WB16)
                    load_a_b16(apk[(t + 2) % 3], tap2, cb2);
                else
                    load_a_raw(araw[(t + 2) % 3], tap2, cb2);
            }
            // the next patch is requested behind the weights of tap 2 (no wait covers it before the one for tap 3's
            // weights), unconditionally: the last stage re-reads its own patch and drops it
            if (t == 0) stage_load(more ? cb + 1 : cb);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int step = t * 2 + ks;
                if (step + 1 < 18) {
                    const int t2 = (step + 1) >> 1, k2 = (step + 1) & 1;
                    bload(bfr[(step + 1) & 1], t2 / 3, t2 % 3, k2);
                }
                const bf16x8 av = apk[t % 3][ks];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = MFMA16(av, bfr[step & 1][j], acc[j], 0, 0, 0);
            }
            // fp32 weights: tap t+1 (of this stage or the first of the next) was loaded one tap ago, packed now
            if (!WB16 && (t + 1 < 9 || more)) {
                apk[(t + 1) % 3][0] = pack8(araw[(t + 1) % 3][0], araw[(t + 1) % 3][1]);
                apk[(t + 1) % 3][1] = pack8(araw[(t + 1) % 3][2], araw[(t + 1) % 3][3]);
            }
            // issue order of this tap: weight loads of tap t+2, then per k-step the LDS reads of the next sub-step
            // ahead of the 4 MFMAs, then the packs of tap t+1 (left alone the compiler sinks the loads to their use)
            if (ld) __builtin_amdgcn_sched_group_barrier(0x020, WB16 ? 2 : 4, 0);
            if (t == 0) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (t < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (!WB16 && (t + 1 < 9 || more)) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
        if (more) stage_store((cb + 1) & 1);
        __syncthreads();
    }

    if constexpr (FLAT) {
        // the lane's pixels (0,1) and (2,3) are two pairs, each inside one row (W even): 8-byte accesses
        const int mb = m0 + wave * 32 + 4 * khalf;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (fY[2 * h] >= NH) continue;
            const int n_img = fY[2 * h] / g.H;
            const size_t ob = ((size_t)n_img * g.Cout + mb) * HW + (size_t)(fY[2 * h] - n_img * g.H) * g.W + fx[2 * h];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int mo = (e & 3) + 8 * (e >> 2);
                if (mb + mo >= g.Cout) continue;
                const float sc = ep.scale ? ep.scale[mb + mo] : 1.0f;
                const float bi = ep.bias ? ep.bias[mb + mo] : 0.0f;
                const size_t o = ob + (size_t)mo * HW;
                float v0 = acc[2 * h][e] * sc + bi, v1 = acc[2 * h + 1][e] * sc + bi;
                if (ep.residual) {
                    const float2 rr = *reinterpret_cast<const float2 *>(ep.residual + o);
                    v0 += rr.x;
                    v1 += rr.y;
                }
                if (ep.relu) {
                    v0 = fmaxf(v0, 0.0f);
                    v1 = fmaxf(v1, 0.0f);
                }
                if (ep.gate) {
                    const float2 gg = *reinterpret_cast<const float2 *>(ep.gate + o);
                    v0 = gg.x > 0.0f ? v0 : 0.0f;
                    v1 = gg.y > 0.0f ? v1 : 0.0f;
                }
                *reinterpret_cast<float2 *>(y + o) = make_float2(v0, v1);
            }
        }
        return;
    }
    // ---- epilogue: the lane's quad x 16 channels (same layout as conv_bf16_fwd_kernel<128>) ----------------
    const int xo = X0 + 4 * q;
    if (Yo >= NH || xo >= g.W) return;
    const int n_img = Yo / g.H;
    const int mb = m0 + wave * 32 + 4 * khalf;
    const size_t obase = ((size_t)n_img * g.Cout + mb) * HW + (size_t)yo * g.W + xo;
    if (m0 + 128 <= g.Cout) {
        if (ep.gate) {
            if (ep.residual)
                epilogue_full_nchw<4, true, true>(acc, ep, y, obase, (int)HW, mb);
            else
                epilogue_full_nchw<4, false, true>(acc, ep, y, obase, (int)HW, mb);
        } else if (ep.residual)
            epilogue_full_nchw<4, true, false>(acc, ep, y, obase, (int)HW, mb);
        else
            epilogue_full_nchw<4, false, false>(acc, ep, y, obase, (int)HW, mb);
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m >= g.Cout) continue;
        const float sc = ep.scale ? ep.scale[m] : 1.0f;
        const float bi = ep.bias ? ep.bias[m] : 0.0f;
        const size_t o = obase + (size_t)((e & 3) + 8 * (e >> 2)) * HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[j][e] * sc + bi;
            if (ep.residual) v += ep.residual[o + j];
            v = ep.relu ? fmaxf(v, 0.0f) : v;
            if (ep.gate) v = ep.gate[o + j] > 0.0f ? v : 0.0f;
            y[o + j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 1x1 / stride 1 forward + data gradient: weights in registers, pixel tile staged 64 channels at a time
// (the bf16 twin of conv1x1_reg_kernel in conv_igemm.hip)
// ------------------------------------------------------------------------------------------------
// A workgroup owns 128 consecutive pixels (H*W % 4 == 0: a quad of 4 pixels never straddles two images) x 128
// output channels.  Per stage of 64 input channels a thread loads ONE quad of 8 channels (eight 16-byte loads, the
// 32 lanes of a half-wavefront read 512 contiguous bytes per channel), packs it and writes four 16-byte LDS slots,
// one per pixel = one MFMA B-fragment; slot = (pixel mod 4) * 32 + quad, so that the lanes of a wavefront -- lane
// l31 reads pixel 4*l31 + j -- read consecutive slots.  The bf16 weights (converted once per step by the caller) go
// straight into the MFMA A-operand: one 16-byte load per k-step.  16 MFMAs per wavefront and barrier.
constexpr int P1_KC = 64;                         // channels per stage = 8 groups of 8

struct Conv1x1GeomB {
    int N, Cin, HW, Cout, P;                      // P = N * HW
    int ptiles, mtiles;
};

__global__ __launch_bounds__(kThreads, 2) void conv1x1_bf16_kernel(const float *__restrict__ x,
                                                                  const E16 *__restrict__ wb, Epi ep,
                                                                  float *__restrict__ y, Conv1x1GeomB g)
{
    __shared__ __align__(16) bf16x8 Ps[2][P1_KC / 8][128];

    // XCD-aware order: XCD c owns a contiguous band of pixel tiles, Cout tiles innermost
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_xcd = (g.ptiles + 7) >> 3;
    const int mt = local % g.mtiles;
    const int pt = xcd * per_xcd + local / g.mtiles;
    if (pt >= g.ptiles) return;
    const int m0 = mt * 128, P0 = pt * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;

    // ---- A operand: 8 consecutive channels of one output channel per k-step ----------------------------
    const int am = min(m0 + wave * 32 + l31, g.Cout - 1);            // rows past Cout re-read the last row
    const E16 *__restrict__ a_base = wb + (size_t)am * g.Cin + khalf * 8;
    auto load_a = [&](bf16x8 (&a)[4], int cb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(a_base + cb * P1_KC + ks * 16);
    };

    // ---- staging: thread = (channel group of 8, quad) ----------------------------------------------------
    const int s_kg = tid >> 5, s_q = tid & 31;
    const int s_p = P0 + 4 * s_q;
    const bool s_ok = s_p < g.P;                                     // P % 4 == 0: whole quads
    const int s_n = s_ok ? s_p / g.HW : 0;
    const float *__restrict__ s_base =
        s_ok ? x + ((size_t)s_n * g.Cin + s_kg * 8) * g.HW + (s_p - s_n * g.HW) : g_zero_quad;
    const size_t s_cs = s_ok ? (size_t)g.HW : 0;                     // channel stride (0: the quad of zeros)
    f32x4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7;
    auto stage_load = [&](int cb) {
        const float *__restrict__ p = s_base + (size_t)cb * P1_KC * s_cs;
        sr0 = *reinterpret_cast<const f32x4 *>(p);
        sr1 = *reinterpret_cast<const f32x4 *>(p + s_cs);
        sr2 = *reinterpret_cast<const f32x4 *>(p + 2 * s_cs);
        sr3 = *reinterpret_cast<const f32x4 *>(p + 3 * s_cs);
        sr4 = *reinterpret_cast<const f32x4 *>(p + 4 * s_cs);
        sr5 = *reinterpret_cast<const f32x4 *>(p + 5 * s_cs);
        sr6 = *reinterpret_cast<const f32x4 *>(p + 6 * s_cs);
        sr7 = *reinterpret_cast<const f32x4 *>(p + 7 * s_cs);
    };
    auto stage_store = [&](int buf) {
        bf16x8 *__restrict__ d = &Ps[buf][s_kg][s_q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 v;
            v[0] = (E16)sr0[i]; v[1] = (E16)sr1[i]; v[2] = (E16)sr2[i]; v[3] = (E16)sr3[i];
            v[4] = (E16)sr4[i]; v[5] = (E16)sr5[i]; v[6] = (E16)sr6[i]; v[7] = (E16)sr7[i];
            d[i * 32] = v;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    const int ncb = g.Cin / P1_KC;
    bf16x8 a0[4], a1[4];                                  // weights of the even / odd stages (static register sets)
    stage_load(0);
    load_a(a0, 0);
    stage_store(0);
    __syncthreads();
    auto stage = [&](int cb, const bf16x8 (&acur)[4], bf16x8 (&anxt)[4]) {
        const bf16x8 *__restrict__ pbuf = &Ps[cb & 1][khalf][l31];
        const bool more = cb + 1 < ncb;
        // weights of the next stage, then the next pixel tile (unconditional: the last stage re-reads its own)
        load_a(anxt, more ? cb + 1 : cb);
        stage_load(more ? cb + 1 : cb);
        // issue order: the 12 global loads first (left alone the compiler sinks the 8 staging loads to their use at
        // the end of the stage), then per k-step the LDS reads of the next k-step ahead of the 4 MFMAs
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 bv[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[0][j] = pbuf[j * 32];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[(ks + 1) & 1][j] = pbuf[(ks + 1) * 2 * 128 + j * 32];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = MFMA16(acur[ks], bv[ks & 1][j], acc[j], 0, 0, 0);
            if (ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        // unconditional (the last stage writes its own tile into the idle buffer): under `if (more)` the compiler
        // sinks the staging loads into the branch, i.e. behind the MFMAs
        stage_store((cb + 1) & 1);
        __syncthreads();
    };
    for (int cb = 0; cb < ncb; cb += 2) {
        stage(cb, a0, a1);
        if (cb + 1 < ncb) stage(cb + 1, a1, a0);
    }

    // ---- epilogue: the lane's quad x 16 channels -----------------------------------------------------------
    const int p = P0 + 4 * l31;
    if (p >= g.P) return;
    const int n_img = p / g.HW;
    const int mb = m0 + wave * 32 + 4 * khalf;
    const size_t obase = ((size_t)n_img * g.Cout + mb) * g.HW + (p - n_img * g.HW);
    if (m0 + 128 <= g.Cout) {
        if (ep.gate) {
            if (ep.residual)
                epilogue_full_nchw<4, true, true>(acc, ep, y, obase, g.HW, mb);
            else
                epilogue_full_nchw<4, false, true>(acc, ep, y, obase, g.HW, mb);
        } else if (ep.residual)
            epilogue_full_nchw<4, true, false>(acc, ep, y, obase, g.HW, mb);
        else
            epilogue_full_nchw<4, false, false>(acc, ep, y, obase, g.HW, mb);
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m >= g.Cout) continue;
        const float sc = ep.scale ? ep.scale[m] : 1.0f;
        const float bi = ep.bias ? ep.bias[m] : 0.0f;
        const size_t o = obase + (size_t)((e & 3) + 8 * (e >> 2)) * g.HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[j][e] * sc + bi;
            if (ep.residual) v += ep.residual[o + j];
            v = ep.relu ? fmaxf(v, 0.0f) : v;
            if (ep.gate) v = ep.gate[o + j] > 0.0f ? v : 0.0f;
            y[o + j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient (tap-major dW [Cout][R*S][Cin]), stride 1 or general
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void conv_bf16_wgrad_generic_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ dy,
                                                                   float *__restrict__ dw, Geom g,
                                                                   int cin_tiles, int chunks_per_image,
                                                                   int chunk_pixels)
{
    constexpr int BM = 128;
    __shared__ __align__(16) E16 As[2][BM][LP];      // dY  [cout][pixel]
    __shared__ __align__(16) E16 Bs[2][TN][LP];      // X   [ci][pixel]   (for one tap)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int RS = g.R * g.S;
    const int OHW = g.OH * g.OW;
    const size_t HW = (size_t)g.H * g.W;

    const int m0 = blockIdx.x * BM;
    const int tap = blockIdx.y / cin_tiles;
    const int ci0 = (blockIdx.y - tap * cin_tiles) * TN;
    const int r = tap / g.S, s = tap - r * g.S;
    // this workgroup's share of the reduction: images and pixel chunks z, z + gridDim.z, ...
    const int total_chunks = g.N * chunks_per_image;

    // loader mapping (both tiles): thread -> (row = tid / 2 of 128, 16 consecutive pixels = 4 quads)
    const int row = tid >> 1, half = tid & 1;
    const int a_m = min(m0 + row, g.Cout - 1);
    const int b_c = min(ci0 + row, g.Cin - 1);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    f32x4 ra[4], rb[4];
    int cur_chunk = blockIdx.z;
    int kpos = 0;                                       // pixel offset inside the chunk of the NEXT tile to load
    bool more = cur_chunk < total_chunks;

    auto load_tile = [&]() {
        const int n = cur_chunk / chunks_per_image;
        const int pbase = (cur_chunk - n * chunks_per_image) * chunk_pixels + kpos + half * 16;
        const int pend = min(OHW, (cur_chunk - n * chunks_per_image + 1) * chunk_pixels);
        const float *pa = dy + ((size_t)n * g.Cout + a_m) * OHW;
        const float *pb = x + ((size_t)n * g.Cin + b_c) * HW;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int p = pbase + 4 * v;
            if (p + 3 < pend && (OHW & 3) == 0) {
                ra[v] = *reinterpret_cast<const f32x4_a4 *>(pa + p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[v][e] = (p + e < pend) ? pa[p + e] : 0.0f;
            }
            // the same 4 output pixels seen through the tap
            const int oh = p / g.OW, ow = p - oh * g.OW;
            const int ih = oh * g.sh - g.ph + r, iw = ow * g.sw - g.pw + s;
            if (g.sw == 1 && p + 3 < pend && ow + 3 < g.OW && ih >= 0 && ih < g.H && iw >= 0 && iw + 3 < g.W) {
                rb[v] = *reinterpret_cast<const f32x4_a4 *>(pb + (size_t)ih * g.W + iw);
            } else {
                int o2 = oh, w2 = ow;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = 0.0f;
                    if (p + e < pend) {
                        const int ih2 = o2 * g.sh - g.ph + r, iw2 = w2 * g.sw - g.pw + s;
                        if (ih2 >= 0 && ih2 < g.H && iw2 >= 0 && iw2 < g.W) t = pb[(size_t)ih2 * g.W + iw2];
                    }
                    rb[v][e] = t;
                    if (++w2 == g.OW) {
                        w2 = 0;
                        ++o2;
                    }
                }
            }
        }
        // advance to the next K-tile of this workgroup
        kpos += TK;
        const int clen = pend - (cur_chunk - n * chunks_per_image) * chunk_pixels;
        if (kpos >= clen) {
            kpos = 0;
            cur_chunk += gridDim.z;
            more = cur_chunk < total_chunks;
        }
    };

    auto store_tile = [&](int buf) {
        bf16x8 a0, a1, b0, b1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0[e] = (E16)ra[0][e];
            a0[4 + e] = (E16)ra[1][e];
            a1[e] = (E16)ra[2][e];
            a1[4 + e] = (E16)ra[3][e];
            b0[e] = (E16)rb[0][e];
            b0[4 + e] = (E16)rb[1][e];
            b1[e] = (E16)rb[2][e];
            b1[4 + e] = (E16)rb[3][e];
        }
        *reinterpret_cast<bf16x8 *>(&As[buf][row][half * 16]) = a0;
        *reinterpret_cast<bf16x8 *>(&As[buf][row][half * 16 + 8]) = a1;
        *reinterpret_cast<bf16x8 *>(&Bs[buf][row][half * 16]) = b0;
        *reinterpret_cast<bf16x8 *>(&Bs[buf][row][half * 16 + 8]) = b1;
    };

    if (!more) return;                                  // uniform: no share of the reduction
    load_tile();
    store_tile(0);
    __syncthreads();
    int cur = 0;
    while (true) {
        const bool have_next = more;
        if (have_next) load_tile();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&As[cur][wm * 64 + i * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][wn * 64 + j * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (!have_next) break;
        store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // dW[m][tap][ci]: lanes run over ci (contiguous) -> coalesced fp32 atomics
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 64 + j * 32 + l31;
        if (ci >= g.Cin) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < g.Cout) atomicAdd(dw + ((size_t)m * RS + tap) * g.Cin + ci, acc[i][j][e]);
            }
        }
    }
}

// Weight gradient, fast form (column stride 1 or 2, any row stride).  The reduction index runs over the same VIRTUAL pixel space as the
// forward kernel (rows padded to quads); a K-tile is 8 consecutive quads.  Loader: thread -> (quad of the
// tile = tid & 7, rows (tid >> 3) + 32 v): the 8 lanes of a row fetch its whole 128-byte slice of the
// K-tile, a thread owns ONE quad, so the quad's geometry (image, row, clamped columns, halo shift) is
// advanced incrementally once per tile -- no divisions, no per-element validity tests in the loop.
// Both operands use the clamped 16-byte load + register shift of the forward kernel: dY where the row
// width is not a multiple of 4, X where the tap pushes the quad over the halo.
template <int BM, int BNC, int SWT>
__global__ __launch_bounds__(kThreads, 2) void conv_bf16_wgrad_kernel(const float *__restrict__ x,
                                                                      const float *__restrict__ dy,
                                                                      float *__restrict__ dw, Geom g, int cin_tiles,
                                                                      int ktiles_total, int ktiles_per_split)
{
    constexpr int MT = BM / 64, NT = BNC / 64;
    constexpr int AV = BM / 32, BV = BNC / 32;          // row passes of the loaders
    constexpr int NL = SWT;                             // 16-byte loads per X row and quad (column stride 1 or 2)
    __shared__ __align__(16) E16 As[2][BM][LP];       // dY  [cout][pixel]
    __shared__ __align__(16) E16 Bs[2][BNC][LP];      // X   [ci][pixel]   (for one tap)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int RS = g.R * g.S;
    const int OHW = g.OH * g.OW;
    const int OWQ = (g.OW + 3) >> 2;
    const int TQ = g.N * g.OH * OWQ;                    // quads in the virtual pixel space
    const size_t HW = (size_t)g.H * g.W;

    const int m0 = blockIdx.x * BM;
    const int tap = blockIdx.y / cin_tiles;
    const int ci0 = (blockIdx.y - tap * cin_tiles) * BNC;
    const int r = tap / g.S, s_ = tap - r * g.S;
    const int kt_beg = blockIdx.z * ktiles_per_split;
    const int kt_end = min(ktiles_total, kt_beg + ktiles_per_split);
    if (kt_beg >= kt_end) return;

    const int q = tid & 7, r0 = tid >> 3;
    // rows this thread stages (clamped: rows past the end re-read the last row, their sums are never stored)
    size_t a_row_off[AV], b_row_off[BV];
#pragma unroll
    for (int v = 0; v < AV; ++v) a_row_off[v] = (size_t)min(m0 + r0 + 32 * v, g.Cout - 1) * OHW;
#pragma unroll
    for (int v = 0; v < BV; ++v) b_row_off[v] = (size_t)min(ci0 + r0 + 32 * v, g.Cin - 1) * HW;

    // the thread's quad: position and derived load state
    int qn, qoh, qx;
    {
        const int Q = kt_beg * 8 + q;
        const int row = Q / OWQ;
        qx = Q - row * OWQ;
        qn = row / g.OH;
        qoh = row - qn * g.OH;
    }
    const float *pa = dy;
    const float *pb = x;
    int da = -99, db = -99;
    auto locate = [&]() {
        const bool live = qn < g.N;
        const int n = min(qn, g.N - 1);
        const int ow0 = 4 * qx;
        const int owc = min(ow0, g.OW - 4);
        pa = dy + (size_t)n * g.Cout * OHW + (size_t)qoh * g.OW + owc;
        da = live ? (ow0 - owc) : -99;                    // elements j with j + da > 3 lie past the row end
        const int ih = qoh * g.sh - g.ph + r;
        const int iw0 = ow0 * SWT - g.pw + s_;
        const int iwc = min(max(iw0, 0), g.W - 4 * SWT);
        const bool ok = live && ih >= 0 && ih < g.H && iw0 > -4 * SWT && iw0 < g.W;
        pb = x + (size_t)n * g.Cin * HW + (size_t)min(max(ih, 0), g.H - 1) * g.W + iwc;
        db = ok ? (iw0 - iwc) : -99;
    };
    auto advance = [&]() {
        qx += 8;
        while (qx >= OWQ) {
            qx -= OWQ;
            if (++qoh == g.OH) {
                qoh = 0;
                ++qn;
            }
        }
        locate();
    };
    locate();

    struct Regs {
        f32x4 a[AV], b[BV][NL];
        int da, db;
    };
    auto load_tile = [&](Regs &R) {
#pragma unroll
        for (int v = 0; v < AV; ++v) R.a[v] = *reinterpret_cast<const f32x4_a4 *>(pa + a_row_off[v]);
#pragma unroll
        for (int v = 0; v < BV; ++v)
#pragma unroll
            for (int l = 0; l < NL; ++l) R.b[v][l] = *reinterpret_cast<const f32x4_a4 *>(pb + b_row_off[v] + 4 * l);
        R.da = da;
        R.db = db;
        advance();
    };
    auto shifted = [&](const f32x4 &v, int d) {
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = j + d;
            t[j] = ((unsigned)k < 4u) ? pick4(v, k) : 0.0f;
        }
        return t;
    };
    auto store_tile = [&](int buf, const Regs &R) {
        // dY: the quad may hang over the END of its row (da > 0): those pixels are padding -> zero.  A
        // shift never brings in pixels of the previous quad: j + da >= da.
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            f32x4 t = R.a[v];
            if (R.da != 0) t = shifted(t, R.da);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (E16)t[j];
            *reinterpret_cast<bf16x4 *>(&As[buf][r0 + 32 * v][4 * q]) = o;
        }
#pragma unroll
        for (int v = 0; v < BV; ++v) {
            f32x4 t = R.b[v][0];
            if (SWT == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 2 * j + R.db;               // element of the 8 loaded floats
                    const float e = k >= 4 ? pick4(R.b[v][NL - 1], k & 3) : pick4(R.b[v][0], k & 3);
                    t[j] = ((unsigned)k < 8u) ? e : 0.0f;
                }
            } else if (R.db != 0) {
                t = shifted(t, R.db);
            }
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (E16)t[j];
            *reinterpret_cast<bf16x4 *>(&Bs[buf][r0 + 32 * v][4 * q]) = o;
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto mma = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&As[cur][wm * (BM / 2) + i * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][wn * (BNC / 2) + j * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    // NOTE: a padding pixel of dY that is zeroed makes its whole K-column vanish, so the matching X values
    // (whatever the clamped load fetched) do not matter.
    const int ktiles = kt_end - kt_beg;
    Regs R0, R1;
    load_tile(R0);
    if (ktiles > 1) load_tile(R1);
    store_tile(0, R0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; kt += 2) {
        if (kt + 2 < ktiles) load_tile(R0);
        mma(0);
        if (kt + 1 < ktiles) store_tile(1, R1);
        __syncthreads();
        if (kt + 1 < ktiles) {
            if (kt + 3 < ktiles) load_tile(R1);
            mma(1);
            if (kt + 2 < ktiles) store_tile(0, R0);
        }
        __syncthreads();
    }

    // dW[m][tap][ci]: lanes run over ci (contiguous) -> coalesced fp32 atomics
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ci = ci0 + wn * (BNC / 2) + j * 32 + l31;
        if (ci >= g.Cin) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < g.Cout) atomicAdd(dw + ((size_t)m * RS + tap) * g.Cin + ci, acc[i][j][e]);
            }
        }
    }
    (void)TQ;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of same-size stride-1 layers (3x3 / pad 1 and 1x1): FLAT pixel space, 64-pixel K-tiles.
//
// conv_bf16_wgrad_kernel above spends ~50 vector instructions per MFMA (dynamic register shifts for the tap
// column, quad geometry with clamps, per-element picks), and on CDNA4 a wavefront's v_mfma and its other vector
// instructions do not overlap: it runs at 8 % of the bf16 peak.  For the layers that carry the time the input
// pixel of tap (r, s) is simply q + (r-1)*W + (s-1) for output pixel q of the same image, so 4 consecutive output
// pixels read 4 consecutive input floats -- across row ends too: ONE (4-byte aligned) 16-byte buffer load per
// group, no shift.  What is left per thread and 64-pixel tile: 16 loads, 32 v_cvt_pk, 16 ds_write_b64 and ~40
// instructions of pixel bookkeeping against 16 MFMAs (512 cycles) per wavefront.  Halo elements are zeroed on a
// divergent path that only wavefronts with a row end inside their tile take.  The pixel axis is the plain
// (n, y, x) order (requires H*W % 4 == 0 so that a group never straddles two images), i.e. no padded pixels.
// ------------------------------------------------------------------------------------------------
// operand pointers of a batched launch (fi_conv2d_weight_grad_batch_{bf16,f16}; see csrc/conv_igemm.hip), by value
struct WgradBatch16 {
    int n;
    const float *x[FI_WGRAD_BATCH_MAX];
    const float *dy[FI_WGRAD_BATCH_MAX];
    float *dw[FI_WGRAD_BATCH_MAX];
    float *db[FI_WGRAD_BATCH_MAX];
};

constexpr int FK = 64;             // pixels per K-tile: 4 MFMA k-steps of 16
constexpr int FLP = FK + 8;        // LDS row pitch in bf16 (144 B = 9 16-byte chunks: 16 rows cover all banks once)

template <int BM, int BNC, bool K3>
__global__ __launch_bounds__(kThreads, 2) void conv_bf16_wgrad_flat_kernel(const float *__restrict__ x_arg,
                                                                           const float *__restrict__ dy_arg,
                                                                           float *__restrict__ dw_arg, Geom g, int cin_tiles,
                                                                           int p_per_split, int mtiles, int splits,
                                                                           float *__restrict__ dbias_arg, WgradBatch16 wb)
{
    const float *__restrict__ x = x_arg;
    const float *__restrict__ dy = dy_arg;
    float *__restrict__ dw = dw_arg;
    float *__restrict__ dbias = dbias_arg;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int MT = BM / 64, NT = BNC / 64;
    constexpr int AV = BM / 16, BV = BNC / 16;          // row passes of the loaders: thread = (group tid & 15, row tid >> 4)
    __shared__ __align__(16) E16 As[2][BM][FLP];      // dY  [cout][pixel]
    __shared__ __align__(16) E16 Bs[2][BNC][FLP];     // X   [ci][pixel]   (one tap)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int RS = K3 ? 9 : 1;
    const int HW = g.H * g.W;

    // XCD-aware order (1-D launch, workgroup b runs on XCD b % 8): the workgroups, sorted by (pixel split, tile),
    // are cut into 8 contiguous bands.  All (tap, ci tile, cout tile) workgroups of a split walk the same dY / X
    // ranges, so a split is fetched into ONE L2 and re-read there (plain order: 36 tiles of a split on 8 XCDs,
    // 9.8 GB of operand reads per launch on the P2-level layers -- the kernel ran at the Infinity-Cache rate).
    const int tiles_per_split = mtiles * RS * cin_tiles;
    const int nblk1 = tiles_per_split * splits;
    const int nblk = wb.n ? nblk1 * wb.n : nblk1;
    const int per_xcd = (nblk + 7) >> 3;
    int idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (idx >= nblk) return;
    if (wb.n) {                                       // batched launch: (problem, split, tile) order
        const int prob = idx / nblk1;
        idx -= prob * nblk1;
        x = wb.x[prob];
        dy = wb.dy[prob];
        dw = wb.dw[prob];
        dbias = wb.db[prob];
    }
    const int bz = idx / tiles_per_split;
    const int tl = idx - bz * tiles_per_split;
    const int by = tl / mtiles, bx = tl - by * mtiles;
    const int m0 = bx * BM;
    if (g.n_live && m0 >= *g.n_live) return;             // the kernel as a GEMM (conv.linear): rows past the live count
    const int tap = by / cin_tiles;
    const int ci0 = (by - tap * cin_tiles) * BNC;
    const int dr = K3 ? tap / 3 - 1 : 0, ds = K3 ? tap - (tap / 3) * 3 - 1 : 0;
    const int off = dr * g.W + ds;
    const int p_begin = bz * p_per_split;
    const int p_end = min(g.P, p_begin + p_per_split);
    if (p_begin >= p_end) return;

    const int gq = tid & 15, lr = tid >> 4;
    const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(dy), 0, (int)((size_t)g.N * g.Cout * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (int)((size_t)g.N * g.Cin * HW * 4), 0x00020000);
    const int kOutOfRange = 0x7ffffff0;
    const int rs16 = 16 * HW * 4;                       // byte distance of two row passes
    // a NEGATIVE offset would zero the whole 16 bytes: the few groups in front of the tensor (first rows of the first
    // image, channel 0) are loaded from offset 0 and shifted when the tile is stored
    const bool neg_block = ci0 == 0 && off < 0;
    const int o_floor = neg_block ? 0 : (int)0x80000000;

    // pixel state of the thread's group, advanced by 64 pixels per tile with adds and compares
    int cp = p_begin + 4 * gq;
    int cq, coh, cow, a_cur, b_cur;
    {
        const int cn = cp / HW;
        cq = cp - cn * HW;
        coh = cq / g.W;
        cow = cq - coh * g.W;
        a_cur = (cn * g.Cout * HW + cq + (m0 + lr) * HW) * 4;
        b_cur = (cn * g.Cin * HW + cq + (ci0 + lr) * HW + off) * 4;
    }
    const int adv_h = FK / g.W, adv_w = FK - adv_h * g.W;
    const int a_wrap = (g.Cout - 1) * HW * 4, b_wrap = (g.Cin - 1) * HW * 4;

    struct Regs {
        u32x4 a[AV], b[BV];
        unsigned mk;       // validity of the 4 elements of the X group (tap halo, pixel range)
        int bo;            // unclamped offset of the X group in row pass 0
    };
    auto load_tile = [&](Regs &R) {
        const bool ok = cp < p_end;
        unsigned mk = 0xFu;
        if (K3) {
            const int wrap = g.W - cow;                        // elements e >= wrap sit on the next row
            const unsigned low = wrap >= 4 ? 0xFu : ((1u << wrap) - 1u);
            const bool row0 = (unsigned)(coh + dr) < (unsigned)g.H;
            const bool row1 = (unsigned)(coh + 1 + dr) < (unsigned)g.H;
            mk = (row0 ? low : 0u) | (row1 ? (0xFu & ~low) : 0u);
            if (ds < 0) {                                      // the element in column 0 has no left neighbour
                const unsigned kill = (cow == 0) ? 1u : (wrap < 4 ? (1u << wrap) : 0u);
                mk &= ~kill;
            } else if (ds > 0) {                               // the element in column W-1 has no right neighbour
                const int e1 = g.W - 1 - cow;
                mk &= ~(e1 < 4 ? (1u << e1) : 0u);
            }
        }
        mk = ok ? mk : 0u;
        // dY needs no validity test: a tile only runs past p_end in the last split, where p_end == P and the offsets
        // of pixels >= P lie behind the tensor (the buffer returns zeros)
#pragma unroll
        for (int i = 0; i < AV; ++i) R.a[i] = __builtin_amdgcn_raw_buffer_load_b128(dy_rsrc, a_cur + i * rs16, 0, 0);
        const int b0 = mk ? max(b_cur, o_floor) : kOutOfRange;
        const int bi = mk ? b_cur : kOutOfRange;
        R.b[0] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, b0, 0, 0);
#pragma unroll
        for (int i = 1; i < BV; ++i) R.b[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, bi + i * rs16, 0, 0);
        R.mk = mk;
        R.bo = b_cur;
        cp += FK;
        cq += FK;
        cow += adv_w;
        coh += adv_h;
        a_cur += FK * 4;
        b_cur += FK * 4;
        if (cow >= g.W) {
            cow -= g.W;
            coh += 1;
        }
        if (cq >= HW) {                                        // H*W >= 64: at most one image boundary per tile
            cq -= HW;
            coh -= g.H;
            a_cur += a_wrap;
            b_cur += b_wrap;
        }
    };
    auto cvt4 = [](const u32x4 &v) {
        bf16x4 o;
        o[0] = (E16)__uint_as_float(v.x);
        o[1] = (E16)__uint_as_float(v.y);
        o[2] = (E16)__uint_as_float(v.z);
        o[3] = (E16)__uint_as_float(v.w);
        return o;
    };
    // dbias[m] = sum over pixels of dy[m]: the workgroups of (tap 0, ci tile 0) add up the fp32 dY values they stage
    // anyway (every pixel split has exactly one such workgroup per Cout tile)
    const bool bias_sums = dbias != nullptr && by == 0;
    float bsum[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) bsum[i] = 0.0f;
    auto store_tile = [&](int buf, Regs &R) {
#pragma unroll
        for (int i = 0; i < AV; ++i) *reinterpret_cast<bf16x4 *>(&As[buf][lr + 16 * i][4 * gq]) = cvt4(R.a[i]);
        if (bias_sums) {
#pragma unroll
            for (int i = 0; i < AV; ++i)
                bsum[i] += (__uint_as_float(R.a[i].x) + __uint_as_float(R.a[i].y)) +
                           (__uint_as_float(R.a[i].z) + __uint_as_float(R.a[i].w));
        }
        if (K3 && R.mk != 0xFu) {                               // a row end / image edge inside the group: rare
            const unsigned mk = R.mk;
            if (neg_block && R.bo < 0 && R.bo > -16) {          // loaded from offset 0: element e holds x[e], wanted x[e - k]
                const int k = (-R.bo) >> 2;                    // 1..3
                const u32x4 v = R.b[0];
                u32x4 t;
                t.x = 0u;
                t.y = k == 1 ? v.x : 0u;
                t.z = k == 1 ? v.y : (k == 2 ? v.x : 0u);
                t.w = k == 1 ? v.z : (k == 2 ? v.y : v.x);
                R.b[0] = t;
            }
#pragma unroll
            for (int i = 0; i < BV; ++i) {
                R.b[i].x = (mk & 1u) ? R.b[i].x : 0u;
                R.b[i].y = (mk & 2u) ? R.b[i].y : 0u;
                R.b[i].z = (mk & 4u) ? R.b[i].z : 0u;
                R.b[i].w = (mk & 8u) ? R.b[i].w : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < BV; ++i) *reinterpret_cast<bf16x4 *>(&Bs[buf][lr + 16 * i][4 * gq]) = cvt4(R.b[i]);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto mma = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < FK / 16; ++ks) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&As[cur][wm * (BM / 2) + i * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][wn * (BNC / 2) + j * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    const int tiles = (p_end - p_begin + FK - 1) / FK;
    Regs R;
    load_tile(R);
    store_tile(0, R);
    __syncthreads();
    for (int t = 0; t < tiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < tiles) load_tile(R);
        mma(cur);
        if (t + 1 < tiles) store_tile(cur ^ 1, R);
        __syncthreads();
    }

    if (bias_sums) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            float v = bsum[i];                      // the 16 threads of a row (gq = lane & 15) hold its pixel groups
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 1, 64);
            if (gq == 0) atomicAdd(dbias + m0 + lr + 16 * i, v);
        }
    }
    // dW[m][tap][ci]: lanes run over ci (contiguous) -> coalesced fp32 atomics
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ci = ci0 + wn * (BNC / 2) + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                atomicAdd(dw + ((size_t)m * RS + tap) * g.Cin + ci, acc[i][j][e]);
            }
        }
    }
}

// dbias[c] += sum over images and pixels of dy[n][c][p] for the layers the flat weight-gradient kernel does not serve
__global__ __launch_bounds__(256) void channel_sum_kernel(const float *__restrict__ dy, float *__restrict__ dbias, int N,
                                                          int C, int HW)
{
    __shared__ float part[4];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const float *__restrict__ p = dy + ((size_t)n * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dbias + c, (part[0] + part[1]) + (part[2] + part[3]));
}

int make_geom(Geom &g, int N, int Cin, int H, int W, int Cout, int R, int S, int sh, int sw, int ph, int pw,
              int out_h, int out_w)
{
    FI_REQUIRE(N >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1 && R >= 1 && S >= 1, "sizes must be positive");
    FI_REQUIRE(sh >= 1 && sw >= 1 && ph >= 0 && pw >= 0, "bad stride / padding");
    g = Geom{};
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.R = R; g.S = S;
    g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw;
    g.OH = out_h > 0 ? out_h : (H + 2 * ph - R) / sh + 1;
    g.OW = out_w > 0 ? out_w : (W + 2 * pw - S) / sw + 1;
    FI_REQUIRE(g.OH >= 1 && g.OW >= 1, "empty output");
    FI_REQUIRE((long)N * g.OH * g.OW < 2147483647L, "too many output pixels");
    g.P = N * g.OH * g.OW;
    return FI_OK;
}

}  // namespace

extern "C" {

int FI16(fi_conv2d_forward, )(const float *x, const float *weight, const float *bias, const float *scale,
                           const float *residual, float *y, int N, int Cin, int H, int W, int Cout, int R,
                           int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                           int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream)
{
    return FI16(fi_conv2d_forward_gated, )(x, weight, bias, scale, residual, nullptr, y, N, Cin, H, W, Cout, R, S, stride_h,
                                           stride_w, pad_h, pad_w, relu, weight_layout, out_h, out_w, output_layout,
                                           stream);
}

int FI16(fi_conv2d_forward_gated, )(const float *x, const float *weight, const float *bias, const float *scale,
                                 const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                 int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                                 int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream)
{
    return FI16(fi_conv2d_forward_live, )(x, weight, bias, scale, residual, gate, y, N, Cin, H, W, Cout, R, S, stride_h,
                                          stride_w, pad_h, pad_w, relu, weight_layout, out_h, out_w, output_layout, nullptr,
                                          stream);
}

int FI16(fi_conv2d_forward_live, )(const float *x, const float *weight, const float *bias, const float *scale,
                                const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                                int weight_layout, int out_h, int out_w, int output_layout, const int32_t *n_live_dev,
                                fi_stream_t stream)
{
    Geom g;
    int rc = make_geom(g, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, out_h, out_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(x && weight && y, "null pointer");
    if (Cin % TK != 0 || !(weight_layout == 1 || weight_layout == 2 || R * S == 1)) {
        fi::set_error("the bf16 path needs Cin %% 32 == 0 and tap-major weights (got Cin = %d, layout %d)", Cin,
                      weight_layout);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE(output_layout == 0 || (Cout % 4 == 0 && residual == nullptr && gate == nullptr && (uintptr_t)y % 16 == 0),
               "channels-last output needs Cout % 4 == 0, a 16-byte aligned y and no fused residual / gate");
    FI_REQUIRE(((uintptr_t)weight % 16) == 0, "weights must be 16-byte aligned");
    g.flip = weight_layout == 2;
    g.out_nhwc = output_layout == 1;
    g.n_live = n_live_dev;               // honoured by conv_bf16_fwd_kernel; the patch kernel computes every image
    const Epi ep = {bias, scale, residual, relu, gate};
    hipStream_t st = (hipStream_t)stream;
    // 3x3 / stride 1 / pad 1 on maps at least 16 columns wide (whole 4-column groups): input patch in LDS
    if (!getenv("FI_NO_PATCH") && R == 3 && S == 3 && stride_h == 1 && stride_w == 1 && pad_h == 1 && pad_w == 1 &&
        g.OH == H && g.OW == W && W % 4 == 0 && W >= PB_TW && !g.out_nhwc && Cout > 64 && weight_layout >= 1 &&
        (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (residual == nullptr || (uintptr_t)residual % 16 == 0) &&
        (gate == nullptr || (uintptr_t)gate % 16 == 0) && (long)N * Cin * H * W < 2147483647L && (long)N * Cout * H * W < 2147483647L) {
        PatchGeomB pg;
        pg.N = N; pg.Cin = Cin; pg.H = H; pg.W = W; pg.Cout = Cout;
        pg.flip = g.flip;
        pg.tiles_x = fi::ceil_div(W, PB_TW);                 // W % 16 != 0: the last column tile is partly empty
        pg.ptiles = fi::ceil_div(N * H, PB_TH) * pg.tiles_x;
        pg.mtiles = fi::ceil_div(Cout, 128);
        if ((long)pg.ptiles * pg.mtiles >= 192) {
            const long blocks = (long)fi::ceil_div(pg.ptiles, 8) * 8 * pg.mtiles;
            fi::ProfScope prof(FI_K_CONV_BF16_FWD, st);
            hipLaunchKernelGGL(conv3x3_patch_bf16_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, st, x,
                               static_cast<const void *>(weight), ep, y, pg);
            FI_HIP_CHECK(hipGetLastError());
            return FI_OK;
        }
    }
    const long pv = (long)g.N * g.OH * ((g.OW + 3) / 4) * 4;          // virtual pixel space (rows padded to quads)
    FI_REQUIRE(pv < 2147483647L, "too many output pixels");
    const int ptiles = fi::ceil_div((int)pv, TN);
    // 64-row tiles for narrow layers and for grids that would leave CUs idle (C5 at batch 4: 128 workgroups)
    const int bm = (Cout <= 64 || (long)fi::ceil_div(Cout, 128) * ptiles < 512) ? 64 : 128;
    const int mtiles = fi::ceil_div(Cout, bm);
    const long grid = (long)mtiles * fi::ceil_div(ptiles, 8) * 8;
    FI_REQUIRE(grid < 2147483647L, "grid too large");
    fi::ProfScope prof(FI_K_CONV_BF16_FWD, st);
    const bool s2 = (stride_w == 2 && W >= 8);
    auto k = bm == 64 ? (s2 ? conv_bf16_fwd_kernel<64, 2> : conv_bf16_fwd_kernel<64, 1>)
                      : (s2 ? conv_bf16_fwd_kernel<128, 2> : conv_bf16_fwd_kernel<128, 1>);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), 0, st, x, weight, ep, y, g, mtiles, ptiles);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int FI16(fi_conv3x3_forward, w)(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                             const float *residual, float *y, int N, int Cin, int H, int W, int Cout, int relu,
                             int flip_taps, fi_stream_t stream)
{
    return FI16(fi_conv3x3_forward_gated, w)(x, weight_bf16, bias, scale, residual, nullptr, y, N, Cin, H, W, Cout, relu,
                                             flip_taps, stream);
}

int FI16(fi_conv3x3_forward_gated, w)(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                                   const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                   int Cout, int relu, int flip_taps, fi_stream_t stream)
{
    FI_REQUIRE(N >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1, "sizes must be positive");
    FI_REQUIRE(x && weight_bf16 && y, "null pointer");
    // 2-D tiles: width a multiple of 4, at least 16 (a width that is not a multiple of 16 leaves the last column tile
    // partly empty).  Flat tiles: even widths 12..14 (the 14 x 14 RoI maps)
    const bool tiled = W % 4 == 0 && W >= PB_TW;
    const bool flat = !tiled && W < 16 && W % 2 == 0 && (W + 126) / W + 2 <= PB_ROWS_FLAT;
    if (!((tiled || flat) && Cin % PB_CB == 0 && Cout > 64)) {
        fi::set_error("fi_conv3x3_forward_bf16w needs W %% 4 == 0 (W %% 16 == 0 for full tiles) and W >= 16, or W in {12, 14}; Cin %% 32 == 0 and Cout > 64 (got W = %d, Cin = %d, Cout = %d)",
                      W, Cin, Cout);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)weight_bf16 % 16 == 0 &&
               (residual == nullptr || (uintptr_t)residual % 16 == 0) && (gate == nullptr || (uintptr_t)gate % 16 == 0),
               "16-byte aligned tensors required");
    FI_REQUIRE((long)N * Cin * H * W < 2147483647L && (long)N * Cout * H * W < 2147483647L, "tensor too large");
    PatchGeomB pg;
    pg.N = N; pg.Cin = Cin; pg.H = H; pg.W = W; pg.Cout = Cout;
    pg.flip = flip_taps ? 1 : 0;
    pg.tiles_x = flat ? 1 : fi::ceil_div(W, PB_TW);
    pg.ptiles = flat ? fi::ceil_div(N * H * W, 128) : fi::ceil_div(N * H, PB_TH) * pg.tiles_x;
    pg.mtiles = fi::ceil_div(Cout, 128);
    const Epi ep = {bias, scale, residual, relu, gate};
    hipStream_t st = (hipStream_t)stream;
    const long blocks = (long)fi::ceil_div(pg.ptiles, 8) * 8 * pg.mtiles;
    FI_REQUIRE(blocks < 2147483647L, "grid too large");
    fi::ProfScope prof(FI_K_CONV_BF16_FWD, st);
    if (flat)
        hipLaunchKernelGGL((conv3x3_patch_bf16_kernel<true, true>), dim3((unsigned)blocks), dim3(kThreads), 0, st, x,
                           static_cast<const void *>(weight_bf16), ep, y, pg);
    else
        hipLaunchKernelGGL((conv3x3_patch_bf16_kernel<true, false>), dim3((unsigned)blocks), dim3(kThreads), 0, st, x,
                           static_cast<const void *>(weight_bf16), ep, y, pg);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int FI16(fi_conv1x1_forward, w)(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                             const float *residual, float *y, int N, int Cin, int HW, int Cout, int relu,
                             fi_stream_t stream)
{
    return FI16(fi_conv1x1_forward_gated, w)(x, weight_bf16, bias, scale, residual, nullptr, y, N, Cin, HW, Cout, relu, stream);
}

int FI16(fi_conv1x1_forward_gated, w)(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                                   const float *residual, const float *gate, float *y, int N, int Cin, int HW, int Cout,
                                   int relu, fi_stream_t stream)
{
    FI_REQUIRE(N >= 1 && Cin >= 1 && HW >= 1 && Cout >= 1, "sizes must be positive");
    FI_REQUIRE(x && weight_bf16 && y, "null pointer");
    if (!(HW % 4 == 0 && Cin % P1_KC == 0 && Cout > 64)) {
        fi::set_error("fi_conv1x1_forward_bf16w needs H*W %% 4 == 0, Cin %% 64 == 0 and Cout > 64 (got H*W = %d, Cin = %d, Cout = %d)",
                      HW, Cin, Cout);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)weight_bf16 % 16 == 0 &&
               (residual == nullptr || (uintptr_t)residual % 16 == 0) && (gate == nullptr || (uintptr_t)gate % 16 == 0),
               "16-byte aligned tensors required");
    FI_REQUIRE((long)N * Cin * HW < 2147483647L && (long)N * Cout * HW < 2147483647L, "tensor too large");
    Conv1x1GeomB pg;
    pg.N = N; pg.Cin = Cin; pg.HW = HW; pg.Cout = Cout; pg.P = N * HW;
    pg.ptiles = fi::ceil_div(pg.P, 128);
    pg.mtiles = fi::ceil_div(Cout, 128);
    const Epi ep = {bias, scale, residual, relu, gate};
    hipStream_t st = (hipStream_t)stream;
    const long blocks = (long)fi::ceil_div(pg.ptiles, 8) * 8 * pg.mtiles;
    FI_REQUIRE(blocks < 2147483647L, "grid too large");
    fi::ProfScope prof(FI_K_CONV_BF16_FWD, st);
    hipLaunchKernelGGL(conv1x1_bf16_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, x,
                       reinterpret_cast<const E16 *>(weight_bf16), ep, y, pg);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int FI16(fi_conv2d_weight_grad, )(const float *x, const float *dy, float *dweight, int N, int Cin, int H, int W,
                               int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                               int flags, fi_stream_t stream)
{
    return FI16(fi_conv2d_weight_grad_db, )(x, dy, dweight, nullptr, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h,
                                            pad_w, flags, stream);
}

static int wgrad16_impl(const float *x, const float *dy, float *dweight, float *dbias, int N, int Cin, int H,
                        int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                        int flags, fi_stream_t stream, const WgradBatch16 *batch);

int FI16(fi_conv2d_weight_grad_db, )(const float *x, const float *dy, float *dweight, float *dbias, int N, int Cin, int H,
                                  int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                                  int flags, fi_stream_t stream)
{
    return wgrad16_impl(x, dy, dweight, dbias, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, flags, stream,
                        nullptr);
}

// the weight gradient used as a GEMM (conv.linear on the 16-bit kernels: dweight [Cout = rows][Cin]) with a DEVICE count of
// live rows: row tiles past it are skipped and stay at the zeros the call fills dweight with
static const int32_t *g_rows_live = nullptr;      // (set around one launch by the entry point below; host-side only)
int FI16(fi_conv2d_weight_grad_rows, )(const float *x, const float *dy, float *dweight, int N, int Cin, int H, int W,
                                    int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int flags,
                                    const int32_t *rows_live_dev, fi_stream_t stream)
{
    g_rows_live = rows_live_dev;
    const int rc = wgrad16_impl(x, dy, dweight, nullptr, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, flags,
                                stream, nullptr);
    g_rows_live = nullptr;
    return rc;
}

// n weight gradients of one geometry in one launch (see fi_conv2d_weight_grad_batch); loops when the geometry is not
// the flat kernel's or the outputs are not pre-zeroed
int FI16(fi_conv2d_weight_grad_batch, )(const float *const *x, const float *const *dy, float *const *dweight,
                                     float *const *dbias, int n, int N, int Cin, int H, int W, int Cout, int R, int S,
                                     int stride_h, int stride_w, int pad_h, int pad_w, int weight_layout, int flags,
                                     fi_stream_t stream)
{
    (void)weight_layout;                                  // the 16-bit kernels always write tap-major
    FI_REQUIRE(n >= 1 && x && dy && dweight, "empty batch / null pointer table");
    const bool k3 = R == 3 && S == 3 && pad_h == 1 && pad_w == 1, k1 = R == 1 && S == 1 && pad_h == 0 && pad_w == 0;
    const int HWf = H * W;
    bool ok = (k3 || k1) && stride_h == 1 && stride_w == 1 && HWf % 4 == 0 && HWf >= FK && W >= 4 && Cout % 64 == 0 &&
              Cin % 64 == 0 && (flags & FI_OUTPUTS_ZEROED) && n > 1 && !getenv("FI_NO_BF16_FLAT");
    bool any_db = false, all_db = true;
    for (int i = 0; i < n; ++i) {
        FI_REQUIRE(x[i] && dy[i] && dweight[i], "null pointer in the batch");
        ok = ok && (uintptr_t)x[i] % 16 == 0 && (uintptr_t)dy[i] % 16 == 0;
        const bool has = dbias && dbias[i];
        any_db = any_db || has;
        all_db = all_db && has;
    }
    if (!ok || (any_db && !all_db)) {
        for (int i = 0; i < n; ++i) {
            const int rc = wgrad16_impl(x[i], dy[i], dweight[i], dbias ? dbias[i] : nullptr, N, Cin, H, W, Cout, R, S,
                                        stride_h, stride_w, pad_h, pad_w, flags, stream, nullptr);
            if (rc != FI_OK) return rc;
        }
        return FI_OK;
    }
    for (int i0 = 0; i0 < n; i0 += FI_WGRAD_BATCH_MAX) {
        WgradBatch16 wb;
        wb.n = n - i0 < FI_WGRAD_BATCH_MAX ? n - i0 : FI_WGRAD_BATCH_MAX;
        for (int i = 0; i < FI_WGRAD_BATCH_MAX; ++i) {
            const int j = i0 + (i < wb.n ? i : 0);
            wb.x[i] = x[j]; wb.dy[i] = dy[j]; wb.dw[i] = dweight[j]; wb.db[i] = all_db ? dbias[j] : nullptr;
        }
        const int rc = wgrad16_impl(wb.x[0], wb.dy[0], wb.dw[0], wb.db[0], N, Cin, H, W, Cout, R, S, stride_h, stride_w,
                                    pad_h, pad_w, flags, stream, &wb);
        if (rc != FI_OK) return rc;
    }
    return FI_OK;
}

static int wgrad16_impl(const float *x, const float *dy, float *dweight, float *dbias, int N, int Cin, int H,
                        int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                        int flags, fi_stream_t stream, const WgradBatch16 *batch)
{
    Geom g;
    int rc = make_geom(g, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, 0, 0);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(x && dy && dweight, "null pointer");
    g.n_live = g_rows_live;              // only the flat kernel reads it
    hipStream_t st = (hipStream_t)stream;
    const int RS = R * S;
    if (!(flags & FI_OUTPUTS_ZEROED)) {
        FI_HIP_CHECK(hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)Cout * RS * Cin, st));
        if (dbias) FI_HIP_CHECK(hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st));
    }
    const int OHW = g.OH * g.OW;
    fi::ProfScope prof(FI_K_CONV_BF16_WGRAD, st);
    // same-size stride-1 layers with whole channel tiles: flat pixel space (conv_bf16_wgrad_flat_kernel)
    {
        const bool k3 = R == 3 && S == 3 && pad_h == 1 && pad_w == 1, k1 = R == 1 && S == 1 && pad_h == 0 && pad_w == 0;
        const int HWf = H * W;
        const int bm = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 0), bnc = Cin % 128 == 0 ? 128 : (Cin % 64 == 0 ? 64 : 0);
        if ((k3 || k1) && stride_h == 1 && stride_w == 1 && g.OH == H && g.OW == W && HWf % 4 == 0 && HWf >= FK && W >= 4 &&
            bm && bnc && (size_t)N * Cin * HWf * 4 < 0x7fffff00ULL && (size_t)N * Cout * HWf * 4 < 0x7fffff00ULL &&
            (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && !getenv("FI_NO_BF16_FLAT")) {
            const int mt = Cout / bm, cin_tiles = Cin / bnc;
            const long tiles = (long)mt * RS * cin_tiles;
            const int ptiles = fi::ceil_div(g.P, FK);
            // split the pixel range so that ~2048 workgroups exist, at least 8 K-tiles each
            const int nb = batch ? batch->n : 1;            // a batch fills the chip together: fewer, longer splits each
            static const int target_wg = getenv("FI_WG16_TARGET") ? atoi(getenv("FI_WG16_TARGET")) : 2048;    // (tuning knob)
            long z = target_wg / (tiles * nb);
            if (z < 1) z = 1;
            if (z > ptiles / 8) z = ptiles / 8 > 0 ? ptiles / 8 : 1;
            if (N == 1 && H == 1 && !batch) {
                // the kernel as a GEMM (conv.linear: one long reduction): every split ends with a 64 KB atomic epilogue, so
                // a split should cover >= 3072 elements of the reduction as long as the chip still gets a workgroup per CU
                // (scripts/gemm16_probe.py: 2048 x 1024 x 12544 163 -> 111 us, 1408 x 1024 x 25088 180 -> 150 us)
                long zmax = g.P / 3072, zmin = fi::ceil_div(256, (int)tiles);
                if (zmax < zmin) zmax = zmin;
                if (z > zmax) z = zmax;
            }
            if (z > 65535) z = 65535;
            const int per = fi::ceil_div(ptiles, (int)z);
            z = fi::ceil_div(ptiles, per);
            FI_REQUIRE((long)RS * cin_tiles <= 65535, "too many (tap, ci) tiles");
            const long nblk = tiles * z * nb;
            FI_REQUIRE(nblk + 8 < 2147483647L, "too many workgroups");
            const dim3 grid((unsigned)(((nblk + 7) / 8) * 8));
            auto k = k3 ? (bm == 128 ? (bnc == 128 ? conv_bf16_wgrad_flat_kernel<128, 128, true> : conv_bf16_wgrad_flat_kernel<128, 64, true>)
                                     : (bnc == 128 ? conv_bf16_wgrad_flat_kernel<64, 128, true> : conv_bf16_wgrad_flat_kernel<64, 64, true>))
                        : (bm == 128 ? (bnc == 128 ? conv_bf16_wgrad_flat_kernel<128, 128, false> : conv_bf16_wgrad_flat_kernel<128, 64, false>)
                                     : (bnc == 128 ? conv_bf16_wgrad_flat_kernel<64, 128, false> : conv_bf16_wgrad_flat_kernel<64, 64, false>));
            WgradBatch16 wb;
            wb.n = 0;
            if (batch) wb = *batch;
            hipLaunchKernelGGL(k, grid, dim3(kThreads), 0, st, x, dy, dweight, g, cin_tiles, per * FK, mt, (int)z, dbias, wb);
            FI_HIP_CHECK(hipGetLastError());
            return FI_OK;
        }
    }
    if (dbias) {
        const int chunks = std::max(1, std::min(N, 2048 / std::max(1, Cout)));
        hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)Cout, (unsigned)chunks), dim3(256), 0, st, dy, dbias, N, Cout, OHW);
        FI_HIP_CHECK(hipGetLastError());
    }
    if ((stride_w == 1 || stride_w == 2) && g.OW >= 4 && W >= 4 * stride_w) {
        const int bm = Cout <= 64 ? 64 : 128, bnc = Cin <= 64 ? 64 : 128;
        const int mt = fi::ceil_div(Cout, bm), cin_tiles = fi::ceil_div(Cin, bnc);
        const long tiles = (long)mt * RS * cin_tiles;
        const long tq = (long)N * g.OH * ((g.OW + 3) / 4);
        FI_REQUIRE(tq * 4 < 2147483647L, "too many pixels");
        const int ktiles_total = (int)((tq + 7) / 8);
        // split the reduction so that ~2048 workgroups exist, at least 8 K-tiles each
        long z = 2048 / tiles;
        if (z < 1) z = 1;
        if (z > ktiles_total / 8) z = ktiles_total / 8 > 0 ? ktiles_total / 8 : 1;
        if (z > 65535) z = 65535;
        const int per = fi::ceil_div(ktiles_total, (int)z);
        z = fi::ceil_div(ktiles_total, per);
        FI_REQUIRE((long)RS * cin_tiles <= 65535, "too many (tap, ci) tiles");
        const dim3 grid(mt, RS * cin_tiles, (unsigned)z);
        const bool s2 = stride_w == 2;
        auto k = bm == 128 ? (bnc == 128 ? (s2 ? conv_bf16_wgrad_kernel<128, 128, 2> : conv_bf16_wgrad_kernel<128, 128, 1>)
                                         : (s2 ? conv_bf16_wgrad_kernel<128, 64, 2> : conv_bf16_wgrad_kernel<128, 64, 1>))
                           : (bnc == 128 ? (s2 ? conv_bf16_wgrad_kernel<64, 128, 2> : conv_bf16_wgrad_kernel<64, 128, 1>)
                                         : (s2 ? conv_bf16_wgrad_kernel<64, 64, 2> : conv_bf16_wgrad_kernel<64, 64, 1>));
        hipLaunchKernelGGL(k, grid, dim3(kThreads), 0, st, x, dy, dweight, g, cin_tiles, ktiles_total, per);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    const int mt = fi::ceil_div(Cout, 128), cin_tiles = fi::ceil_div(Cin, TN);
    const long tiles = (long)mt * RS * cin_tiles;
    // generic form (strided layers, maps narrower than 4): split the reduction (N images x chunks of pixels)
    // so that ~2048 workgroups exist; a chunk is a multiple of 32 pixels and never crosses an image
    long want = 2048 / tiles;
    if (want < 1) want = 1;
    int chunks_per_image = 1;
    if (want > N) chunks_per_image = (int)((want + N - 1) / N);
    int chunk_pixels = fi::ceil_div(fi::ceil_div(OHW, chunks_per_image), TK) * TK;
    if (chunk_pixels < 4 * TK) chunk_pixels = 4 * TK;
    chunks_per_image = fi::ceil_div(OHW, chunk_pixels);
    long z = (long)N * chunks_per_image;
    if (z > want) z = want;
    if (z > 65535) z = 65535;
    FI_REQUIRE((long)RS * cin_tiles <= 65535, "too many (tap, ci) tiles");
    hipLaunchKernelGGL(conv_bf16_wgrad_generic_kernel, dim3(mt, RS * cin_tiles, (unsigned)z), dim3(kThreads), 0, st, x, dy,
                       dweight, g, cin_tiles, chunks_per_image, chunk_pixels);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
