// conv_igemm.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores (gfx950):
// forward / data-gradient (one kernel) and weight-gradient, NCHW, arbitrary R x S,
// stride and zero padding.  This is the MFMA path for the dense 1x1/3x3/7x7 convolutions
// of the ResNet-FPN backbone, RPN, Dev make-up layer and mask head
// (lib/sub_module.py:38-128, 147-228, 234-280, 308-345, 750-787 of the reference, which
// ran them through cuDNN).
//
// Arithmetic: exact fp32 -- v_mfma_f32_32x32x2_f32 is bit-for-bit an fp32 fma chain, so no
// precision is given up against the reference's fp32 convolutions (there is no TF32-like
// mode on gfx950).  Peak for this instruction is the fp32 vector peak, 157 TFLOP/s.
//
// GEMM view of the forward pass   Y[m][p] = sum_k A[m][k] * B[k][p]
//   m = output channel, p = (image, oh, ow) flattened, k = (ci, r, s)
//   A = the weight tensor as stored ([Cout][Cin*R*S], k contiguous),
//   B = the im2col matrix, NEVER materialised: each workgroup gathers its
//       [BK x BN] slice straight from the NCHW input (for a fixed k, consecutive p are
//       consecutive pixels of one input row -> coalesced), zero-filling the halo.
// Tile: BM x 128 x 16 per 256-thread workgroup (BM = 128 or 64), 4 wavefronts as 2 x 2,
// each owning a (BM/2) x 64 accumulator block of 32x32 MFMA tiles; <= 128 VGPRs so that 4
// workgroups share a CU.  LDS tiles are stored k-major ([BK][BM+2], [BK][BN+2]: an MFMA operand
// read is 32 consecutive floats per half-wavefront, and the row pitch of 2 banks (mod 32) makes
// the transposed tile stores conflict free as well).  Global loads for K-step t+1 are issued
// into registers before the MFMAs of step t and written to the other LDS buffer afterwards:
// one barrier per K-step, L2/HBM latency hidden behind 16..32 MFMAs (64 cycles each).
//
// What the K loop does NOT contain (each item was worth 3..8 % of the kernel): divergent
// branches (weight rows past Cout re-read the last row; their sums are never stored), per-value
// halo selects (a tap outside the image reads a zero page instead of being masked after the
// load, so loaded registers go to LDS untouched), integer division (tap-major K order + magic
// numbers), integer multiplies.  The fused epilogue applies scale/bias (an eval-mode BatchNorm
// folded by the caller), the bottleneck shortcut and ReLU, and stores NCHW rows of 32 consecutive
// pixels or, for maps that only the channels-last RoIAlign reads, NHWC 16-byte channel groups.
//
// The data gradient of a stride-1 convolution is the same kernel run on dY with the
// transposed weights applied with reversed taps (weight_layout 2) and padding R-1-pad.  The
// weight gradient is a second GEMM with the reduction over pixels (split across workgroups,
// fp32 atomics into dW).
#include <stdint.h>
#include <stdlib.h>

#include "fi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_v __attribute__((ext_vector_type(4)));
typedef f32x4_v f32x4_u __attribute__((aligned(4)));    // 16-byte load from a 4-byte aligned address

constexpr int kThreads = 256;
constexpr int BN = 128;
constexpr int BK = 16;
constexpr int PAD = 2;   // row pitch = 2 (mod 32) banks: the transposed tile stores of both kernels are conflict free

// epilogue: y = acc * scale[m] + bias[m] (+ residual) (ReLU) -- scale/bias carry an eval-mode
// BatchNorm folded by the caller, residual the bottleneck shortcut
// gate (optional, shaped like y): the result is multiplied by (gate > 0) -- the data gradient of a layer whose input
// is a ReLU output leaves the kernel already masked (the separate masking pass of that ReLU's backward disappears)
struct Epilogue {
    const float *bias;
    const float *scale;
    const float *residual;
    int relu;
    const float *gate;
};

struct ConvGeom {
    int N, Cin, H, W, Cout, R, S, sh, sw, ph, pw, OH, OW;
    int K;        // Cin*R*S
    int P;        // N*OH*OW
    int out_nhwc; // 1: y is [N][OH][OW][Cout] (channels-last), else [N][Cout][OH][OW]
    int flip;     // 1: tap-major weights are applied with the taps in reverse order (data gradient)
    int swz;      // 1: XCD-aware tile order (grid padded to a multiple of 8 tiles along x)
    int vec_out;  // 1: NCHW output rows can be written 4 pixels at a time (OH*OW % 4 == 0, 16-byte aligned)
    int p_base;   // first output pixel of this launch (a layer may be split into a main and a tail launch)
    unsigned mul_ohw, sft_ohw, mul_ow, sft_ow;   // magic numbers: n / d == umulhi(n, mul) >> sft
    const float *zero;   // device address of g_zero_page (a kernel argument: no GOT load inside the K loop)
    int wg_gx, wg_gy, wg_splits;   // weight gradient with swz: logical grid (column tiles, Cout tiles, pixel splits) of a 1-D launch
    const int *n_live;  // optional DEVICE count: forward kernels skip tiles whose pixels all belong to images >= *n_live, the
                        // slab-mode weight gradient (fi_gemm_nt) tiles whose rows are all >= *n_live (a batch of static
                        // capacity of which only the first *n_live images / rows are real: the Dev stage's big branch)
    long dw_slab; // weight gradient: 0 = all pixel splits add into ONE dW (atomics); > 0 = split s STORES its partial
                  // sums at dw + s * dw_slab (one writer per element: deterministic; the caller reduces the slabs)
};

// Several weight gradients of ONE geometry in one launch (fi_conv2d_weight_grad_batch): the operand pointers of every
// problem travel by value in the kernel arguments -- no device-side table, no host-to-device copy per launch.
constexpr int kWgBatchMax = FI_WGRAD_BATCH_MAX;
struct WgradBatch {
    int n;                                   // 0: a plain launch (the kernel's own x / dy / dw / dbias arguments)
    const float *x[kWgBatchMax];
    const float *dy[kWgBatchMax];
    float *dw[kWgBatchMax];
    float *db[kWgBatchMax];
};

// Out-of-image taps of the tap-major gather read this instead of being masked after the load:
// the loaded registers go to LDS untouched (no per-value select in the K loop).
__device__ __attribute__((aligned(16))) float g_zero_page[64];

// exact for 0 <= n < 2^31 (mul = ceil(2^(32+sft) / d), 32 + sft = 31 + ceil(log2 d))
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sft)
{
    return mul ? (int)(__umulhi((unsigned)n, mul) >> sft) : n;   // mul == 0 encodes d == 1
}

// One K-step of MFMAs on a staged [BK][BM+PAD] x [BK][BN+PAD] tile pair.  The operand
// fragments of sub-step kk+1 are read from LDS while the MFMAs of sub-step kk issue
// (two register sets, statically indexed), so LDS latency is not exposed per sub-step.
template <int BM, int BNT = BN>
__device__ __forceinline__ void mma_tile(const float (*__restrict__ As)[BM + PAD],
                                         const float (*__restrict__ Bs)[BNT + PAD], int a_col, int b_col,
                                         int khalf, f32x16 (&acc)[BM / 64][BNT / 64])
{
    constexpr int MT = BM / 64;
    constexpr int NT = BNT / 64;
    float af[2][MT], bf[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = As[khalf][a_col + i * 32];
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[0][j] = Bs[khalf][b_col + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
            const int kr = (kk + 1) * 2 + khalf;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[nxt][i] = As[kr][a_col + i * 32];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[nxt][j] = Bs[kr][b_col + j * 32];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        // issue order: the LDS reads of sub-step kk+1 BEFORE the MFMAs of sub-step kk (the compiler otherwise
        // reuses one register set and reads after the MFMAs issue: LDS latency exposed once per sub-step)
        __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
    }
}

// NCHW epilogue of a FULL tile (every row < Cout, every pixel < P): no predicate anywhere, so the code is one
// basic block and the compiler batches the loads.  (The general path below tests scale / bias / residual /
// bounds per group of 4 pixels: its ISA is a chain of `global_load; s_waitcnt vmcnt(0)` -- scale, bias, shortcut,
// 48 dependent round trips per lane -- which on the short-K 1x1 layers, where all resident workgroups reach
// the epilogue together, cost 10..15 % of the kernel.)  Absent scale / bias are read from a dummy page and
// dropped by a select, so that the arithmetic stays exactly `acc [*scale] [+bias] [+shortcut] [relu]`.
template <int BM, int BNT, bool HAS_RES, bool HAS_GATE>
__device__ __forceinline__ void conv_epilogue_full(const f32x16 (&acc)[BM / 64][BNT / 64], const Epilogue &ep,
                                                   float *__restrict__ y, const ConvGeom &g, int m0, int p0,
                                                   int wm, int wn, int l31, int khalf,
                                                   float *__restrict__ scratch)
{
    constexpr int MT = BM / 64;
    constexpr int NT = BNT / 64;
    constexpr int SP = 36;                       // scratch row pitch (floats): 16-byte aligned rows
    const int OHW = g.OH * g.OW;
    const int lane = l31 + 32 * khalf;
    const int rr = lane >> 3;                    // 0..7 (+8 for the second read)
    const int c4 = (lane & 7) * 4;
    const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr, relu = ep.relu != 0;
    const float *__restrict__ sp = has_sc ? ep.scale : g.zero;
    const float *__restrict__ bp = has_bi ? ep.bias : g.zero;
    const int smul = has_sc ? 1 : 0, bmul = has_bi ? 1 : 0;
    const int mrow = m0 + wm * (BM / 2) + rr;
    float scv[MT][4], biv[MT][4];                // rows mrow + i*32 + 8*q, q = half*2 + t
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            scv[i][q] = sp[(mrow + i * 32 + 8 * q) * smul];
            biv[i][q] = bp[(mrow + i * 32 + 8 * q) * bmul];
        }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int pp = p0 + wn * (BNT / 2) + j * 32 + c4;
        const int on = fast_div(pp, g.mul_ohw, g.sft_ohw);
        const size_t obase = (size_t)on * g.Cout * OHW + (pp - on * OHW) + (size_t)mrow * OHW;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 r[4], gt[4];
            if (HAS_RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r[q] = *reinterpret_cast<const float4 *>(ep.residual + obase + (size_t)(i * 32 + 8 * q) * OHW);
            }
            if (HAS_GATE) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    gt[q] = *reinterpret_cast<const float4 *>(ep.gate + obase + (size_t)(i * 32 + 8 * q) * OHW);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = half * 8 + e8;
                    scratch[((e & 3) + 8 * ((e >> 2) & 1) + 4 * khalf) * SP + l31] = acc[i][j][e];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q = half * 2 + t;
                    float4 v = *reinterpret_cast<const float4 *>(&scratch[(rr + 8 * t) * SP + c4]);
                    const float sc = scv[i][q], bi = biv[i][q];
                    v.x = has_sc ? v.x * sc : v.x; v.y = has_sc ? v.y * sc : v.y;
                    v.z = has_sc ? v.z * sc : v.z; v.w = has_sc ? v.w * sc : v.w;
                    v.x = has_bi ? v.x + bi : v.x; v.y = has_bi ? v.y + bi : v.y;
                    v.z = has_bi ? v.z + bi : v.z; v.w = has_bi ? v.w + bi : v.w;
                    if (HAS_RES) {
                        v.x += r[q].x; v.y += r[q].y; v.z += r[q].z; v.w += r[q].w;
                    }
                    v.x = relu ? fmaxf(v.x, 0.0f) : v.x; v.y = relu ? fmaxf(v.y, 0.0f) : v.y;
                    v.z = relu ? fmaxf(v.z, 0.0f) : v.z; v.w = relu ? fmaxf(v.w, 0.0f) : v.w;
                    if (HAS_GATE) {
                        v.x = gt[q].x > 0.0f ? v.x : 0.0f; v.y = gt[q].y > 0.0f ? v.y : 0.0f;
                        v.z = gt[q].z > 0.0f ? v.z : 0.0f; v.w = gt[q].w > 0.0f ? v.w : 0.0f;
                    }
                    *reinterpret_cast<float4 *>(y + obase + (size_t)(i * 32 + 8 * q) * OHW) = v;
                }
            }
        }
    }
}

// Epilogue shared by the forward kernels.  C/D layout of v_mfma_f32_32x32x2_f32:
// col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int BM, bool ONHWC, int BNT = BN>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[BM / 64][BNT / 64], const Epilogue &ep,
                                              float *__restrict__ y, const ConvGeom &g, int m0, int p0,
                                              int wm, int wn, int l31, int khalf, float *__restrict__ scratch)
{
    constexpr int MT = BM / 64;
    constexpr int NT = BNT / 64;
    const int OHW = g.OH * g.OW;
    if (!ONHWC && g.vec_out && p0 + BNT <= g.P && m0 + BM <= g.Cout) {
        if (ep.gate) {
            if (ep.residual)
                conv_epilogue_full<BM, BNT, true, true>(acc, ep, y, g, m0, p0, wm, wn, l31, khalf, scratch);
            else
                conv_epilogue_full<BM, BNT, false, true>(acc, ep, y, g, m0, p0, wm, wn, l31, khalf, scratch);
        } else if (ep.residual)
            conv_epilogue_full<BM, BNT, true, false>(acc, ep, y, g, m0, p0, wm, wn, l31, khalf, scratch);
        else
            conv_epilogue_full<BM, BNT, false, false>(acc, ep, y, g, m0, p0, wm, wn, l31, khalf, scratch);
        return;
    }
    if (!ONHWC && g.vec_out) {
        // NCHW through a wavefront-private LDS transpose: the MFMA layout gives a lane 16 different
        // channels of ONE pixel (64 scalar stores, and 64 scalar shortcut loads, per lane and tile); after
        // a [16 channels][32 pixels] round trip through LDS a lane owns 4 consecutive pixels of a channel:
        // 16-byte stores / shortcut loads, 4x fewer memory instructions.  `scratch` is this wavefront's
        // slice of the (no longer needed) operand tiles; LDS operations of one wavefront execute in order,
        // so no barrier is involved.
        constexpr int SP = 36;                       // scratch row pitch (floats): 16-byte aligned rows
        const int lane = l31 + 32 * khalf;
        const int rr = lane >> 3;                    // 0..7 (+8 for the second read)
        const int c4 = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int pp = p0 + wn * (BNT / 2) + j * 32 + c4;
                const bool pv = pp < g.P;            // P % 4 == 0 here: the 4 pixels are valid together
                const int ppc = pv ? pp : 0;
                const int on = fast_div(ppc, g.mul_ohw, g.sft_ohw);
                const size_t obase = (size_t)on * g.Cout * OHW + (ppc - on * OHW);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const int e = half * 8 + e8;
                        scratch[((e & 3) + 8 * ((e >> 2) & 1) + 4 * khalf) * SP + l31] = acc[i][j][e];
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = rr + 8 * t;                                  // 0..15 within the half
                        const int m = m0 + wm * (BM / 2) + i * 32 + half * 16 + row;
                        float4 v = *reinterpret_cast<const float4 *>(&scratch[row * SP + c4]);
                        if (pv && m < g.Cout) {
                            const size_t o = obase + (size_t)m * OHW;
                            if (ep.scale) {
                                const float sc = ep.scale[m];
                                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                            }
                            if (ep.bias) {
                                const float bi = ep.bias[m];
                                v.x += bi; v.y += bi; v.z += bi; v.w += bi;
                            }
                            if (ep.residual) {
                                const float4 r = *reinterpret_cast<const float4 *>(ep.residual + o);
                                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                            }
                            if (ep.relu) {
                                v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f);
                                v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
                            }
                            if (ep.gate) {
                                const float4 t4 = *reinterpret_cast<const float4 *>(ep.gate + o);
                                v.x = t4.x > 0.0f ? v.x : 0.0f; v.y = t4.y > 0.0f ? v.y : 0.0f;
                                v.z = t4.z > 0.0f ? v.z : 0.0f; v.w = t4.w > 0.0f ? v.w : 0.0f;
                            }
                            *reinterpret_cast<float4 *>(y + o) = v;
                        }
                    }
                }
            }
        return;
    }
    if (ONHWC) {     // a separate instantiation: the wider stores must not raise the NCHW kernel's VGPR count
        // channels-last output: a lane owns 4 consecutive channels (e & 3) of its pixel -> one 16-byte
        // store; lanes l and l+32 are adjacent (32 bytes), the 4 e-groups x MT tiles complete the
        // 128..256-byte channel run of the pixel within this wavefront (merged in L2).
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = p0 + wn * (BNT / 2) + j * 32 + l31;
            if (pp >= g.P) continue;
            float *__restrict__ yb = y + (size_t)pp * g.Cout;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + 8 * eg + 4 * khalf;
                    if (m < g.Cout) {            // Cout % 4 == 0 (checked by the launcher)
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q] = acc[i][j][eg * 4 + q];
                            if (ep.scale) v[q] = v[q] * ep.scale[m + q];
                            if (ep.bias) v[q] += ep.bias[m + q];
                            if (ep.relu) v[q] = fmaxf(v[q], 0.0f);
                        }
                        *reinterpret_cast<float4 *>(yb + m) = make_float4(v[0], v[1], v[2], v[3]);
                        __builtin_amdgcn_sched_barrier(0);     // one group at a time: keeps the accumulators in AGPRs
                    }
                }
        }
        return;
    }
    // NCHW: per-channel scale / bias are read once per accumulator row (they do not depend on the
    // pixel column j); a wavefront store covers 32 consecutive pixels of one channel plane.
    size_t obase[NT];
    bool pok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int pp = p0 + wn * (BNT / 2) + j * 32 + l31;
        pok[j] = pp < g.P;
        const int ppc = pok[j] ? pp : 0;
        const int on = fast_div(ppc, g.mul_ohw, g.sft_ohw);
        obase[j] = (size_t)on * g.Cout * OHW + (ppc - on * OHW);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
            if (m < g.Cout) {
                const float sc = ep.scale ? ep.scale[m] : 1.0f;
                const float bi = ep.bias ? ep.bias[m] : 0.0f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (!pok[j]) continue;
                    const size_t o = obase[j] + (size_t)m * OHW;
                    float v = acc[i][j][e];
                    if (ep.scale) v = v * sc;
                    if (ep.bias) v += bi;
                    if (ep.residual) v += ep.residual[o];
                    if (ep.relu) v = fmaxf(v, 0.0f);
                    if (ep.gate) v = ep.gate[o] > 0.0f ? v : 0.0f;
                    y[o] = v;
                }
            }
        }
}

// -------------------------------------------------------------------------------------
// forward / dgrad
// -------------------------------------------------------------------------------------
// HWC = true: the reduction index runs tap-major, k = (r*S + s)*Cin + ci, over weights stored
// [Cout][R][S][Cin] (requires Cin % BK == 0).  A K-step then has ONE tap and BK consecutive
// channels, so the halo test is a bit test of a per-thread tap mask and the gather address is
// base + i*2*H*W: ~3 instructions per load instead of ~35 for the (ci, r, s) order.
// 4 wavefronts per SIMD (<= 128 VGPRs, accumulators included): 1024 resident workgroups, so the
// 4096 / 8192-tile grids of the P2-level layers run as whole waves of workgroups instead of 5.33
template <int BM, int TR, int TS, bool HWC, bool ONHWC = false, int BNT = BN>
__global__ __launch_bounds__(kThreads, 4) void conv_fwd_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ w,
                                                            Epilogue ep,
                                                            float *__restrict__ y, ConvGeom g)
{
    constexpr int MT = BM / 64;                 // 32-row MFMA tiles per wave along M
    constexpr int NT = BNT / 64;                // 32-column MFMA tiles per wave along the pixels
    constexpr int B_RSTEP = kThreads / BNT;     // im2col rows covered per load pass: 2 (BNT 128) or 4 (BNT 64)
    constexpr int B_LOADS = BK / B_RSTEP;       // 8 or 4 loads per thread and K-step
    // one buffer: [2][BK][BM+PAD] weight tiles, then [2][BK][BNT+PAD] im2col tiles; the epilogue reuses it
    constexpr int A_FLOATS = 2 * BK * (BM + PAD), B_FLOATS = 2 * BK * (BNT + PAD);
    static_assert((A_FLOATS + B_FLOATS) / 4 >= 16 * 36, "epilogue scratch does not fit");
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
    float (*As)[BK][BM + PAD] = reinterpret_cast<float (*)[BK][BM + PAD]>(smem);
    float (*Bs)[BK][BNT + PAD] = reinterpret_cast<float (*)[BK][BNT + PAD]>(smem + A_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = TR ? TR : g.R, S = TS ? TS : g.S;
    const int RS = R * S;
    const int K = g.K;
    // XCD-aware tile order.  Workgroups are handed to the 8 XCDs round-robin by linear id, and each
    // XCD has its own L2: with the plain (x fast) order the 3 image rows a 3x3 tile needs are fetched
    // by 3 different XCDs and the Cout tiles of one pixel tile run a whole grid apart (PMC: 2.6 GB
    // fetched for a 268 MB input).  Here XCD c owns a contiguous band of pixel tiles and walks it with
    // the Cout tiles innermost, so halo rows and the second Cout pass hit the same L2.
    int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (g.swz) {             // the launcher padded gridDim.x to a multiple of 8
        const int ny = gridDim.y;
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        const int per_xcd = gridDim.x >> 3;                // pixel tiles per XCD band
        const int xcd = id & 7, local = id >> 3;
        tile_y = local % ny;
        tile_x = xcd * per_xcd + local / ny;
        if (g.p_base + tile_x * BNT >= g.P) return;         // padding tile
    }
    const int m0 = tile_y * BM;
    const int p0 = g.p_base + tile_x * BNT;
    const int OHW = g.OH * g.OW;
    const int HW = g.H * g.W;
    if (g.n_live && p0 >= *g.n_live * OHW) return;       // every pixel of the tile lies in an image that is not there

    // ---- per-thread constants of the B (im2col) gather: one pixel column, 8 k rows ----
    const int bj = tid & (BNT - 1);
    const int bk0 = tid / BNT;                  // first row; rows bk0, bk0 + B_RSTEP, ... (wave-uniform)
    const int p = p0 + bj;
    const bool p_ok = p < g.P;
    int n = 0, oh = 0, ow = 0;
    if (p_ok) {
        n = fast_div(p, g.mul_ohw, g.sft_ohw);
        const int q = p - n * OHW;
        oh = fast_div(q, g.mul_ow, g.sft_ow);
        ow = q - oh * g.OW;
    }
    const int ih0 = oh * g.sh - g.ph;
    const int iw0 = ow * g.sw - g.pw;
    const float *__restrict__ xn = x + (size_t)n * g.Cin * HW;
    const int pix_off = ih0 * g.W + iw0;        // may be negative; only used when in range

    // ---- A (weights) loader: 4 consecutive k of one output channel per load -----------
    constexpr int A_TPR = BK / 4;                       // threads per weight row (float4 each)
    constexpr int A_ROWS = kThreads / A_TPR;            // rows covered per pass
    constexpr int A_LOADS = BM / A_ROWS;                // float4 loads per thread
    const int ak4 = (tid % A_TPR) * 4;
    const int am = tid / A_TPR;                         // + A_ROWS per further load
    const bool k_vec = (K & 3) == 0;

    // Loaded values are NOT touched until store_tiles(): any select/move right after a load makes
    // the compiler wait for it on the spot (s_waitcnt vmcnt) and the prefetch would be synchronous.
    // The halo / tail zeroing is therefore a bit mask applied when the tile is written to LDS.
    float a_reg[A_LOADS][4];
    float b_reg[B_LOADS];
    unsigned b_mask = 0;

    // HWC state: tap and channel base of the NEXT K-step to be loaded (calls are sequential)
    int cur_rs = 0, cur_ci0 = 0;
    unsigned long long tap_mask = 0;
    if (HWC) {
        for (int rs = 0; rs < RS; ++rs) {
            const int r = rs / S, s = rs - (rs / S) * S;
            if (p_ok && ((unsigned)(ih0 + r) < (unsigned)g.H) && ((unsigned)(iw0 + s) < (unsigned)g.W))
                tap_mask |= 1ULL << rs;
        }
    }

    // tap-major path: K % BK == 0, so a weight load is never partial; rows past Cout re-read the last
    // row (their accumulators are never stored) -- branch-free, unconditional 16-byte loads
    const float *__restrict__ a_row[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) a_row[i] = w + (size_t)min(m0 + am + i * A_ROWS, g.Cout - 1) * K + ak4;

    auto load_tiles = [&](int kt) {
        const int kbase = kt * BK;
        if (HWC) {
            const int kw = g.flip ? ((RS - 1 - cur_rs) * g.Cin + cur_ci0) : kbase;    // wave-uniform
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const float4 v = *reinterpret_cast<const float4 *>(a_row[i] + kw);
                a_reg[i][0] = v.x; a_reg[i][1] = v.y; a_reg[i][2] = v.z; a_reg[i][3] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < (HWC ? 0 : A_LOADS); ++i) {
            const int m = m0 + am + i * A_ROWS;
            const int k = kbase + ak4;
            if (m < g.Cout && k_vec && k + 3 < K) {
                const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)m * K + k);
                a_reg[i][0] = v.x; a_reg[i][1] = v.y; a_reg[i][2] = v.z; a_reg[i][3] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a_reg[i][q] = 0.0f;
                    if (m < g.Cout && k + q < K) a_reg[i][q] = w[(size_t)m * K + k + q];
                }
            }
        }
        if (HWC) {
            const int rs = cur_rs;
            const int r = rs / S, s = rs - (rs / S) * S;
            const bool ok = (tap_mask >> rs) & 1ULL;
            const int off0 = pix_off + r * g.W + s + (cur_ci0 + bk0) * HW;
            const float *__restrict__ bp = ok ? (xn + off0) : g.zero;
            const int stride = ok ? B_RSTEP * HW : 0;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) b_reg[i] = bp[i * stride];
            cur_ci0 += BK;
            if (cur_ci0 >= g.Cin) {
                cur_ci0 = 0;
                ++cur_rs;
            }
            return;
        }
        b_mask = 0;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int k = __builtin_amdgcn_readfirstlane(kbase + bk0 + B_RSTEP * i);
            const int kc = min(k, K - 1);
            const int ci = kc / RS;
            const int rs = kc - ci * RS;
            const int r = rs / S;
            const int s = rs - r * S;
            const int koff = ci * HW + r * g.W + s;              // scalar
            const bool ok = p_ok && (k < K) && ((unsigned)(ih0 + r) < (unsigned)g.H) &&
                            ((unsigned)(iw0 + s) < (unsigned)g.W);
            b_reg[i] = xn[ok ? (pix_off + koff) : 0];
            b_mask |= ok ? (1u << i) : 0u;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) As[buf][ak4 + q][am + i * A_ROWS] = a_reg[i][q];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            Bs[buf][bk0 + B_RSTEP * i][bj] = (HWC || ((b_mask >> i) & 1u)) ? b_reg[i] : 0.0f;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = (K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);          // lands while the MFMAs below run
        mma_tile<BM, BNT>(As[buf], Bs[buf], wm * (BM / 2) + l31, wn * (BNT / 2) + l31, khalf, acc);
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<BM, ONHWC, BNT>(acc, ep, y, g, m0, p0, wm, wn, l31, khalf,
                                  smem + wave * (((A_FLOATS + B_FLOATS) / 4) & ~3));
}

// -------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 (forward and data gradient): the input PATCH lives in LDS
// -------------------------------------------------------------------------------------
// conv_fwd_kernel gathers the im2col tile of every tap from global memory: for a 3x3 layer each input element
// is fetched 9 times (from L2), every K-step of 16 channels x 1 tap costs 10 global loads, 16 LDS stores and
// a barrier per thread for 32 MFMAs per wavefront.  Here a workgroup owns a tile of 8 rows x 16 columns of
// output pixels and stages, per block of 16 input channels, the (8+2) x (16+8) input patch ONCE; the nine
// taps are nine different LDS offsets into it.  Weights do not go through LDS at all: the 4 wavefronts of a
// workgroup own 32 output channels each (4 x 1 layout), so a lane loads its MFMA A-operand -- 8 consecutive
// input channels of one output channel and tap -- straight from the tap-major weight into registers.
// Per 16 channels: 288 MFMAs per wavefront against 4 patch loads + 18 weight loads, 8 LDS stores and ONE
// barrier per thread (conv_fwd_kernel: 90 loads, 144 LDS stores, 9 barriers).
//
// Rows are addressed in the STACKED row space Y = n*H + y of the whole batch, so that 14x14 RoI maps fill
// 8-row tiles without per-RoI padding; a tap that would cross into the neighbouring image reads a row of
// zeros kept at the end of the patch instead.  MFMA column n = l31 + 32*j is pixel 4*l31 + j of the tile
// (quad l31 = 4 consecutive columns), so a lane ends up with 4 adjacent pixels of 16 channels: 16-byte
// NCHW stores without a transpose.
constexpr int PT_TH = 8, PT_TW = 16;        // 2-D tile: 8 stacked rows x 16 columns = 128 pixels
constexpr int PT_PW = 24;                   // patch row: six 16-byte groups (2-D: columns X0-4 .. X0+19; flat: -4 .. 19)

// FLAT = false: 2-D tiles (maps whose width is a multiple of 16).
// FLAT = true : maps of 12..16 columns (14 x 14 RoI maps): a tile is 128 CONSECUTIVE pixels of the flattened
//               (stacked row, column) space -- no tile column is wasted on a 14-wide map; a lane's 4 pixels may
//               then sit on two rows, so every pixel carries its own patch offsets.
template <bool FLAT>
struct PatchShape {
    static constexpr int ROWS = FLAT ? 13 : PT_TH + 2;         // patch rows (flat: up to 11 rows of 12+ pixels, + halo)
    static constexpr int ZR = ROWS;                              // + one row of zeros
    static constexpr int PP = FLAT ? 338 : 266;                  // floats per channel: (ROWS+1)*24 + 2 -> 8*PP = 16 (mod 32) banks
    // staged 16-byte groups per patch row.  Flat maps are at most 16 columns wide and a patch row starts at image
    // column -4: groups 0 and 5 (columns -4..-1 and 16..19) are outside every image -- zeroed once, never staged.
    static constexpr int GROUPS = FLAT ? 4 : PT_PW / 4;
    static constexpr int ITEMS = BK * ROWS * GROUPS;             // 16-byte groups per channel block
    static constexpr int PER_THREAD = (ITEMS + kThreads - 1) / kThreads;
    static constexpr int STAGE_LOADS = PER_THREAD * (FLAT ? 2 : 1);   // global loads per thread and channel block
    static constexpr int NJ = FLAT ? 4 : 1;                      // distinct patch offsets per lane and kernel row
};

struct PatchGeom {
    int N, Cin, H, W, Cout;
    int flip;            // taps applied in reverse order (data gradient)
    int tiles_x;         // 2-D: column tiles per row
    int ptiles, mtiles;  // pixel tiles, Cout tiles
    int vec4;            // 2-D: W % 4 == 0 and 16-byte aligned tensors -> 16-byte stores / shortcut loads
    int out_nhwc;        // 2-D only: y is [N][H][W][Cout] (Cout % 128 == 0, no shortcut): 16-byte channel groups
    const float *zero;
};

// predicate-free NCHW epilogue of conv3x3_patch_kernel<false>: the loads of a group of 4 rows are issued together
template <bool HAS_RES, bool HAS_GATE>
__device__ __forceinline__ void patch_epilogue_vec(const f32x16 (&acc)[4], const Epilogue &ep, float *__restrict__ y,
                                                   size_t obase, size_t HW, int mb, const float *__restrict__ sp,
                                                   const float *__restrict__ bp, int smul, int bmul, bool has_sc,
                                                   bool has_bi, bool relu)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float sc[4], bi[4];
        float4 rr[4], gt[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int m = mb + 8 * q + e4;
            sc[e4] = sp[m * smul];
            bi[e4] = bp[m * bmul];
            if (HAS_RES) rr[e4] = *reinterpret_cast<const float4 *>(ep.residual + obase + (size_t)(8 * q + e4) * HW);
            if (HAS_GATE) gt[e4] = *reinterpret_cast<const float4 *>(ep.gate + obase + (size_t)(8 * q + e4) * HW);
        }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j][4 * q + e4];
                v = has_sc ? v * sc[e4] : v;
                v = has_bi ? v + bi[e4] : v;
                t[j] = v;
            }
            if (HAS_RES) {
                t[0] += rr[e4].x; t[1] += rr[e4].y; t[2] += rr[e4].z; t[3] += rr[e4].w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = relu ? fmaxf(t[j], 0.0f) : t[j];
            if (HAS_GATE) {
                t[0] = gt[e4].x > 0.0f ? t[0] : 0.0f; t[1] = gt[e4].y > 0.0f ? t[1] : 0.0f;
                t[2] = gt[e4].z > 0.0f ? t[2] : 0.0f; t[3] = gt[e4].w > 0.0f ? t[3] : 0.0f;
            }
            *reinterpret_cast<float4 *>(y + obase + (size_t)(8 * q + e4) * HW) = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}

// flat tiles: the lane's pixels (0,1) and (2,3) are two pairs, each inside one row (W even): 8-byte accesses.
// Loads are unconditional (clamped rows / pixels), only the stores are predicated.
template <bool HAS_RES, bool HAS_GATE>
__device__ __forceinline__ void patch_epilogue_pairs(const f32x16 (&acc)[4], const Epilogue &ep, float *__restrict__ y,
                                                     const size_t (&opair)[2], const bool (&okp)[2], size_t HW, int mb,
                                                     int Cout, const float *__restrict__ sp,
                                                     const float *__restrict__ bp, int smul, int bmul, bool has_sc,
                                                     bool has_bi, bool relu)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float sc[4], bi[4];
        float2 rr[4][2], gt[4][2];
        int mrow[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            mrow[e4] = min(mb + 8 * q + e4, Cout - 1) - mb;          // clamped row, relative to mb
            sc[e4] = sp[(mb + mrow[e4]) * smul];
            bi[e4] = bp[(mb + mrow[e4]) * bmul];
            if (HAS_RES) {
                rr[e4][0] = *reinterpret_cast<const float2 *>(ep.residual + opair[0] + (size_t)mrow[e4] * HW);
                rr[e4][1] = *reinterpret_cast<const float2 *>(ep.residual + opair[1] + (size_t)mrow[e4] * HW);
            }
            if (HAS_GATE) {
                gt[e4][0] = *reinterpret_cast<const float2 *>(ep.gate + opair[0] + (size_t)mrow[e4] * HW);
                gt[e4][1] = *reinterpret_cast<const float2 *>(ep.gate + opair[1] + (size_t)mrow[e4] * HW);
            }
        }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j][4 * q + e4];
                v = has_sc ? v * sc[e4] : v;
                v = has_bi ? v + bi[e4] : v;
                t[j] = v;
            }
            if (HAS_RES) {
                t[0] += rr[e4][0].x; t[1] += rr[e4][0].y; t[2] += rr[e4][1].x; t[3] += rr[e4][1].y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = relu ? fmaxf(t[j], 0.0f) : t[j];
            if (HAS_GATE) {
                t[0] = gt[e4][0].x > 0.0f ? t[0] : 0.0f; t[1] = gt[e4][0].y > 0.0f ? t[1] : 0.0f;
                t[2] = gt[e4][1].x > 0.0f ? t[2] : 0.0f; t[3] = gt[e4][1].y > 0.0f ? t[3] : 0.0f;
            }
            const bool row_ok = mb + 8 * q + e4 < Cout;
            if (row_ok && okp[0]) *reinterpret_cast<float2 *>(y + opair[0] + (size_t)mrow[e4] * HW) = make_float2(t[0], t[1]);
            if (row_ok && okp[1]) *reinterpret_cast<float2 *>(y + opair[1] + (size_t)mrow[e4] * HW) = make_float2(t[2], t[3]);
        }
    }
}

template <bool FLAT>
__global__ __launch_bounds__(kThreads, 3) void conv3x3_patch_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ w, Epilogue ep,
                                                                   float *__restrict__ y, PatchGeom g)
{
    using SH = PatchShape<FLAT>;
    __shared__ __attribute__((aligned(16))) float Ps[2][BK][SH::PP];

    // XCD-aware order: XCD c owns a contiguous band of pixel tiles, Cout tiles innermost
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_xcd = (g.ptiles + 7) >> 3;
    const int mt = local % g.mtiles;
    const int pt = xcd * per_xcd + local / g.mtiles;
    if (pt >= g.ptiles) return;
    const int m0 = mt * 128;
    const int NH = g.N * g.H;
    const size_t HW = (size_t)g.H * g.W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;

    // ---- tile geometry: stacked row / column of the lane's 4 pixels, patch origin ------------------------
    int Yj[4], xj[4];                 // stacked row (n*H + y) and column of pixel j of the lane's quad
    int Yp0, Xp0;                     // stacked row of patch row 0, image column of patch column 0
    if (FLAT) {
        const int P0 = pt * 128;
        Yp0 = P0 / g.W - 1;
        Xp0 = -4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = P0 + 4 * l31 + j;
            Yj[j] = p / g.W;
            xj[j] = p - Yj[j] * g.W;
        }
    } else {
        const int tile_y = pt / g.tiles_x, tile_x = pt - tile_y * g.tiles_x;
        Yp0 = tile_y * PT_TH - 1;
        Xp0 = tile_x * PT_TW - 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Yj[j] = tile_y * PT_TH + (l31 >> 2);
            xj[j] = tile_x * PT_TW + (l31 & 3) * 4 + j;
        }
    }

    // ---- A operand: 8 consecutive channels of (output channel, tap), straight into registers --------
    const int am = min(m0 + wave * 32 + l31, g.Cout - 1);          // rows past Cout re-read the last row
    const float *__restrict__ a_base = w + (size_t)am * 9 * g.Cin + khalf * 8;
    auto load_a = [&](float (&a)[8], int tap, int cb) {
        const float *__restrict__ p = a_base + (size_t)(g.flip ? 8 - tap : tap) * g.Cin + cb * BK;
        const float4 v0 = *reinterpret_cast<const float4 *>(p);
        const float4 v1 = *reinterpret_cast<const float4 *>(p + 4);
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
        a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    };

    // ---- B operand: patch offset of (row of pixel j + r - 1, column of pixel j - 1), per kernel row r ----
    // a tap that would leave the pixel's image reads the row of zeros (ZR) instead
    int rowbase[3][SH::NJ];
#pragma unroll
    for (int j = 0; j < SH::NJ; ++j) {
        const int yo = Yj[j] - (Yj[j] / g.H) * g.H;                // row inside its image
        const int pr = Yj[j] - Yp0;                                // patch row of the pixel's own row (r = 1)
        const int col = xj[j] - Xp0 - 1;
        rowbase[0][j] = (yo != 0 ? pr - 1 : SH::ZR) * PT_PW + col;
        rowbase[1][j] = pr * PT_PW + col;
        rowbase[2][j] = (yo != g.H - 1 ? pr + 1 : SH::ZR) * PT_PW + col;
    }

    // ---- patch staging: work items of one 16-byte group (4 columns of one patch row and channel) ------
    int it_goff[SH::PER_THREAD], it_lds[SH::PER_THREAD], it_mask[SH::PER_THREAD];
#pragma unroll
    for (int i = 0; i < SH::PER_THREAD; ++i) {
        const int wi = tid + kThreads * i;
        const int ch = wi / (SH::ROWS * SH::GROUPS);
        const int rem = wi - ch * (SH::ROWS * SH::GROUPS);
        const int prow = rem / SH::GROUPS, grp = rem - prow * SH::GROUPS + (FLAT ? 1 : 0);
        const int Ys = Yp0 + prow;
        const int xx = Xp0 + grp * 4;
        const bool item = wi < SH::ITEMS;
        const bool row_ok = item && Ys >= 0 && Ys < NH;
        const int n = row_ok ? Ys / g.H : 0;
        const int yy = row_ok ? Ys - n * g.H : 0;
        int mask = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (row_ok && xx + e >= 0 && xx + e < g.W) mask |= 1 << e;
        it_mask[i] = item ? mask : -1;                             // -1: no item
        it_goff[i] = (int)(((size_t)n * g.Cin + ch) * HW + (size_t)yy * g.W) + xx;   // < 2^31 (launcher)
        it_lds[i] = ch * SH::PP + prow * PT_PW + grp * 4;
    }
    float4 preg[SH::PER_THREAD];
    auto stage_load = [&](int cb) {
        const float *__restrict__ xb = x + (size_t)cb * BK * HW;
#pragma unroll
        for (int i = 0; i < SH::PER_THREAD; ++i) {
            const int m = it_mask[i];
            if (FLAT) {
                // W is even and the group starts at a multiple of 4: each half of the group is inside or outside the
                // image as a whole -> two 8-byte loads, the outside ones redirected to the page of zeros (no branches,
                // so the loads of all items are in flight together)
                const float *__restrict__ q = xb + it_goff[i];
                const float *__restrict__ q0 = (m & 3) == 3 ? q : g.zero;
                const float *__restrict__ q1 = (m & 12) == 12 ? q + 2 : g.zero;
                const float2 v0 = *reinterpret_cast<const float2 *>(q0);
                const float2 v1 = *reinterpret_cast<const float2 *>(q1);
                preg[i] = make_float4(v0.x, v0.y, v1.x, v1.y);
            } else {
                // W is a multiple of 16 and the group starts at a multiple of 4: a group is inside the image as a
                // whole (mask 15) or not at all -> one 16-byte load, redirected to the page of zeros when outside
                const float *__restrict__ q = m == 15 ? xb + it_goff[i] : g.zero;
                const f32x4_v v = *reinterpret_cast<const f32x4_u *>(q);
                preg[i] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto stage_store = [&](int buf) {
        float *__restrict__ pb = &Ps[buf][0][0];
#pragma unroll
        for (int i = 0; i < SH::PER_THREAD; ++i) {
            if (it_mask[i] < 0) continue;
            float2 *__restrict__ d = reinterpret_cast<float2 *>(pb + it_lds[i]);        // 8-byte aligned
            d[0] = make_float2(preg[i].x, preg[i].y);
            d[1] = make_float2(preg[i].z, preg[i].w);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // zero rows of both buffers (never overwritten; flat: also the never-staged outer column groups), first patch
    if (FLAT) {
        float *__restrict__ pz = &Ps[0][0][0];
        for (int i = tid; i < 2 * BK * SH::PP; i += kThreads) pz[i] = 0.0f;
        __syncthreads();
    } else {
        for (int i = tid; i < 2 * BK * PT_PW; i += kThreads) {
            const int b = i / (BK * PT_PW), r = i - b * (BK * PT_PW);
            const int ch = r / PT_PW;
            Ps[b][ch][SH::ZR * PT_PW + (r - ch * PT_PW)] = 0.0f;
        }
    }
    const int ncb = g.Cin / BK;
    stage_load(0);
    stage_store(0);
    __syncthreads();

    // 72 sub-steps (9 taps x 8 channel pairs) of 4 MFMAs per channel block, software pipelined: the 4 LDS reads
    // of sub-step i+1 and (at the start of a tap) the 2 weight loads of the NEXT tap are issued before the MFMAs
    // of sub-step i.  Register sets rotate statically (weights: tap % 3, LDS values: sub-step & 1).
    float areg[3][8];
    float breg[2][4];
    auto bload = [&](float (&bv)[4], const float *__restrict__ pbuf, int r, int sft) {
        if (FLAT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = pbuf[rowbase[r][j] + sft];
        } else {
            const float *__restrict__ bp = pbuf + rowbase[r][0] + sft;
            bv[0] = bp[0]; bv[1] = bp[1]; bv[2] = bp[2]; bv[3] = bp[3];
        }
    };
    load_a(areg[0], 0, 0);
    for (int cb = 0; cb < ncb; ++cb) {
        const float *__restrict__ pbuf = &Ps[cb & 1][0][0] + khalf * 8 * SH::PP;
        const bool more = cb + 1 < ncb;
        bload(breg[0], pbuf, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);          // the reads of sub-step 0 (keeps the groups below aligned:
                                                                    // without it every read group slides one sub-step back)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#ifdef FI_EXP_NO_A
            if (t < 8 && cb == 0) load_a(areg[(t + 1) % 3], t + 1, cb);      // timing experiment: weights loaded once (wrong results)
#else
            if (t < 8) load_a(areg[(t + 1) % 3], t + 1, cb);
#endif
            // the next patch is requested behind the weights of tap 2: the first wait that covers it is the one for
            // the weights of tap 3, two taps (64 MFMAs) later -- at the top of the block it would sit in front of the
            // wait for tap 0's weights
            // (unconditional -- a branch here would make the waits behind it count for the path without the loads;
            // the last block re-reads its own patch from L2 and drops it)
            if (t == 1) stage_load(more ? cb + 1 : cb);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int step = t * 8 + kk;
                if (step + 1 < 72) {
                    const int t2 = (step + 1) / 8, k2 = (step + 1) % 8;
                    bload(breg[(step + 1) & 1], pbuf, t2 / 3, (t2 % 3) + k2 * SH::PP);
                }
                const float av = areg[t % 3][kk];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[step & 1][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[step & 1][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[step & 1][2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[step & 1][3], acc[3], 0, 0, 0);
                if (kk == 0 && t < 8) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // weight loads of tap t+1
                if (kk == 0 && t == 1) __builtin_amdgcn_sched_group_barrier(0x020, SH::STAGE_LOADS, 0);   // next patch
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                            // LDS reads of sub-step i+1
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                            // MFMAs of sub-step i
            }
        }
        if (more) {
#ifndef FI_EXP_NO_A
            load_a(areg[0], 0, cb + 1);
#endif
            stage_store((cb + 1) & 1);
        }
        __syncthreads();
    }

    // ---- epilogue: lane = 4 pixels x 16 channels (rows (e&3) + 8*(e>>2) + 4*khalf) ---------------------------
    const int mb = m0 + wave * 32 + 4 * khalf;
    const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr, relu = ep.relu != 0;
    const float *__restrict__ sp = has_sc ? ep.scale : g.zero;
    const float *__restrict__ bp = has_bi ? ep.bias : g.zero;
    const int smul = has_sc ? 1 : 0, bmul = has_bi ? 1 : 0;
    if (FLAT) {
        size_t opair[2];
        bool okp[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            okp[h] = Yj[2 * h] < NH;                       // W even: a pair is inside one row, valid together
            const int Yc = min(Yj[2 * h], NH - 1);
            const int n_img = Yc / g.H;
            opair[h] = ((size_t)n_img * g.Cout + mb) * HW + (size_t)(Yc - n_img * g.H) * g.W + xj[2 * h];
        }
        if (ep.gate) {
            if (ep.residual)
                patch_epilogue_pairs<true, true>(acc, ep, y, opair, okp, HW, mb, g.Cout, sp, bp, smul, bmul, has_sc, has_bi, relu);
            else
                patch_epilogue_pairs<false, true>(acc, ep, y, opair, okp, HW, mb, g.Cout, sp, bp, smul, bmul, has_sc, has_bi, relu);
        } else if (ep.residual)
            patch_epilogue_pairs<true, false>(acc, ep, y, opair, okp, HW, mb, g.Cout, sp, bp, smul, bmul, has_sc, has_bi, relu);
        else
            patch_epilogue_pairs<false, false>(acc, ep, y, opair, okp, HW, mb, g.Cout, sp, bp, smul, bmul, has_sc, has_bi, relu);
        return;
    }
    const int Yo = Yj[0], xo = xj[0];
    if (Yo >= NH || xo >= g.W) return;
    if (g.out_nhwc) {
        // channels-last output (maps that only the channels-last RoIAlign reads): the lane's accumulator rows
        // (e & 3) are 4 consecutive channels of one pixel -> one 16-byte store per pixel and group of rows
        float *__restrict__ yp = y + ((size_t)Yo * g.W + xo) * g.Cout + mb;      // stacked row == n*H + y
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sc[4], bi[4];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                sc[e4] = sp[(mb + 8 * q + e4) * smul];
                bi[e4] = bp[(mb + 8 * q + e4) * bmul];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    float v = acc[j][4 * q + e4];
                    v = has_sc ? v * sc[e4] : v;
                    v = has_bi ? v + bi[e4] : v;
                    t[e4] = relu ? fmaxf(v, 0.0f) : v;
                }
                *reinterpret_cast<float4 *>(yp + (size_t)j * g.Cout + 8 * q) = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
        return;
    }
    const int n_img = Yo / g.H;
    const size_t obase = ((size_t)n_img * g.Cout + mb) * HW + (size_t)(Yo - n_img * g.H) * g.W + xo;
    if (g.vec4 && m0 + 128 <= g.Cout) {      // xo % 4 == 0 and W % 4 == 0: the quad is inside the row
        if (ep.gate) {
            if (ep.residual)
                patch_epilogue_vec<true, true>(acc, ep, y, obase, HW, mb, sp, bp, smul, bmul, has_sc, has_bi, relu);
            else
                patch_epilogue_vec<false, true>(acc, ep, y, obase, HW, mb, sp, bp, smul, bmul, has_sc, has_bi, relu);
        } else if (ep.residual)
            patch_epilogue_vec<true, false>(acc, ep, y, obase, HW, mb, sp, bp, smul, bmul, has_sc, has_bi, relu);
        else
            patch_epilogue_vec<false, false>(acc, ep, y, obase, HW, mb, sp, bp, smul, bmul, has_sc, has_bi, relu);
        return;
    }
    // general: rows past Cout and columns past W are skipped; 4-byte accesses
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m >= g.Cout) continue;
        const float sc = sp[m * smul], bi = bp[m * bmul];
        const size_t o = obase + (size_t)((e & 3) + 8 * (e >> 2)) * HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (xo + j >= g.W) continue;
            float v = acc[j][e];
            v = has_sc ? v * sc : v;
            v = has_bi ? v + bi : v;
            if (ep.residual) v += ep.residual[o + j];
            v = relu ? fmaxf(v, 0.0f) : v;
            if (ep.gate) v = ep.gate[o + j] > 0.0f ? v : 0.0f;
            y[o + j] = v;
        }
    }
}

// -------------------------------------------------------------------------------------
// 1x1 / stride 1 (forward and data gradient) on the structure of conv3x3_patch_kernel
// -------------------------------------------------------------------------------------
// A workgroup owns 128 consecutive pixels of the flattened [n][H*W] space x 128 output channels.  Per block of 32
// input channels the pixel tile [32 ch][128 px] is staged with four 16-byte loads + four 16-byte LDS stores per
// thread (conv_fwd_kernel: 16 scalar loads + 16 scalar LDS stores, and the weights through LDS as well); weights
// go straight from memory into the MFMA A-operand registers (4 wavefronts x 32 output channels).  MFMA column
// n = l31 + 32j is pixel 4*l31 + j, so the lane's four B values are ONE 16-byte LDS read per channel, and its
// results are 4 adjacent pixels x 16 channels (16-byte NCHW stores, predicate-free epilogue on full tiles).
constexpr int P1_CB = 32;                   // channels per stage
constexpr int P1_PP = 132;                  // floats per channel row: 128 pixels + 4 (16-byte aligned, bank shift 4)

struct Conv1x1Geom {
    int N, Cin, HW, Cout;
    int ptiles, mtiles;
    int vec4;                               // 16-byte aligned y / residual
    const float *zero;
};

__global__ __launch_bounds__(kThreads, 3) void conv1x1_reg_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ w, Epilogue ep,
                                                                 float *__restrict__ y, Conv1x1Geom g)
{
    __shared__ __attribute__((aligned(16))) float Ps[2][P1_CB][P1_PP];
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_xcd = (g.ptiles + 7) >> 3;
    const int mt = local % g.mtiles;
    const int pt = xcd * per_xcd + local / g.mtiles;
    if (pt >= g.ptiles) return;
    const int m0 = mt * 128;
    const int P = g.N * g.HW;
    const int P0 = pt * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;

    // ---- A operand: 8 consecutive input channels of one output channel ------------------------------
    const int am = min(m0 + wave * 32 + l31, g.Cout - 1);
    const float *__restrict__ a_base = w + (size_t)am * g.Cin + khalf * 8;
    auto load_a = [&](float (&a)[8], int c16) {           // c16: 16-channel group index
        const float *__restrict__ p = a_base + c16 * 16;
        const float4 v0 = *reinterpret_cast<const float4 *>(p);
        const float4 v1 = *reinterpret_cast<const float4 *>(p + 4);
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
        a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    };

    // ---- staging: thread = (pixel group of 4, channels (tid >> 5) + 8 i) ----------------------------
    const int sg = tid & 31;                               // pixel group of the tile
    const int sc = tid >> 5;                               // first channel (of the stage)
    const int sp = P0 + 4 * sg;
    const bool s_ok = sp < P;                              // P % 4 == 0: a group is valid as a whole
    const int s_n = s_ok ? sp / g.HW : 0;
    const size_t s_off = ((size_t)s_n * g.Cin + sc) * g.HW + (s_ok ? sp - s_n * g.HW : 0);
    const size_t cstep = (size_t)8 * g.HW;
    float4 pr0, pr1, pr2, pr3;              // (separate variables: as an array the compiler kept them in scratch)
    auto stage_load = [&](int cb) {
        const float *__restrict__ q = s_ok ? x + s_off + (size_t)cb * P1_CB * g.HW : g.zero;
        const size_t st = s_ok ? cstep : 0;
        pr0 = *reinterpret_cast<const float4 *>(q);
        pr1 = *reinterpret_cast<const float4 *>(q + st);
        pr2 = *reinterpret_cast<const float4 *>(q + 2 * st);
        pr3 = *reinterpret_cast<const float4 *>(q + 3 * st);
    };
    auto stage_store = [&](int buf) {
        *reinterpret_cast<float4 *>(&Ps[buf][sc][4 * sg]) = pr0;
        *reinterpret_cast<float4 *>(&Ps[buf][sc + 8][4 * sg]) = pr1;
        *reinterpret_cast<float4 *>(&Ps[buf][sc + 16][4 * sg]) = pr2;
        *reinterpret_cast<float4 *>(&Ps[buf][sc + 24][4 * sg]) = pr3;
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

#ifdef FI_PROBE_1X1
    if (ep.relu & (1 << 21)) return;                      // probe: empty launch
    const int ncb = (ep.relu & (1 << 22)) ? 1 : g.Cin / P1_CB;      // probe: one stage only
#else
    const int ncb = g.Cin / P1_CB;
#endif
    stage_load(0);
    stage_store(0);
    __syncthreads();

    // 16 sub-steps (2 channel groups x 8 channel pairs) of 4 MFMAs per stage; the 16-byte LDS read of sub-step
    // i+1 and the weight loads of the next channel group are issued before the MFMAs of sub-step i
    float areg[2][8];
    float4 breg[2];
    load_a(areg[0], 0);
    for (int cb = 0; cb < ncb; ++cb) {
        const float *__restrict__ pbuf = &Ps[cb & 1][khalf * 8][4 * l31];
        const bool more = cb + 1 < ncb;
        breg[0] = *reinterpret_cast<const float4 *>(pbuf);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // the read of sub-step 0
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) {
                load_a(areg[1], 2 * cb + 1);
                // the next pixel tile is requested BEHIND the weights of the second channel group and without a
                // branch (the last stage re-reads its own tile): at the top of the stage, or inside `if (more)`, the
                // loads sit in front of the wait for the first group's weights and every stage starts with the
                // memory latency
                stage_load(more ? cb + 1 : cb);
            } else {
                // the first channel group of the NEXT stage: its register set is free from here on (at the end of the
                // stage the loads sat 0 MFMAs in front of their first use: one exposed L2 round trip per stage)
                load_a(areg[0], more ? 2 * cb + 2 : 2 * cb);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int step = h * 8 + kk;
                if (step + 1 < 16) {
                    const int h2 = (step + 1) / 8, k2 = (step + 1) % 8;
                    breg[(step + 1) & 1] = *reinterpret_cast<const float4 *>(pbuf + (h2 * 16 + k2) * P1_PP);
                }
                const float av = areg[h][kk];
                const float4 bv = breg[step & 1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[3], 0, 0, 0);
                if (kk == 0 && h == 0) __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);     // 2 weight + 4 staging loads
                if (kk == 0 && h == 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // 2 weight loads
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        if (more) stage_store((cb + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    const int po = P0 + 4 * l31;
    if (po >= P) return;
    const int n_img = po / g.HW;
    const int mb = m0 + wave * 32 + 4 * khalf;
    const size_t HW = (size_t)g.HW;
    const size_t obase = ((size_t)n_img * g.Cout + mb) * HW + (po - n_img * g.HW);
    const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr, relu = (ep.relu & 1) != 0;
#ifdef FI_PROBE_1X1
    if (ep.relu & (1 << 20)) return;       // probe: no epilogue at all
#endif
    const float *__restrict__ spp = has_sc ? ep.scale : g.zero;
    const float *__restrict__ bpp = has_bi ? ep.bias : g.zero;
    const int smul = has_sc ? 1 : 0, bmul = has_bi ? 1 : 0;
    if (g.vec4 && m0 + 128 <= g.Cout) {
        if (ep.gate) {
            if (ep.residual)
                patch_epilogue_vec<true, true>(acc, ep, y, obase, HW, mb, spp, bpp, smul, bmul, has_sc, has_bi, relu);
            else
                patch_epilogue_vec<false, true>(acc, ep, y, obase, HW, mb, spp, bpp, smul, bmul, has_sc, has_bi, relu);
        } else if (ep.residual)
            patch_epilogue_vec<true, false>(acc, ep, y, obase, HW, mb, spp, bpp, smul, bmul, has_sc, has_bi, relu);
        else
            patch_epilogue_vec<false, false>(acc, ep, y, obase, HW, mb, spp, bpp, smul, bmul, has_sc, has_bi, relu);
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m >= g.Cout) continue;
        const float sc = spp[m * smul], bi = bpp[m * bmul];
        const size_t o = obase + (size_t)((e & 3) + 8 * (e >> 2)) * HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[j][e];
            v = has_sc ? v * sc : v;
            v = has_bi ? v + bi : v;
            if (ep.residual) v += ep.residual[o + j];
            v = relu ? fmaxf(v, 0.0f) : v;
            if (ep.gate) v = ep.gate[o + j] > 0.0f ? v : 0.0f;
            y[o + j] = v;
        }
    }
}

// -------------------------------------------------------------------------------------
// weight gradient:  dW[m][k] = sum_p dY[m][p] * Xcol[k][p]
//   GEMM rows m = output channel, columns k = (ci, r, s), reduction over pixels p.
//   blockIdx.z splits the pixel range; partial sums are added with fp32 atomics.
// -------------------------------------------------------------------------------------
// HWC = true: dW columns run tap-major (k = (r*S+s)*Cin + ci, dW stored [Cout][R][S][Cin]) and
// Cin % 128 == 0, so a workgroup's 128 columns share one tap: one halo test per pixel and
// constant-stride channel gathers.
template <int BM, int TR, int TS, bool HWC>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ dy,
                                                              float *__restrict__ dw, ConvGeom g,
                                                              int p_per_split, float *__restrict__ dbias)
{
    constexpr int MT = BM / 64;
    __shared__ float As[2][BK][BM + PAD];       // dY tile  [p][m]
    __shared__ float Bs[2][BK][BN + PAD];       // Xcol tile [p][k]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = TR ? TR : g.R, S = TS ? TS : g.S;
    const int RS = R * S;
    const int K = g.K;
    const int m0 = blockIdx.y * BM;
    const int k0 = blockIdx.x * BN;
    const int OHW = g.OH * g.OW;
    const int HW = g.H * g.W;
    const int p_begin = blockIdx.z * p_per_split;
    const int p_end = min(g.P, p_begin + p_per_split);
    if (p_begin >= p_end) return;

    // A loader (dY): lane -> pixel (contiguous in memory), 16 pixels x BM channels per step
    constexpr int W_COLS = kThreads / BK;         // channels / k-columns covered per pass
    constexpr int A_LOADS = BM / W_COLS;          // scalar loads per thread
    const int ap = tid % BK;
    const int am = tid / BK;                      // + W_COLS per further load
    // B loader (input patches): lane -> pixel, column k = (ci,r,s) fixed per thread
    constexpr int B_LOADS = BN / W_COLS;
    const int bp = tid % BK;
    const int bk = tid / BK;                      // + W_COLS per further load
    int b_off[B_LOADS], b_r[B_LOADS], b_s[B_LOADS];
    bool b_ok[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int k = k0 + bk + W_COLS * i;
        b_ok[i] = k < K;
        const int kc = b_ok[i] ? k : 0;
        const int ci = HWC ? (kc % g.Cin) : (kc / RS);
        const int rs = HWC ? (kc / g.Cin) : (kc - ci * RS);
        b_r[i] = rs / S;
        b_s[i] = rs - b_r[i] * S;
        b_off[i] = ci * HW + b_r[i] * g.W + b_s[i];
    }
    // HWC: every column of this workgroup has the same tap; channels step by W_COLS
    const int h_r = b_r[0], h_s = b_s[0];
    const int h_stride = W_COLS * HW;
    int a_off[A_LOADS];                            // dY row offsets of this thread's output channels
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) a_off[i] = min(m0 + am + W_COLS * i, g.Cout - 1) * OHW;

    // as in the forward kernel: loaded values stay untouched until store_tiles(); zeroing is a mask
    float a_reg[A_LOADS], b_reg[B_LOADS];
    unsigned a_mask = 0, b_mask = 0;
    auto load_tiles = [&](int pt) {
        // pixel handled by this thread in this step (same for the A and B loaders: ap == bp)
        const int p = pt + ap;
        const bool ok = p < p_end;
        const int pc = ok ? p : p_begin;
        const int n = fast_div(pc, g.mul_ohw, g.sft_ohw);
        const int q = pc - n * OHW;
        const int oh = fast_div(q, g.mul_ow, g.sft_ow);
        const int ow = q - oh * g.OW;
        const float *__restrict__ dyn = dy + (size_t)n * g.Cout * OHW + q;
        const int ih0 = oh * g.sh - g.ph, iw0 = ow * g.sw - g.pw;
        const float *__restrict__ xn = x + (size_t)n * g.Cin * HW;
        const int pix_off = ih0 * g.W + iw0;
        if (HWC) {
            // no per-value selects: pixels past the split and taps outside the image read the zero
            // page; rows past Cout re-read the last row (their sums are never written)
            // (no integer multiplies in the loop: v_mul_lo_u32 is quarter rate)
            const float *__restrict__ ap = ok ? dyn : g.zero;
            const int a_sel = ok ? -1 : 0;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) a_reg[i] = ap[a_off[i] & a_sel];
            const bool inb = ok && ((unsigned)(ih0 + h_r) < (unsigned)g.H) && ((unsigned)(iw0 + h_s) < (unsigned)g.W);
            const float *__restrict__ bp = inb ? (xn + pix_off + b_off[0]) : g.zero;
            const size_t stride = inb ? (size_t)h_stride : 0;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                b_reg[i] = *bp;
                bp += stride;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int m = m0 + am + W_COLS * i;
            const bool inm = ok && m < g.Cout;
            a_reg[i] = dyn[inm ? m * OHW : 0];
            a_mask = (i == 0 ? 0u : a_mask) | (inm ? (1u << i) : 0u);
        }
        b_mask = 0;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const bool inb = ok && b_ok[i] && ((unsigned)(ih0 + b_r[i]) < (unsigned)g.H) &&
                             ((unsigned)(iw0 + b_s[i]) < (unsigned)g.W);
            b_reg[i] = xn[inb ? (pix_off + b_off[i]) : 0];
            b_mask |= inb ? (1u << i) : 0u;
        }
    };
    // bias gradient (sum of dY over images and pixels): the workgroups of the first column tile see
    // every dY element of their rows exactly once on its way to LDS
    const bool do_bias = dbias != nullptr && blockIdx.x == 0;
    float bsum[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) bsum[i] = 0.0f;
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const float v = (HWC || ((a_mask >> i) & 1u)) ? a_reg[i] : 0.0f;
            As[buf][ap][am + W_COLS * i] = v;
            if (do_bias) bsum[i] += v;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            Bs[buf][bp][bk + W_COLS * i] = (HWC || ((b_mask >> i) & 1u)) ? b_reg[i] : 0.0f;
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int steps = (p_end - p_begin + BK - 1) / BK;
    load_tiles(p_begin);
    store_tiles(0);
    __syncthreads();
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        if (st + 1 < steps) load_tiles(p_begin + (st + 1) * BK);
        mma_tile<BM>(As[buf], Bs[buf], wm * (BM / 2) + l31, wn * 64 + l31, khalf, acc);
        if (st + 1 < steps) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wn * 64 + j * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (m < g.Cout) atomicAdd(dw + (size_t)m * K + k, acc[i][j][e]);
            }
    }
    if (do_bias) {                 // the 16 lanes that share `am` hold the 16 pixels of a K-step
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            const int m = m0 + am + W_COLS * i;
            if (ap == 0 && m < g.Cout) atomicAdd(dbias + m, v);
        }
    }
}

// -------------------------------------------------------------------------------------
// weight gradient, "same"-size stride-1 convolutions (OH == H, OW == W), tap-major dW.
// In this GEMM BOTH operands are contiguous along the reduction index (pixels), so the tiles are
// kept row-major in LDS ([channel][16 pixels + 4]) instead of being transposed on the way in:
//   * global -> LDS: one 16-byte load and one ds_write_b128 per 4 pixels (the scalar kernel above:
//     4 loads + 4 ds_write_b32 into 4 different LDS rows) -- the transposing store pass was the
//     largest single cost of that kernel (ablation: 88 -> 117 TFLOP/s without it);
//   * LDS -> MFMA: the two half-wavefronts of v_mfma_f32_32x32x2_f32 supply k and k' of each
//     product pair; pairing (kk, kk + 8) makes a lane's 8 operands of a K-step CONTIGUOUS
//     (pixels 8*khalf .. 8*khalf+7 of its row): 2 ds_read_b128 instead of 8 ds_read_b32.
//     Row pitch 20 floats: 16 lanes x 16 bytes cover all 64 banks exactly once.
// For a same-size convolution the input pixel of tap (r, s) is q + (r-ph)*W + (s-pw) for output
// pixel q, so 4 consecutive output pixels read 4 consecutive input floats (across row ends too);
// halo validity is a 4-bit mask applied when the tile is written to LDS.  Loads go through buffer
// descriptors: an offset past the tensor returns 0 per dword, which covers the ragged pixel tail and
// the last rows of the tensor.  A NEGATIVE start offset (first channel of the first image only)
// would zero the whole 16 bytes, so those few K-steps use 4-byte loads (workgroup-uniform branch).
// -------------------------------------------------------------------------------------
// HALF: Cin == 64 -- a workgroup's 128 columns are the 64 channels of TWO consecutive taps (rows 0..63
// of the X tile belong to tap t, rows 64..127 to tap t+1), so the C2-stage layers use this kernel too.
// U16: OW % 16 == 0 -- the 16 pixels of a K-step lie inside ONE row, so image / row / column of the step are
// wave-uniform: the pixel state advances on the scalar unit, the row halo becomes a scalar select of the load
// offset and the column halo a fix-up of one element on the 2 steps of a row that touch an edge (uniform
// branch).  v_mfma and the other vector instructions of a wavefront's SIMD do not overlap (64 extra v_add per
// K-step cost 8 % on this kernel), so every vector instruction removed from the K loop is MFMA time: the general
// path spends ~80 of them per 32 MFMAs on per-lane pixel bookkeeping and halo masks, this one ~15.
template <int BM, int TR, int TS, bool HALF = false, bool U16 = false>
__global__ __launch_bounds__(kThreads, 4) void conv_wgrad_vec_kernel(const float *__restrict__ x_arg,
                                                                     const float *__restrict__ dy_arg,
                                                                     float *__restrict__ dw_arg, ConvGeom g,
                                                                     int p_per_split, float *__restrict__ dbias_arg,
                                                                     WgradBatch wb)
{
    const float *__restrict__ x = x_arg;
    const float *__restrict__ dy = dy_arg;
    float *__restrict__ dw = dw_arg;
    float *__restrict__ dbias = dbias_arg;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int MT = BM / 64;
    constexpr int PITCH = BK + 4;
    __shared__ __attribute__((aligned(16))) float As[2][BM][PITCH];    // dY   [m][pixel]
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][PITCH];    // Xcol [column][pixel]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int S = TS ? TS : g.S;
    const int K = g.K;
    // XCD-aware order (1-D launch): the workgroups, sorted by (pixel split, tile), are cut into 8 contiguous
    // bands, one per XCD.  The (tap, channel-block, Cout-block) tiles of a pixel split walk the same dY / X
    // ranges in lockstep, so a split is fetched into ONE L2 and re-read there; in the plain order its 36 tiles
    // sit on all 8 XCDs and each streams its ranges from memory (PMC: 6.0 GB fetched per launch for 0.54 GB of
    // operands on the P2-level layers -- the kernel ran at the HBM/Infinity-Cache rate, not at the MFMA rate).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.swz) {
        const int tiles = g.wg_gx * g.wg_gy;
        const int nblk1 = tiles * g.wg_splits;                   // workgroups of one problem
        const int nblk = wb.n ? nblk1 * wb.n : nblk1;
        const int per_xcd = (nblk + 7) >> 3;
        int idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (idx >= nblk) return;
        if (wb.n) {
            // batched launch: (problem, pixel split, tile) order, cut into the 8 XCD bands like a single problem --
            // a problem's operands stay in one L2
            const int prob = idx / nblk1;
            idx -= prob * nblk1;
            x = wb.x[prob];
            dy = wb.dy[prob];
            dw = wb.dw[prob];
            dbias = wb.db[prob];
        }
        bz = idx / tiles;
        const int t = idx - bz * tiles;
        by = t / g.wg_gx;
        bx = t - by * g.wg_gx;
    }
    const int m0 = by * BM;
    const int k0 = bx * BN;                      // 128 columns = 128 input channels of ONE tap
    if (g.n_live && m0 >= *g.n_live) return;     // fi_gemm_nt: rows past the live count (their slabs are not reduced either)
    const int HW = g.H * g.W;                    // == OH*OW
    const int p_begin = bz * p_per_split;
    const int p_end = min(g.P, p_begin + p_per_split);
    if (p_begin >= p_end) return;

    constexpr int NTAP = HALF ? 2 : 1;
    const int RS = (TR ? TR : g.R) * S;
    const int rs0 = k0 / g.Cin;
    const int ci0 = HALF ? 0 : (k0 - rs0 * g.Cin);
    int dr_t[NTAP], ds_t[NTAP], off_t[NTAP];
    bool tap_ok[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int rs = rs0 + t;
        tap_ok[t] = rs < RS;
        const int rc = tap_ok[t] ? rs : 0;
        const int r = rc / S, s = rc - (rc / S) * S;
        dr_t[t] = r - g.ph;
        ds_t[t] = s - g.pw;
        off_t[t] = dr_t[t] * g.W + ds_t[t];
    }
    const int off_min = (NTAP == 2) ? min(off_t[0], off_t[1]) : off_t[0];

    const int lj = tid & 3;                      // which 4 of the step's 16 pixels
    // row 0..63 (+64 per further load).  The 4 lane quads of a 16-lane group take rows 0, 4, 8, 12 (+ group index):
    // with a pitch of 5 16-byte chunks their ds_write_b128 start at chunks 0, 4, 8, 12 (mod 16) -- all 64 banks once;
    // consecutive rows would start at 0, 5, 10, 15 and the last quad would run into the first one's banks.
    const int lr = (wave << 4) | (((tid >> 2) & 3) << 2) | ((tid >> 4) & 3);
    constexpr int A_LOADS = BM / 64;
    constexpr int B_LOADS = BN / 64;
    int a_row4[A_LOADS], b_row4[B_LOADS];        // byte offsets of this thread's rows
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) a_row4[i] = min(m0 + lr + 64 * i, g.Cout - 1) * HW * 4;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) b_row4[i] = (HALF ? lr : (ci0 + lr + 64 * i)) * HW * 4;
    const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(dy), 0, (int)((size_t)g.N * g.Cout * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (int)((size_t)g.N * g.Cin * HW * 4), 0x00020000);
    const int kOutOfRange = 0x7ffffff0;
    // K-steps whose input offsets can be negative: first channel block, first image, first rows
    const bool neg_block = (ci0 == 0) && (off_min < 0);

    u32x4 a_reg[A_LOADS], b_reg[B_LOADS];
    unsigned b_mask[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) b_mask[t] = 0;
    // Pixel state of this thread's 4-pixel group, advanced by 16 pixels per K-step with adds and
    // compares only (VALU instructions in the K loop come straight out of the MFMA issue budget):
    // image cn, pixel cq = coh*OW + cow, and the byte offsets of the group in dY / X.
    constexpr bool k1x1 = (TR == 1 && TS == 1);
    constexpr bool k3x3 = (TR == 3 && TS == 3);
    int cp = p_begin + 4 * lj;
    int cn = fast_div(cp, g.mul_ohw, g.sft_ohw);
    int cq = cp - cn * HW;
    int coh = fast_div(cq, g.mul_ow, g.sft_ow);
    int cow = cq - coh * g.OW;
    int a_cur = (cn * g.Cout * HW + cq) * 4;
    int b_cur = (cn * g.Cin * HW + cq) * 4;                    // + off_t[t]*4: may be negative when neg_block
    const int adv_h = BK / g.OW, adv_w = BK - adv_h * g.OW;
    const int a_wrap = (g.Cout - 1) * HW * 4, b_wrap = (g.Cin - 1) * HW * 4;
    auto halo_mask = [&](int dr, int ds, bool ok) -> unsigned {
        unsigned mk = 0;
        if (k1x1) {
            mk = ok ? 0xFu : 0u;
        } else if (k3x3) {
            const int wrap = g.OW - cow;                        // elements e >= wrap sit on the next row
            const unsigned low = wrap >= 4 ? 0xFu : ((1u << wrap) - 1u);
            const bool row0 = (unsigned)(coh + dr) < (unsigned)g.H;
            const bool row1 = (unsigned)(coh + 1 + dr) < (unsigned)g.H;
            mk = (row0 ? low : 0u) | (row1 ? (0xFu & ~low) : 0u);
            if (ds < 0) {                                       // the element in column 0 has no left neighbour
                const unsigned kill = (cow == 0) ? 1u : (wrap < 4 ? (1u << wrap) : 0u);
                mk &= ~kill;
            } else if (ds > 0) {                                // the element in column OW-1 has no right neighbour
                const int e1 = g.OW - 1 - cow;
                mk &= ~(e1 < 4 ? (1u << e1) : 0u);
            }
            mk = ok ? mk : 0u;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int owe = cow + e, ohe = coh;
                if (owe >= g.OW) {
                    owe -= g.OW;
                    ohe += 1;
                }
                const bool v = ok && ((unsigned)(ohe + dr) < (unsigned)g.H) && ((unsigned)(owe + ds) < (unsigned)g.W);
                mk |= v ? (1u << e) : 0u;
            }
        }
        return mk;
    };
    // U16 state (uniform -> SGPRs): first pixel of the step, its image / row / column, byte offsets of the step
    int u_cp = p_begin, u_cq = 0, u_coh = 0, u_cow = 0, u_a = 0, u_b = 0;
    int a_lane[A_LOADS], b_lane[B_LOADS];        // per-thread part of the load offsets (row + the 4-pixel group)
    bool u_edge[NTAP];                           // this step's tile has a column-halo element (latched for the store)
    int u_bl = 0;                                // u_b of the step whose tile is in the registers
#pragma unroll
    for (int t = 0; t < NTAP; ++t) u_edge[t] = false;
    if constexpr (U16) {
        const int n0 = fast_div(p_begin, g.mul_ohw, g.sft_ohw);
        u_cq = p_begin - n0 * HW;
        u_coh = fast_div(u_cq, g.mul_ow, g.sft_ow);
        u_cow = u_cq - u_coh * g.OW;
        u_a = (n0 * g.Cout * HW + u_cq) * 4;
        u_b = (n0 * g.Cin * HW + u_cq) * 4;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) a_lane[i] = a_row4[i] + 16 * lj;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) b_lane[i] = b_row4[i] + 16 * lj;
    }
    auto load_tiles = [&](int pt) {
        if constexpr (U16) {
            const bool ok = u_cp < p_end;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i)
                a_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(dy_rsrc, ok ? a_lane[i] + u_a : kOutOfRange, 0, 0);
            bool rv[NTAP];                       // the tap's input row exists (else: zeros via an out-of-range offset)
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                rv[t] = ok && tap_ok[t] && (unsigned)(u_coh + dr_t[t]) < (unsigned)g.H;
                u_edge[t] = rv[t] && ((ds_t[t] < 0 && u_cow == 0) || (ds_t[t] > 0 && u_cow == g.OW - BK));
            }
            const int b_base = u_b;
            u_cp += BK;
            u_cq += BK;
            u_cow += BK;
            u_a += BK * 4;
            u_b += BK * 4;
            if (u_cow >= g.OW) {
                u_cow = 0;
                u_coh += 1;
            }
            if (u_cq >= HW) {
                u_cq -= HW;
                u_coh = 0;
                u_a += a_wrap;
                u_b += b_wrap;
            }
            // The only negative offset of a valid row: tap (0, -1) at the first pixel of the tensor (-4 bytes; a
            // negative offset would zero all 16 bytes).  That group is loaded from offset 0 instead and shifted by one
            // element when the tile is stored (it is a column-halo step: the fix-up branch below runs anyway).
            u_bl = b_base;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int t = HALF ? i : 0;
                int o = b_base + off_t[t] * 4 + b_lane[i];
                if (neg_block) o = max(o, 0);
                b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, rv[t] ? o : kOutOfRange, 0, 0);
            }
            return;
        }
        const bool ok = cp < p_end;
        const int a_off = ok ? a_cur : kOutOfRange;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            a_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(dy_rsrc, a_off + (ok ? a_row4[i] : 0), 0, 0);
        // halo masks of the 4 pixels (they may continue on the next row), one per tap of this tile
#pragma unroll
        for (int t = 0; t < NTAP; ++t) b_mask[t] = halo_mask(dr_t[t], ds_t[t], ok && tap_ok[t]);
        const int b_base = b_cur;
        // advance to the next K-step
        cp += BK;
        cq += BK;
        cow += adv_w;
        coh += adv_h;
        a_cur += BK * 4;
        b_cur += BK * 4;
        if (cow >= g.OW) {
            cow -= g.OW;
            coh += 1;
        }
        if (cq >= HW) {
            cq -= HW;
            coh -= g.OH;
            a_cur += a_wrap;
            b_cur += b_wrap;
        }
        if (neg_block && pt + off_min < 0 && pt < HW) {             // workgroup-uniform, a handful of K-steps
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                // per-element offsets: an element before the tensor start is always a halo element, and
                // giving it an out-of-range offset also keeps the compiler from re-merging the four loads
                const int t = HALF ? i : 0;
                const unsigned mk = b_mask[t];
                const int o = b_base + off_t[t] * 4 + b_row4[i];
                b_reg[i].x = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (mk & 1u) ? o : kOutOfRange, 0, 0);
                b_reg[i].y = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (mk & 2u) ? o + 4 : kOutOfRange, 0, 0);
                b_reg[i].z = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (mk & 4u) ? o + 8 : kOutOfRange, 0, 0);
                b_reg[i].w = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (mk & 8u) ? o + 12 : kOutOfRange, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int t = HALF ? i : 0;
            const unsigned mk = b_mask[t];
            b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(
                x_rsrc, mk ? (b_base + off_t[t] * 4 + b_row4[i]) : kOutOfRange, 0, 0);
        }
    };
    const bool do_bias = dbias != nullptr && bx == 0;      // see conv_wgrad_kernel
    f32x4_v bsum[A_LOADS];                                 // per 4-pixel group element: two v_pk_add_f32 per load
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) bsum[i] = f32x4_v{0.0f, 0.0f, 0.0f, 0.0f};
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            *reinterpret_cast<u32x4 *>(&As[buf][lr + 64 * i][4 * lj]) = a_reg[i];
            if (do_bias) {
                asm volatile("");                    // keep this a (uniform) branch: if-converted, every launch pays for it
                bsum[i] += __builtin_bit_cast(f32x4_v, a_reg[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            if constexpr (U16) {
                // rows outside the image were loaded as zeros.  The element beside the map (2 steps of a row) is
                // zeroed in LDS behind the thread's own 16-byte store -- patching the registers instead would put a
                // copy of the tile on the common path.
                const int t = HALF ? i : 0;
                float *__restrict__ dst = &Bs[buf][lr + 64 * i][4 * lj];
                *reinterpret_cast<u32x4 *>(dst) = b_reg[i];
                if (!k1x1 && u_edge[t]) {
                    if (ds_t[t] < 0) {
                        if (neg_block && u_bl + off_t[t] * 4 + b_lane[i] < 0) {       // loaded one element to the right
                            dst[3] = __uint_as_float(b_reg[i].z);
                            dst[2] = __uint_as_float(b_reg[i].y);
                            dst[1] = __uint_as_float(b_reg[i].x);
                        }
                        if (lj == 0) dst[0] = 0.0f;
                    } else if (lj == 3) {
                        dst[3] = 0.0f;
                    }
                }
            } else {
                u32x4 v = b_reg[i];
                if (!k1x1) {           // a 1x1 tile has no halo: an invalid group was loaded as zeros
                    const unsigned mk = b_mask[HALF ? i : 0];
                    v.x = (mk & 1u) ? v.x : 0u;
                    v.y = (mk & 2u) ? v.y : 0u;
                    v.z = (mk & 4u) ? v.z : 0u;
                    v.w = (mk & 8u) ? v.w : 0u;
                }
                *reinterpret_cast<u32x4 *>(&Bs[buf][lr + 64 * i][4 * lj]) = v;
            }
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int steps = (p_end - p_begin + BK - 1) / BK;
    load_tiles(p_begin);
    store_tiles(0);
    __syncthreads();
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        if (st + 1 < steps) load_tiles(p_begin + (st + 1) * BK);
        float4 af[MT][2], bf[2][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float *rowp = &As[buf][wm * (BM / 2) + i * 32 + l31][8 * khalf];
            af[i][0] = *reinterpret_cast<const float4 *>(rowp);
            af[i][1] = *reinterpret_cast<const float4 *>(rowp + 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float *rowp = &Bs[buf][wn * 64 + j * 32 + l31][8 * khalf];
            bf[j][0] = *reinterpret_cast<const float4 *>(rowp);
            bf[j][1] = *reinterpret_cast<const float4 *>(rowp + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 a4 = af[i][kk >> 2], b4 = bf[j][kk >> 2];
                    const float a = (kk & 3) == 0 ? a4.x : (kk & 3) == 1 ? a4.y : (kk & 3) == 2 ? a4.z : a4.w;
                    const float b = (kk & 3) == 0 ? b4.x : (kk & 3) == 1 ? b4.y : (kk & 3) == 2 ? b4.z : b4.w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                }
        if (st + 1 < steps) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wn * 64 + j * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (m < g.Cout) {
                    if (g.dw_slab)
                        dw[(size_t)bz * g.dw_slab + (size_t)m * K + k] = acc[i][j][e];
                    else
                        atomicAdd(dw + (size_t)m * K + k, acc[i][j][e]);
                }
            }
    }
    if (do_bias) {                 // the 4 lanes that share `lr` hold the 16 pixels of a K-step
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            float v = (bsum[i].x + bsum[i].y) + (bsum[i].z + bsum[i].w);
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            const int m = m0 + lr + 64 * i;
            if (lj == 0 && m < g.Cout) atomicAdd(dbias + m, v);
        }
    }
}

// 64-row tiles for narrow layers, and for grids that would leave the chip under-filled with
// 128-row tiles (fewer than 2 workgroups per CU): C4/C5 of the backbone at batch 4.
bool use_bm64(int Cout, int P)
{
    if (Cout <= 64) return true;
    const long tiles128 = (long)fi::ceil_div(P, BN) * fi::ceil_div(Cout, 128);
    return tiles128 < 512;
}

// conv3x3_patch_kernel: 3x3 / stride 1 / pad 1, same-size NCHW output, tap-major weights, 16-channel blocks,
// 128-row Cout tiles, and at least 256 workgroups (one per CU: measured faster than the 64x64-tile kernel down
// to there -- C4 of ResNet at batch 4 -- and slower below).
// Returns 0 (not eligible), 1 (2-D tiles: width a multiple of 16) or 2 (flat tiles: even widths 12..16, e.g. the
// 14 x 14 RoI maps; 8-byte aligned tensors).
int patch_eligible(const ConvGeom &g, bool hwc, int weight_layout, const float *x, const float *y,
                   const float *residual)
{
    if (!(g.R == 3 && g.S == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1)) return 0;
    if (!hwc || weight_layout < 1 || g.Cout <= 64 || g.OH != g.H || g.OW != g.W) return 0;
    if (g.out_nhwc && !(g.W % PT_TW == 0 && g.Cout % 128 == 0 && residual == nullptr && (uintptr_t)y % 16 == 0)) return 0;
    if ((long)g.N * g.Cin * g.H * g.W >= 2147483647L || (long)g.N * g.Cout * g.H * g.W >= 2147483647L) return 0;
    const long mt = fi::ceil_div(g.Cout, 128);
    if (g.W % PT_TW == 0)
        return (long)fi::ceil_div(g.N * g.H, PT_TH) * (g.W / PT_TW) * mt >= 256 ? 1 : 0;
    // flat tiles: 128 consecutive pixels touch at most (W + 126) / W rows, + 2 halo rows <= 13 patch rows
    if (g.W < 16 && g.W % 2 == 0 && (g.W + 126) / g.W + 2 <= PatchShape<true>::ROWS && (uintptr_t)x % 8 == 0 &&
        (uintptr_t)y % 8 == 0 &&
        (residual == nullptr || (uintptr_t)residual % 8 == 0))
        return (long)fi::ceil_div(g.N * g.H * g.W, 128) * mt >= 512 ? 2 : 0;
    return 0;
}

int window_class(int R, int S)
{
    if (R == 1 && S == 1) return 0;
    if (R == 3 && S == 3) return 1;
    if (R == 7 && S == 7) return 2;
    return 3;
}

// device address of g_zero_page on the current device (looked up once per device)
const float *zero_page()
{
    static const float *cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!cache[dev]) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
        cache[dev] = static_cast<const float *>(p);
    }
    return cache[dev];
}

int make_geom(ConvGeom &g, int N, int Cin, int H, int W, int Cout, int R, int S, int sh, int sw,
              int ph, int pw, int out_h = 0, int out_w = 0)
{
    FI_REQUIRE(N >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1, "sizes must be positive");
    FI_REQUIRE(R >= 1 && S >= 1 && sh >= 1 && sw >= 1 && ph >= 0 && pw >= 0, "bad window");
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.R = R; g.S = S;
    g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw;
    g.out_nhwc = 0;
    g.flip = 0;
    g.swz = 0;
    g.dw_slab = 0;
    g.n_live = nullptr;
    g.vec_out = 0;
    g.zero = nullptr;      // set by the entry points once the arguments are validated (needs the device)
    g.p_base = 0;
    g.OH = out_h > 0 ? out_h : (H + 2 * ph - R) / sh + 1;   // explicit size: taps past the input read zeros
    g.OW = out_w > 0 ? out_w : (W + 2 * pw - S) / sw + 1;
    FI_REQUIRE(g.OH >= 1 && g.OW >= 1, "empty output");
    const long K = (long)Cin * R * S, P = (long)N * g.OH * g.OW;
    FI_REQUIRE(K < 2147483647L && P < 2147483647L, "problem too large for int32 indexing");
    FI_REQUIRE((long)N * Cin * H * W < 2147483647L * 2 && (long)Cout * K < 2147483647L, "tensor too large");
    g.K = (int)K;
    g.P = (int)P;
    auto magic = [](unsigned d, unsigned &mul, unsigned &sft) {
        if (d <= 1) {
            mul = 0;
            sft = 0;
            return;
        }
        unsigned L = 0;
        while ((1ull << L) < d) ++L;                   // ceil(log2 d)
        const unsigned total = 31 + L;                 // 2^total / d < 2^32
        const unsigned long long num = 1ull << total;
        mul = (unsigned)((num + d - 1) / d);
        sft = total - 32;                              // L >= 1 for d >= 2
    };
    magic((unsigned)(g.OH * g.OW), g.mul_ohw, g.sft_ohw);
    magic((unsigned)g.OW, g.mul_ow, g.sft_ow);
    return FI_OK;
}

template <int BM>
void launch_fwd(const ConvGeom &g_in, const float *x, const float *w, const Epilogue &ep, float *y,
                bool hwc, hipStream_t st)
{
    ConvGeom g = g_in;
    g.vec_out = (!g.out_nhwc && (g.OH * g.OW) % 4 == 0 && ((uintptr_t)y % 16 == 0) &&
                 (ep.residual == nullptr || (uintptr_t)ep.residual % 16 == 0) &&
                 (ep.gate == nullptr || (uintptr_t)ep.gate % 16 == 0)) ? 1 : 0;
    // 64-pixel tiles when even 64-row tiles leave fewer than 4 workgroups per CU (C4/C5 of the backbone
    // at batch 4): twice the workgroups, so that 4 wavefronts share each SIMD's MFMA pipe
    if constexpr (BM == 64) {
        const long wgs = (long)fi::ceil_div(g.P, BN) * fi::ceil_div(g.Cout, BM);
        if (hwc && !g.out_nhwc && wgs < 1024 && g.P >= 64) {
            dim3 grid64(fi::ceil_div(g.P, 64), fi::ceil_div(g.Cout, BM));
            if (g.R == 3 && g.S == 3)
                hipLaunchKernelGGL((conv_fwd_kernel<64, 3, 3, true, false, 64>), grid64, dim3(kThreads), 0, st, x, w, ep, y, g);
            else if (g.R == 1 && g.S == 1)
                hipLaunchKernelGGL((conv_fwd_kernel<64, 1, 1, true, false, 64>), grid64, dim3(kThreads), 0, st, x, w, ep, y, g);
            else
                hipLaunchKernelGGL((conv_fwd_kernel<64, 0, 0, true, false, 64>), grid64, dim3(kThreads), 0, st, x, w, ep, y, g);
            return;
        }
    }
    // Tail launch.  All tiles of a layer cost the same, so a grid of r.f "rounds" of the 1024 resident
    // workgroups ends with a round that uses a fraction f of the chip (mask head: 6272 tiles = 6.125
    // rounds).  A small remainder is cut off and run as 64x64 tiles -- 4x the workgroups, a quarter of the
    // work each -- so that it spreads over all CUs.
    if constexpr (BM == 128) {
        const long ny = fi::ceil_div(g.Cout, BM);
        const long nx_all = fi::ceil_div(g.P, BN);
        const long wgs = nx_all * ny, slots = 1024;
        const long rem = wgs % slots;
        if (hwc && !g.out_nhwc && g.p_base == 0 && wgs > slots && rem > 0 && rem * 3 < slots && slots % ny == 0) {
            const long nx_main = (wgs / slots) * (slots / ny);
            ConvGeom gm = g, gt = g;
            gm.P = (int)(nx_main * BN);
            launch_fwd<128>(gm, x, w, ep, y, hwc, st);          // p_base == 0, wgs % slots == 0: no recursion
            gt.p_base = gm.P;
            dim3 grid64(fi::ceil_div(g.P - gm.P, 64), fi::ceil_div(g.Cout, 64));
            if (g.R == 3 && g.S == 3)
                hipLaunchKernelGGL((conv_fwd_kernel<64, 3, 3, true, false, 64>), grid64, dim3(kThreads), 0, st, x, w, ep, y, gt);
            else if (g.R == 1 && g.S == 1)
                hipLaunchKernelGGL((conv_fwd_kernel<64, 1, 1, true, false, 64>), grid64, dim3(kThreads), 0, st, x, w, ep, y, gt);
            else
                hipLaunchKernelGGL((conv_fwd_kernel<64, 0, 0, true, false, 64>), grid64, dim3(kThreads), 0, st, x, w, ep, y, gt);
            return;
        }
    }
    int nx = fi::ceil_div(g.P, BN);
    // XCD-aware order for large compute-bound grids; short-K 1x1 layers are bound by their output
    // stream and small grids by occupancy -- both measured faster in the plain order
    // (not with a device-side live count: the live tiles are the FIRST ones, and a band per XCD would hand them all to one
    // or two XCDs)
    if (nx >= 512 && (g.R * g.S > 1 || g.K >= 128) && !g.n_live) {
        g.swz = 1;
        nx = fi::ceil_div(nx, 8) * 8;
    }
    dim3 grid(nx, fi::ceil_div(g.Cout, BM));
    if (hwc && g.out_nhwc) {
        if (g.R == 3 && g.S == 3)
            hipLaunchKernelGGL((conv_fwd_kernel<BM, 3, 3, true, true>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
        else if (g.R == 1 && g.S == 1)
            hipLaunchKernelGGL((conv_fwd_kernel<BM, 1, 1, true, true>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
        else
            hipLaunchKernelGGL((conv_fwd_kernel<BM, 0, 0, true, true>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
        return;
    }
    if (hwc) {
        if (g.R == 3 && g.S == 3)
            hipLaunchKernelGGL((conv_fwd_kernel<BM, 3, 3, true>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
        else if (g.R == 1 && g.S == 1)
            hipLaunchKernelGGL((conv_fwd_kernel<BM, 1, 1, true>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
        else
            hipLaunchKernelGGL((conv_fwd_kernel<BM, 0, 0, true>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
        return;
    }
    if (g.R == 3 && g.S == 3)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 3, 3, false>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
    else if (g.R == 7 && g.S == 7)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 7, 7, false>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 0, 0, false>), grid, dim3(kThreads), 0, st, x, w, ep, y, g);
}

// same-size stride-1 layers (every 3x3/pad-1 and 1x1 layer of the model) can use the row-major kernel
bool wgrad_same_size(const ConvGeom &g, const float *x, const float *dy)
{
    return g.sh == 1 && g.sw == 1 && g.OH == g.H && g.OW == g.W && (g.H * g.W) % 4 == 0 && g.W >= 4 &&
           (size_t)g.N * g.Cin * g.H * g.W * 4 < 0x7fffff00ULL &&
           (size_t)g.N * g.Cout * g.H * g.W * 4 < 0x7fffff00ULL && ((uintptr_t)x % 16 == 0) &&
           ((uintptr_t)dy % 16 == 0);
}

template <int BM>
void launch_wgrad(const ConvGeom &g, const float *x, const float *dy, float *dw, int splits,
                  int p_per_split, bool hwc, float *dbias, hipStream_t st, const WgradBatch *batch = nullptr)
{
    dim3 grid(fi::ceil_div(g.K, BN), fi::ceil_div(g.Cout, BM), splits);
    const bool same = wgrad_same_size(g, x, dy);
    WgradBatch wbatch;
    wbatch.n = 0;
    if (batch) wbatch = *batch;              // (only the row-major kernel below reads it; checked by the caller)
    if (hwc && same && g.Cin == 64) {            // two taps per 128-column tile
        if (g.R == 3 && g.S == 3)
            if (g.OW % BK == 0)
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 3, 3, true, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias, wbatch);
            else
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 3, 3, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias, wbatch);
        else if (g.R == 1 && g.S == 1)
            if (g.OW % BK == 0)
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 1, 1, true, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias, wbatch);
            else
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 1, 1, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias, wbatch);
        else
            hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 0, 0, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias, wbatch);
        return;
    }
    if (hwc && same) {
        ConvGeom gs = g;
        dim3 vgrid = grid;
        if ((g.P >= 32768 && !getenv("FI_NO_WG_SWZ")) || wbatch.n) {       // XCD-aware 1-D launch (see the kernel)
            gs.swz = 1;
            gs.wg_gx = (int)grid.x; gs.wg_gy = (int)grid.y; gs.wg_splits = splits;
            const long nblk = (long)grid.x * grid.y * splits * (wbatch.n ? wbatch.n : 1);
            vgrid = dim3((unsigned)(((nblk + 7) / 8) * 8), 1, 1);
        }
        if (g.R == 3 && g.S == 3)
            if (g.OW % BK == 0)
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 3, 3, false, true>), vgrid, dim3(kThreads), 0, st, x, dy, dw, gs, p_per_split, dbias, wbatch);
            else
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 3, 3>), vgrid, dim3(kThreads), 0, st, x, dy, dw, gs, p_per_split, dbias, wbatch);
        else if (g.R == 1 && g.S == 1)
            if (g.OW % BK == 0)
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 1, 1, false, true>), vgrid, dim3(kThreads), 0, st, x, dy, dw, gs, p_per_split, dbias, wbatch);
            else
                hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 1, 1>), vgrid, dim3(kThreads), 0, st, x, dy, dw, gs, p_per_split, dbias, wbatch);
        else
            hipLaunchKernelGGL((conv_wgrad_vec_kernel<BM, 0, 0>), vgrid, dim3(kThreads), 0, st, x, dy, dw, gs, p_per_split, dbias, wbatch);
        return;
    }
    if (hwc) {
        if (g.R == 3 && g.S == 3)
            hipLaunchKernelGGL((conv_wgrad_kernel<BM, 3, 3, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
        else if (g.R == 1 && g.S == 1)
            hipLaunchKernelGGL((conv_wgrad_kernel<BM, 1, 1, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
        else
            hipLaunchKernelGGL((conv_wgrad_kernel<BM, 0, 0, true>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
        return;
    }
    if (g.R == 3 && g.S == 3)
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, 3, 3, false>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
    else if (g.R == 1 && g.S == 1)
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, 1, 1, false>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
    else if (g.R == 7 && g.S == 7)
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, 7, 7, false>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, 0, 0, false>), grid, dim3(kThreads), 0, st, x, dy, dw, g, p_per_split, dbias);
}

// -------------------------------------------------------------------------------------
// Backward of the fused epilogue  y = act(z*scale + shift [+ residual]),  z = conv output:
//   g  = dy * (y > 0)            (relu)   or dy
//   dz = g * scale[c]            -> feeds dgrad / wgrad
//   dshift[c] = sum g            (= d beta; the conv-bias gradient is dshift*scale)
//   dgamma[c] = sum g * (y - beta[c]) / gamma[c]      (x_hat recovered from y where g != 0)
// One pass over dy and y; one workgroup per (channel, image chunk); fp32 atomics for the sums.
// -------------------------------------------------------------------------------------
// HAS_RES / HAS_G are template parameters so that the loop body is ONE basic block: with run-time tests the
// shortcut load sat behind a branch and was issued only after dy and y had arrived (two dependent memory
// round trips per 16 bytes and lane); here the 2 x (2..3) loads of two consecutive 16-byte groups are issued
// together.
template <bool HAS_RES, bool HAS_G>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float *__restrict__ dy,
                                                         const float *__restrict__ y,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta,
                                                         const float *__restrict__ residual, int N, int C,
                                                         int HW, int relu,
                                                         float *__restrict__ dz, float *__restrict__ g_out,
                                                         float *__restrict__ dshift,
                                                         float *__restrict__ dgamma, float *__restrict__ dbias,
                                                         int imgs_per_block)
{
    __shared__ float s_a[4], s_b[4];
    const int c = blockIdx.x;
    const int n0 = blockIdx.y * imgs_per_block;
    const int n1 = min(N, n0 + imgs_per_block);
    const float sc = scale[c];
    const float be = beta ? beta[c] : 0.0f;
    float sum_g = 0.0f, sum_gy = 0.0f;
    const bool vec = (HW & 3) == 0;
    const bool norelu = relu == 0;
    auto group = [&](const float4 d, float4 v, const float4 r, size_t at) {
        float4 g;
        g.x = (norelu || v.x > 0.0f) ? d.x : 0.0f;
        g.y = (norelu || v.y > 0.0f) ? d.y : 0.0f;
        g.z = (norelu || v.z > 0.0f) ? d.z : 0.0f;
        g.w = (norelu || v.w > 0.0f) ? d.w : 0.0f;
        if (HAS_RES) {   // BN output = y - shortcut wherever the gradient is non-zero
            v.x -= r.x; v.y -= r.y; v.z -= r.z; v.w -= r.w;
        }
        sum_g += (g.x + g.y) + (g.z + g.w);
        sum_gy += (g.x * (v.x - be) + g.y * (v.y - be)) + (g.z * (v.z - be) + g.w * (v.w - be));
        if (HAS_G) *reinterpret_cast<float4 *>(g_out + at) = g;
        float4 o;
        o.x = g.x * sc; o.y = g.y * sc; o.z = g.z * sc; o.w = g.w * sc;
        *reinterpret_cast<float4 *>(dz + at) = o;
    };
    for (int n = n0; vec && n < n1; ++n) {
        const size_t base = ((size_t)n * C + c) * HW;
        {
            int i = threadIdx.x * 4;
            for (; i + 1024 < HW; i += 2048) {          // two groups per trip, all loads first
                const float4 d0 = *reinterpret_cast<const float4 *>(dy + base + i);
                const float4 v0 = *reinterpret_cast<const float4 *>(y + base + i);
                const float4 d1 = *reinterpret_cast<const float4 *>(dy + base + i + 1024);
                const float4 v1 = *reinterpret_cast<const float4 *>(y + base + i + 1024);
                float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
                if (HAS_RES) {
                    r0 = *reinterpret_cast<const float4 *>(residual + base + i);
                    r1 = *reinterpret_cast<const float4 *>(residual + base + i + 1024);
                }
                group(d0, v0, r0, base + i);
                group(d1, v1, r1, base + i + 1024);
            }
            if (i < HW) {
                const float4 d0 = *reinterpret_cast<const float4 *>(dy + base + i);
                const float4 v0 = *reinterpret_cast<const float4 *>(y + base + i);
                float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (HAS_RES) r0 = *reinterpret_cast<const float4 *>(residual + base + i);
                group(d0, v0, r0, base + i);
            }
        }
    }
    if (!vec) {
        // planes whose size is not a multiple of 4 (7x7 = 49: RoI heads): the workgroup walks the flattened
        // (image, element) index of its channel, so all 256 lanes stay busy on planes smaller than 256
        const int total = (n1 - n0) * HW;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int n = idx / HW;
            const size_t at = ((size_t)(n0 + n) * C + c) * HW + (idx - n * HW);
            const float d = dy[at], v = y[at];
            const float g = (norelu || v > 0.0f) ? d : 0.0f;
            const float vb = HAS_RES ? (v - residual[at]) : v;
            sum_g += g;
            sum_gy += g * (vb - be);
            if (HAS_G) g_out[at] = g;
            dz[at] = g * sc;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sum_g += __shfl_xor(sum_g, off, 64);
        sum_gy += __shfl_xor(sum_gy, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_a[wave] = sum_g;
        s_b[wave] = sum_gy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        const float b = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
        atomicAdd(dshift + c, a);
        if (dbias) atomicAdd(dbias + c, a * sc);       // gradient of a conv bias folded into the shift
        if (dgamma) {
            const float ga = gamma[c];
            atomicAdd(dgamma + c, ga != 0.0f ? b / ga : 0.0f);
        }
    }
}

// Channels-last variant: dy and y are [N*HW][C] (the gradient arrives from the channels-last
// RoIAlign backward, y was written channels-last by the conv epilogue); dz leaves as [N][C][HW]
// for the dgrad / wgrad kernels.  64 pixels x 64 channels are transposed through LDS per step:
// reads run along channels, writes along pixels, both as 256-byte wavefront accesses.
__global__ __launch_bounds__(256) void bn_act_bwd_cl_kernel(const float *__restrict__ dy,
                                                            const float *__restrict__ y,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int P, int C,
                                                            int HW, int relu, float *__restrict__ dz,
                                                            float *__restrict__ dshift,
                                                            float *__restrict__ dgamma, float *__restrict__ dbias,
                                                            int tiles_per_block)
{
    __shared__ float s_t[64][65];
    __shared__ float s_a[4][64], s_b[4][64];
    const int c0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63;
    const int row = threadIdx.x >> 6;          // 0..3
    const int c = c0 + lane;
    const bool c_ok = c < C;
    const float sc = c_ok ? scale[c] : 0.0f;
    const float be = (c_ok && beta) ? beta[c] : 0.0f;
    float sum_g = 0.0f, sum_gy = 0.0f;
    const int t0 = blockIdx.y * tiles_per_block;
    for (int t = t0; t < t0 + tiles_per_block; ++t) {
        const int pbase = t * 64;
        if (pbase >= P) break;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int pl = row + 4 * i;
            const int pix = pbase + pl;
            float o = 0.0f;
            if (c_ok && pix < P) {
                const float d = dy[(size_t)pix * C + c];
                const float v = y[(size_t)pix * C + c];
                const float gq = (!relu || v > 0.0f) ? d : 0.0f;
                sum_g += gq;
                sum_gy += gq * (v - be);
                o = gq * sc;
            }
            s_t[lane][pl] = o;
        }
        __syncthreads();
        const int pix = pbase + lane;
        if (pix < P) {
            const int n = pix / HW;
            const int q = pix - n * HW;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int cc = row + 4 * j;
                if (c0 + cc < C) dz[((size_t)n * C + c0 + cc) * HW + q] = s_t[cc][lane];
            }
        }
        __syncthreads();
    }
    s_a[row][lane] = sum_g;
    s_b[row][lane] = sum_gy;
    __syncthreads();
    if (row == 0 && c_ok) {
        const float a = (s_a[0][lane] + s_a[1][lane]) + (s_a[2][lane] + s_a[3][lane]);
        const float b = (s_b[0][lane] + s_b[1][lane]) + (s_b[2][lane] + s_b[3][lane]);
        atomicAdd(dshift + c, a);
        if (dbias) atomicAdd(dbias + c, a * scale[c]);
        if (dgamma) {
            const float ga = gamma[c];
            atomicAdd(dgamma + c, ga != 0.0f ? b / ga : 0.0f);
        }
    }
}

// ---- batched weight transposes ---------------------------------------------------------------------
// The data-gradient kernel wants W^T in the tap-major layout [Cin][R][S][Cout]; parameters live as
// [Cout][R][S][Cin].  All layers are re-laid-out by ONE launch per step (fi_weight_transpose_batch) instead
// of one strided copy per layer and step (~150 launches of 9 us).  A workgroup transposes one 32 x 32 tile
// of one tap of one layer through LDS; the layer is found by bisection of the tile prefix table.
__global__ __launch_bounds__(256) void weight_transpose_kernel(const FiTransposeDesc *__restrict__ descs, int n,
                                                               long total_tiles)
{
    __shared__ float s_t[32][33];
    for (long t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {                              // last descriptor with tile_base <= t
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].tile_base <= t) lo = mid; else hi = mid - 1;
        }
        const FiTransposeDesc d = descs[lo];
        const int tr = (d.rows + 31) >> 5, tc = (d.cols + 31) >> 5;
        if (d.pad_ & 1) {
            // fragment-major output for conv1x1_ring_kernel (taps == 1, rows and cols multiples of 32): the matrix D [M][K]
            // (D = src^T * row_scale, or D = src with flag 2) stored per block of 32 rows x 16 columns as
            // [2 halves of 4 k][64 lanes = (k / 8, row)][4 k]: a wavefront's A-operand load is 1 KB contiguous
            const long local = t - d.tile_base;
            const int r0 = (int)(local / tc) * 32, c0 = (int)(local % tc) * 32;
            const float *__restrict__ rs = (const float *)d.row_scale;
            const bool keep = (d.pad_ & 2) != 0;
            const int K = keep ? d.cols : d.rows;
            const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = r0 + ty + 8 * k, c = c0 + tx;
                const float v = ((const float *)d.src)[(size_t)r * d.cols + c];
                s_t[ty + 8 * k][tx] = (rs && !keep) ? v * rs[r] : v;
            }
            __syncthreads();
            // element (m, kq) of the 32 x 32 tile of D, 4 consecutive k per thread: 256 threads x 4 = 1024
            const int m = threadIdx.x & 31, kq = (threadIdx.x >> 5) * 4;      // kq = 0, 4, .., 28
            float4 v;
            if (keep) v = make_float4(s_t[m][kq], s_t[m][kq + 1], s_t[m][kq + 2], s_t[m][kq + 3]);
            else v = make_float4(s_t[kq][m], s_t[kq + 1][m], s_t[kq + 2][m], s_t[kq + 3][m]);
            const int Mrow = (keep ? r0 : c0) + m, Kcol = (keep ? c0 : r0) + kq;
            const int kg = Kcol >> 4, kh = (Kcol & 15) >> 3, k8 = Kcol & 7;           // k8 = 0 or 4
            float *__restrict__ dstf = (float *)d.dst + ((size_t)(Mrow >> 5) * (K >> 4) + kg) * 512 + (k8 >> 2) * 256 +
                                       (kh * 32 + (Mrow & 31)) * 4;
            *reinterpret_cast<float4 *>(dstf) = v;
            continue;
        }
        long local = t - d.tile_base;
        const int tap = (int)(local / ((long)tr * tc));
        local -= (long)tap * tr * tc;
        const int r0 = (int)(local / tc) * 32, c0 = (int)(local % tc) * 32;
        const float *__restrict__ src = (const float *)d.src + (size_t)tap * d.cols;           // [rows][taps][cols]
        float *__restrict__ dst = (float *)d.dst + (size_t)tap * d.rows;                       // [cols][taps][rows]
        const size_t src_pitch = (size_t)d.taps * d.cols, dst_pitch = (size_t)d.taps * d.rows;
        const float *__restrict__ rs = (const float *)d.row_scale;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            if (r < d.rows && c < d.cols)
                s_t[ty + 8 * k][tx] = rs ? src[(size_t)r * src_pitch + c] * rs[r] : src[(size_t)r * src_pitch + c];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < d.rows && c < d.cols) dst[(size_t)c * dst_pitch + r] = s_t[tx][ty + 8 * k];
        }
    }
}

}  // namespace

extern "C" {

int fi_conv2d_forward(const float *x, const float *weight, const float *bias, const float *scale,
                      const float *residual, float *y, int N, int Cin, int H, int W, int Cout, int R,
                      int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                      int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream)
{
    return fi_conv2d_forward_gated(x, weight, bias, scale, residual, nullptr, y, N, Cin, H, W, Cout, R, S, stride_h,
                                   stride_w, pad_h, pad_w, relu, weight_layout, out_h, out_w, output_layout, stream);
}

int fi_conv2d_forward_gated(const float *x, const float *weight, const float *bias, const float *scale,
                            const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                            int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                            int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream)
{
    return fi_conv2d_forward_live(x, weight, bias, scale, residual, gate, y, N, Cin, H, W, Cout, R, S, stride_h, stride_w,
                                  pad_h, pad_w, relu, weight_layout, out_h, out_w, output_layout, nullptr, stream);
}

int fi_conv2d_forward_live(const float *x, const float *weight, const float *bias, const float *scale,
                           const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                           int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                           int weight_layout, int out_h, int out_w, int output_layout, const int32_t *n_live_dev,
                           fi_stream_t stream)
{
    ConvGeom g;
    int rc = make_geom(g, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, out_h, out_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(x && weight && y, "null pointer");
    FI_REQUIRE(output_layout == 0 || output_layout == 1, "output_layout: 0 = [N][Cout][OH][OW], 1 = [N][OH][OW][Cout]");
    if (output_layout == 1) {
        FI_REQUIRE(Cout % 4 == 0 && residual == nullptr && gate == nullptr && (uintptr_t)y % 16 == 0,
                   "channels-last output needs Cout % 4 == 0, a 16-byte aligned y and no fused residual / gate");
        FI_REQUIRE(Cin % BK == 0 && R * S <= 64 && (weight_layout >= 1 || R * S == 1),
                   "channels-last output is implemented on the tap-major path (Cin % 16 == 0, weight_layout 1)");
        g.out_nhwc = 1;
    }
    FI_REQUIRE(weight_layout >= 0 && weight_layout <= 3,
               "weight_layout: 0 = [Cout][Cin][R][S], 1 = [Cout][R][S][Cin], 2 = 1 with the taps reversed, 3 = fragment-major 1x1");
    // tap-major fast path: channels-last weights (any 1x1 weight is both layouts at once)
    const bool hwc = (Cin % BK == 0) && (R * S <= 64) && (weight_layout >= 1 || R * S == 1);
    FI_REQUIRE(hwc || weight_layout == 0, "weight_layout 1/2/3 needs Cin % 16 == 0 and R*S <= 64");
    g.flip = (weight_layout == 2) ? 1 : 0;
    g.zero = zero_page();
    FI_REQUIRE(g.zero != nullptr, "zero page lookup failed (no HIP device?)");
    g.n_live = n_live_dev;        // honoured by conv_fwd_kernel (every instantiation); the patch / 1x1 kernels compute all
    hipStream_t st = (hipStream_t)stream;
#ifdef FI_PROBE_1X1
    if (getenv("FI_DBG_1X1")) relu |= (int)strtol(getenv("FI_DBG_1X1"), nullptr, 0);
#endif
    const Epilogue ep = {bias, scale, residual, relu, gate};
    const bool bm64 = use_bm64(Cout, g.P);
    // 3x3 / stride 1 / pad 1 layers with enough tiles to fill the chip: input patch in LDS (conv3x3_patch_kernel)
    int patch_mode = getenv("FI_NO_PATCH") ? 0 : patch_eligible(g, hwc, weight_layout, x, y, residual);
    if (patch_mode == 2 && gate != nullptr && (uintptr_t)gate % 8 != 0) patch_mode = 0;
    // 1x1 / stride 1 layers: weights in registers, pixel tile staged 32 channels at a time (conv1x1_reg_kernel);
    // layers with fewer than 128 input channels are bound by their output stream and measured faster on
    // conv_fwd_kernel (4 workgroups per CU)
    const bool reg1x1 = !getenv("FI_NO_REG1X1") && R == 1 && S == 1 && stride_h == 1 && stride_w == 1 && pad_h == 0 &&
                        pad_w == 0 && !g.out_nhwc && Cin % P1_CB == 0 && Cin >= 128 && Cout > 64 && (H * W) % 4 == 0 &&
                        (uintptr_t)x % 16 == 0 && (long)N * Cin * H * W < 2147483647L &&
                        (long)fi::ceil_div(N * H * W, 128) * fi::ceil_div(Cout, 128) >= 256;
    // ... and with fragment-major weights (weight_layout 3, made by fi_weight_transpose_batch): the persistent ring form
    if (weight_layout == 3) {
        FI_REQUIRE(fi_conv1x1_ring_eligible(N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, output_layout, x, y,
                                            residual, gate) == 1,
                   "weight_layout 3 (fragment-major 1x1 weights): shape or alignment not eligible for conv1x1_ring_kernel");
        fi::ProfScope prof(FI_K_CONV1X1_REG, st);
        const fi::RingArgs ra = {x, weight, bias, scale, residual, gate, y, g.zero, N, Cin, H * W, Cout, relu & 1};
        fi::launch_conv1x1_ring(ra, st);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    fi::ProfScope prof(patch_mode ? FI_K_CONV3X3_PATCH + (patch_mode - 1)
                       : reg1x1 ? FI_K_CONV1X1_REG
                                : FI_K_CONV_FWD + (bm64 ? 0 : 4) + window_class(R, S), st);
    if (reg1x1) {
        Conv1x1Geom cg;
        cg.N = N; cg.Cin = Cin; cg.HW = H * W; cg.Cout = Cout;
        cg.ptiles = fi::ceil_div(N * H * W, 128);
        cg.mtiles = fi::ceil_div(Cout, 128);
        cg.vec4 = ((uintptr_t)y % 16 == 0 && (residual == nullptr || (uintptr_t)residual % 16 == 0) &&
                   (gate == nullptr || (uintptr_t)gate % 16 == 0)) ? 1 : 0;
        cg.zero = g.zero;
        const long blocks = (long)fi::ceil_div(cg.ptiles, 8) * 8 * cg.mtiles;
        hipLaunchKernelGGL(conv1x1_reg_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, x, weight, ep, y, cg);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    if (patch_mode) {
        PatchGeom pg;
        pg.N = N; pg.Cin = Cin; pg.H = H; pg.W = W; pg.Cout = Cout;
        pg.flip = g.flip;
        pg.tiles_x = fi::ceil_div(W, PT_TW);
        pg.ptiles = patch_mode == 2 ? fi::ceil_div(N * H * W, 128) : fi::ceil_div(N * H, PT_TH) * pg.tiles_x;
        pg.mtiles = fi::ceil_div(Cout, 128);
        pg.vec4 = (W % 4 == 0 && (uintptr_t)y % 16 == 0 && (residual == nullptr || (uintptr_t)residual % 16 == 0) &&
                   (gate == nullptr || (uintptr_t)gate % 16 == 0)) ? 1 : 0;
        pg.out_nhwc = g.out_nhwc;
        pg.zero = g.zero;
        const long blocks = (long)fi::ceil_div(pg.ptiles, 8) * 8 * pg.mtiles;
        if (patch_mode == 2)
            hipLaunchKernelGGL(conv3x3_patch_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, st, x, weight, ep, y, pg);
        else
            hipLaunchKernelGGL(conv3x3_patch_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, st, x, weight, ep, y, pg);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    if (bm64)
        launch_fwd<64>(g, x, weight, ep, y, hwc, st);
    else
        launch_fwd<128>(g, x, weight, ep, y, hwc, st);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_bn_act_backward(const float *dy, const float *y, const float *scale, const float *gamma,
                       const float *beta, const float *residual, int N, int C, int HW, int relu,
                       float *dz, float *g_out, float *dshift, float *dgamma, float *dbias, int layout,
                       int flags, fi_stream_t stream)
{
    FI_REQUIRE(N >= 1 && C >= 1 && HW >= 1, "sizes must be positive");
    FI_REQUIRE(dy && y && scale && dz && dshift, "null pointer");
    FI_REQUIRE(!dgamma || gamma, "dgamma needs gamma");
    FI_REQUIRE(layout == 0 || layout == 1, "layout: 0 = dy,y [N][C][HW], 1 = dy,y [N][HW][C]");
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & FI_OUTPUTS_ZEROED)) {
        if (dgamma == dshift + C && (!dbias || dbias == dshift + 2 * C)) {      // adjacent buffers: one fill
            FI_HIP_CHECK(hipMemsetAsync(dshift, 0, sizeof(float) * (dbias ? 3 : 2) * C, st));
        } else {
            FI_HIP_CHECK(hipMemsetAsync(dshift, 0, sizeof(float) * C, st));
            if (dgamma) FI_HIP_CHECK(hipMemsetAsync(dgamma, 0, sizeof(float) * C, st));
            if (dbias) FI_HIP_CHECK(hipMemsetAsync(dbias, 0, sizeof(float) * C, st));
        }
    }
    if (layout == 1) {
        FI_REQUIRE(residual == nullptr && g_out == nullptr, "channels-last backward has no fused residual");
        FI_REQUIRE((long)N * HW < 2147483647L, "too many pixels");
        const int P = N * HW;
        const int tiles = fi::ceil_div(P, 64);
        const int cblk = fi::ceil_div(C, 64);
        int tpb = fi::ceil_div(tiles * cblk, 2048);          // ~2048 workgroups
        if (tpb < 1) tpb = 1;
        fi::ProfScope prof(FI_K_BN_ACT_BWD, st);
        hipLaunchKernelGGL(bn_act_bwd_cl_kernel, dim3(cblk, fi::ceil_div(tiles, tpb)), dim3(256), 0, st, dy, y,
                           scale, gamma, beta, P, C, HW, relu, dz, dshift, dgamma, dbias, tpb);
        FI_HIP_CHECK(hipGetLastError());
        return FI_OK;
    }
    FI_REQUIRE((long)N * HW < 2147483647L, "too many elements per channel");
    // enough workgroups to fill the chip: C * chunks >= ~2048
    int chunks = fi::ceil_div(2048, C);
    if (chunks > N) chunks = N;
    if (chunks < 1) chunks = 1;
    const int ipb = fi::ceil_div(N, chunks);
    chunks = fi::ceil_div(N, ipb);
    fi::ProfScope prof(FI_K_BN_ACT_BWD, st);
    auto k = residual ? (g_out ? bn_act_bwd_kernel<true, true> : bn_act_bwd_kernel<true, false>)
                      : (g_out ? bn_act_bwd_kernel<false, true> : bn_act_bwd_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3(C, chunks), dim3(256), 0, st, dy, y, scale, gamma, beta, residual,
                       N, C, HW, relu, dz, g_out, dshift, dgamma, dbias, ipb);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

// 1: this geometry runs on conv_wgrad_vec_kernel's plain (non-HALF) instantiation -- the one that takes a WgradBatch
static bool wgrad_batchable(const ConvGeom &g, const float *x, const float *dy, int weight_layout)
{
    const bool hwc = (g.Cin % BN == 0) && (weight_layout == 1 || g.R * g.S == 1);
    return hwc && wgrad_same_size(g, x, dy);
}

static int wgrad_impl(const float *x, const float *dy, float *dweight, int N, int Cin, int H,
                      int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h,
                      int pad_w, int weight_layout, float *dbias, int flags, fi_stream_t stream, const WgradBatch *batch);

int fi_conv2d_weight_grad(const float *x, const float *dy, float *dweight, int N, int Cin, int H,
                          int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h,
                          int pad_w, int weight_layout, float *dbias, int flags, fi_stream_t stream)
{
    return wgrad_impl(x, dy, dweight, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w, weight_layout, dbias,
                      flags, stream, nullptr);
}

int fi_conv2d_weight_grad_batch(const float *const *x, const float *const *dy, float *const *dweight,
                                float *const *dbias, int n, int N, int Cin, int H, int W, int Cout, int R, int S,
                                int stride_h, int stride_w, int pad_h, int pad_w, int weight_layout, int flags,
                                fi_stream_t stream)
{
    FI_REQUIRE(n >= 1 && x && dy && dweight, "empty batch / null pointer table");
    ConvGeom g;
    int rc = make_geom(g, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w);
    if (rc != FI_OK) return rc;
    bool ok = true, any_db = false, all_db = true;
    for (int i = 0; i < n; ++i) {
        FI_REQUIRE(x[i] && dy[i] && dweight[i], "null pointer in the batch");
        ok = ok && wgrad_batchable(g, x[i], dy[i], weight_layout);
        const bool has = dbias && dbias[i];
        any_db = any_db || has;
        all_db = all_db && has;
    }
    // one launch only for the row-major kernel, with the outputs pre-zeroed by the caller and uniform bias use;
    // anything else: one launch per problem (same results)
    if (!ok || n == 1 || !(flags & FI_OUTPUTS_ZEROED) || (any_db && !all_db)) {
        for (int i = 0; i < n; ++i) {
            rc = wgrad_impl(x[i], dy[i], dweight[i], N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w,
                            weight_layout, dbias ? dbias[i] : nullptr, flags, stream, nullptr);
            if (rc != FI_OK) return rc;
        }
        return FI_OK;
    }
    for (int i0 = 0; i0 < n; i0 += kWgBatchMax) {
        WgradBatch wb;
        wb.n = n - i0 < kWgBatchMax ? n - i0 : kWgBatchMax;
        for (int i = 0; i < kWgBatchMax; ++i) {
            const int j = i0 + (i < wb.n ? i : 0);
            wb.x[i] = x[j]; wb.dy[i] = dy[j]; wb.dw[i] = dweight[j]; wb.db[i] = all_db ? dbias[j] : nullptr;
        }
        rc = wgrad_impl(wb.x[0], wb.dy[0], wb.dw[0], N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w,
                        weight_layout, wb.db[0], flags, stream, &wb);
        if (rc != FI_OK) return rc;
    }
    return FI_OK;
}

static int wgrad_impl(const float *x, const float *dy, float *dweight, int N, int Cin, int H,
                      int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h,
                      int pad_w, int weight_layout, float *dbias, int flags, fi_stream_t stream, const WgradBatch *batch)
{
    ConvGeom g;
    int rc = make_geom(g, N, Cin, H, W, Cout, R, S, stride_h, stride_w, pad_h, pad_w);
    if (rc != FI_OK) return rc;
    FI_REQUIRE(x && dy && dweight, "null pointer");
    FI_REQUIRE(weight_layout == 0 || weight_layout == 1, "weight_layout: 0 = [Cout][Cin][R][S], 1 = [Cout][R][S][Cin]");
    // tap-major dW: 128 input channels of one tap per column tile, or (Cin == 64, row-major kernel only)
    // the 64 channels of two taps
    const bool half = (Cin == 64) && wgrad_same_size(g, x, dy);
    const bool hwc = ((Cin % BN == 0) || half) && (weight_layout == 1 || R * S == 1);
    FI_REQUIRE(hwc || weight_layout == 0,
               "weight_layout 1 needs Cin % 128 == 0, or Cin == 64 on a same-size stride-1 layer");
    g.zero = zero_page();
    FI_REQUIRE(g.zero != nullptr, "zero page lookup failed (no HIP device?)");
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & FI_OUTPUTS_ZEROED)) {
        FI_HIP_CHECK(hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)Cout * g.K, st));
        if (dbias) FI_HIP_CHECK(hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st));
    }
    // 64-row tiles for narrow layers, and when 128-row tiles x the admissible splits (>= 512 pixels each)
    // would fill less than 3/4 of the resident slots (C4/C5 1x1 layers at batch 4)
    static const int min_pix = getenv("FI_WG_MINPIX") ? atoi(getenv("FI_WG_MINPIX")) : 512;      // tuning knobs (scripts/)
    static const int force_bm = getenv("FI_WG_BM") ? atoi(getenv("FI_WG_BM")) : 0;
    const long max_splits0 = (g.P + min_pix - 1) / min_pix;
    const long tiles128 = (long)fi::ceil_div(g.K, BN) * fi::ceil_div(Cout, 128);
    const long reach128 = (batch ? batch->n : 1) * tiles128 * (1024 / tiles128 < max_splits0 ? (1024 / tiles128 < 1 ? 1 : 1024 / tiles128) : max_splits0);
    const int BMsel = force_bm ? (Cout <= 64 ? 64 : force_bm) : ((Cout <= 64 || reach128 < 768) ? 64 : 128);
    const long tiles = (long)fi::ceil_div(g.K, BN) * fi::ceil_div(Cout, BMsel);
    // Split the pixel range so that tiles x splits fills the 1024 resident workgroup slots (256 CUs x 4)
    // in ONE round: every workgroup has the same amount of work, so 1044 workgroups on 1024 slots take
    // two rounds (the first version rounded UP and paid exactly that).  At least 512 pixels per split.
    // (a batch of nb problems fills the slots together: fewer, longer pixel splits per problem -- the fixed cost of a
    // workgroup, prologue and atomic epilogue, is paid per split)
    const int nb = batch ? batch->n : 1;
    long want = 1024 / (tiles * nb);
    long max_splits = (g.P + min_pix - 1) / min_pix;
    int splits = (int)(want < 1 ? 1 : (want > max_splits ? max_splits : want));
    if (splits < 1) splits = 1;
    if (nb > 1) {
        // a batch rarely fits ONE round exactly (12 layers x 36 tiles x 2 splits = 864 of 1024 slots): take the split
        // count that minimises rounds x (pixels per split + the fixed cost of a workgroup, ~256 pixels' worth)
        long best_cost = -1;
        const long smax = max_splits < 64 ? max_splits : 64;
        for (long sp = 1; sp <= smax; ++sp) {
            const long total = tiles * nb * sp;
            const long rounds = (total + 1023) / 1024;
            const long cost = rounds * ((g.P + sp - 1) / sp + 256);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                splits = (int)sp;
            }
        }
    }
    int pps = fi::ceil_div(g.P, splits);
    pps = fi::ceil_div(pps, BK) * BK;
    splits = fi::ceil_div(g.P, pps);
    fi::ProfScope prof(FI_K_CONV_WGRAD + (BMsel == 64 ? 0 : 4) + window_class(R, S), st);
    if (BMsel == 64)
        launch_wgrad<64>(g, x, dy, dweight, splits, pps, hwc, dbias, st, batch);
    else
        launch_wgrad<128>(g, x, dy, dweight, splits, pps, hwc, dbias, st, batch);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

// c[i] = act(sum_s slab_s[i] * scale[i % N] + bias[i % N]): the ordered reduction of fi_gemm_nt's split-K partial sums
// (scale: the eval-mode BatchNorm behind a fully connected layer, fi_gemm_nt_affine)
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float *__restrict__ ws, int splits, long slab,
                                                               const float *__restrict__ bias, int N, int relu,
                                                               float *__restrict__ c, long total4,
                                                               const int *__restrict__ n_live, int bm,
                                                               const float *__restrict__ scale)
{
    // rows past the live count (rounded up to the tile height): their tiles were never computed -- zeros, not the slabs
    const long live4 = n_live ? (long)((*n_live + bm - 1) / bm) * bm * N / 4 : total4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        if (i >= live4) {
            *reinterpret_cast<float4 *>(c + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float4 a = *reinterpret_cast<const float4 *>(ws + 4 * i);
        for (int sidx = 1; sidx < splits; ++sidx) {
            const float4 v = *reinterpret_cast<const float4 *>(ws + (size_t)sidx * slab + 4 * i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (scale) {
            const float4 m = *reinterpret_cast<const float4 *>(scale + (4 * i) % N);
            a.x *= m.x; a.y *= m.y; a.z *= m.z; a.w *= m.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + (4 * i) % N);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (relu) {
            a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        }
        *reinterpret_cast<float4 *>(c + 4 * i) = a;
    }
}

static void gemm_nt_plan(int M, int N, int K, int *bm, int *splits, int *pps)
{
    // the tile / split choice of fi_conv2d_weight_grad with Cout = M, Cin = N, pixels = K
    const long max_splits0 = ((long)K + 511) / 512;
    const long tiles128 = (long)fi::ceil_div(N, BN) * fi::ceil_div(M, 128);
    const long per = 1024 / tiles128 < 1 ? 1 : 1024 / tiles128;
    const long reach128 = tiles128 * (per < max_splits0 ? per : max_splits0);
    *bm = (M <= 64 || reach128 < 768) ? 64 : 128;
    const long tiles = (long)fi::ceil_div(N, BN) * fi::ceil_div(M, *bm);
    static const long target_wg = getenv("FI_GEMM_TARGET") ? atol(getenv("FI_GEMM_TARGET")) : 1024;      // (tuning knob)
    long want = target_wg / tiles;
    long sp = want < 1 ? 1 : (want > max_splits0 ? max_splits0 : want);
    int p = fi::ceil_div(K, (int)sp);
    p = fi::ceil_div(p, BK) * BK;
    *pps = p;
    *splits = fi::ceil_div(K, p);
}

size_t fi_gemm_nt_workspace_bytes(int M, int N, int K)
{
    if (M < 1 || N < 1 || K < 1) return 0;
    int bm, splits, pps;
    gemm_nt_plan(M, N, K, &bm, &splits, &pps);
    return sizeof(float) * (size_t)splits * (size_t)M * (size_t)N;
}

int fi_gemm_nt(const float *a, const float *b, const float *bias, float *c, int M, int N, int K, int relu,
               float *workspace, fi_stream_t stream)
{
    return fi_gemm_nt_rows(a, b, bias, c, M, N, K, relu, workspace, nullptr, stream);
}

int fi_gemm_nt_rows(const float *a, const float *b, const float *bias, float *c, int M, int N, int K, int relu,
                    float *workspace, const int32_t *m_live_dev, fi_stream_t stream)
{
    return fi_gemm_nt_affine(a, b, nullptr, bias, c, M, N, K, relu, workspace, m_live_dev, stream);
}

int fi_gemm_nt_affine(const float *a, const float *b, const float *scale, const float *bias, float *c, int M, int N, int K,
                      int relu, float *workspace, const int32_t *m_live_dev, fi_stream_t stream)
{
    FI_REQUIRE(a && b && c && workspace, "null pointer");
    FI_REQUIRE(((uintptr_t)scale & 15) == 0, "fi_gemm_nt needs 16-byte aligned operands");
    FI_REQUIRE(M >= 1 && N >= 1 && K >= 4, "sizes must be positive");
    FI_REQUIRE(N % BN == 0 && K % 4 == 0, "fi_gemm_nt needs N % 128 == 0 and K % 4 == 0");
    FI_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)workspace | (uintptr_t)bias) & 15) == 0,
               "fi_gemm_nt needs 16-byte aligned operands");
    FI_REQUIRE((long)M * K * 4 < 0x7fffff00L && (long)N * K * 4 < 0x7fffff00L, "operand larger than 2 GB");
    ConvGeom g;
    int rc = make_geom(g, 1, N, 1, K, M, 1, 1, 1, 1, 0, 0);
    if (rc != FI_OK) return rc;
    g.zero = zero_page();
    FI_REQUIRE(g.zero != nullptr, "zero page lookup failed (no HIP device?)");
    FI_REQUIRE(wgrad_same_size(g, b, a), "operands do not meet the row-major kernel's alignment rules");
    int bm, splits, pps;
    gemm_nt_plan(M, N, K, &bm, &splits, &pps);
    g.dw_slab = (long)M * N;
    g.n_live = m_live_dev;
    hipStream_t st = (hipStream_t)stream;
    {
        fi::ProfScope prof(FI_K_CONV_WGRAD + (bm == 64 ? 0 : 4) + window_class(1, 1), st);
        if (bm == 64)
            launch_wgrad<64>(g, b, a, workspace, splits, pps, true, nullptr, st);
        else
            launch_wgrad<128>(g, b, a, workspace, splits, pps, true, nullptr, st);
        FI_HIP_CHECK(hipGetLastError());
    }
    const long total4 = (long)M * N / 4;
    const long blocks = (total4 + 255) / 256;
    fi::ProfScope prof2(FI_K_GEMM_REDUCE, st);
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, workspace,
                       splits, g.dw_slab, bias, N, relu, c, total4, m_live_dev, bm, scale);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_weight_transpose_batch(const FiTransposeDesc *descs_dev, int n, long total_tiles, fi_stream_t stream)
{
    FI_REQUIRE(n >= 0 && total_tiles >= 0, "bad sizes");
    if (n == 0 || total_tiles == 0) return FI_OK;
    FI_REQUIRE(descs_dev != nullptr, "null descriptor table");
    const long grid = total_tiles < 65536 ? total_tiles : 65536;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, descs_dev, n,
                       total_tiles);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
