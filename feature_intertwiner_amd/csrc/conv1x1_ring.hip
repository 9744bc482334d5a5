// conv1x1_ring.hip -- the persistent form of the 1x1 / stride-1 convolution (forward and data gradient of the ResNet
// bottleneck's conv1 / conv3, lib/sub_module.py:90-128 of the reference) for gfx950.  Reached from fi_conv2d_forward(_gated)
// with weight_layout 3 (csrc/conv_igemm.hip); its own translation unit so that tests/test_isa_audit.py can compile and
// audit it in seconds.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "fi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -------------------------------------------------------------------------------------
// 1x1 / stride 1, persistent form (round 6): conv1x1_ring_kernel
// -------------------------------------------------------------------------------------
// Same tile as conv1x1_reg_kernel (128 pixels x 128 output channels, 4 wavefronts x 32 channels, weights in registers),
// restructured around what scripts/micro/conv1x1_ring.hip measured on the C3/C4 bottleneck layers (profiles/r06_ring_*):
//   * PERSISTENT workgroups (<= 2 per CU) walk several tiles: 1024 tiles on 768 resident slots were 1.33 rounds, and every
//     workgroup paid its own start-up (first tile's memory latency);
//   * the pixel tile travels through a 3-deep LDS ring by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
//     ds_write pass), the ring runs across tile boundaries (the next tile's first stages are in flight under the current
//     tile's last ones), ONE barrier per 32-channel stage, placed mid-stage so that no LDS read waits behind it;
//   * weights come FRAGMENT-MAJOR (weight_layout 3, fi_weight_transpose_batch flag 1): per block of 32 rows x 16 channels
//     [2 halves][64 lanes][4 floats], so a wavefront's A-operand load is 1 KB contiguous -- the row-major form is 32 rows x
//     64 bytes per instruction (32 cache lines, 12 % of the kernel: r06_ring_v1_exp.txt, "no weight loads");
//   * tile t's epilogue (scale, bias, residual, ReLU, gate, store) is executed by the same wavefronts under tile t+1's
//     first four stages, one group of 4 channel rows per stage, from a copy of the sums.
// The LDS-DMA pieces and the weight loads are inline asm: hipcc treats the LDS-DMA builtin as a store that may alias every
// later ds_read and drains it (vmcnt(0)) at the top of every stage.  What that costs in care, all measured here:
//   - the hazard recogniser does not look into asm: VALU-written SGPRs (v_readlane of a spilled SGPR) need 5 wait states
//     before a VMEM instruction reads them as its scalar base -> s_nop 4 in front of every asm VMEM instruction;
//   - an asm load with a register OUTPUT may be copied by the register allocator before it has landed -> only the weight
//     registers are asm outputs (tied into the wait that covers them); the epilogue's loads and stores are ordinary C++;
//   - loads and stores do NOT retire in order relative to each other on gfx950: a counted wait must never leave a store in
//     flight in front of a needed load.  Outstanding stores only make a counted wait stricter.
// VMEM issue order per stage:  s = 0: wait vmcnt(4) [weights A0 + epilogue loads landed, 4 DMA pieces in flight]; A1 x2
//                              s = 8: wait vmcnt(0); s_barrier; epilogue group: compute + 4 stores; next group's loads;
//                                     A0(next) x2; DMA(stage + 2) x4
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ring_dma16(const float *g, unsigned lds_byte_addr)
{
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_byte_addr)
                 : "memory", "m0");
}
__device__ __forceinline__ void ring_load_a(f32x4_t &lo, f32x4_t &hi, unsigned off, const float *base)   // halves 1 KB apart
{
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(lo), "=&v"(hi) : "v"(off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait(f32x4_t &a, f32x4_t &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ void ring_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct RingEGroup {
    f32x4_t res[4], gate[4];
    float sc[4], bi[4];
};

constexpr int RING_CB = 32;                 // channels per stage
constexpr int RING_LDS_BYTES = 3 * RING_CB * 128 * 4;
constexpr int kThreads = 256;

struct RingEpilogue {
    const float *bias, *scale, *residual, *gate;
    int relu;
};
struct RingGeom {
    int N, Cin, HW, Cout, ptiles, mtiles;
    const float *zero;
};

template <bool HAS_RES, bool HAS_GATE>
__global__ __launch_bounds__(kThreads, 2) void conv1x1_ring_kernel(const float *__restrict__ x, const float *__restrict__ wF,
                                                                  RingEpilogue ep, float *__restrict__ y, RingGeom g, int nwg)
{
    constexpr int CB = RING_CB, GI = 4;
    extern __shared__ __attribute__((aligned(16))) float Pr[];      // [3][32][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_xcd = (g.ptiles + 7) >> 3;
    const int tiles_xcd = per_xcd * g.mtiles;
    const int stride = nwg >> 3;
    const int P = g.N * g.HW;
    const int ncb = g.Cin / CB;                        // >= 4
    const size_t HW = (size_t)g.HW;
    const int kgroups = g.Cin / 16;
    const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr;
    const float *__restrict__ spp = has_sc ? ep.scale : g.zero;
    const float *__restrict__ bpp = has_bi ? ep.bias : g.zero;
    const float relu_lo = (ep.relu & 1) ? 0.0f : -INFINITY;
    const unsigned sc_off = has_sc ? khalf * 16u : 0u, bi_off = has_bi ? khalf * 16u : 0u;   // lane's first row: + 4 * khalf
    const unsigned a_off = lane * 16u;
    const unsigned row_bytes = (unsigned)(HW * 4);

    auto tile_valid = [&](int li) { return li < tiles_xcd && xcd * per_xcd + li / g.mtiles < g.ptiles; };
    if (!tile_valid(local)) return;
    // a lane past the pixel range works on the LAST pixel group: it recomputes and re-stores that group's values bit for
    // bit (no predicate anywhere, and every wavefront issues the same memory instructions)
    auto lane_pixel = [&](int li, int &n, int &pix) {
        const int pt = xcd * per_xcd + li / g.mtiles;
        const int spc = min(pt * 128 + 4 * l31, P - 4);
        n = spc / g.HW;
        pix = spc - n * g.HW;
    };
    auto dma_source = [&](int li) {
        int n, pix;
        lane_pixel(li, n, pix);
        return x + ((size_t)n * g.Cin + wave * 8 + khalf) * HW + pix;
    };
    auto out_offset = [&](int li) {                    // byte offset of (lane's first row, first pixel) in y / residual / gate
        int n, pix;
        lane_pixel(li, n, pix);
        const int m0w = (li % g.mtiles) * 128 + wave * 32;
        return (unsigned)((((size_t)n * g.Cout + m0w + 4 * khalf) * HW + pix) * 4);
    };
    auto a_tile = [&](int li) {                        // uniform: this wavefront's 32 rows, channel group 0
        return wF + ((size_t)((li % g.mtiles) * 4 + wave) * kgroups) * 512;
    };

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float *)(Pr + wave * 8 * 128));
    const size_t dma_row2 = 2 * HW, dma_stage = (size_t)CB * HW;
    const float *ld_ptr = dma_source(local);           // load cursor: two stages ahead of the compute cursor, across tiles
    int ld_cb = 0, ld_li = local;
    auto issue_dma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GI; ++i) ring_dma16(ld_ptr + i * dma_row2, lds0 + (unsigned)(buf * CB * 128 + i * 256) * 4u);
        if (++ld_cb == ncb) {
            ld_cb = 0;
            if (tile_valid(ld_li + stride)) {
                ld_li += stride;
                ld_ptr = dma_source(ld_li);
            } else {
                ld_ptr -= (size_t)(ncb - 1) * dma_stage;       // past the last tile: re-read it (nobody consumes those stages)
            }
        } else {
            ld_ptr += dma_stage;
        }
    };

    f32x16 acc[4], out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[j][e] = 0.0f; out[j][e] = 0.0f; }

    int li = local;
    bool pend = false;
    unsigned cur_off = out_offset(li), prev_off = 0;
    int prev_li = li;
    RingEGroup eg;
#pragma unroll
    for (int i = 0; i < 4; ++i) { eg.sc[i] = 0.f; eg.bi[i] = 0.f; eg.res[i] = f32x4_t{0, 0, 0, 0}; eg.gate[i] = f32x4_t{0, 0, 0, 0}; }

    // group q = channel rows 8q .. 8q+3 of the lane's 16.  Ordinary loads / stores between asm statements with memory
    // clobbers: issued where they are written, waited for by the compiler (over-waits only: vmcnt counts in issue order).
    auto eload = [&](int tli, unsigned off, int q) {
        const int m0w = (tli % g.mtiles) * 128 + wave * 32;
        const unsigned o = off + (unsigned)(8 * q) * row_bytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            eg.sc[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(spp + (has_sc ? m0w + 8 * q + i : 0)) + sc_off);
            eg.bi[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(bpp + (has_bi ? m0w + 8 * q + i : 0)) + bi_off);
            if (HAS_RES)
                eg.res[i] = *reinterpret_cast<const f32x4_t *>(reinterpret_cast<const char *>(ep.residual) + (size_t)(o + i * row_bytes));
            if (HAS_GATE)
                eg.gate[i] = *reinterpret_cast<const f32x4_t *>(reinterpret_cast<const char *>(ep.gate) + (size_t)(o + i * row_bytes));
        }
    };
    auto econsume = [&](const f32x16 (&o)[4], unsigned off, auto QT) {
        constexpr int q = decltype(QT)::value;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            f32x4_t t;
            // absent scale / bias / ReLU as exact identities: * 1, + (-0), max(., -inf)
            const float sc = has_sc ? eg.sc[e4] : 1.0f, bi = has_bi ? eg.bi[e4] : -0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = o[j][4 * q + e4];
                v = v * sc;
                v = v + bi;
                if (HAS_RES) v += eg.res[e4][j];
                v = fmaxf(v, relu_lo);
                if (HAS_GATE) v = eg.gate[e4][j] > 0.0f ? v : 0.0f;
                t[j] = v;
            }
            *reinterpret_cast<f32x4_t *>(reinterpret_cast<char *>(y) + (size_t)(off + (unsigned)(8 * q + e4) * row_bytes)) = t;
        }
    };

    // ---- prologue of the first tile
    f32x4_t a0lo, a0hi, a1lo, a1hi;
    float4 breg[2];
    const float *a_cur = a_tile(li);
    ring_load_a(a0lo, a0hi, a_off, a_cur);
    issue_dma(0);
    issue_dma(1);
    a1lo = a0lo; a1hi = a0hi;
    ring_wait<4>(a0lo, a0hi);
    ring_barrier();
    int rbuf = 0;
    breg[0] = *reinterpret_cast<const float4 *>(Pr + (khalf * 8) * 128 + 4 * l31);

    for (;;) {
        const bool has_next = tile_valid(li + stride);
        const float *a_next = has_next ? a_tile(li + stride) : a_cur;
        auto stage = [&](auto JT, int cb) {
            constexpr int J = decltype(JT)::value;        // epilogue group of the previous tile handled in this stage (-1: none)
            const float *__restrict__ pbuf = Pr + rbuf * (CB * 128) + (khalf * 8) * 128 + 4 * l31;
            const int nbuf = rbuf == 2 ? 0 : rbuf + 1;
            const int wbuf = nbuf == 2 ? 0 : nbuf + 1;
            const float *__restrict__ pnext = Pr + nbuf * (CB * 128) + (khalf * 8) * 128 + 4 * l31;
            const bool last = cb + 1 == ncb;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int h = s / 8, kk = s % 8;
                if (s == 0) {
                    // A0 and, older, the epilogue loads landed; the 4 DMA pieces (the youngest loads) stay in flight.  The
                    // count does not depend on how many epilogue loads the compiler emitted (it merges loads of one address)
                    ring_wait<4>(a0lo, a0hi);
                    ring_load_a(a1lo, a1hi, a_off, a_cur + (size_t)(2 * cb + 1) * 512);
                }
                if (s == 8) {
                    // A1 (s = 0) and, older, DMA(cb+1) landed; the previous stage's stores are 16 sub-steps old
                    ring_wait<0>(a1lo, a1hi);
                    ring_barrier();                       // stage cb+1 visible to all; everyone is past stage cb-1
                    if (J >= 0) {
                        if (pend) econsume(out, prev_off, fi::Int<(J >= 0 ? J : 0)>{});
                    }
                    if (J >= 0 && J < 3) {
                        if (pend) eload(prev_li, prev_off, J + 1);
                    }
                    if (last) eload(li, cur_off, 0);      // (ncb >= 4: the last stage is never one of J = 0..2)
                    ring_load_a(a0lo, a0hi, a_off, last ? a_next : a_cur + (size_t)(2 * cb + 2) * 512);
                    issue_dma(wbuf);
                }
                if (s == 15) {
                    if (!last || has_next) breg[(s + 1) & 1] = *reinterpret_cast<const float4 *>(pnext);
                } else {
                    const int h2 = (s + 1) / 8, k2 = (s + 1) % 8;
                    breg[(s + 1) & 1] = *reinterpret_cast<const float4 *>(pbuf + (h2 * 16 + k2) * 128);
                }
                const f32x4_t alo = h == 0 ? a0lo : a1lo, ahi = h == 0 ? a0hi : a1hi;
                const float av = kk < 4 ? alo[kk] : ahi[kk - 4];
                const float4 bv = breg[s & 1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[3], 0, 0, 0);
            }
            rbuf = nbuf;
        };
        stage(fi::Int<0>{}, 0);
        stage(fi::Int<1>{}, 1);
        stage(fi::Int<2>{}, 2);
        stage(fi::Int<3>{}, 3);
        for (int cb = 4; cb < ncb; ++cb) stage(fi::Int<-1>{}, cb);
        // ---- tile done: hand the sums to the deferred epilogue (group 0's loads are in flight)
        if (has_next) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                out[j] = acc[j];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
            }
            prev_off = cur_off; prev_li = li;
            pend = true;
            li += stride;
            cur_off = out_offset(li);
            a_cur = a_next;
            continue;
        }
        // ---- last tile of this workgroup: epilogue now.  Group 0 was loaded under the last stage.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (HAS_GATE) {                 // (residual + gate: 40 registers per group -- one group at a time)
            econsume(acc, cur_off, fi::Int<0>{});
            eload(li, cur_off, 1);
            econsume(acc, cur_off, fi::Int<1>{});
            eload(li, cur_off, 2);
            econsume(acc, cur_off, fi::Int<2>{});
            eload(li, cur_off, 3);
            econsume(acc, cur_off, fi::Int<3>{});
        } else {
            // two groups in flight: the load of group q+1 is issued before group q is consumed
            RingEGroup ea = eg, eb;
            eload(li, cur_off, 1); eb = eg;
            eg = ea; econsume(acc, cur_off, fi::Int<0>{});
            eload(li, cur_off, 2); ea = eg;
            eg = eb; econsume(acc, cur_off, fi::Int<1>{});
            eload(li, cur_off, 3); eb = eg;
            eg = ea; econsume(acc, cur_off, fi::Int<2>{});
            eg = eb; econsume(acc, cur_off, fi::Int<3>{});
        }
        break;
    }
}


}  // namespace

namespace fi {

int launch_conv1x1_ring(const RingArgs &a, hipStream_t st)
{
    RingGeom g;
    g.N = a.N; g.Cin = a.Cin; g.HW = a.HW; g.Cout = a.Cout;
    g.ptiles = ceil_div(a.N * a.HW, 128);
    g.mtiles = a.Cout / 128;
    g.zero = a.zero;
    const RingEpilogue ep = {a.bias, a.scale, a.residual, a.gate, a.relu};
    const long tiles8 = (long)ceil_div(g.ptiles, 8) * 8 * g.mtiles;
    const int nwg = (int)(tiles8 < 512 ? tiles8 : 512);       // <= 2 workgroups per CU, a multiple of 8
    if (a.gate && a.residual)
        hipLaunchKernelGGL((conv1x1_ring_kernel<true, true>), dim3(nwg), dim3(kThreads), RING_LDS_BYTES, st, a.x, a.wF, ep, a.y, g, nwg);
    else if (a.gate)
        hipLaunchKernelGGL((conv1x1_ring_kernel<false, true>), dim3(nwg), dim3(kThreads), RING_LDS_BYTES, st, a.x, a.wF, ep, a.y, g, nwg);
    else if (a.residual)
        hipLaunchKernelGGL((conv1x1_ring_kernel<true, false>), dim3(nwg), dim3(kThreads), RING_LDS_BYTES, st, a.x, a.wF, ep, a.y, g, nwg);
    else
        hipLaunchKernelGGL((conv1x1_ring_kernel<false, false>), dim3(nwg), dim3(kThreads), RING_LDS_BYTES, st, a.x, a.wF, ep, a.y, g, nwg);
    return FI_OK;
}

}  // namespace fi

extern "C" {

int fi_conv1x1_ring_eligible(int N, int Cin, int H, int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h,
                             int pad_w, int output_layout, const float *x, const float *y, const float *residual,
                             const float *gate)
{
    const long P = (long)N * H * W;
    if (!(R == 1 && S == 1 && stride_h == 1 && stride_w == 1 && pad_h == 0 && pad_w == 0 && output_layout == 0)) return 0;
    if (Cin < 128 || Cin % RING_CB != 0 || Cout % 128 != 0 || (H * W) % 4 != 0 || P < 128) return 0;
    if ((uintptr_t)x % 16 || (uintptr_t)y % 16 || (uintptr_t)residual % 16 || (uintptr_t)gate % 16) return 0;
    if (P * Cout * 4 >= 4294967296L || (long)N * Cin * H * W >= 2147483647L) return 0;      // 32-bit byte offsets into y
    if ((long)fi::ceil_div((int)P, 128) * (Cout / 128) < 256) return 0;                      // fewer tiles than CUs
    static const bool off = getenv("FI_NO_RING1X1") != nullptr;
    return off ? 0 : 1;
}

}  // extern "C"
