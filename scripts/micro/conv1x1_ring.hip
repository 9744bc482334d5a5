// Development harness for the short-K 1x1 convolution kernel (C3/C4 bottleneck layers of ResNet-101-FPN,
// lib/sub_module.py:90-128 of the reference): variants of the pixel-tile pipeline against the library's
// conv1x1_reg_kernel (through fi_conv2d_forward_gated), results compared element by element.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include scripts/micro/conv1x1_ring.hip \
//         -L feature_intertwiner_amd -lfi_hip -o scripts/micro/bin/conv1x1_ring
//   LD_LIBRARY_PATH=feature_intertwiner_amd scripts/micro/bin/conv1x1_ring
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "fi_capi.h"

#define CK(e)                                                                          \
    do {                                                                               \
        hipError_t _e = (e);                                                           \
        if (_e != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Epi {
    const float *bias, *scale, *residual, *gate;
    int relu;
};
struct Geom {
    int N, Cin, HW, Cout, ptiles, mtiles;
    const float *zero;
    int no_epilogue;
    float *sink;     // 1 KB that lanes outside the pixel range store to (unconditional stores: exact vmcnt bookkeeping)
};

__device__ __attribute__((aligned(16))) float d_zero_page[64];


typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA and weight loads as inline asm: hipcc treats the builtin LDS-DMA as a store that may alias every later
// ds_read and drains it (s_waitcnt vmcnt(0)) at the top of every stage; with asm the waits are the counted ones below.
// (vmcnt counts in issue order; a compiler-inserted vmcnt(N) for its own loads can only over-wait.)
__device__ __forceinline__ void glds16(const float *g, unsigned lds_byte_addr)
{
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_byte_addr) : "memory", "m0");
}
__device__ __forceinline__ void gload16x2(f32x4 &lo, f32x4 &hi, const float *p)
{
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm(f32x4 &a, f32x4 &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_only()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier()
{
    asm volatile("s_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Ring kernel: a workgroup owns 128 pixels x 128 output channels; the pixel tile travels through a 3-deep LDS ring in
// stages of CB channels by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), weights go from memory
// straight into the MFMA A operand.  One barrier per stage, placed one sub-step before the end of the stage: loads have a
// full stage of slack and no LDS read ever waits behind the barrier.
// ---------------------------------------------------------------------------------------------------------------------
// EXP (timing experiments, results wrong): 1 = pixel tile always from the same (cached) lines, 2 = no weight loads,
// 3 = no barrier, 4 = no LDS reads, 5 = MFMAs only
template <int CB, bool PERSIST, int EXP = 0>
__global__ __launch_bounds__(256, 3) void ring_kernel(const float *__restrict__ x, const float *__restrict__ w, Epi ep,
                                                      float *__restrict__ y, Geom g, int nwg)
{
    constexpr int NS = CB / 2;                         // sub-steps (channel pairs) per stage
    constexpr int GI = CB / 8;                         // LDS-DMA instructions per wave and stage (2 rows of 128 px each)
    extern __shared__ __attribute__((aligned(16))) float Ps[];      // [3][CB][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_xcd = (g.ptiles + 7) >> 3;
    const int tiles_xcd = per_xcd * g.mtiles;
    const int stride = PERSIST ? (nwg >> 3) : tiles_xcd;
    const int P = g.N * g.HW;
    const int ncb = g.Cin / CB;
    const size_t HW = (size_t)g.HW;

    for (int li = local; li < tiles_xcd; li += stride) {
        const int mt = li % g.mtiles;
        const int pt = xcd * per_xcd + li / g.mtiles;
        if (pt >= g.ptiles) break;
        const int m0 = mt * 128, P0 = pt * 128;
        // ---- A operand: 8 consecutive input channels of one output channel per lane
        const int am = min(m0 + wave * 32 + l31, g.Cout - 1);
        const float *__restrict__ a_base = w + (size_t)am * g.Cin + khalf * 8;
        // ---- LDS-DMA source: lane = (row of the pair, pixel group)
        const int sp = P0 + 4 * l31;
        const bool s_ok = sp < P;
        const int s_n = s_ok ? sp / g.HW : 0;
        const float *__restrict__ s_src = s_ok ? x + ((size_t)s_n * g.Cin + wave * (CB / 4) + khalf) * HW + (sp - s_n * g.HW)
                                                : g.zero;
        const size_t s_row2 = s_ok ? 2 * HW : 0;
        const size_t s_stage = (s_ok && EXP != 1) ? (size_t)CB * HW : 0;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) float *)(Ps + wave * (CB / 4) * 128));
        auto issue_stage = [&](int cb, int buf) {
            const float *__restrict__ q = s_src + (size_t)cb * s_stage;
#pragma unroll
            for (int i = 0; i < GI; ++i)
                if (EXP != 5) glds16(q + i * s_row2, lds0 + (unsigned)(buf * CB * 128 + i * 256) * 4u);
        };

        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

        // (a previous tile's LDS reads are all behind a barrier: its last stage ended with one)
        f32x4 a0lo, a0hi, a1lo, a1hi;      // two 16-channel groups of weights: [khalf*8 .. +8) of the group, per lane
        float4 breg[2];
        gload16x2(a0lo, a0hi, a_base);
        if (EXP == 2 || EXP == 5) gload16x2(a1lo, a1hi, a_base + 16);
        issue_stage(0, 0);
        if (ncb > 1) issue_stage(1, 1); else issue_stage(0, 1);
        wait_vm_only<GI>();                  // weights + stage 0 landed (stage 1 may be in flight)
        wg_barrier();
        breg[0] = *reinterpret_cast<const float4 *>(Ps + (khalf * 8) * 128 + 4 * l31);

        int rbuf = 0;
        for (int cb = 0; cb < ncb; ++cb) {
            const float *__restrict__ pbuf = Ps + rbuf * (CB * 128) + (khalf * 8) * 128 + 4 * l31;
            const int nbuf = rbuf == 2 ? 0 : rbuf + 1;
            const int wbuf = nbuf == 2 ? 0 : nbuf + 1;
            const float *__restrict__ pnext = Ps + nbuf * (CB * 128) + (khalf * 8) * 128 + 4 * l31;
            const bool more = cb + 1 < ncb;
            static_assert(CB == 32, "this variant: 32 channels per stage");
            // VMEM issue order per stage: [s=0] A1(cb) x2 ... [s=8] barrier, A0(cb+1) x2, DMA(cb+2) x GI
            // waits:  s=8 needs A1(cb): the youngest -> vmcnt(0), which also covers DMA(cb+1) (issued a stage ago), then barrier
            //         s=0 of the next stage needs A0(cb+1): issued BEFORE DMA(cb+2) -> vmcnt(GI) leaves the DMA in flight
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int h = s / 8, kk = s % 8;
                if (s == 0) {
                    if (cb > 0 && EXP != 5) wait_vm<GI>(a0lo, a0hi);
                    if (EXP != 2 && EXP != 5) gload16x2(a1lo, a1hi, a_base + (2 * cb + 1) * 16);
                }
                if (s == 8) {
                    if (EXP != 5) wait_vm<0>(a1lo, a1hi);
                    if (EXP != 3 && EXP != 5) wg_barrier();             // stage cb+1 visible; everyone is past stage cb-1
                    if (EXP != 2 && EXP != 5) gload16x2(a0lo, a0hi, a_base + (more ? 2 * cb + 2 : 2 * cb) * 16);
                    issue_stage(cb + 2 < ncb ? cb + 2 : cb, wbuf);       // (past the end: a dummy DMA into the free buffer keeps
                }                                                        //  the vmcnt bookkeeping uniform; nobody reads it)
                if (EXP == 4 || EXP == 5) {
                    breg[(s + 1) & 1] = breg[s & 1];
                } else if (s == NS - 1) {
                    if (more) breg[(s + 1) & 1] = *reinterpret_cast<const float4 *>(pnext);
                } else {
                    const int h2 = (s + 1) / 8, k2 = (s + 1) % 8;
                    breg[(s + 1) & 1] = *reinterpret_cast<const float4 *>(pbuf + (h2 * 16 + k2) * 128);
                }
                const f32x4 alo = h == 0 ? a0lo : a1lo, ahi = h == 0 ? a0hi : a1hi;
                const float av = kk < 4 ? alo[kk] : ahi[kk - 4];
                const float4 bv = breg[s & 1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[3], 0, 0, 0);
            }
            rbuf = nbuf;
        }
        wait_vm_only<0>();                                               // the dummy DMA / last weight loads
        // the last stage's LDS reads of every wave must be done before the next tile's first DMA lands in the ring
        if (PERSIST) __builtin_amdgcn_s_barrier();

        // ---- epilogue (as the library's patch_epilogue_vec)
        if (g.no_epilogue) {
            if (acc[0][0] == 1.2345e-30f) y[0] = acc[1][1] + acc[2][2] + acc[3][3];
            continue;
        }
        const int po = P0 + 4 * l31;
        if (po >= P) continue;
        const int n_img = po / g.HW;
        const int mb = m0 + wave * 32 + 4 * khalf;
        const size_t obase = ((size_t)n_img * g.Cout + mb) * HW + (po - n_img * g.HW);
        const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr, relu = ep.relu != 0;
        const float *__restrict__ spp = has_sc ? ep.scale : g.zero;
        const float *__restrict__ bpp = has_bi ? ep.bias : g.zero;
        const int smul = has_sc ? 1 : 0, bmul = has_bi ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sc[4], bi[4];
            float4 rr[4], gt[4];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int m = mb + 8 * q + e4;
                sc[e4] = spp[m * smul];
                bi[e4] = bpp[m * bmul];
                if (ep.residual) rr[e4] = *reinterpret_cast<const float4 *>(ep.residual + obase + (size_t)(8 * q + e4) * HW);
                if (ep.gate) gt[e4] = *reinterpret_cast<const float4 *>(ep.gate + obase + (size_t)(8 * q + e4) * HW);
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
                if (ep.residual) {
                    t[0] += rr[e4].x; t[1] += rr[e4].y; t[2] += rr[e4].z; t[3] += rr[e4].w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = relu ? fmaxf(t[j], 0.0f) : t[j];
                if (ep.gate) {
                    t[0] = gt[e4].x > 0.0f ? t[0] : 0.0f; t[1] = gt[e4].y > 0.0f ? t[1] : 0.0f;
                    t[2] = gt[e4].z > 0.0f ? t[2] : 0.0f; t[3] = gt[e4].w > 0.0f ? t[3] : 0.0f;
                }
                *reinterpret_cast<float4 *>(y + obase + (size_t)(8 * q + e4) * HW) = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// V2: persistent workgroups, fragment-major weights (coalesced A loads), ring running across tile boundaries, and tile t's
// epilogue executed by the same wavefronts under tile t+1's first four stages (one group of 4 channel rows per stage).
//   wF: weights re-laid-out per 32-row x 16-channel block: [Cout/32][Cin/16][2 halves][64 lanes][4 floats] -- a wave's
//       A-fragment load is 1 KB contiguous.
// VMEM issue order per stage (all asm, so that the only waits in the loop are the two written here):
//   s = 0 : wait vmcnt(4) [A0 + epilogue loads landed, the 4 DMA pieces stay in flight]; A1 x2; epilogue group: compute + 4 stores
//   s = 8 : wait vmcnt(0); s_barrier; epilogue loads of the next group; A0(next) x2; DMA(stage + 2) x4
// Addresses: every load / store is (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): the per-lane state of a tile
// is ONE offset register; only the DMA source is a per-lane 64-bit pointer (the input may exceed 4 GB).
// ---------------------------------------------------------------------------------------------------------------------
struct EGroup {
    f32x4 res[4], gate[4];
    float sc[4], bi[4];
};

__device__ __forceinline__ void sload4(f32x4 &d, unsigned off, const float *base)
{
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void sload1(float &d, unsigned off, const float *base)
{
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void sstore4(unsigned off, const f32x4 &v, float *base)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void sload4x2(f32x4 &lo, f32x4 &hi, unsigned off, const float *base)     // halves 1 KB apart
{
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(lo), "=&v"(hi) : "v"(off), "s"(base) : "memory");
}

template <int N, bool HAS_RES, bool HAS_GATE>
__device__ __forceinline__ void wait_vm_e(f32x4 &a, f32x4 &b, EGroup &e)
{
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(a), "+v"(b), "+v"(e.sc[0]), "+v"(e.sc[1]), "+v"(e.sc[2]), "+v"(e.sc[3]), "+v"(e.bi[0]), "+v"(e.bi[1]),
                   "+v"(e.bi[2]), "+v"(e.bi[3])
                 : "n"(N) : "memory");
    if (HAS_RES) asm volatile("" : "+v"(e.res[0]), "+v"(e.res[1]), "+v"(e.res[2]), "+v"(e.res[3]) :: "memory");
    if (HAS_GATE) asm volatile("" : "+v"(e.gate[0]), "+v"(e.gate[1]), "+v"(e.gate[2]), "+v"(e.gate[3]) :: "memory");
}

template <bool HAS_RES, bool HAS_GATE, int DBG = 0>
__global__ __launch_bounds__(256, 2) void ring2_kernel(const float *__restrict__ x, const float *__restrict__ wF, Epi ep,
                                                       float *__restrict__ y, Geom g, int nwg)
{
    constexpr int CB = 32, GI = 4;
    extern __shared__ __attribute__((aligned(16))) float Ps[];      // [3][32][128]
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
    const bool has_sc = ep.scale != nullptr, has_bi = ep.bias != nullptr, relu = ep.relu != 0;
    const float *__restrict__ spp = has_sc ? ep.scale : g.zero;
    const float *__restrict__ bpp = has_bi ? ep.bias : g.zero;
    const float relu_lo = relu ? 0.0f : -INFINITY;
    const unsigned sc_off = has_sc ? khalf * 16u : 0u, bi_off = has_bi ? khalf * 16u : 0u;   // lane's first row: + 4 * khalf
    const unsigned a_off = lane * 16u;

    auto tile_valid = [&](int li) { return li < tiles_xcd && xcd * per_xcd + li / g.mtiles < g.ptiles; };
    if (!tile_valid(local)) return;
    // experiment: start the second workgroup of every CU a fraction of a stage late (g.no_epilogue = number of 64-clock sleeps)
    if ((int)blockIdx.x >= (nwg >> 1))
        for (int i = 0; i < g.no_epilogue; ++i) __builtin_amdgcn_s_sleep(32);
    // per-lane pixel position of tile li: image, pixel, validity (clamped to the last pixel group: its results are not stored)
    auto lane_pixel = [&](int li, int &n, int &pix) {
        const int pt = xcd * per_xcd + li / g.mtiles;
        const int sp = pt * 128 + 4 * l31;
        const int spc = min(sp, P - 4);
        n = spc / g.HW;
        pix = spc - n * g.HW;
        return sp < P;
    };
    auto dma_source = [&](int li) {
        int n, pix;
        lane_pixel(li, n, pix);
        return x + ((size_t)n * g.Cin + wave * 8 + khalf) * HW + pix;
    };
    auto out_offset = [&](int li, bool &ok) {          // byte offset of (lane's first row, first pixel) in y / residual / gate
        int n, pix;
        ok = lane_pixel(li, n, pix);
        const int m0w = (li % g.mtiles) * 128 + wave * 32;
        return (unsigned)((((size_t)n * g.Cout + m0w + 4 * khalf) * HW + pix) * 4);
    };
    auto a_tile = [&](int li) {                        // uniform: this wave's 32 rows, channel group 0
        return wF + ((size_t)((li % g.mtiles) * 4 + wave) * kgroups) * 512;
    };

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float *)(Ps + wave * 8 * 128));
    const size_t dma_row2 = 2 * HW, dma_stage = (size_t)CB * HW;
    const float *ld_ptr = dma_source(local);           // load cursor: two stages ahead of the compute cursor, across tiles
    int ld_cb = 0, ld_li = local;
    auto issue_dma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GI; ++i)
            if (DBG != 3) glds16(ld_ptr + i * dma_row2, lds0 + (unsigned)(buf * CB * 128 + i * 256) * 4u);
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
    bool cur_ok, prev_ok = false, pend = false;
    unsigned cur_off = out_offset(li, cur_ok), prev_off = 0;
    int prev_li = li;
    EGroup eg;
#pragma unroll
    for (int i = 0; i < 4; ++i) { eg.sc[i] = 0.f; eg.bi[i] = 0.f; eg.res[i] = f32x4{0, 0, 0, 0}; eg.gate[i] = f32x4{0, 0, 0, 0}; }

    // group q = channel rows 8q .. 8q+3 of the lane's 16.  Epilogue loads and stores are ordinary (compiler-visible) memory
    // operations: the compiler places their waits itself (it never moves or copies a value that has not landed -- an asm load
    // with a register output gives no such guarantee) and handles the store-data / SGPR hazards.  They sit between two asm
    // statements with memory clobbers, so they are issued where they are written.
    const unsigned row_bytes = (unsigned)(HW * 4);
    auto eload = [&](int tli, unsigned off, bool ok, int q) {
        const int m0w = (tli % g.mtiles) * 128 + wave * 32;
        const unsigned o = off + (unsigned)(8 * q) * row_bytes;
        if (DBG == 2 || DBG == 3) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            eg.sc[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(spp + (has_sc ? m0w + 8 * q + i : 0)) + sc_off);
            eg.bi[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(bpp + (has_bi ? m0w + 8 * q + i : 0)) + bi_off);
            if (HAS_RES)
                eg.res[i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(ep.residual) + (size_t)(o + i * row_bytes));
            if (HAS_GATE)
                eg.gate[i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(ep.gate) + (size_t)(o + i * row_bytes));
        }
    };
    auto econsume = [&](const f32x16 (&o)[4], unsigned off, bool ok, auto QT) {
        constexpr int q = decltype(QT)::value;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 t;
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
            // no predicate: a lane past the pixel range works on the LAST pixel group (lane_pixel clamps), so it recomputes and
            // re-stores that group's values bit for bit -- and every wave issues exactly 4 stores per group (the counted waits)
            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(y) + (size_t)(off + (unsigned)(8 * q + e4) * row_bytes)) = t;
        }
    };

    // ---- prologue of the first tile
    f32x4 a0lo, a0hi, a1lo, a1hi;
    float4 breg[2];
    const float *a_cur = a_tile(li);
    sload4x2(a0lo, a0hi, a_off, a_cur);
    issue_dma(0);
    issue_dma(1);
    a1lo = a0lo; a1hi = a0hi;
    wait_vm<4>(a0lo, a0hi);
    wg_barrier();
    int rbuf = 0;
    breg[0] = *reinterpret_cast<const float4 *>(Ps + (khalf * 8) * 128 + 4 * l31);

    for (;;) {
        const bool has_next = tile_valid(li + stride);
        const float *a_next = has_next ? a_tile(li + stride) : a_cur;
        auto stage = [&](auto JT, int cb) {
            constexpr int J = decltype(JT)::value;        // epilogue group of the previous tile handled in this stage (-1: none)
            const float *__restrict__ pbuf = Ps + rbuf * (CB * 128) + (khalf * 8) * 128 + 4 * l31;
            const int nbuf = rbuf == 2 ? 0 : rbuf + 1;
            const int wbuf = nbuf == 2 ? 0 : nbuf + 1;
            const float *__restrict__ pnext = Ps + nbuf * (CB * 128) + (khalf * 8) * 128 + 4 * l31;
            const bool last = cb + 1 == ncb;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int h = s / 8, kk = s % 8;
                if (s == 0) {
                    // A0 and, older, the epilogue loads landed; the 4 DMA pieces (the youngest loads) stay in flight.  The count
                    // does not depend on how many epilogue loads the compiler emitted (it merges loads of one address).
                    wait_vm<4>(a0lo, a0hi);
                    sload4x2(a1lo, a1hi, a_off, a_cur + (size_t)(2 * cb + 1) * 512);
                }
                if (s == 8) {
                    // A1 (s = 0) and, older, DMA(cb+1) landed; the previous stage's stores are 16 sub-steps old.
                    // (Measured: loads and stores do NOT retire in order relative to each other on gfx950 -- a count that leaves
                    // stores in flight in front of a needed load is wrong; outstanding stores only make a counted wait stricter.)
                    wait_vm<0>(a1lo, a1hi);
                    wg_barrier();
                    if (J >= 0) {
                        if (pend) econsume(out, prev_off, prev_ok, std::integral_constant<int, (J >= 0 ? J : 0)>{});
                    }
                    if (J >= 0 && J < 3) {
                        if (pend) eload(prev_li, prev_off, prev_ok, J + 1);
                    }
                    if (last) eload(li, cur_off, cur_ok, 0);             // (ncb >= 4: the last stage is never one of J = 0..2)
                    sload4x2(a0lo, a0hi, a_off, last ? a_next : a_cur + (size_t)(2 * cb + 2) * 512);
                    issue_dma(wbuf);
                }
                if (s == 15) {
                    if (!last || has_next) breg[(s + 1) & 1] = *reinterpret_cast<const float4 *>(pnext);
                } else {
                    const int h2 = (s + 1) / 8, k2 = (s + 1) % 8;
                    breg[(s + 1) & 1] = *reinterpret_cast<const float4 *>(pbuf + (h2 * 16 + k2) * 128);
                }
                const f32x4 alo = h == 0 ? a0lo : a1lo, ahi = h == 0 ? a0hi : a1hi;
                const float av = kk < 4 ? alo[kk] : ahi[kk - 4];
                const float4 bv = breg[s & 1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[3], 0, 0, 0);
            }
            rbuf = nbuf;
        };
        stage(std::integral_constant<int, 0>{}, 0);
        stage(std::integral_constant<int, 1>{}, 1);
        stage(std::integral_constant<int, 2>{}, 2);
        stage(std::integral_constant<int, 3>{}, 3);
        for (int cb = 4; cb < ncb; ++cb) stage(std::integral_constant<int, -1>{}, cb);
        // ---- tile done: hand the sums to the deferred epilogue (group 0's loads are in flight)
        if (has_next) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                out[j] = acc[j];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
            }
            prev_off = cur_off; prev_ok = cur_ok; prev_li = li;
            pend = true;
            li += stride;
            cur_off = out_offset(li, cur_ok);
            a_cur = a_next;
            continue;
        }
        // ---- last tile of this workgroup: epilogue now.  Group 0 was loaded under the last stage; the loads of groups 1..3 are
        // issued together (the deferred-sum registers are free here), so the tail is one memory round trip, not three.
        wait_vm_only<0>();
        {
            if (HAS_GATE) {                 // (residual + gate: 40 registers per group -- one group at a time)
                econsume(acc, cur_off, cur_ok, std::integral_constant<int, 0>{});
                eload(li, cur_off, cur_ok, 1);
                econsume(acc, cur_off, cur_ok, std::integral_constant<int, 1>{});
                eload(li, cur_off, cur_ok, 2);
                econsume(acc, cur_off, cur_ok, std::integral_constant<int, 2>{});
                eload(li, cur_off, cur_ok, 3);
                econsume(acc, cur_off, cur_ok, std::integral_constant<int, 3>{});
            } else {
                // two groups in flight: the load of group q+1 is issued before group q is consumed
                EGroup ea = eg, eb;
                eload(li, cur_off, cur_ok, 1); eb = eg;
                eg = ea; econsume(acc, cur_off, cur_ok, std::integral_constant<int, 0>{});
                eload(li, cur_off, cur_ok, 2); ea = eg;
                eg = eb; econsume(acc, cur_off, cur_ok, std::integral_constant<int, 1>{});
                eload(li, cur_off, cur_ok, 3); eb = eg;
                eg = ea; econsume(acc, cur_off, cur_ok, std::integral_constant<int, 2>{});
                eg = eb; econsume(acc, cur_off, cur_ok, std::integral_constant<int, 3>{});
            }
        }
        break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.0f;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// W [Cout][Cin] -> fragment-major [Cout/32][Cin/16][2][64][4] (see ring2_kernel)
__global__ void to_fragment_major(const float *__restrict__ w, float *__restrict__ wF, int Cout, int Cin)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cout * Cin) return;
    const int m = (int)(i / Cin), k = (int)(i % Cin);
    const int mb = m / 32, l31 = m % 32, kg = k / 16, khalf = (k % 16) / 8, kk = k % 8;
    const int lane = khalf * 32 + l31;
    wF[((size_t)mb * (Cin / 16) + kg) * 512 + (kk / 4) * 256 + lane * 4 + (kk % 4)] = w[i];
}

static float *dev_random(size_t n, unsigned seed, float scale, float shift = 0.0f)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((float)(s >> 8) / 16777216.0f * 2.0f - 1.0f) * scale + shift;
    }
    float *d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

struct Shape {
    const char *name;
    int N, Cin, HW_h, HW_w, Cout;
};

template <typename F>
static float time_us(F f, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char **argv)
{
    const int iters = 40;
    setvbuf(stdout, NULL, _IONBF, 0);
    float *zero, *sink;
    CK(hipMalloc(&sink, 1024));
    CK(hipGetSymbolAddress((void **)&zero, HIP_SYMBOL(d_zero_page)));
    CK(hipMemset(zero, 0, 256));
    {   // the practical MFMA ceiling on this box (clocks under load)
        float *out;
        CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
        for (int wgs : {256 * 2, 256 * 4}) {
            const int it = 4000;
            float us = time_us([&] { hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(wgs), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 5);
            double fl = (double)wgs * 4 * it * 8 * 4 * 4096.0;
            printf("mfma_peak f32 32x32x2, %d workgroups: %.1f TFLOP/s\n", wgs, fl / us / 1e6);
        }
        CK(hipFree(out));
    }
    if (argc > 1 && !strncmp(argv[1], "dbg", 3)) {
        const int d = atoi(argv[1] + 3);
        const int N = 4, Cin = 256, HW = 4096, Cout = 1024;
        float *x = dev_random((size_t)N * Cin * HW, 1, 1.0f), *w = dev_random((size_t)Cout * Cin, 2, 0.05f);
        float *sc = dev_random(Cout, 3, 0.25f, 1.0f), *bi = dev_random(Cout, 4, 1.0f), *res = dev_random((size_t)N * Cout * HW, 5, 1.0f);
        float *wF, *y1;
        CK(hipMalloc(&wF, (size_t)Cout * Cin * 4));
        CK(hipMalloc(&y1, (size_t)N * Cout * HW * 4));
        hipLaunchKernelGGL(to_fragment_major, dim3((Cout * Cin + 255) / 256), dim3(256), 0, 0, w, wF, Cout, Cin);
        Epi ep = {bi, sc, res, nullptr, 1};
        Geom g = {N, Cin, HW, Cout, N * HW / 128, Cout / 128, zero, 0, sink};
        const int nwg = 256;
        printf("dbg %d ...\n", d);
        if (d == 0) hipLaunchKernelGGL((ring2_kernel<true, false, 0>), dim3(nwg), dim3(256), 49152, 0, x, wF, ep, y1, g, nwg);
        if (d == 1) hipLaunchKernelGGL((ring2_kernel<true, false, 1>), dim3(nwg), dim3(256), 49152, 0, x, wF, ep, y1, g, nwg);
        if (d == 2) hipLaunchKernelGGL((ring2_kernel<true, false, 2>), dim3(nwg), dim3(256), 49152, 0, x, wF, ep, y1, g, nwg);
        if (d == 4) hipLaunchKernelGGL((ring2_kernel<true, false, 4>), dim3(nwg), dim3(256), 49152, 0, x, wF, ep, y1, g, nwg);
        if (d == 5) hipLaunchKernelGGL((ring2_kernel<true, false, 5>), dim3(nwg), dim3(256), 49152, 0, x, wF, ep, y1, g, nwg);
        if (d == 3) hipLaunchKernelGGL((ring2_kernel<true, false, 3>), dim3(nwg), dim3(256), 49152, 0, x, wF, ep, y1, g, nwg);
        CK(hipDeviceSynchronize());
        printf("dbg %d ok\n", d);
        return 0;
    }
    const Shape shapes[] = {{"C4 256->1024", 4, 256, 64, 64, 1024}, {"C4 1024->256", 4, 1024, 64, 64, 256},
                            {"C3 128->512", 4, 128, 128, 128, 512}, {"C3 512->128", 4, 512, 128, 128, 128},
                            {"C5 512->2048", 4, 512, 32, 32, 2048}, {"C5 2048->512", 4, 2048, 32, 32, 512},
                            {"roi 256->1024", 1376, 256, 14, 14, 1024}, {"roi 1024->256", 672, 1024, 14, 14, 256},
                            {"odd 7x7 256->256", 3, 256, 7, 4, 256}};
    for (const Shape &s : shapes) {
        const int HW = s.HW_h * s.HW_w;
        const size_t nx = (size_t)s.N * s.Cin * HW, ny = (size_t)s.N * s.Cout * HW;
        float *x = dev_random(nx, 1, 1.0f), *w = dev_random((size_t)s.Cout * s.Cin, 2, 0.05f);
        float *sc = dev_random(s.Cout, 3, 0.25f, 1.0f), *bi = dev_random(s.Cout, 4, 1.0f);
        float *res = dev_random(ny, 5, 1.0f), *gate = dev_random(ny, 6, 1.0f);
        float *wF;
        CK(hipMalloc(&wF, (size_t)s.Cout * s.Cin * sizeof(float)));
        hipLaunchKernelGGL(to_fragment_major, dim3((unsigned)(((size_t)s.Cout * s.Cin + 255) / 256)), dim3(256), 0, 0, w, wF, s.Cout,
                           s.Cin);
        float *y0, *y1;
        CK(hipMalloc(&y0, ny * sizeof(float)));
        CK(hipMalloc(&y1, ny * sizeof(float)));
        const double fl = 2.0 * s.N * HW * (double)s.Cin * s.Cout;
        for (int mode = 0; mode < 3; ++mode) {       // 0 fwd (scale, bias, residual, relu); 1 dgrad (residual, gate); 2 plain
            Epi ep = {mode == 1 ? nullptr : bi, mode == 0 ? sc : nullptr, mode == 2 ? nullptr : res, mode == 1 ? gate : nullptr,
                      mode == 0 ? 1 : 0};
            const char *mname = mode == 0 ? "fwd  " : mode == 1 ? "dgrad" : "plain";
            auto lib = [&] {
                fi_conv2d_forward_gated(x, w, ep.bias, ep.scale, ep.residual, ep.gate, y0, s.N, s.Cin, s.HW_h, s.HW_w, s.Cout, 1, 1,
                                        1, 1, 0, 0, ep.relu, 0, 0, 0, 0, nullptr);
            };
            const float t_lib = time_us(lib, iters);
            printf("%-14s %s library           %7.1f us %6.1f TFLOP/s\n", s.name, mname, t_lib, fl / t_lib / 1e6);
            Geom g = {s.N, s.Cin, HW, s.Cout, (s.N * HW + 127) / 128, (s.Cout + 127) / 128, zero, 0, sink};
            const long tiles = (long)((g.ptiles + 7) / 8) * 8 * g.mtiles;
            std::vector<float> h0(ny), h1(ny);
            CK(hipMemcpy(h0.data(), y0, ny * sizeof(float), hipMemcpyDeviceToHost));
            auto check = [&](const char *tag, float t) {
                CK(hipMemcpy(h1.data(), y1, ny * sizeof(float), hipMemcpyDeviceToHost));
                double md = 0.0, mr = 0.0;
                for (size_t i = 0; i < ny; ++i) {
                    md = fmax(md, fabs((double)h1[i] - h0[i]));
                    mr = fmax(mr, fabs((double)h0[i]));
                }
                printf("%-14s %s %-17s %7.1f us %6.1f TFLOP/s   max|diff| %.3g (max|ref| %.3g)\n", s.name, mname, tag, t,
                       fl / t / 1e6, md, mr);
                if (md > 1e-3 * mr && getenv("SHOW_BAD")) {
                    long bad = 0;
                    long rows_bad[16] = {0}, px_bad[8] = {0};
                    for (size_t i = 0; i < ny; ++i)
                        if (fabs((double)h1[i] - h0[i]) > 1e-4 * mr) {
                            const int pix = (int)(i % HW), c = (int)((i / HW) % s.Cout), n = (int)(i / ((size_t)HW * s.Cout));
                            if (bad < 12) printf("   bad n=%d c=%d pix=%d got %.5f ref %.5f\n", n, c, pix, h1[i], h0[i]);
                            ++bad;
                            rows_bad[c % 16]++;
                            px_bad[pix % 8]++;
                        }
                    printf("   %ld of %zu elements differ; by channel %% 16:", bad, ny);
                    for (int k = 0; k < 16; ++k) printf(" %ld", rows_bad[k]);
                    printf("; by pixel %% 8:");
                    for (int k = 0; k < 8; ++k) printf(" %ld", px_bad[k]);
                    printf("\n");
                }
            };
            for (int per_cu : {1, 2, 3, 4}) {
                g.no_epilogue = per_cu == 3 ? 1 : per_cu == 4 ? 3 : 0;        // variants 3 / 4: pers2 with a start offset
                if (per_cu > 2) per_cu = 2;
                const int nwg = (int)((tiles < 256L * per_cu ? tiles : 256L * per_cu) / 8 * 8);
                CK(hipMemset(y1, 0, ny * sizeof(float)));
                auto f = [&] {
                    if (mode == 0)
                        hipLaunchKernelGGL((ring2_kernel<true, false>), dim3((unsigned)nwg), dim3(256), 3 * 32 * 128 * 4, 0, x, wF, ep, y1,
                                           g, nwg);
                    else if (mode == 1)
                        hipLaunchKernelGGL((ring2_kernel<true, true>), dim3((unsigned)nwg), dim3(256), 3 * 32 * 128 * 4, 0, x, wF, ep, y1,
                                           g, nwg);
                    else
                        hipLaunchKernelGGL((ring2_kernel<false, false>), dim3((unsigned)nwg), dim3(256), 3 * 32 * 128 * 4, 0, x, wF, ep,
                                           y1, g, nwg);
                };
                float t = time_us(f, iters);
                char tag[64];
                snprintf(tag, sizeof tag, "ring2 pers%d delay%d", per_cu, g.no_epilogue);
                check(tag, t);
                g.no_epilogue = 0;
            }
            for (int noep = 0; noep < 2; ++noep) {
                if (noep && mode != 0) continue;
                if (getenv("V2_ONLY")) continue;
                g.no_epilogue = noep;
                char tag[64];
                {
                    CK(hipMemset(y1, 0, ny * sizeof(float)));
                    auto f = [&] {
                        hipLaunchKernelGGL((ring_kernel<32, false>), dim3((unsigned)tiles), dim3(256), 3 * 32 * 128 * 4, 0, x, w,
                                           ep, y1, g, (int)tiles);
                    };
                    float t = time_us(f, iters);
                    snprintf(tag, sizeof tag, "ring32%s", noep ? " noepi" : "");
                    if (noep) printf("%-14s %s %-17s %7.1f us %6.1f TFLOP/s\n", s.name, mname, tag, t, fl / t / 1e6);
                    else check(tag, t);
                }
                if (noep) {
                    auto run_exp = [&](auto kern, const char *what) {
                        auto f = [&] { hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), 3 * 32 * 128 * 4, 0, x, w, ep, y1, g, (int)tiles); };
                        float t = time_us(f, iters);
                        printf("%-14s %s %-28s %7.1f us %6.1f TFLOP/s\n", s.name, mname, what, t, fl / t / 1e6);
                    };
                    run_exp(ring_kernel<32, false, 1>, "noepi exp1 cached pixels");
                    run_exp(ring_kernel<32, false, 2>, "noepi exp2 no weight loads");
                    run_exp(ring_kernel<32, false, 3>, "noepi exp3 no barrier");
                    run_exp(ring_kernel<32, false, 4>, "noepi exp4 no LDS reads");
                    run_exp(ring_kernel<32, false, 5>, "noepi exp5 MFMA only");
                }
                for (int per_cu : {2, 3}) {
                    const int nwg = (int)((tiles < 256L * per_cu ? tiles : 256L * per_cu) / 8 * 8);
                    CK(hipMemset(y1, 0, ny * sizeof(float)));
                    auto f = [&] {
                        hipLaunchKernelGGL((ring_kernel<32, true>), dim3((unsigned)nwg), dim3(256), 3 * 32 * 128 * 4, 0, x, w, ep,
                                           y1, g, nwg);
                    };
                    float t = time_us(f, iters);
                    snprintf(tag, sizeof tag, "ring32 pers%d%s", per_cu, noep ? " noepi" : "");
                    if (noep) printf("%-14s %s %-17s %7.1f us %6.1f TFLOP/s\n", s.name, mname, tag, t, fl / t / 1e6);
                    else check(tag, t);
                }
            }
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(sc)); CK(hipFree(bi)); CK(hipFree(res)); CK(hipFree(gate));
        CK(hipFree(y0)); CK(hipFree(y1));
    }
    return 0;
}
