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
};

__device__ __attribute__((aligned(16))) float d_zero_page[64];


typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA and weight loads as inline asm: hipcc treats the builtin LDS-DMA as a store that may alias every later
// ds_read and drains it (s_waitcnt vmcnt(0)) at the top of every stage; with asm the waits are the counted ones below.
// (vmcnt counts in issue order; a compiler-inserted vmcnt(N) for its own loads can only over-wait.)
__device__ __forceinline__ void glds16(const float *g, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_byte_addr) : "memory", "m0");
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
    float *zero;
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
    const Shape shapes[] = {{"C4 256->1024", 4, 256, 64, 64, 1024}, {"C4 1024->256", 4, 1024, 64, 64, 256},
                            {"C3 128->512", 4, 128, 128, 128, 512}};
    for (const Shape &s : shapes) {
        const int HW = s.HW_h * s.HW_w;
        const size_t nx = (size_t)s.N * s.Cin * HW, ny = (size_t)s.N * s.Cout * HW;
        float *x = dev_random(nx, 1, 1.0f), *w = dev_random((size_t)s.Cout * s.Cin, 2, 0.05f);
        float *sc = dev_random(s.Cout, 3, 0.25f, 1.0f), *bi = dev_random(s.Cout, 4, 1.0f);
        float *res = dev_random(ny, 5, 1.0f), *gate = dev_random(ny, 6, 1.0f);
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
            Geom g = {s.N, s.Cin, HW, s.Cout, (s.N * HW + 127) / 128, (s.Cout + 127) / 128, zero, 0};
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
            };
            for (int noep = 0; noep < 2; ++noep) {
                if (noep && mode != 0) continue;
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
