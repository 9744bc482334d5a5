// sinkhorn.hip -- batched entropic-OT (Sinkhorn) term of the intertwiner loss, gfx950.
//
// Specification: OptTrans._sinkhorn_iterate, lib/OT_module.py:104-135 (oracle:
// orc_sinkhorn).  The reference evaluates each problem with 2L+6 tiny torch kernels
// inside a Python loop over classes and loss terms (up to 240 problems, :100-101).
//
// One launch solves every problem: one 1024-thread workgroup (16 wavefronts, one
// CU) per problem, S <= 256 samples.  The S x S Gibbs kernel K = exp(-C/eps) never
// touches memory: it lives in REGISTERS as a 32 x 32 grid of 8x8 tiles, one tile
// (64 VGPRs) per thread -- 256 KB of K would not fit the 160 KB LDS.  Each of the
// 2L dependent mat-vecs is then
//   K b   : 64 FMAs/thread, row sums reduced across the 32 lanes that share a row
//           tile (cross-lane shuffles inside a half-wavefront),
//   K^T a : 64 FMAs/thread, column sums reduced across row tiles: one xor-32
//           shuffle inside the wavefront, then a 16-way sum through LDS.
// The scaling vectors a, b live in LDS.  The cost matrix is recomputed for the
// final <P, C> instead of being held in 64 more registers.  The roofline that
// binds is on-chip (fp32 FMA issue + LDS/barrier latency); HBM traffic is 2*S*D
// floats in and 1 float out per problem.
#include "fi_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxS = 256;
constexpr int kT = 8;         // tile edge held per thread
constexpr int kDChunk = 16;   // feature columns staged per pass (D > 1)
constexpr int kStride = 260;  // LDS row stride of the staged, transposed chunk
#define FI_OT_EPS 1e-20f

// v (DPP-moved) with zero where the control selects no source lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}

struct Smem {
    float a[kMaxS];
    float b[kMaxS];
    float nx[kMaxS];  // ||x_i|| + eps (cosine) -- divisor
    float ny[kMaxS];
    float red[16][kMaxS];
    float xs[kDChunk][kStride];
    float ys[kDChunk][kStride];
    float wsum[16];
};

// cost tile C[r][c] for rows ti*8+r, cols tj*8+c of problem (x, y).
__device__ void cost_tile(Smem &sm, const float *__restrict__ x, const float *__restrict__ y, int S,
                          int D, int l2_cost, int normalize, int ti, int tj, float (&C)[kT][kT])
{
    const int tid = threadIdx.x;
    float acc[kT][kT];
#pragma unroll
    for (int r = 0; r < kT; ++r)
#pragma unroll
        for (int c = 0; c < kT; ++c) acc[r][c] = 0.0f;

    for (int d0 = 0; d0 < D; d0 += kDChunk) {
        const int dc = min(kDChunk, D - d0);
        __syncthreads();  // previous chunk fully consumed
        for (int e = tid; e < kMaxS * kDChunk; e += kThreads) {
            const int i = e / kDChunk;
            const int d = e - i * kDChunk;
            float xv = 0.0f, yv = 0.0f;
            if (i < S && d < dc) {
                xv = x[(size_t)i * D + d0 + d];
                yv = y[(size_t)i * D + d0 + d];
                if (normalize) {  // x /= (||x|| + eps), OT_module.py:111-112
                    xv = xv / sm.nx[i];
                    yv = yv / sm.ny[i];
                }
            }
            sm.xs[d][i] = xv;
            sm.ys[d][i] = yv;
        }
        __syncthreads();
        for (int d = 0; d < dc; ++d) {
            float xr[kT], yc[kT];
#pragma unroll
            for (int r = 0; r < kT; ++r) xr[r] = sm.xs[d][ti * kT + r];
#pragma unroll
            for (int c = 0; c < kT; ++c) yc[c] = sm.ys[d][tj * kT + c];
#pragma unroll
            for (int r = 0; r < kT; ++r)
#pragma unroll
                for (int c = 0; c < kT; ++c) {
                    if (l2_cost) {
                        const float df = xr[r] - yc[c];
                        acc[r][c] = fmaf(df, df, acc[r][c]);
                    } else {
                        acc[r][c] = fmaf(xr[r], yc[c], acc[r][c]);
                    }
                }
        }
    }
#pragma unroll
    for (int r = 0; r < kT; ++r)
#pragma unroll
        for (int c = 0; c < kT; ++c) C[r][c] = l2_cost ? sqrtf(acc[r][c]) : (1.0f - acc[r][c]);
}

__global__ __launch_bounds__(kThreads) void sinkhorn_kernel(const float *__restrict__ xs_all,
                                                            const float *__restrict__ ys_all, int S,
                                                            int D, float eps_inv, int L, int cost_mode,
                                                            float *__restrict__ loss,
                                                            float *__restrict__ plan,
                                                            float *__restrict__ xn_out,
                                                            float *__restrict__ yn_out)
{
    __shared__ Smem sm;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tj = tid & 31;   // column tile
    const int ti = tid >> 5;   // row tile (lanes 0-31 and 32-63 of a wave hold ti = 2w, 2w+1)
    const size_t prob = blockIdx.x;
    const int l2_cost = (cost_mode == 1);
    const int normalize = (cost_mode == 0);  // mode 2: rows are already normalised
    const float *__restrict__ x = xs_all + prob * (size_t)S * D;
    const float *__restrict__ y = ys_all + prob * (size_t)S * D;

    // ---- row norms (cosine) ----------------------------------------------------
    if (normalize) {
        if (tid < 2 * kMaxS) {
            const int i = tid & (kMaxS - 1);
            const float *__restrict__ src = (tid < kMaxS) ? x : y;
            float ss = 0.0f;
            if (i < S)
                for (int d = 0; d < D; ++d) {
                    const float v = src[(size_t)i * D + d];
                    ss = fmaf(v, v, ss);
                }
            const float nrm = sqrtf(ss) + FI_OT_EPS;
            if (tid < kMaxS) sm.nx[i] = nrm; else sm.ny[i] = nrm;
        }
        __syncthreads();
        if (xn_out && yn_out) {
            for (int e = tid; e < S * D; e += kThreads) {
                const int i = e / D;
                xn_out[prob * (size_t)S * D + e] = x[e] / sm.nx[i];
                yn_out[prob * (size_t)S * D + e] = y[e] / sm.ny[i];
            }
        }
    }

    // ---- K tile in registers ------------------------------------------------------
    float K[kT][kT];
    {
        float C[kT][kT];
        cost_tile(sm, x, y, S, D, l2_cost, normalize, ti, tj, C);
#pragma unroll
        for (int r = 0; r < kT; ++r)
#pragma unroll
            for (int c = 0; c < kT; ++c) {
                const bool in = (ti * kT + r < S) && (tj * kT + c < S);
                K[r][c] = in ? expf(-eps_inv * C[r][c]) : 0.0f;
            }
    }

    const float u = 1.0f / (float)S;
    if (tid < kMaxS) {
        sm.b[tid] = (tid < S) ? u : 0.0f;
        sm.a[tid] = (tid < S) ? u : 0.0f;
    }
    __syncthreads();

    for (int it = 0; it < L; ++it) {
        // ---- a = u / (K b + eps) ---------------------------------------------------
        {
            float bv[kT];
#pragma unroll
            for (int c = 0; c < kT; ++c) bv[c] = sm.b[tj * kT + c];
            float part[kT];
#pragma unroll
            for (int r = 0; r < kT; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int c = 0; c < kT; ++c) s = fmaf(K[r][c], bv[c], s);
                part[r] = s;
            }
            // Reduce over the 32 lanes that share this row tile, HALVING the values a lane carries at every step: the pair
            // (lane, partner) splits its rows -- the lower lane keeps the first half and is sent the partner's, the
            // upper lane the second half -- so 8 values need 4 + 2 + 1 exchanges (quad_perm xor 1, xor 2, ds_swizzle xor 4: partners
            // always carry the same rows), then the one value left is folded by row_ror:8 and ds_swizzle xor 16.
            // Every lane ends with the FULL sum of one row, r = 4*bit0 + 2*bit1 + bit2 of its lane number, and does ONE
            // division.  (Before: five DPP steps on all 8 values, the sums in lane 31 only, which then did 8 divisions
            // one after the other -- under an exec mask, so at the full cost for the wavefront: 136 instructions, now ~40.)
            float v4[4], v2[2], v;
            {
                const bool up = lane & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float keep = up ? part[k + 4] : part[k], send = up ? part[k] : part[k + 4];
                    v4[k] = keep + dpp_mov<0xB1, 0xF>(send);          // quad_perm [1,0,3,2]
                }
            }
            {
                const bool up = lane & 2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float keep = up ? v4[k + 2] : v4[k], send = up ? v4[k] : v4[k + 2];
                    v2[k] = keep + dpp_mov<0x4E, 0xF>(send);          // quad_perm [2,3,0,1]
                }
            }
            {
                const bool up = lane & 4;
                const float keep = up ? v2[1] : v2[0], send = up ? v2[0] : v2[1];
                v = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), 0x101F));   // lane ^ 4
            }
            v += dpp_mov<0x128, 0xF>(v);                              // row_ror:8 = lane ^ 8
            v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // lane ^ 16
            const int r = ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
            const int i = ti * kT + r;
            const float ai = u / (v + FI_OT_EPS);
            if ((tj & 24) == 0 && i < S) sm.a[i] = ai;
        }
        __syncthreads();
        // ---- b = u / (K^T a + eps) -------------------------------------------------
        {
            float av[kT];
#pragma unroll
            for (int r = 0; r < kT; ++r) av[r] = sm.a[ti * kT + r];
            float part[kT];
#pragma unroll
            for (int c = 0; c < kT; ++c) {
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < kT; ++r) s = fmaf(K[r][c], av[r], s);
                part[c] = s;
            }
#pragma unroll
            for (int c = 0; c < kT; ++c) {          // lane l + lane l^32 in every lane (v_permlane32_swap: no LDS crossbar)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(part[c]), __float_as_int(part[c]), false, false);
                part[c] = __int_as_float(sw[0]) + __int_as_float(sw[1]);
            }
            if (lane < 32) {
#pragma unroll
                for (int c = 0; c < kT; ++c) sm.red[wave][tj * kT + c] = part[c];
            }
        }
        __syncthreads();
        if (tid < kMaxS) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += sm.red[w][tid];
            if (tid < S) sm.b[tid] = u / (s + FI_OT_EPS);
        }
        __syncthreads();
    }

    // ---- P = a K b^T, loss = <P, C> -----------------------------------------------
    float C[kT][kT];
    cost_tile(sm, x, y, S, D, l2_cost, normalize, ti, tj, C);
    float local = 0.0f;
#pragma unroll
    for (int r = 0; r < kT; ++r) {
        const int i = ti * kT + r;
        const float ai = sm.a[i];
#pragma unroll
        for (int c = 0; c < kT; ++c) {
            const int j = tj * kT + c;
            const float ak = ai * K[r][c];
            const float p = ak * sm.b[j];
            if (i < S && j < S) {
                local = fmaf(p, C[r][c], local);
                if (plan) plan[prob * (size_t)S * S + (size_t)i * S + j] = p;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off, 64);
    if (lane == 0) sm.wsum[wave] = local;
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int w = 0; w < 16; ++w) s += sm.wsum[w];
        loss[prob] = s;
    }
}

}  // namespace

extern "C" {

int fi_sinkhorn_forward(const float *x, const float *y, int num_problems, int S, int D,
                        float eps_inv, int L, int cost_mode, float *loss, float *plan, float *xn_out,
                        float *yn_out, fi_stream_t stream)
{
    FI_REQUIRE(num_problems >= 0 && D >= 1 && L >= 1, "num_problems >= 0, D >= 1, L >= 1");
    FI_REQUIRE(cost_mode >= 0 && cost_mode <= 2, "cost_mode in {0 cosine, 1 l2, 2 dot}");
    if (S < 1 || S > kMaxS) {
        fi::set_error("fi_sinkhorn_forward supports 1 <= S <= %d samples (got %d)", kMaxS, S);
        return FI_ERR_UNSUPPORTED;
    }
    if (num_problems == 0) return FI_OK;
    FI_REQUIRE(x && y && loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    fi::ProfScope prof(FI_K_SINKHORN, st);
    hipLaunchKernelGGL(sinkhorn_kernel, dim3(num_problems), dim3(kThreads), 0, st, x, y, S, D,
                       eps_inv, L, cost_mode, loss, plan, xn_out, yn_out);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
