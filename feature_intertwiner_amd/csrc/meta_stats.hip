// meta_stats.hip -- the statistics side of the intertwiner's meta loss in two launches (one rank, history of one step).
//
// Specification: MaskRCNN.meta_loss, lib/model.py:143-210 of the reference -- _merge_feat_vec (:217-224: count-weighted
// mean over the (gpu, scale) axes) for the big and the small class features, the history-buffer update (:150-166), the
// class selection (:176-181) -- as restated on static shapes by feature_intertwiner_amd.intertwiner.meta_loss, whose
// tensor formulation (≈35 small launches forward, ≈25 autograd nodes backward) is the oracle of these kernels: the same
// fp32 operations in the same order, so the results are bit-identical (tests/test_gpu_meta.py).
//
// Layout: feat[g][s][f][k] = base[g * lg + s * lk + f * ld + k] (strides in floats): both the stacked [G, S, F, K] tensor
// of the reference (ld = K, lk = F K, lg = S F K; G = the ranks of its nn.DataParallel) and the [F, S K] result of one
// class-mean launch over S K classes (G = 1, ld = S K, lk = K).
#include "fi_common.h"

namespace {

constexpr float kEps = 1e-20f;

struct StatsArgs {
    const float *big_feat, *big_cnt, *small_feat, *small_cnt;    // cnt[l][k] = cnt_base[l * K + k]
    int G, S, F, K;                      // feat[g][s][f][k] = base[g * lg + s * lk + f * ld + k]
    int big_ld, big_lk, big_lg, small_ld, small_lk, small_lg;
    float *b_feat;        // [F][K]   merged big features (before the history)
    float *s_feat;        // [F][K]   merged small features
    float *b_cnt, *s_cnt; // [K]
    unsigned int *active; // != 0: the step has small-object statistics (lib/workflow.py:190: small_feat.sum() != 0)
};

// phase 1: the count-weighted sums over the levels and the two merged means; one thread per (f, k)
__global__ __launch_bounds__(256) void meta_merge_kernel(StatsArgs a)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    bool any = false;
    if (i < (long)a.F * a.K) {
        const int f = (int)(i / a.K), k = (int)(i - (long)f * a.K);
        float sb = 0.f, cb = 0.f, ss = 0.f, cs = 0.f;
        // (feat * cnt).sum(0).sum(0): over the ranks' axis first, then over the levels, each in order
        for (int l = 0; l < a.S; ++l) {
            float tb = 0.f, ub = 0.f, ts = 0.f, us = 0.f;
            for (int g = 0; g < a.G; ++g) {
                const float wb = a.big_cnt[(g * a.S + l) * a.K + k], ws = a.small_cnt[(g * a.S + l) * a.K + k];
                const float pb = a.big_feat[(size_t)g * a.big_lg + (size_t)l * a.big_lk + (size_t)f * a.big_ld + k] * wb;
                const float ps = a.small_feat[(size_t)g * a.small_lg + (size_t)l * a.small_lk + (size_t)f * a.small_ld + k] * ws;
                tb = g ? tb + pb : pb;
                ts = g ? ts + ps : ps;
                ub = g ? ub + wb : wb;
                us = g ? us + ws : ws;
            }
            sb = l ? sb + tb : tb;
            ss = l ? ss + ts : ts;
            cb = l ? cb + ub : ub;
            cs = l ? cs + us : us;
        }
        a.b_feat[i] = sb / (cb + kEps);
        a.s_feat[i] = ss / (cs + kEps);
        if (f == 0) {
            a.b_cnt[k] = cb;
            a.s_cnt[k] = cs;
        }
        any = ss != 0.0f;
    }
    // the guard reads `s_sum.sum() != 0`; the sums are >= 0 (class means of ReLU / sigmoid / softmax outputs times counts),
    // so the total vanishes exactly when every term does
    if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) atomicOr(a.active, 1u);
}

// data-parallel form, phase 1a: the LOCAL count-weighted sums only -- sums[0 .. FK) big, [FK .. 2FK) small, then the big
// and the small counts [K] each: one flat vector, all-reduced (sum) across the ranks before phase 1b
__global__ __launch_bounds__(256) void meta_sums_kernel(StatsArgs a, float *__restrict__ sums)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)a.F * a.K;
    if (i >= n) return;
    const int f = (int)(i / a.K), k = (int)(i - (long)f * a.K);
    float sb = 0.f, cb = 0.f, ss = 0.f, cs = 0.f;
    for (int l = 0; l < a.S; ++l) {
        float tb = 0.f, ub = 0.f, ts = 0.f, us = 0.f;
        for (int g = 0; g < a.G; ++g) {
            const float wb = a.big_cnt[(g * a.S + l) * a.K + k], ws = a.small_cnt[(g * a.S + l) * a.K + k];
            const float pb = a.big_feat[(size_t)g * a.big_lg + (size_t)l * a.big_lk + (size_t)f * a.big_ld + k] * wb;
            const float ps = a.small_feat[(size_t)g * a.small_lg + (size_t)l * a.small_lk + (size_t)f * a.small_ld + k] * ws;
            tb = g ? tb + pb : pb;
            ts = g ? ts + ps : ps;
            ub = g ? ub + wb : wb;
            us = g ? us + ws : ws;
        }
        sb = l ? sb + tb : tb;
        ss = l ? ss + ts : ts;
        cb = l ? cb + ub : ub;
        cs = l ? cs + us : us;
    }
    sums[i] = sb;
    sums[n + i] = ss;
    if (f == 0) {
        sums[2 * n + k] = cb;
        sums[2 * n + a.K + k] = cs;
    }
}

// phase 1b: the merged means from the (reduced) sums -- what meta_merge_kernel leaves for the phases below
__global__ __launch_bounds__(256) void meta_means_kernel(const float *__restrict__ sums, int F, int K, float *__restrict__ b_feat,
                                                         float *__restrict__ s_feat, float *__restrict__ b_cnt,
                                                         float *__restrict__ s_cnt, unsigned int *__restrict__ active)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)F * K;
    bool any = false;
    if (i < n) {
        const int k = (int)(i % K);
        const float cb = sums[2 * n + k], cs = sums[2 * n + K + k], ss = sums[n + i];
        b_feat[i] = sums[i] / (cb + kEps);
        s_feat[i] = ss / (cs + kEps);
        if (i < K) {
            b_cnt[k] = cb;
            s_cnt[k] = cs;
        }
        any = ss != 0.0f;
    }
    if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) atomicOr(active, 1u);
}

struct UpdateArgs {
    const float *b_feat, *s_feat, *b_cnt, *s_cnt;
    const unsigned int *active;
    float *buffer, *buffer_cnt;      // [F][K], [K]: the history of ONE step, updated in place
    int F, K;
    float *SMALL, *BIG;              // [K-1][F]: rows = foreground classes
    float *on;                       // [K-1]
    float *active_f;                 // [1]: 1.0 / 0.0
};

// phase 2: history update (new = (buf * cnt + b * b_cnt) / (cnt + b_cnt + eps), kept when the step is not active), the
// transposed operands of the pair loss and the class selection.  32 x 32 tiles through LDS: reads along k, writes along f.
__global__ __launch_bounds__(256) void meta_update_kernel(UpdateArgs a)
{
    __shared__ float t_big[32][33], t_small[32][33];
    const bool act = *a.active != 0;
    const int f0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, k = k0 + tx;
        float nb = 0.f, sm = 0.f;
        if (f < a.F && k < a.K) {
            const size_t i = (size_t)f * a.K + k;
            const float oc = a.buffer_cnt[k], bc = a.b_cnt[k];
            const float sum = a.buffer[i] * oc + a.b_feat[i] * bc;
            const float nc = oc + bc;
            nb = act ? sum / (nc + kEps) : a.buffer[i];
            sm = a.s_feat[i];
        }
        t_big[r][tx] = nb;
        t_small[r][tx] = sm;
    }
    __syncthreads();
    // (every workgroup reads the OLD history counts: they are rewritten by meta_finish_kernel, a launch of its own)
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, k = k0 + tx;
        if (f < a.F && k < a.K) a.buffer[(size_t)f * a.K + k] = t_big[r][tx];
    }
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, f = f0 + tx;                 // transposed: row = class, column = feature
        if (k >= 1 && k < a.K && f < a.F) {
            a.BIG[(size_t)(k - 1) * a.F + f] = t_big[tx][r];
            a.SMALL[(size_t)(k - 1) * a.F + f] = t_small[tx][r];
        }
    }
}

// phase 3 (one workgroup): the history counts, the class selection, the activity flag as a float
__global__ __launch_bounds__(128) void meta_finish_kernel(UpdateArgs a)
{
    const bool act = *a.active != 0;
    for (int k = threadIdx.x; k < a.K; k += 128) {
        const float oc = a.buffer_cnt[k];
        const float nc = act ? oc + a.b_cnt[k] : oc;
        a.buffer_cnt[k] = nc;
        if (k >= 1) a.on[k - 1] = (a.s_cnt[k] > 0.0f && nc > 0.0f) ? 1.0f : 0.0f;
    }
    if (threadIdx.x == 0) *a.active_f = act ? 1.0f : 0.0f;
}

// backward of the small branch: d small_feat[g][l][f][k] = (d SMALL[k-1][f] / (cs[k] + eps)) * small_cnt[g][l][k], 0 for k = 0
__global__ __launch_bounds__(256) void meta_merge_bwd_kernel(const float *__restrict__ dsmall, const float *__restrict__ s_cnt,
                                                             const float *__restrict__ small_cnt, int G, int S, int F, int K,
                                                             int out_ld, int out_lk, int out_lg, float *__restrict__ dfeat)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)F * K) return;
    const int f = (int)(i / K), k = (int)(i - (long)f * K);
    const float g = k >= 1 ? dsmall[(size_t)(k - 1) * F + f] / (s_cnt[k] + kEps) : 0.0f;
    for (int r = 0; r < G; ++r)
        for (int l = 0; l < S; ++l)
            dfeat[(size_t)r * out_lg + (size_t)l * out_lk + (size_t)f * out_ld + k] = g * small_cnt[(r * S + l) * K + k];
}

}  // namespace

extern "C" {

int fi_meta_stats_forward(const float *big_feat, const float *big_cnt, int big_ld, int big_lk, int big_lg,
                          const float *small_feat, const float *small_cnt, int small_ld, int small_lk, int small_lg, int G,
                          int S, int F, int K, float *buffer, float *buffer_cnt, float *s_cnt_out, float *SMALL, float *BIG,
                          float *on, float *active_f, float *workspace, fi_stream_t stream)
{
    FI_REQUIRE(G >= 1 && S >= 1 && F >= 1 && K >= 2, "G, S, F >= 1, K >= 2");
    FI_REQUIRE(big_feat && big_cnt && small_feat && small_cnt && buffer && buffer_cnt && s_cnt_out && SMALL && BIG && on &&
                   active_f && workspace, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    // workspace: b_feat [F K], s_feat [F K], b_cnt [K], flag [1]  (fi_meta_stats_workspace_bytes)
    float *b_feat = workspace, *s_feat = workspace + (size_t)F * K, *b_cnt = workspace + 2 * (size_t)F * K;
    unsigned int *flag = reinterpret_cast<unsigned int *>(b_cnt + K);
    FI_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(unsigned int), st));
    StatsArgs sa = {big_feat, big_cnt, small_feat, small_cnt, G, S, F, K, big_ld, big_lk, big_lg, small_ld, small_lk, small_lg,
                    b_feat, s_feat, b_cnt, s_cnt_out, flag};
    const long n = (long)F * K;
    hipLaunchKernelGGL(meta_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sa);
    UpdateArgs ua = {b_feat, s_feat, b_cnt, s_cnt_out, flag, buffer, buffer_cnt, F, K, SMALL, BIG, on, active_f};
    hipLaunchKernelGGL(meta_update_kernel, dim3((unsigned)((F + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0, st, ua);
    hipLaunchKernelGGL(meta_finish_kernel, dim3(1), dim3(128), 0, st, ua);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_meta_stats_sums(const float *big_feat, const float *big_cnt, int big_ld, int big_lk, int big_lg, const float *small_feat,
                       const float *small_cnt, int small_ld, int small_lk, int small_lg, int G, int S, int F, int K, float *sums,
                       fi_stream_t stream)
{
    FI_REQUIRE(G >= 1 && S >= 1 && F >= 1 && K >= 2, "G, S, F >= 1, K >= 2");
    FI_REQUIRE(big_feat && big_cnt && small_feat && small_cnt && sums, "null pointer");
    StatsArgs sa = {big_feat, big_cnt, small_feat, small_cnt, G, S, F, K, big_ld, big_lk, big_lg, small_ld, small_lk, small_lg,
                    nullptr, nullptr, nullptr, nullptr, nullptr};
    const long n = (long)F * K;
    hipLaunchKernelGGL(meta_sums_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sa, sums);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_meta_stats_from_sums(const float *sums, int F, int K, float *buffer, float *buffer_cnt, float *s_cnt_out, float *SMALL,
                            float *BIG, float *on, float *active_f, float *workspace, fi_stream_t stream)
{
    FI_REQUIRE(F >= 1 && K >= 2, "F >= 1, K >= 2");
    FI_REQUIRE(sums && buffer && buffer_cnt && s_cnt_out && SMALL && BIG && on && active_f && workspace, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    float *b_feat = workspace, *s_feat = workspace + (size_t)F * K, *b_cnt = workspace + 2 * (size_t)F * K;
    unsigned int *flag = reinterpret_cast<unsigned int *>(b_cnt + K);
    FI_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(unsigned int), st));
    const long n = (long)F * K;
    hipLaunchKernelGGL(meta_means_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sums, F, K, b_feat, s_feat, b_cnt,
                       s_cnt_out, flag);
    UpdateArgs ua = {b_feat, s_feat, b_cnt, s_cnt_out, flag, buffer, buffer_cnt, F, K, SMALL, BIG, on, active_f};
    hipLaunchKernelGGL(meta_update_kernel, dim3((unsigned)((F + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0, st, ua);
    hipLaunchKernelGGL(meta_finish_kernel, dim3(1), dim3(128), 0, st, ua);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

size_t fi_meta_stats_workspace_bytes(int F, int K)
{
    if (F < 1 || K < 1) return 0;
    return sizeof(float) * (2 * (size_t)F * K + (size_t)K + 4);
}

int fi_meta_stats_backward(const float *dsmall, const float *s_cnt, const float *small_cnt, int G, int S, int F, int K,
                           int out_ld, int out_lk, int out_lg, float *dfeat, fi_stream_t stream)
{
    FI_REQUIRE(G >= 1 && S >= 1 && F >= 1 && K >= 2, "G, S, F >= 1, K >= 2");
    FI_REQUIRE(dsmall && s_cnt && small_cnt && dfeat, "null pointer");
    const long n = (long)F * K;
    hipLaunchKernelGGL(meta_merge_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dsmall, s_cnt,
                       small_cnt, G, S, F, K, out_ld, out_lk, out_lg, dfeat);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
