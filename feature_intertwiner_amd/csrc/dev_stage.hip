// dev_stage.hip -- the index side of the intertwiner RoI stage with static shapes (Dev._forward_static).
//
// Specification: lib/sub_module.py:437-598 of the reference sizes everything by the RoIs' pyramid levels with one
// nonzero / .any() per level (a host synchronisation each): the 'small' rows of levels 2..4 in level-major order
// (:583-598), and for every level l < 5 the 'big' rows of the levels above it (:456-541, _find_big_box2 :366-378).
// Here one launch of one workgroup turns level[N] (2..5) and the class ids gt[N] into
//   order[N]            stable level-major permutation (== torch.sort(level, stable=True)[1])
//   small_cls[N]        (level - 2) * K + gt for the rows of levels 2..4 with gt > 0, else 0 -- the class index of ONE
//                       class-mean launch over 3 K classes (sorted order)
//   small_gt[N] (float) gt on levels 2..4, else 0 (sorted order: small_gt_all of the reference)
//   small_on[N] (u8)    level <= 4 (sorted order)
//   big_idx[cap], big_level[cap], big_cls[cap]: every (level l, RoI of a higher level) pair in (l, RoI) order, compacted
//                       to the front; big_level = l (or -1 behind the live count: the crop does not write such a row),
//                       big_cls = (l - 2) * K + gt when level l has small rows at all (:456-467) and gt > 0, else 0
//   counts[5]           n2, n3, n4, n5 and the live count n3 + 2 n4 + 3 n5 (device integers: no host read)
// Oracle: the tensor formulation Dev._static_index_tensors (tests/test_gpu_static_dev.py compares the two).
#include "fi_common.h"

namespace {

constexpr int kT = 1024;

__global__ __launch_bounds__(kT) void dev_index_kernel(const int *__restrict__ level, const int *__restrict__ gt, int N, int K,
                                                       int cap, long long *__restrict__ order, int *__restrict__ small_cls,
                                                       float *__restrict__ small_gt, unsigned char *__restrict__ small_on,
                                                       long long *__restrict__ big_idx, int *__restrict__ big_level,
                                                       int *__restrict__ big_cls, int *__restrict__ counts)
{
    __shared__ int s_tot[4];            // rows per level
    __shared__ int s_wave[16][4];       // per wavefront, per level: rows of this chunk
    __shared__ int s_run[4];            // rows per level in the chunks before this one
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 4) { s_tot[tid] = 0; s_run[tid] = 0; }
    __syncthreads();
    // pass 1: rows per level
    int mine[4] = {0, 0, 0, 0};
    for (int i = tid; i < N; i += kT) {
        const int l = min(max(level[i], 2), 5) - 2;
        mine[l]++;
    }
    for (int l = 0; l < 4; ++l) {
        int v = mine[l];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && v) atomicAdd(&s_tot[l], v);
    }
    __syncthreads();
    const int n2 = s_tot[0], n3 = s_tot[1], n4 = s_tot[2], n5 = s_tot[3];
    const int start[4] = {0, n2, n2 + n3, n2 + n3 + n4};                    // first sorted position of each level
    const int bcount[3] = {n3 + n4 + n5, n4 + n5, n5};                       // big rows of levels 2, 3, 4
    const int bbase[3] = {0, bcount[0], bcount[0] + bcount[1]};
    const int live = bcount[0] + bcount[1] + bcount[2];
    const bool has_small[3] = {n2 > 0, n3 > 0, n4 > 0};
    if (tid == 0 && counts) {
        counts[0] = n2; counts[1] = n3; counts[2] = n4; counts[3] = n5; counts[4] = live;
    }
    // pass 2: chunk by chunk, the number of earlier rows of every level (ballots inside a wavefront, LDS across them)
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int c0 = 0; c0 < N; c0 += kT) {
        const int i = c0 + tid;
        const bool ok = i < N;
        const int l = ok ? min(max(level[i], 2), 5) - 2 : -1;
        int before[4];
        for (int q = 0; q < 4; ++q) {
            const unsigned long long b = __ballot(l == q);
            before[q] = __popcll(b & lt);
            if (lane == 0) s_wave[wave][q] = __popcll(b);
        }
        __syncthreads();
        int chunk_tot[4];
        for (int q = 0; q < 4; ++q) {
            int pre = 0, tot = 0;
            for (int w = 0; w < kT / 64; ++w) {
                const int v = s_wave[w][q];
                if (w < wave) pre += v;
                tot += v;
            }
            before[q] += pre + s_run[q];
            chunk_tot[q] = tot;
        }
        if (ok) {
            const int g = gt ? gt[i] : 0;
            const int pos = start[l] + before[l];
            order[pos] = i;
            const bool on = l < 3;
            if (small_on) small_on[pos] = on ? 1 : 0;
            if (small_gt) small_gt[pos] = on ? (float)g : 0.0f;
            if (small_cls) small_cls[pos] = (on && g > 0 && g < K) ? l * K + g : 0;
            if (big_idx) {
                // level index q < l: this row is a 'big' row of level q; its rank there = earlier rows of the levels above q
                int above = 0;
                for (int q = 3; q >= 1; --q) {
                    above += before[q];                                   // rows j < i with level >= q (index q..3)
                    const int ql = q - 1;                                 // the lower level it supervises
                    if (l >= q) {
                        const int p = bbase[ql] + above;
                        big_idx[p] = i;
                        big_level[p] = 2 + ql;
                        big_cls[p] = (has_small[ql] && g > 0 && g < K) ? ql * K + g : 0;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 4) s_run[tid] += chunk_tot[tid];
        __syncthreads();
    }
    if (big_idx)
        for (int p = live + tid; p < cap; p += kT) {
            big_idx[p] = 0;
            big_level[p] = -1;
            big_cls[p] = 0;
        }
}

}  // namespace

extern "C" {

int fi_dev_stage_index(const int32_t *level, const int32_t *gt, int N, int num_classes, int capacity, int64_t *order,
                       int32_t *small_cls, float *small_gt, uint8_t *small_on, int64_t *big_idx, int32_t *big_level,
                       int32_t *big_cls, int32_t *counts, fi_stream_t stream)
{
    FI_REQUIRE(N >= 1 && num_classes >= 1, "N, num_classes >= 1");
    FI_REQUIRE(level && order, "null pointer");
    FI_REQUIRE(!big_idx || (big_level && big_cls && capacity >= 3 * N), "the big batch needs big_level, big_cls and capacity >= 3 N");
    hipLaunchKernelGGL(dev_index_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, level, gt, N, num_classes, capacity,
                       (long long *)order, small_cls, small_gt, small_on, (long long *)big_idx, big_level, big_cls, counts);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
