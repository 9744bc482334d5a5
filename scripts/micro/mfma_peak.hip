// Sustained rate of v_mfma_f32_32x32x2_f32 (and v_mfma_f32_32x32x16_bf16) with nothing else in the loop:
// the practical ceiling of the MFMA conv kernels on this part (clocks under load included).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
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

template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float *out, int iters, float a0)
{
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(a0 + e);
        b[e] = (__bf16)(a0 - e);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.0f;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// MFMA operands from LDS: 1 A + 4 B reads per 4 MFMAs, read DIST sub-steps ahead of their use
template <int DIST>
__global__ __launch_bounds__(256) void k_f32_lds(float *out, int iters)
{
    __shared__ float As[16][130], Bs[16][130];
    for (int i = threadIdx.x; i < 16 * 130; i += 256) {
        (&As[0][0])[i] = (float)(i & 7);
        (&Bs[0][0])[i] = (float)(i & 3);
    }
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const int l31 = threadIdx.x & 31, kh = (threadIdx.x >> 5) & 1;
    float a[DIST + 1], b[DIST + 1][4];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int d = 0; d < DIST; ++d) {
            a[d] = As[2 * d + kh][l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[d][j] = Bs[2 * d + kh][l31 + 32 * j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u + DIST < 8) {
                a[(u + DIST) % (DIST + 1)] = As[2 * (u + DIST) + kh][l31];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[(u + DIST) % (DIST + 1)][j] = Bs[2 * (u + DIST) + kh][l31 + 32 * j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % (DIST + 1)], b[u % (DIST + 1)][j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    }
    float s = 0.0f;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 8192);
    const int iters = 4000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu;       // 256 threads = 4 wavefronts = 1 per SIMD
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
            double flops = (double)grid * 4 * iters * 8 * 4 * 4096.0;
            printf("f32  32x32x2   %d wave(s)/SIMD, 4 accumulators: %8.3f ms  %7.1f TFLOP/s\n", blocks_per_cu, ms, flops / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); }, 5);
            double flops = (double)grid * 4 * iters * 8 * 4 * 32768.0;
            printf("bf16 32x32x16  %d wave(s)/SIMD, 4 accumulators: %8.3f ms  %7.1f TFLOP/s\n", blocks_per_cu, ms, flops / ms / 1e9);
        }
    }
    // duration vs occupancy: the same total work as 4 waves/SIMD at 2 waves/SIMD, and 4 waves/SIMD for half as long
    {
        const int grid = 256 * 2;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, out, 2 * iters, 1.0f, 2.0f); }, 5);
        printf("f32  2 waves/SIMD, 2x iterations: %8.3f ms  %7.1f TFLOP/s\n", ms, (double)grid * 4 * 2 * iters * 8 * 4 * 4096.0 / ms / 1e9);
        const int grid4 = 256 * 4;
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid4), dim3(256), 0, 0, out, iters / 4, 1.0f, 2.0f); }, 5);
        printf("f32  4 waves/SIMD, 1/4 iterations: %8.3f ms  %7.1f TFLOP/s\n", ms, (double)grid4 * 4 * (iters / 4) * 8 * 4 * 4096.0 / ms / 1e9);
        const int grid8 = 256 * 8;
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid8), dim3(256), 0, 0, out, iters / 4, 1.0f, 2.0f); }, 5);
        printf("f32  8 waves/SIMD (2 rounds of 4?), 1/4 iterations: %8.3f ms  %7.1f TFLOP/s\n", ms, (double)grid8 * 4 * (iters / 4) * 8 * 4 * 4096.0 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<2>, dim3(grid4), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
        printf("f32  4 waves/SIMD, 2 accumulators: %8.3f ms  %7.1f TFLOP/s\n", ms, (double)grid4 * 4 * iters * 8 * 2 * 4096.0 / ms / 1e9);
    }
    for (int bpc = 1; bpc <= 4; ++bpc) {
        const int grid = 256 * bpc * 4;          // several rounds
        double ms = time_ms([&] { hipLaunchKernelGGL(k_f32_lds<1>, dim3(grid), dim3(256), 0, 0, out, iters / 4); }, 5);
        printf("f32 operands from LDS, 1 sub-step ahead, grid %d: %8.3f ms  %7.1f TFLOP/s\n", grid, ms, (double)grid * 4 * (iters / 4) * 8 * 4 * 4096.0 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32_lds<2>, dim3(grid), dim3(256), 0, 0, out, iters / 4); }, 5);
        printf("f32 operands from LDS, 2 sub-steps ahead, grid %d: %8.3f ms  %7.1f TFLOP/s\n", grid, ms, (double)grid * 4 * (iters / 4) * 8 * 4 * 4096.0 / ms / 1e9);
    }
    {
        const int grid = 256 * 2;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
        double flops = (double)grid * 4 * iters * 8 * 1 * 4096.0;
        printf("f32  32x32x2   2 waves/SIMD, 1 accumulator (dependent chain): %8.3f ms  %7.1f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    return 0;
}
