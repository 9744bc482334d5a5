// LDS atomic throughput on gfx950, measured (scripts/micro: not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/lds_atomic_rate.hip -o scripts/micro/bin/lds_atomic_rate
// One workgroup of 256 threads per CU slot, K rounds of ds_add to an LDS array; patterns:
//   0 ds_add_f32, lane-consecutive addresses (no bank conflict)      1 ds_add_f32, stride 32 floats (32-way conflict)
//   2 ds_add_f32, all lanes of a quad on one address (same-address)  3 ds_add_u32, lane-consecutive
//   4 plain ds_read + add + ds_write, lane-consecutive (what a non-atomic accumulate costs)
//   5 ds_add_f32, random addresses in a 1024-float tile             6 ds_add_rtn_f32 (returning), lane-consecutive
//   7 ds_add_f64 consecutive    8 fp32 add as a compare-and-swap loop (ds_cmpst_rtn_b32), consecutive
//   9 the same, 4 lanes per address   10 ds_add_f32 with 8 of 64 lanes active
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int rounds, const int *rnd)
{
    __shared__ float s[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) s[i] = 0.0f;
    __syncthreads();
    int a;
    if (MODE == 1) a = (tid * 32) & 8191;
    else if (MODE == 2 || MODE == 9) a = tid >> 2;
    else if (MODE == 5) a = rnd[tid] & 1023;
    else a = tid;
    float acc = 0.0f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = (a + u * 1024) & 8191;
            if (MODE == 3) atomicAdd((unsigned *)s + idx, 1u);
            else if (MODE == 4) { volatile float *vs = s; vs[idx] = vs[idx] + 1.0f; }
            else if (MODE == 7) atomicAdd((double *)s + (idx & 4095), 1.0);
            else if (MODE == 8 || MODE == 9) {
                unsigned *w = (unsigned *)s + idx;
                unsigned old = *(volatile unsigned *)w, assumed;
                do {
                    assumed = old;
                    old = atomicCAS(w, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f));
                } while (old != assumed);
            }
            else if (MODE == 10) { if ((tid & 7) == 0) atomicAdd(s + idx, 1.0f); }
            else if (MODE == 6) acc += atomicAdd(s + idx, 1.0f);
            else atomicAdd(s + idx, 1.0f);
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) out[tid] = s[tid] + acc;
}

template <int MODE>
void run(const char *name, float *out, const int *rnd)
{
    const int rounds = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, rounds, rnd);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, rounds, rnd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * rounds * 8;
    printf("%-58s %8.1f us  %8.1f G lane-ops/s  (%.2f lane-ops / clk / CU at 2.4 GHz)\n", name, ms * 1e3, lane_ops / ms / 1e6,
           lane_ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main()
{
    float *out;
    int *rnd, h[256];
    (void)hipMalloc(&out, 1024);
    (void)hipMalloc(&rnd, 1024);
    srand(1);
    for (int i = 0; i < 256; ++i) h[i] = rand();
    (void)hipMemcpy(rnd, h, 1024, hipMemcpyHostToDevice);
    run<0>("ds_add_f32 consecutive", out, rnd);
    run<1>("ds_add_f32 stride 32 (bank conflict)", out, rnd);
    run<2>("ds_add_f32 4 lanes per address", out, rnd);
    run<3>("ds_add_u32 consecutive", out, rnd);
    run<4>("ds_read + v_add + ds_write consecutive", out, rnd);
    run<5>("ds_add_f32 random in 1024", out, rnd);
    run<6>("ds_add_rtn_f32 consecutive", out, rnd);
    run<7>("ds_add_f64 consecutive", out, rnd);
    run<8>("fp32 add by CAS loop, consecutive", out, rnd);
    run<9>("fp32 add by CAS loop, 4 lanes per address", out, rnd);
    run<10>("ds_add_f32, 8 of 64 lanes active (lane-ops counted as 64)", out, rnd);
    return 0;
}
