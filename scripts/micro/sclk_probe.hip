// Shader clock seen by a kernel that occupies ONE workgroup (the serial tails: nms_scan, proposal sort, Sinkhorn run on
// a handful of CUs): s_memtime (shader clock) against s_memrealtime (100 MHz) over a ~200 us spin, first launch after
// idle and steady state.   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/sclk_probe scripts/micro/sclk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long *out, int iters)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    int x = threadIdx.x;
    for (int i = 0; i < iters; ++i) { x = x * 3 + 1; asm volatile("" : "+v"(x)); }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = c1 - c0; out[blockIdx.x * 3 + 1] = w1 - w0; out[blockIdx.x * 3 + 2] = x; }
}
int main()
{
    long long *d, h[3 * 1024];
    hipMalloc(&d, sizeof(h));
    for (int wgs : {1, 4, 256, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, 0, d, 100000);
            hipMemcpy(h, d, sizeof(long long) * 3 * wgs, hipMemcpyDeviceToHost);
            printf("workgroups %4d rep %d: %lld shader clocks in %lld ticks of 100 MHz -> %.0f MHz; %.2f clocks per dependent multiply-add\n",
                   wgs, rep, h[0], h[1], (double)h[0] / (double)h[1] * 100.0, (double)h[0] / 100000.0);
        }
    }
    return 0;
}
