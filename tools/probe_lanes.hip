// Micro-probe: does a wave64 fp64 VALU instruction get cheaper when only 16 / 32 lanes are active?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, unsigned long long* cyc, int active, int iters)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-3, b = 0.999, c = 1e-9, d = 2.0 + lane, e = 0.5, f = 3.0;
    unsigned long long t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a = fma(a, b, c);
                d = fma(d, b, c);
                e = fma(e, b, c);
                f = fma(f, b, c);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 64 + lane] = a + d + e + f;
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    }
}
int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 1024 * 8); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    for (int active : {64, 48, 32, 16, 8, 1}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, out, cyc, active, iters);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("active lanes %2d: %.2f cycles per fp64 FMA instruction (4 independent chains)\n", active,
               (double)s / 256 / (iters * 64.0));
    }
    return 0;
}
