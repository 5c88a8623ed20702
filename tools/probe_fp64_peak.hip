// Micro-probe: what the fp64 vector pipe of the MI355X sustains, and at which clock - the ceiling the lane-per-instance
// kernels are priced against (MI355X_MICROARCH.md quotes 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16
// lanes x 2 flop x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_fp64_peak.hip -o tools/_build/probe_fp64_peak && tools/_build/probe_fp64_peak
// Every wave runs ITER x 16 v_fma_f64 over 16 independent accumulators (no memory in the loop); launches of
// waves_per_simd x 1024 waves; s_memtime (shader clock) and s_memrealtime (100 MHz) of the first wave of block 0 give the
// clock under that load; the launch time by hipEvents gives the rate.  Durations: about 70 us (a 1 M-instance tick) and
// about 2 ms (sustained).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MIX, int REGS = 0>
__global__ __launch_bounds__(64) void k_fma(double* __restrict__ out, unsigned long long* __restrict__ clk, int iters, double seed)
{
    extern __shared__ double lds_pad[];     // (dynamic LDS only bounds how many of these waves a CU takes)
    if (iters < 0) lds_pad[threadIdx.x] = seed;
    // REGS: allocate registers as the tick kernels do (1: 248 VGPRs, two waves fit a SIMD; 2: 256 + AGPRs, one wave) -
    // unlike the LDS bound this also decides WHICH SIMD takes the next wave
    if constexpr (REGS == 1) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    if constexpr (REGS == 2) asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a63, v255" ::: "v255", "a63");
    if constexpr (REGS == 3) asm volatile("v_mov_b32 v255, 0" ::: "v255");     // (allocation sizes: does the size alone cost issue rate?)
    if constexpr (REGS == 4) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (REGS == 5) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (REGS == 6) asm volatile("v_mov_b32 v217, 0" ::: "v217");
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = seed + k + threadIdx.x;
    const double m = 0.9999999, c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    // (iters counts groups of 16 instructions; 16 groups = 256 instructions per trip of the loop, so that the branch
    // and its fetch bubble - some 30 cycles - do not show: with 16 per trip a lone wave reads 6.0 cycles per fma)
#pragma unroll 1
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
        for (int kk = 0; kk < 256; ++kk) {
            const int k = kk & 15;
            if constexpr (MIX == 0) a[k] = fma(a[k], m, c);
            else if constexpr (MIX == 1) a[k] = (k & 3) == 3 ? a[k] * m : fma(a[k], m, c);   // 1 mul in 4
            else if constexpr (MIX == 2) a[k] = fma(a[k], a[(k + 1) & 15], c);              // two register operands
            else if constexpr (MIX == 4) a[0] = fma(a[0], a[1], a[2]);                      // one dependent chain
            else if constexpr (MIX == 5) a[k & 1] = fma(a[k & 1], a[2], a[3]);              // two chains
            else if constexpr (MIX == 6) a[k & 3] = fma(a[k & 3], a[4], a[5]);              // four chains
            else a[k] = fma(a[k], a[(k + 1) & 15], a[(k + 2) & 15] * 1e-30);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MIX, int REGS = 0>
int run(const char* what, int waves_per_simd, int iters, double* out, unsigned long long* clk, int resident_per_simd = 0)
{
    const int grid = waves_per_simd * 1024;
    iters = (iters + 15) / 16 * 16;
    // resident_per_simd > 0: each block claims 160 KiB / (4 x that) of LDS, so a CU holds exactly 4 x that many waves
    // (what 248 VGPRs do to the lane kernels: two per SIMD)
    const size_t lds = resident_per_simd > 0 ? (size_t)(160 * 1024) / (4 * resident_per_simd) : 0;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)(k_fma<MIX, REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_fma<MIX, REGS>), dim3(grid), dim3(64), lds, 0, out, clk, iters, 1.0);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_fma<MIX, REGS>), dim3(grid), dim3(64), lds, 0, out, clk, iters, 1.0);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / reps;
    const double insts = (double)grid * iters * 16;                      // wave-instructions
    const double flops = insts * 64 * (MIX == 1 ? 1.75 : 2.0);
    char label[96];
    snprintf(label, sizeof(label), "%s%s", what, REGS == 1 ? " [248 VGPRs]" : REGS == 2 ? " [256 VGPRs + 64 AGPRs]" : REGS == 3 ? " [256 VGPRs]" : REGS == 4 ? " [128 VGPRs]" : REGS == 5 ? " [64 VGPRs]" : REGS == 6 ? " [218 VGPRs]" : resident_per_simd == 1 ? " [LDS: 1 resident/SIMD]" : resident_per_simd == 2 ? " [LDS: 2 resident/SIMD]" : "");
    what = label;
    printf("%-52s waves/SIMD %2d iters %6d  %9.2f us/launch  %6.2f TFLOP/s  %5.2f cycles per wave-instruction per SIMD at 2.4 GHz;"
           "  wave 0: %llu shader clocks in %.2f us = %.3f GHz\n", what, waves_per_simd, iters, us, flops / us * 1e-6,
           us * 1e-6 * 2.4e9 / (insts / 1024), h[0], h[1] / 100.0, h[0] / (h[1] / 100.0) * 1e-3);
    return 0;
}

int main()
{
    double* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)16 * 1024 * 64 * sizeof(double)));
    CHECK(hipMalloc(&clk, 64));
    for (int wps : {1, 2, 4, 8}) {
        if (run<0>("independent fma, short", wps, 1100 / wps, out, clk)) return 1;      // ~ 17600 instructions per SIMD
        if (run<0>("independent fma, sustained", wps, 40000 / wps, out, clk)) return 1;
    }
    if (run<1>("3 fma : 1 mul, sustained", 2, 20000, out, clk)) return 1;
    if (run<2>("fma chained over accumulators", 2, 20000, out, clk)) return 1;
    if (run<2>("fma chained over accumulators", 1, 40000, out, clk)) return 1;
    // occupancy pinned as registers pin it for the tick kernels
    if (run<0>("independent fma, sustained", 1, 40000, out, clk, 1)) return 1;
    if (run<0>("independent fma, sustained", 2, 20000, out, clk, 2)) return 1;
    if (run<2>("fma, two register operands, sustained", 1, 40000, out, clk, 1)) return 1;
    if (run<2>("fma, two register operands, sustained", 2, 20000, out, clk, 2)) return 1;
    if (run<1>("3 fma : 1 mul, sustained", 2, 20000, out, clk, 2)) return 1;
    // the shape of a 16384-instance team tick: one wave per SIMD, 1206 instructions
    if (run<0>("1 wave per SIMD, 1280 instr", 1, 80, out, clk, 1)) return 1;
    if (run<2>("1 wave per SIMD, 1280 instr, two reg operands", 1, 80, out, clk, 1)) return 1;
    // the shape of a 131072-instance lane tick: two waves per SIMD, 1760 instructions each
    if (run<0>("2 waves per SIMD, 1792 instr", 2, 112, out, clk, 2)) return 1;
    if (run<2>("2 waves per SIMD, 1792 instr, two reg operands", 2, 112, out, clk, 2)) return 1;
    // the shape of a 1 M-instance lane tick: 16 waves per SIMD in turn, 1760 instructions each
    if (run<0>("16 waves per SIMD in turn, 1792 instr each", 16, 112, out, clk)) return 1;
    if (run<0>("16 waves per SIMD in turn, 1792 instr each", 16, 112, out, clk, 2)) return 1;
    if (run<2>("16 waves per SIMD in turn, 1792 instr, two reg operands", 16, 112, out, clk, 2)) return 1;
    // ... with the occupancy set by registers
    if ((run<2, 2>("1 wave per SIMD, 1280 instr, two reg operands", 1, 80, out, clk))) return 1;
    if ((run<2, 1>("2 waves per SIMD, 1792 instr, two reg operands", 2, 112, out, clk))) return 1;
    if ((run<0, 1>("2 waves per SIMD, 1792 instr", 2, 112, out, clk))) return 1;
    if ((run<2, 1>("16 waves per SIMD in turn, 1792 instr, two reg operands", 16, 112, out, clk))) return 1;
    if ((run<0, 1>("16 waves per SIMD in turn, 1792 instr", 16, 112, out, clk))) return 1;
    if ((run<1, 1>("16 waves per SIMD in turn, 1792 instr, 3 fma : 1 mul", 16, 112, out, clk))) return 1;
    if ((run<2, 1>("sustained, two reg operands", 2, 20000, out, clk))) return 1;
    if ((run<2, 2>("sustained, two reg operands", 1, 40000, out, clk))) return 1;
    if ((run<2, 5>("ONE wave per SIMD, sustained, two reg operands", 1, 40000, out, clk))) return 1;
    if ((run<2, 4>("ONE wave per SIMD, sustained, two reg operands", 1, 40000, out, clk))) return 1;
    if ((run<2, 6>("ONE wave per SIMD, sustained, two reg operands", 1, 40000, out, clk))) return 1;
    if ((run<2, 1>("ONE wave per SIMD, sustained, two reg operands", 1, 40000, out, clk))) return 1;
    if ((run<2, 3>("ONE wave per SIMD, sustained, two reg operands", 1, 40000, out, clk))) return 1;
    if ((run<0, 5>("ONE wave per SIMD, sustained, constant operands", 1, 40000, out, clk))) return 1;
    if ((run<0, 3>("ONE wave per SIMD, sustained, constant operands", 1, 40000, out, clk))) return 1;
    if ((run<4, 2>("ONE dependent chain, 1 wave per SIMD", 1, 4000, out, clk))) return 1;
    if ((run<4, 5>("ONE dependent chain, 1 wave per SIMD", 1, 4000, out, clk))) return 1;
    if ((run<5, 2>("TWO dependent chains, 1 wave per SIMD", 1, 4000, out, clk))) return 1;
    if ((run<6, 2>("FOUR dependent chains, 1 wave per SIMD", 1, 4000, out, clk))) return 1;
    return 0;
}
