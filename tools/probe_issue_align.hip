// Micro-probe (round 5): does the ADDRESS of an instruction decide how fast a lone wave issues it?
// profiles/r4_fp64_issue_probe.txt has the same `v_fma_f64` stream at 4.2 or at 5.2 cycles per instruction depending on
// what was placed in front of the loop (a 4-byte `v_mov_b32` shifts every 8-byte VOP3 that follows to an address that
// is 4 mod 8).  This probe lays out streams by hand:
//   * 8-byte fp64 FMAs starting at a 64-byte boundary + PAD x 4 bytes (PAD = 0 ... 15),
//   * mixed streams (a 4-byte VALU every N FMAs) arranged so that an 8-byte instruction straddles only 32-byte
//     boundaries, only 64-byte boundaries, both, or none,
//   * the same with two waves per SIMD (does a second wave hide it?).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_issue_align.hip -o tools/_build/probe_issue_align
// One wave per SIMD (1024 blocks of 64; 40 KiB of LDS per block keeps it at one per SIMD), wave 0 reads the shader clock
// around ITERS trips of a loop whose body is the laid-out stream; cycles per INSTRUCTION = clocks / (ITERS x instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define FMA "v_fma_f64 v[16:17], v[18:19], v[20:21], v[16:17]\n\t"
#define FMA2 "v_fma_f64 v[22:23], v[18:19], v[20:21], v[22:23]\n\t"
#define MOV4 "v_mov_b32_e32 v24, v25\n\t"          /* a 4-byte VALU */
#define MOV8 "v_mov_b32_e64 v24, v25\n\t"          /* the same instruction in its 8-byte encoding */
#define SNOP "s_nop 0\n\t"                          /* a 4-byte scalar */
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)
#define R64(x) R32(x) R32(x)
#define CLOB : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25"

// LAYOUT: which stream the loop body holds; every body starts at a 64-byte boundary (.p2align 6 pads with s_nop, executed
// once per trip - the bodies are long enough not to see it)
template <int LAYOUT>
__device__ __forceinline__ void body()
{
    // 0 ... 15: PAD x s_nop, then 256 FMAs
    if constexpr (LAYOUT == 0)  asm volatile(".p2align 6\n\t" R64(R4(FMA)) CLOB);
    if constexpr (LAYOUT == 1)  asm volatile(".p2align 6\n\t" SNOP R64(R4(FMA)) CLOB);
    if constexpr (LAYOUT == 2)  asm volatile(".p2align 6\n\t" SNOP SNOP R64(R4(FMA)) CLOB);
    if constexpr (LAYOUT == 3)  asm volatile(".p2align 6\n\t" SNOP SNOP SNOP R64(R4(FMA)) CLOB);
    // 20: per 64 bytes [3 FMA, MOV4, FMA (straddles the 32-byte boundary), 3 FMA, MOV4]: 7 FMA + 2 MOV4, 32-byte straddles only
    if constexpr (LAYOUT == 20) asm volatile(".p2align 6\n\t" R32(FMA FMA FMA MOV4 FMA FMA FMA FMA MOV4) CLOB);
    // 21: the same instructions, none straddling: [3 FMA, MOV4, MOV4, 4 FMA]
    if constexpr (LAYOUT == 21) asm volatile(".p2align 6\n\t" R32(FMA FMA FMA MOV4 MOV4 FMA FMA FMA FMA) CLOB);
    // 22: per 128 bytes, 64-byte straddles only: [MOV4, 3 FMA (4..28), MOV4 (28..32), 3 FMA (32..56), MOV4 (56..60), FMA (60..68
    //     straddles 64), 3 FMA (68..92), MOV4 (92..96), 4 FMA (96..128)]: 14 FMA + 4 MOV4
    if constexpr (LAYOUT == 22) asm volatile(".p2align 6\n\t" R16(MOV4 FMA FMA FMA MOV4 FMA FMA FMA MOV4 FMA FMA FMA FMA MOV4 FMA FMA FMA FMA) CLOB);
    // 23: the same 14 FMA + 4 MOV4 per 128 bytes, none straddling
    if constexpr (LAYOUT == 23) asm volatile(".p2align 6\n\t" R16(MOV4 MOV4 FMA FMA FMA MOV4 MOV4 FMA FMA FMA FMA FMA FMA FMA FMA FMA FMA FMA) CLOB);
    // 24: every FMA at 4 mod 8 but NO straddle of a 32-byte boundary: per 32 bytes [MOV4, 3 FMA (4..28), MOV4]
    if constexpr (LAYOUT == 24) asm volatile(".p2align 6\n\t" R64(MOV4 FMA FMA FMA MOV4) CLOB);
    // 25: the same mix aligned: [MOV4 MOV4 3 FMA]
    if constexpr (LAYOUT == 25) asm volatile(".p2align 6\n\t" R64(MOV4 MOV4 FMA FMA FMA) CLOB);
    // 26: the fix an assembler pass would apply to 20: the MOV4 in front of the straddling FMA in its 8-byte form
    //     [3 FMA, MOV8 (24..32), 4 FMA (32..64)] then [3 FMA MOV4 MOV4 ...]: here simply 3 FMA + MOV8 + 4 FMA per 64 bytes
    if constexpr (LAYOUT == 26) asm volatile(".p2align 6\n\t" R32(FMA FMA FMA MOV8 FMA FMA FMA FMA) CLOB);
    // 27: 3 FMA + MOV4 + 4 FMA per 60 bytes: the phase drifts, straddles come and go (a compiler's stream)
    if constexpr (LAYOUT == 27) asm volatile(".p2align 6\n\t" R32(FMA FMA FMA MOV4 FMA FMA FMA FMA) CLOB);
    // 30 / 31: two independent accumulators alternate (is it the dependence on v16 that costs, not the address?)
    if constexpr (LAYOUT == 30) asm volatile(".p2align 6\n\t" R64(R2(FMA FMA2)) CLOB);
    if constexpr (LAYOUT == 31) asm volatile(".p2align 6\n\t" SNOP R64(R2(FMA FMA2)) CLOB);
    // 40 / 41: 4-byte instructions only (v_mov_b32_e32): 256 of them, aligned / they cannot straddle
    if constexpr (LAYOUT == 40) asm volatile(".p2align 6\n\t" R64(R4(MOV4)) CLOB);
    // 42: 8-byte encodings of the same move at 4 mod 8
    if constexpr (LAYOUT == 42) asm volatile(".p2align 6\n\t" SNOP R64(R4(MOV8)) CLOB);
    if constexpr (LAYOUT == 43) asm volatile(".p2align 6\n\t" R64(R4(MOV8)) CLOB);
}

template <int LAYOUT>
__global__ __launch_bounds__(64) void k(double* __restrict__ out, unsigned long long* __restrict__ clk, int iters, double seed)
{
    extern __shared__ double lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = seed;
    asm volatile("v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\t"
                 "v_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\tv_mov_b32 v24, 0\n\tv_mov_b32 v25, 0" CLOB);
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) body<LAYOUT>();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
    if (iters < 0) out[threadIdx.x] = lds_pad[threadIdx.x];
}

template <int LAYOUT>
int run(const char* what, int n_instr, int waves_per_simd, double* out, unsigned long long* clk)
{
    const int iters = 400;
    const size_t lds = (size_t)(160 * 1024) / (4 * waves_per_simd);
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)(k<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<LAYOUT>), dim3(1024 * waves_per_simd), dim3(64), lds, 0, out, clk, iters, 1.0);
    CHECK(hipDeviceSynchronize());
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL((k<LAYOUT>), dim3(1024 * waves_per_simd), dim3(64), lds, 0, out, clk, iters, 1.0);
        CHECK(hipDeviceSynchronize());
        unsigned long long h; CHECK(hipMemcpy(&h, clk, sizeof(h), hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    printf("%-98s waves/SIMD %d  %6.3f cycles per instruction  (%d instructions per trip)\n", what, waves_per_simd,
           (double)best / ((double)iters * n_instr), n_instr);
    return 0;
}

int main()
{
    double* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&clk, 64));
    for (int wps : {1, 2}) {
        if (run<0>("256 FMA (8-byte) from a 64-byte boundary", 256, wps, out, clk)) return 1;
        if (run<1>("256 FMA from boundary + 4: every FMA at 4 mod 8", 257, wps, out, clk)) return 1;
        if (run<2>("256 FMA from boundary + 8", 258, wps, out, clk)) return 1;
        if (run<3>("256 FMA from boundary + 12", 259, wps, out, clk)) return 1;
        if (run<30>("two accumulators alternating, aligned", 256, wps, out, clk)) return 1;
        if (run<31>("two accumulators alternating, at 4 mod 8", 257, wps, out, clk)) return 1;
        if (run<20>("7 FMA + 2 MOV4 per 64 B, one FMA straddles each 32-byte boundary (not the 64)", 288, wps, out, clk)) return 1;
        if (run<21>("7 FMA + 2 MOV4 per 64 B, no straddle", 288, wps, out, clk)) return 1;
        if (run<22>("14 FMA + 4 MOV4 per 128 B, one FMA straddles each 64-byte boundary only", 288, wps, out, clk)) return 1;
        if (run<23>("14 FMA + 4 MOV4 per 128 B, no straddle", 288, wps, out, clk)) return 1;
        if (run<24>("3 FMA at 4 mod 8 + 2 MOV4 per 32 B, no straddle", 320, wps, out, clk)) return 1;
        if (run<25>("3 FMA aligned + 2 MOV4 per 32 B", 320, wps, out, clk)) return 1;
        if (run<26>("3 FMA + MOV8 + 4 FMA per 64 B (the 4-byte move widened: no straddle)", 256, wps, out, clk)) return 1;
        if (run<27>("3 FMA + MOV4 + 4 FMA per 60 B (phase drifts)", 256, wps, out, clk)) return 1;
        if (run<40>("256 MOV4 (4-byte VALU)", 256, wps, out, clk)) return 1;
        if (run<43>("256 MOV8 (8-byte encoding), aligned", 256, wps, out, clk)) return 1;
        if (run<42>("256 MOV8 at 4 mod 8", 257, wps, out, clk)) return 1;
    }
    return 0;
}
