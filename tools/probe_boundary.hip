// Micro-probe: the fixed cost of one tick-shaped launch on MI355X (no CLIK arithmetic).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_boundary.hip -o tools/_build/probe_boundary && tools/_build/probe_boundary
// Launches, back to back on one stream from a hipGraph of 1000 kernel nodes (what bench.py does),
// grid = 256 blocks x 256 threads (the team kernel's launch at 16384 instances):
//   empty      nothing
//   rw         read the block's q / y rows, write dq (the algorithmic traffic of a tick, no arithmetic)
//   rw_img     ... plus the 13 KiB skill image global -> LDS, barrier, LDS -> registers
//   rw_img_fN  ... plus a chain of N dependent fp64 FMAs per lane (N = 500, 1000, 2000)
// The differences give: kernel boundary, memory round trip, LDS staging, and ns per issued fp64 instruction
// of a lone wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_empty(const double*, const double*, double*, const double*, int) {}

template <int IMG, int NF>
__global__ __launch_bounds__(256) void k_rw(const double* __restrict__ q, const double* __restrict__ y,
                                            double* __restrict__ dq, const double* __restrict__ img, int nf_rt)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * 64 * 7;
    double a = q[b0 + tid], b = y[b0 + tid];
    double c = 0.0, d = 0.0;
    if (tid < 192) { c = q[b0 + 256 + tid]; d = y[b0 + 256 + tid]; }
    double acc = a + b + c + d;
    if constexpr (IMG) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* src = (const d2*)img;
        d2* dst = (d2*)lds;
        d2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = src[k * 256 + tid];
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * 256 + tid] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 48; ++k) acc += lds[(k * 37) & 2047];      // wave-uniform reads of the "image"
    }
    if constexpr (NF > 0) {
        double x = acc;
#pragma unroll 8
        for (int i = 0; i < NF; ++i) x = fma(x, 0.999999, 1e-9);
        acc = x;
    }
    if (nf_rt == 12345) acc += 1.0;
    dq[b0 + tid] = acc;
    if (tid < 192) dq[b0 + 256 + tid] = acc;
}

// straight-line code of the same length as a tick (no loop): does instruction fetch cost anything beyond
// the 4 clocks of issue?  NI independent-ish fp64 FMAs over 8 accumulators, fully unrolled, against the
// same work in a loop of 64 (which stays in the instruction cache / buffer).
template <int NI, bool UNROLLED>
__global__ __launch_bounds__(256) void k_code(const double* __restrict__ q, const double* __restrict__ y,
                                              double* __restrict__ dq, const double* __restrict__ img, int nf_rt)
{
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * 64 * 7;
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = q[b0 + tid] + k;
    const double m = y[b0 + tid] + 0.999999, c = y[b0 + tid + 1] + 1e-9;      // run-time operands: no literals
    if constexpr (UNROLLED) {
#pragma unroll
        for (int i = 0; i < NI / 8; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = fma(a[k], m, c);
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < NI / 64; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = fma(a[k], m, c);
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += a[k];
    if (nf_rt == 12345) acc += 1.0;
    dq[b0 + tid] = acc;
}

typedef void (*kern_t)(const double*, const double*, double*, const double*, int);

int main()
{
    const int B = 16384, grid = B / 64, K = 1000;
    double *q, *y, *dq, *img;
    CHECK(hipMalloc(&q, B * 7 * 8));
    CHECK(hipMalloc(&y, B * 7 * 8));
    CHECK(hipMalloc(&dq, B * 7 * 8));
    CHECK(hipMalloc(&img, 16384));
    CHECK(hipMemset(q, 0, B * 7 * 8));
    CHECK(hipMemset(y, 0, B * 7 * 8));
    CHECK(hipMemset(img, 0, 16384));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    struct Case { const char* name; kern_t fn; size_t shmem; };
    std::vector<Case> cases = {
        {"empty", k_empty, 0},
        {"rw", k_rw<0, 0>, 0},
        {"rw_img", k_rw<1, 0>, 16384},
        {"rw_img_f500", k_rw<1, 500>, 16384},
        {"rw_img_f1000", k_rw<1, 1000>, 16384},
        {"rw_img_f2000", k_rw<1, 2000>, 16384},
        {"code1024_loop", k_code<1024, false>, 0},
        {"code1024_flat", k_code<1024, true>, 0},
        {"code2048_loop", k_code<2048, false>, 0},
        {"code2048_flat", k_code<2048, true>, 0},
        {"code4096_loop", k_code<4096, false>, 0},
        {"code4096_flat", k_code<4096, true>, 0},
    };
    for (auto& c : cases) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), c.shmem, s, q, y, dq, img, 0);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 100; ++r) CHECK(hipGraphLaunch(ge, s));       // clock ramp
        CHECK(hipStreamSynchronize(s));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a, s));
        for (int r = 0; r < 50; ++r) CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(b, s));
        CHECK(hipStreamSynchronize(s));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-14s %7.3f us per launch\n", c.name, ms * 1e3 / (50.0 * K));
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
