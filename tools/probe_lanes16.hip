// Micro-probe for the "8 / 16 lanes per robot instance" layout (profiles/r2_lanes_head_to_head.md section 5).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lanes16.hip -o tools/_build/probe_lanes16 && tools/_build/probe_lanes16
// With 16 lanes per instance the rows of the 6 x 7 task Jacobian live in different lanes and meet through the only
// data-parallel primitive a 64-bit VALU instruction has on gfx950: DPP row_newbcast (one lane of a 16-lane row
// broadcast to the row), fused into v_fmac_f64 / v_mov_b64.  The probe measures
//   (1) what such an instruction costs a lone wave next to a plain v_fma_f64 (is the broadcast free?), and what
//       the two-instruction quad_perm move of a double costs (the four-lane kernel's primitive),
//   (2) the Gram build J J' (21 entries x 7 terms) in both layouts on the same data - one lane per instance:
//       147 v_fma_f64; 16 lanes per instance: 42 v_fmac_f64_dpp per lane (lane r owns row r of J and gets row r of
//       J J'), checked against each other,
//   (3) how both scale with the number of waves per SIMD: 16 lanes per instance need 16 x the lanes, i.e. at
//       16384 instances 4 waves per SIMD that share one fp64 pipe.
// Launch shape: blocks of 256 threads from a hipGraph of 1000 kernel nodes (what bench.py does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int K>
__device__ __forceinline__ void fmac_bcast(double& acc, const double from_lane_k, const double own)
{
    // acc += (value of `from_lane_k` in lane K of this 16-lane row) * own
    if constexpr (K == 0) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(from_lane_k), "v"(own));
    if constexpr (K == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(from_lane_k), "v"(own));
    if constexpr (K == 2) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(from_lane_k), "v"(own));
    if constexpr (K == 3) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(from_lane_k), "v"(own));
    if constexpr (K == 4) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(from_lane_k), "v"(own));
    if constexpr (K == 5) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(from_lane_k), "v"(own));
}

// synthetic 6 x 7 "Jacobian" of an instance from its 7 joint values (no trigonometry: the probe is about the products)
__device__ __forceinline__ double jac_entry(const double (&q)[7], const int r, const int j)
{
    return q[j] * (0.3 + 0.1 * r) + q[(j + r + 1) % 7] * 0.2 - 0.05 * (r - j);
}

// (2a) one lane per instance: the 21 entries of J J' (lower triangle), summed into one number per instance
__global__ __launch_bounds__(256) void gram_lane(const double* __restrict__ q, double* __restrict__ out, const long long B, const int reps)
{
    const long long inst = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = inst < B ? inst : B - 1;
    double qv[7], J[6][7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qv[j] = q[row * 7 + j];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int j = 0; j < 7; ++j) J[r][j] = jac_entry(qv, r, j);
    double total = 0.0;
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
        double G[21];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int k = 0; k <= r; ++k) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < 7; ++j) s = fma(J[r][j], J[k][j], s);
                G[r * (r + 1) / 2 + k] = s;
            }
#pragma unroll
        for (int e = 0; e < 21; ++e) total += G[e] * (1 + e);
        // (every entry moves with the result: nothing of the next repetition is loop invariant)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int j = 0; j < 7; ++j) J[r][j] = fma(1e-9, total, J[r][j]);
    }
    if (inst < B) out[inst] = total;
}

// (2b) 16 lanes per instance: lane r < 6 of the row owns row r of J and computes row r of J J'
__global__ __launch_bounds__(256) void gram_row16(const double* __restrict__ q, double* __restrict__ out, const long long B, const int reps)
{
    const int r = threadIdx.x & 15;
    const long long inst = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const long long row = inst < B ? inst : B - 1;
    double qv[7], Jr[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qv[j] = q[row * 7 + j];
#pragma unroll
    for (int j = 0; j < 7; ++j) Jr[j] = jac_entry(qv, r < 6 ? r : 5, j);
    double total = 0.0;
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
        double G[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            fmac_bcast<0>(G[0], Jr[j], Jr[j]);
            fmac_bcast<1>(G[1], Jr[j], Jr[j]);
            fmac_bcast<2>(G[2], Jr[j], Jr[j]);
            fmac_bcast<3>(G[3], Jr[j], Jr[j]);
            fmac_bcast<4>(G[4], Jr[j], Jr[j]);
            fmac_bcast<5>(G[5], Jr[j], Jr[j]);
        }
        // the same checksum as gram_lane: sum over the lower triangle of G[r][k] (1 + index); lane r holds row r
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) part += (k <= r && r < 6) ? G[k] * (1 + r * (r + 1) / 2 + k) : 0.0;
        // (row sum over the 6 lanes: not part of the Gram build, done with plain shuffles here)
        double sum = part;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) sum += __shfl_xor(sum, off, 16);
        total += sum;
#pragma unroll
        for (int j = 0; j < 7; ++j) Jr[j] = fma(1e-9, total, Jr[j]);
    }
    if (inst < B && r == 0) out[inst] = total;
}

// (1) instruction cost: N dependent-free fp64 operations of one kind per lane
template <int KIND>
__global__ __launch_bounds__(256) void issue(const double* __restrict__ q, double* __restrict__ out, const long long B, const int reps)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = q[(t * 7 + k) % (B * 7)];
    const double m = a[0] * 1e-3 + 0.999, c = a[1] * 1e-9;
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (KIND == 0) a[k] = fma(a[k], m, c);                            // v_fma_f64
                if constexpr (KIND == 1) fmac_bcast<3>(a[k], m, c);                        // v_fmac_f64_dpp row_newbcast
                if constexpr (KIND == 2) {                                                 // quad_perm move of a double + fma
                    int lo = __double2loint(a[k]), hi = __double2hiint(a[k]);
                    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true);
                    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
                    a[k] = fma(__hiloint2double(hi, lo), m, c);
                }
            }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[t % B] = s;
}

template <class F>
static int time_graph(const char* name, F launch, hipStream_t stream, int nodes, double* us_out)
{
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) launch();
    CHECK(hipStreamEndCapture(stream, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CHECK(hipGraphLaunch(exec, stream));
    CHECK(hipStreamSynchronize(stream));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, stream));
        CHECK(hipGraphLaunch(exec, stream));
        CHECK(hipEventRecord(e1, stream));
        CHECK(hipStreamSynchronize(stream));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    *us_out = 1e3 * best / nodes;
    printf("%-58s %8.3f us per launch\n", name, *us_out);
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
    return 0;
}

int main()
{
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    const long long BMAX = 65536;
    std::vector<double> hq(BMAX * 7);
    for (size_t i = 0; i < hq.size(); ++i) hq[i] = 0.3 * std::sin(0.37 * (double)i) + 0.1;
    double *q, *o1, *o2;
    CHECK(hipMalloc(&q, hq.size() * 8));
    CHECK(hipMalloc(&o1, BMAX * 16 * 8));
    CHECK(hipMalloc(&o2, BMAX * 16 * 8));
    CHECK(hipMemcpy(q, hq.data(), hq.size() * 8, hipMemcpyHostToDevice));
    // correctness of the 16-lane Gram against the lane-per-instance one
    {
        const long long B = 4096;
        hipLaunchKernelGGL(gram_lane, dim3((unsigned)(B / 256)), dim3(256), 0, stream, q, o1, B, 3);
        hipLaunchKernelGGL(gram_row16, dim3((unsigned)(B * 16 / 256)), dim3(256), 0, stream, q, o2, B, 3);
        CHECK(hipStreamSynchronize(stream));
        std::vector<double> a(B), b(B);
        CHECK(hipMemcpy(a.data(), o1, B * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), o2, B * 8, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (long long i = 0; i < B; ++i) worst = std::fmax(worst, std::fabs(a[i] - b[i]) / (1.0 + std::fabs(a[i])));
        printf("Gram build, 16 lanes per instance vs 1: max relative difference %.2e over %lld instances\n", worst, B);
    }
    double us;
    const int R = 64;           // repetitions inside a launch (the difference of two R gives the cost per repetition)
    printf("\n(1) issue cost, 256 blocks x 256 threads = one wave per SIMD; 64 x R operations per lane\n");
    double base[3], more[3];
    const char* kinds[3] = {"v_fma_f64", "v_fmac_f64_dpp row_newbcast", "2 x v_mov_b32 quad_perm + v_fma_f64"};
    for (int k = 0; k < 3; ++k) {
        for (int pass = 0; pass < 2; ++pass) {
            const int reps = pass ? 2 * R : R;
            char name[128];
            snprintf(name, sizeof name, "  %s, %d operations", kinds[k], 64 * reps);
            auto go = [&]() {
                if (k == 0) hipLaunchKernelGGL((issue<0>), dim3(256), dim3(256), 0, stream, q, o1, BMAX, reps);
                if (k == 1) hipLaunchKernelGGL((issue<1>), dim3(256), dim3(256), 0, stream, q, o1, BMAX, reps);
                if (k == 2) hipLaunchKernelGGL((issue<2>), dim3(256), dim3(256), 0, stream, q, o1, BMAX, reps);
            };
            if (time_graph(name, go, stream, 500, &us)) return 1;
            (pass ? more : base)[k] = us;
        }
        printf("  -> %.3f ns per operation of a lone wave\n", 1e3 * (more[k] - base[k]) / (64.0 * R));
    }
    printf("\n(2,3) Gram build J J' of R = %d repetitions per launch, cost per repetition = (t(2R) - t(R)) / R\n", R);
    for (long long B : {4096LL, 16384LL, 32768LL}) {
        double t1[2], t16[2];
        for (int pass = 0; pass < 2; ++pass) {
            const int reps = pass ? 2 * R : R;
            char name[128];
            snprintf(name, sizeof name, "  %lld instances, 1 lane per instance (%lld waves), %d reps", B, B / 64, reps);
            if (time_graph(name, [&]() { hipLaunchKernelGGL(gram_lane, dim3((unsigned)(B / 256)), dim3(256), 0, stream, q, o1, B, reps); }, stream, 300, &t1[pass])) return 1;
            snprintf(name, sizeof name, "  %lld instances, 16 lanes per instance (%lld waves), %d reps", B, B * 16 / 64, reps);
            if (time_graph(name, [&]() { hipLaunchKernelGGL(gram_row16, dim3((unsigned)(B * 16 / 256)), dim3(256), 0, stream, q, o2, B, reps); }, stream, 300, &t16[pass])) return 1;
        }
        printf("  -> %lld instances: Gram build per repetition %.3f us with 1 lane per instance, %.3f us with 16 lanes per instance\n",
               B, (t1[1] - t1[0]) / R, (t16[1] - t16[0]) / R);
    }
    return 0;
}
