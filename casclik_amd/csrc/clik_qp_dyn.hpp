// Batched ReactiveQPController tick on gfx950.
//
// Replaces, for B instances per launch, the per-tick body of
//   ReactiveQPController.solve            casclik/controllers/reactive_qp.py:461-528
// i.e. the H/A/lbA/ubA functions (:175-246, :262-298) and the qpOASES call
// through cs.conic (:248-260, :491-513).
//
// The QP   min 1/2 v'Hv   s.t.  lbA <= A v <= ubA,   H = diag(h) > 0,
// v = [robot_vel; virtual_vel; slack]  has a unique minimiser, so the solver is
// free: each lane runs an exact dual active-set method (Goldfarb & Idnani 1983)
// written in CONSTRAINT space.  With nu the signed multipliers,
//      v = H^-1 A' nu,      c = A v = Q nu,      Q = A H^-1 A'   (nc x nc, SPD-ish)
// so the iteration only needs Q (kept in LDS, per lane), nu and c; every change
// of the working set re-factors the masked Schur matrix  S_W = D Q_WW D  by a
// fixed-size LDL^T in registers (no updates/downdates, no drift, the same
// instruction stream for every lane; lanes differ only in masks).
//   - slack columns never materialise: a soft row i adds 1/h_slack,i to Q_ii
//     and its slack is  -nu_i / h_slack,i
//   - rows with lbA == ubA are equalities: once active they stay, their
//     multiplier is sign-free
//   - infeasible problems (hard rows only) are reported per instance, like the
//     reference's RuntimeError from qpOASES
//
// This header holds the dynamic-shape kernels and their launch templates; the instantiations are spread over
// clik_qp_dyn_[a-d].hip (explicit instantiation definitions: four translation units that compile side by side - as one
// unit they were a 12-minute compile) and clik_qp.hip holds the variant table (extern template declarations).
#pragma once
#include "clik_qp_static.hpp"
#include "clik_workspace.hpp"
#include <map>
#include <mutex>
#include <utility>

namespace clik {

// LDS slots per lane: [zs N][ys ny][A rows NC*N (FK frames alias)][Q NC(NC+1)/2][lb NC][ub NC][hinv NC]
template <int N, int NC>
__host__ __device__ constexpr int qp_lds_slots(int ny)
{
    return N + ny + (NC * N > 6 * N ? NC * N : 6 * N) + NC * (NC + 1) / 2 + 3 * NC;
}

// Evaluate every constraint row of the QP for the lane's instance:
//   A_u rows -> As[row*N + j], bounds -> lbs/ubs, slack curvature 1/h_slack (0 = hard) -> hsi
// Returns the number of rows (wave-uniform).  reactive_qp.py:191-246.
template <int N>
__device__ __forceinline__ int qp_rows(const DevSkill* __restrict__ S, const TickArgs& tk, const Kin<N>& K,
                                       const double (&z)[N], const double* ys, const int lane, const int n,
                                       double* As, double* lbs, double* ubs, double* hsi)
{
    const clik_skill_desc& D = S->d;
    int row = 0, slack = 0;
    const double mu = S->qo.weight_shifter;
    for (int ti = 0; ti < D.n_tasks; ++ti) {
        const clik_task& t = D.tasks[ti];
        const int m = t.m;
        double e[N], J[N][N], Jt[N];
        task_eval<N, N>(S, ti, m, -1, false, tk, K, z, ys, lane, n, e, J, Jt);
        double lo[N], hi[N];
        const int cls = t.cls;
        if (cls == CLIK_CLS_EQ) {
            double ke[N];
            gain_apply<N>(t, m, e, ke);
#pragma unroll
            for (int i = 0; i < N; ++i) lo[i] = hi[i] = -Jt[i] - ke[i];
        } else if (cls == CLIK_CLS_SET) {
            double d0[N], g[N];
#pragma unroll
            for (int i = 0; i < N; ++i) d0[i] = (i < m) ? t.set_min[i] - e[i] : 0.0;
            gain_apply<N>(t, m, d0, g);
#pragma unroll
            for (int i = 0; i < N; ++i) lo[i] = -Jt[i] + g[i];
#pragma unroll
            for (int i = 0; i < N; ++i) d0[i] = (i < m) ? t.set_max[i] - e[i] : 0.0;
            gain_apply<N>(t, m, d0, g);
#pragma unroll
            for (int i = 0; i < N; ++i) hi[i] = -Jt[i] + g[i];
        } else if (cls == CLIK_CLS_VELEQ) {
#pragma unroll
            for (int i = 0; i < N; ++i) lo[i] = hi[i] = (i < m) ? t.target[i] - Jt[i] : 0.0;
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                lo[i] = (i < m) ? t.set_min[i] - Jt[i] : 0.0;
                hi[i] = (i < m) ? t.set_max[i] - Jt[i] : 0.0;
            }
        }
        const bool soft = t.soft != 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < m) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j < n) As[((row + i) * N + j) * WAVE + lane] = J[i][j];
                lbs[(row + i) * WAVE + lane] = lo[i];
                ubs[(row + i) * WAVE + lane] = hi[i];
                hsi[(row + i) * WAVE + lane] = soft ? 1.0 / (mu + S->qo.slack_weights[slack + i]) : 0.0;
            }
        }
        if (soft) slack += m;
        row += m;
    }
    return row;
}

// GWS: the per-wave work area (state, rows, dual Hessian, bounds: qp_lds_slots doubles per lane) lies in GLOBAL memory
// handed in by the launch (`gws`, one area per block) instead of LDS - the variants for QPs with more than 16 rows,
// whose work area (up to 490 KB per wave at 42 x 32) no CU holds.  Same code, slower memory: such skills are served,
// not refused (the reference puts no bound on the number of constraints, reactive_qp.py:191-246).
template <int N, int NC, bool EXACT, bool GWS = false>
__global__ __launch_bounds__(WAVE) void qp_solve_kernel(
    const DevSkill* __restrict__ S0, const WarmArgs wa, const TickArgs tk, const long long B,
    const double* __restrict__ q, const double* __restrict__ x, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ dx, double* __restrict__ slack_out,
    int32_t* __restrict__ status_out, double* __restrict__ gws = nullptr, const int ny_slots = 0)
{
    extern __shared__ double lds_shared[];
    double* lds = GWS ? gws + (size_t)blockIdx.x * (size_t)qp_lds_slots<N, NC>(ny_slots) * WAVE : lds_shared;
    constexpr int NT = NC * (NC + 1) / 2;
    const int lane = threadIdx.x;
    const DevSkill* __restrict__ S = warm_descriptor(S0, wa);
    const int n = S->n, nq = S->d.n_q, nx = S->d.n_x, ny = S->d.n_y;
    double* zs = lds;
    double* ys = zs + N * WAVE;
    double* As = ys + ny * WAVE;
    double* Qs = As + (NC * N > 6 * N ? NC * N : 6 * N) * WAVE;
    double* lbs = Qs + NT * WAVE;
    double* ubs = lbs + NC * WAVE;
    double* hsi = ubs + NC * WAVE;
    // one block per 64 instances when the work area is LDS; the global-memory variants launch only as many blocks as the
    // device holds at once (their work area is per RESIDENT block, not per 64 instances) and walk the batch
    const long long nblk = (B + WAVE - 1) / WAVE;
#pragma unroll 1
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long b0 = blk * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
#pragma unroll
    for (int j = 0; j < N; ++j) zs[j * WAVE + lane] = 0.0;
    for (int k = 0; k < ny; ++k) ys[k * WAVE + lane] = 0.0;
    __syncthreads();
    stage_in_dyn(q + b0 * nq, nq, rows_valid, zs, lane);
    if (nx > 0) stage_in_dyn(x + b0 * nx, nx, rows_valid, zs + nq * WAVE, lane);
    if (ny > 0) stage_in_dyn(y + b0 * ny, ny, rows_valid, ys, lane);
    __syncthreads();
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = (j < n) ? zs[j * WAVE + lane] : 0.0;

    Kin<N> K;
    if (S->d.uses_fk) {
        forward_kinematics<N>(S, zs, As, lane, K);      // frames alias the (not yet written) row area
        if (S->d.quat_src != 0) orientation_feature<N>(S, ys, lane, K);
    }
    const int nc = qp_rows<N>(S, tk, K, z, ys, lane, n, As, lbs, ubs, hsi);

    // Q = A_u diag(1/h_u) A_u' + diag(1/h_slack on soft rows)        (H of reactive_qp.py:175-189)
    double hinv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) hinv[j] = (j < n) ? 1.0 / (S->qo.weight_shifter * S->qo.state_weights[j]) : 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (i < nc) {
            double ri[N];
#pragma unroll
            for (int j = 0; j < N; ++j) ri[j] = (j < n) ? As[(i * N + j) * WAVE + lane] * hinv[j] : 0.0;
#pragma unroll
            for (int k = 0; k <= i; ++k) {
                double acc = (k == i) ? hsi[i * WAVE + lane] : 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j < n) acc = fma(ri[j], As[(k * N + j) * WAVE + lane], acc);
                Qs[tri(i, k) * WAVE + lane] = acc;
            }
        }
    }

    // soft equality rows (wave-uniform): start active
    uint32_t softeq = 0u;
    {
        int row = 0;
        for (int ti = 0; ti < S->d.n_tasks; ++ti) {
            const clik_task& t = S->d.tasks[ti];
            const bool se = t.soft != 0 && (t.cls == CLIK_CLS_EQ || t.cls == CLIK_CLS_VELEQ);
            for (int i = 0; i < t.m; ++i)
                if (se) softeq |= 1u << (row + i);
            row += t.m;
        }
    }
    double nu[NC];
    const int status = gi_solve<NC, EXACT>(Qs, lbs, ubs, nullptr, softeq, lane, nc, S->qo.max_iter, valid, nu);

    // v = H^-1 A' nu
    double u[N];
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (i < nc) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j < n) u[j] = fma(nu[i], As[(i * N + j) * WAVE + lane], u[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = u[j] * hinv[j];
    // safety net in the space of the answer: every row  lbA <= A_u v - s <= ubA  must hold for the
    // returned v (s = -nu h_s^-1 on soft rows); see the shape-specialised kernel
    int status_v = status;
    if (status == 0) {
        double worst = 0.0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (i < nc) {
                double cv = nu[i] * hsi[i * WAVE + lane];
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j < n) cv = fma(As[(i * N + j) * WAVE + lane], u[j], cv);
                const double lbi = lbs[i * WAVE + lane], ubi = ubs[i * WAVE + lane];
                worst = fmax(worst, fmax((lbi - cv) / fmax(1.0, fabs(lbi)), (cv - ubi) / fmax(1.0, fabs(ubi))));
            }
        }
        if (!(worst <= 1e-7)) status_v = 2;      // (a net for garbage, not a precision test)
    }
    const unsigned bad = (status_v == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = nan_or(u[j], bad);
    if (slack_out != nullptr && valid) {
        const int ns = S->n_slack;
        int k = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (i < nc) {
                const double hi_ = hsi[i * WAVE + lane];
                if (hi_ != 0.0) {
                    slack_out[(b0 + lane) * ns + k] = nan_or(-nu[i] * hi_, bad);
                    ++k;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (j < n) zs[j * WAVE + lane] = u[j];
    __syncthreads();
    stage_out_dyn(dq + b0 * nq, nq, rows_valid, zs, lane);
    if (nx > 0 && dx != nullptr) stage_out_dyn(dx + b0 * nx, nx, rows_valid, zs + nq * WAVE, lane);
    if (status_out != nullptr && valid) status_out[b0 + lane] = status_v;
    __syncthreads();            // (the work area is reused by the next 64 instances of this block)
    }
}

// H diagonal, A, lbA, ubA exactly as the reference's H_func / A_func / Blb_func /
// Bub_func return them (reactive_qp.py:283-298), for inspection and parity tests.
template <int N, int NC, bool GWS = false>
__global__ __launch_bounds__(WAVE) void qp_data_kernel(
    const DevSkill* __restrict__ S0, const WarmArgs wa, const TickArgs tk, const long long B,
    const double* __restrict__ q, const double* __restrict__ x, const double* __restrict__ y,
    double* __restrict__ Hd, double* __restrict__ A, double* __restrict__ lbA, double* __restrict__ ubA,
    double* __restrict__ gws = nullptr, const int ny_slots = 0)
{
    extern __shared__ double lds_shared[];
    double* lds = GWS ? gws + (size_t)blockIdx.x * (size_t)qp_lds_slots<N, NC>(ny_slots) * WAVE : lds_shared;
    constexpr int NT = NC * (NC + 1) / 2;
    const int lane = threadIdx.x;
    const DevSkill* __restrict__ S = warm_descriptor(S0, wa);
    const int n = S->n, nq = S->d.n_q, nx = S->d.n_x, ny = S->d.n_y;
    double* zs = lds;
    double* ys = zs + N * WAVE;
    double* As = ys + ny * WAVE;
    double* Qs = As + (NC * N > 6 * N ? NC * N : 6 * N) * WAVE;
    double* lbs = Qs + NT * WAVE;
    double* ubs = lbs + NC * WAVE;
    double* hsi = ubs + NC * WAVE;
    const long long nblk = (B + WAVE - 1) / WAVE;
#pragma unroll 1
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long b0 = blk * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
#pragma unroll
    for (int j = 0; j < N; ++j) zs[j * WAVE + lane] = 0.0;
    for (int k = 0; k < ny; ++k) ys[k * WAVE + lane] = 0.0;
    __syncthreads();
    stage_in_dyn(q + b0 * nq, nq, rows_valid, zs, lane);
    if (nx > 0) stage_in_dyn(x + b0 * nx, nx, rows_valid, zs + nq * WAVE, lane);
    if (ny > 0) stage_in_dyn(y + b0 * ny, ny, rows_valid, ys, lane);
    __syncthreads();
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = (j < n) ? zs[j * WAVE + lane] : 0.0;
    Kin<N> K;
    if (S->d.uses_fk) {
        forward_kinematics<N>(S, zs, As, lane, K);
        if (S->d.quat_src != 0) orientation_feature<N>(S, ys, lane, K);
    }
    const int nc = qp_rows<N>(S, tk, K, z, ys, lane, n, As, lbs, ubs, hsi);
    if (valid) {
        const int ns = S->n_slack, nv = n + ns;
        const long long b = b0 + lane;
        for (int j = 0; j < n; ++j) Hd[b * nv + j] = S->qo.weight_shifter * S->qo.state_weights[j];
        for (int k = 0; k < ns; ++k) Hd[b * nv + n + k] = S->qo.weight_shifter + S->qo.slack_weights[k];
        int k = 0;
        for (int i = 0; i < nc; ++i) {
            double* row = A + (b * nc + i) * nv;
            for (int j = 0; j < nv; ++j) row[j] = 0.0;
            for (int j = 0; j < n; ++j) row[j] = As[(i * N + j) * WAVE + lane];
            if (hsi[i * WAVE + lane] != 0.0) {
                row[n + k] = -1.0;
                ++k;
            }
            lbA[b * nc + i] = lbs[i * WAVE + lane];
            ubA[b * nc + i] = ubs[i * WAVE + lane];
        }
    }
    __syncthreads();
    }
}

// ---- host side -----------------------------------------------------------------------
struct QpVariant {
    int N, NC, exact;       // exact: 1 = sizes are the skill's own (no guards), 0 = guarded, 2 = guarded + work area in global memory
    // (GwsOwner*: the handle's global-memory work area, used by the exact == 2 variants only)
    hipError_t (*solve)(const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*,
                        const double*, const double*, double*, double*, double*, int32_t*, hipStream_t, GwsOwner*);
    hipError_t (*data)(const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*,
                       const double*, const double*, double*, double*, double*, double*, hipStream_t, GwsOwner*);
};

template <int N, int NC, bool EXACT>
hipError_t qp_solve_launch(const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                                  const double* q, const double* x, const double* y, double* dq, double* dx,
                                  double* slack, int32_t* status, hipStream_t stream, GwsOwner*)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    const size_t shmem = (size_t)qp_lds_slots<N, NC>(ny) * WAVE * sizeof(double);
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)qp_solve_kernel<N, NC, EXACT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((qp_solve_kernel<N, NC, EXACT>), dim3(grid), dim3(WAVE), shmem, stream, dS, wa, tk, B, q, x, y, dq,
                       dx, slack, status);
    return hipGetLastError();
}

template <int N, int NC>
hipError_t qp_data_launch(const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                                 const double* q, const double* x, const double* y, double* Hd, double* A,
                                 double* lb, double* ub, hipStream_t stream, GwsOwner*)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    const size_t shmem = (size_t)qp_lds_slots<N, NC>(ny) * WAVE * sizeof(double);
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)qp_data_kernel<N, NC>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((qp_data_kernel<N, NC>), dim3(grid), dim3(WAVE), shmem, stream, dS, wa, tk, B, q, x, y, Hd, A,
                       lb, ub);
    return hipGetLastError();
}

// Work area in global memory for the variants no CU's LDS holds: owned by the controller handle (clik_workspace.hpp:
// sized for the blocks the batch needs, at most the resident ones - the kernels walk larger batches with a block
// stride -, grown by retiring the smaller area so that captured graphs stay valid, released with the handle).
// blocks of `kernel` the current device holds at once (64 threads, no LDS); cached per device and kernel
inline hipError_t resident_blocks(const void* kernel, unsigned* out)
{
    static std::mutex m;
    static std::map<std::pair<int, const void*>, unsigned> cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(m);
    auto it = cache.find(std::make_pair(dev, kernel));
    if (it != cache.end()) { *out = it->second; return hipSuccess; }
    int cus = 0, per_cu = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, 0);
    if (e != hipSuccess) return e;
    const unsigned n = (unsigned)(cus > 0 ? cus : 1) * (unsigned)(per_cu > 0 ? per_cu : 1);
    cache[std::make_pair(dev, kernel)] = n;
    *out = n;
    return hipSuccess;
}

template <int N, int NC>
hipError_t qp_solve_launch_gws(const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                                      const double* q, const double* x, const double* y, double* dq, double* dx,
                                      double* slack, int32_t* status, hipStream_t stream, GwsOwner* owner)
{
    if (owner == nullptr) return hipErrorInvalidValue;
    if (B <= 0) return hipSuccess;
    unsigned resident = 0;
    hipError_t e = resident_blocks((const void*)qp_solve_kernel<N, NC, false, true>, &resident);
    if (e != hipSuccess) return e;
    const unsigned long long nblk = (unsigned long long)((B + WAVE - 1) / WAVE);
    const unsigned grid = (unsigned)(nblk < resident ? nblk : resident);
    double* ws = nullptr;
    const size_t per_block = (size_t)qp_lds_slots<N, NC>(ny) * WAVE * sizeof(double);
    e = owner->acquire(stream, (size_t)grid * per_block, (size_t)resident * per_block, &ws);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((qp_solve_kernel<N, NC, false, true>), dim3(grid), dim3(WAVE), 0, stream, dS, wa, tk, B, q, x, y, dq,
                       dx, slack, status, ws, ny);
    return hipGetLastError();
}

template <int N, int NC>
hipError_t qp_data_launch_gws(const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                                     const double* q, const double* x, const double* y, double* Hd, double* A,
                                     double* lb, double* ub, hipStream_t stream, GwsOwner* owner)
{
    if (owner == nullptr) return hipErrorInvalidValue;
    if (B <= 0) return hipSuccess;
    unsigned resident = 0;
    hipError_t e = resident_blocks((const void*)qp_data_kernel<N, NC, true>, &resident);
    if (e != hipSuccess) return e;
    const unsigned long long nblk = (unsigned long long)((B + WAVE - 1) / WAVE);
    const unsigned grid = (unsigned)(nblk < resident ? nblk : resident);
    double* ws = nullptr;
    const size_t per_block = (size_t)qp_lds_slots<N, NC>(ny) * WAVE * sizeof(double);
    e = owner->acquire(stream, (size_t)grid * per_block, (size_t)resident * per_block, &ws);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((qp_data_kernel<N, NC, true>), dim3(grid), dim3(WAVE), 0, stream, dS, wa, tk, B, q, x, y, Hd, A, lb,
                       ub, ws, ny);
    return hipGetLastError();
}


// ---- the variants, in four groups (one translation unit each) --------------------------------------------------
// exact-size instantiations (no guards in the active-set loop) for the common problem sizes, guarded ones for
// everything else up to 16 rows; beyond 16 rows (up to CLIK_MAX_QPROWS) and / or more than eight states or rows per
// constraint: guarded with the work area in global memory; (14, 32): two 7-DoF arms in one skill (CLIK_MAX_DOF = 14)
#define CLIK_QP_VARIANTS_A(EXACT, GUARD, GLOBAL) EXACT(7, 13) EXACT(6, 12) EXACT(7, 10) EXACT(6, 9)
#define CLIK_QP_VARIANTS_B(EXACT, GUARD, GLOBAL) GUARD(6, 8) GUARD(6, 16) GUARD(7, 8) GUARD(7, 16)
#define CLIK_QP_VARIANTS_C(EXACT, GUARD, GLOBAL) GUARD(8, 8) GUARD(8, 16) GLOBAL(8, 32) GLOBAL(12, 16)
#define CLIK_QP_VARIANTS_D(EXACT, GUARD, GLOBAL) GLOBAL(12, 32) GLOBAL(14, 32)

#define CLIK_QP_SOLVE_ARGS                                                                                            \
    const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*, const double*, const double*,   \
        double*, double*, double*, int32_t*, hipStream_t, GwsOwner*
#define CLIK_QP_DATA_ARGS                                                                                             \
    const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*, const double*, const double*,   \
        double*, double*, double*, double*, hipStream_t, GwsOwner*
// X = `template` (definition, clik_qp_dyn_*.hip) or `extern template` (declaration, clik_qp.hip)
#define CLIK_QP_INST_EXACT(X, N, NC)                                  \
    X hipError_t qp_solve_launch<N, NC, true>(CLIK_QP_SOLVE_ARGS);    \
    X hipError_t qp_data_launch<N, NC>(CLIK_QP_DATA_ARGS);
#define CLIK_QP_INST_GUARD(X, N, NC)                                  \
    X hipError_t qp_solve_launch<N, NC, false>(CLIK_QP_SOLVE_ARGS);   \
    X hipError_t qp_data_launch<N, NC>(CLIK_QP_DATA_ARGS);
#define CLIK_QP_INST_GLOBAL(X, N, NC)                                 \
    X hipError_t qp_solve_launch_gws<N, NC>(CLIK_QP_SOLVE_ARGS);      \
    X hipError_t qp_data_launch_gws<N, NC>(CLIK_QP_DATA_ARGS);
#define CLIK_QP_DEF_EXACT(N, NC) CLIK_QP_INST_EXACT(template, N, NC)
#define CLIK_QP_DEF_GUARD(N, NC) CLIK_QP_INST_GUARD(template, N, NC)
#define CLIK_QP_DEF_GLOBAL(N, NC) CLIK_QP_INST_GLOBAL(template, N, NC)
#define CLIK_QP_DECL_EXACT(N, NC) CLIK_QP_INST_EXACT(extern template, N, NC)
#define CLIK_QP_DECL_GUARD(N, NC) CLIK_QP_INST_GUARD(extern template, N, NC)
#define CLIK_QP_DECL_GLOBAL(N, NC) CLIK_QP_INST_GLOBAL(extern template, N, NC)

}  // namespace clik
