// Shape-specialised ReactiveQPController tick (reactive_qp.py:461-528) on gfx950.
//
// Same problem as clik_qp.hip,
//     min 1/2 v'Hv   s.t.  lbA <= A v <= ubA,   v = [robot_vel; slack],  H = diag(h) > 0,
// with the structure of the skill known at compile time (ShapeDesc, as for the
// pseudo-inverse kernels) and an algebraically reduced formulation:
//
//   * SOFT EQUALITY rows  J v - s = b  (cost 1/2 h_s s^2) are eliminated exactly:
//         s = J v - b,      P = H_v + J' diag(h_s) J,      g = J' diag(h_s) b,
//     so the unconstrained minimiser v0 = P^-1 g already satisfies them optimally
//     (n x n LDL^T, n <= 8) and they never enter the active-set iteration.
//   * the remaining rows (hard rows, soft inequality rows) run the dual active-set
//     method of clik_qp.hip in constraint space with
//         Q = A_r P^-1 A_r' (+ 1/h_s on soft inequality rows),   c = A_r v0 + Q nu,
//         v = v0 + P^-1 A_r' nu.
//     For the config-4 skill (6 soft pose rows + 7 joint-speed rows) the masked
//     refactorisation per iteration is 7x7 instead of 13x13.
//
// The minimiser of a strictly convex QP is unique, so this returns the same v and
// slack as the reference's qpOASES call (parity: tests/test_gpu_qp.py against the
// oracle, tolerance QP_RTOL).
#pragma once
#include "clik_pinv_kernels.hpp"

namespace clik {

#ifdef CLIK_QP_DIAG
__device__ double g_qp_dbg[64 * 40];       // diagnostic dump of block 0 (tools only)
#endif

// ---- small LDL^T with a compile-time or run-time size -------------------------------
template <int NC, bool EXACT>
__device__ __forceinline__ void qp_ldl_factor(double (&A)[NC * (NC + 1) / 2], double (&rd)[NC], const int r)
{
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (EXACT || k < r) {
            double t[NC];
            double d = A[tri(k, k)];
#pragma unroll
            for (int j = 0; j < k; ++j) {
                t[j] = A[tri(k, j)] * A[tri(j, j)];
                d = fma(-A[tri(k, j)], t[j], d);
            }
            A[tri(k, k)] = d;
            const double inv = recip(d);
            rd[k] = inv;
#pragma unroll
            for (int i = k + 1; i < NC; ++i) {
                if (EXACT || i < r) {
                    double s = A[tri(i, k)];
#pragma unroll
                    for (int j = 0; j < k; ++j) s = fma(-A[tri(i, j)], t[j], s);
                    A[tri(i, k)] = s * inv;
                }
            }
        }
    }
}

template <int NC, bool EXACT>
__device__ __forceinline__ void qp_ldl_solve(const double (&A)[NC * (NC + 1) / 2], const double (&rd)[NC],
                                             double (&x)[NC], const int r)
{
#pragma unroll
    for (int i = 1; i < NC; ++i)
        if (EXACT || i < r) {
#pragma unroll
            for (int j = 0; j < i; ++j) x[i] = fma(-A[tri(i, j)], x[j], x[i]);
        }
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (EXACT || i < r) x[i] *= rd[i];
#pragma unroll
    for (int i = NC - 2; i >= 0; --i)
        if (EXACT || i < r) {
#pragma unroll
            for (int j = i + 1; j < NC; ++j)
                if (EXACT || j < r) x[i] = fma(-A[tri(j, i)], x[j], x[i]);
        }
}

// ---- dual active set in constraint space -------------------------------------------
// Qs: packed lower triangle of Q per lane (slot tri(i,j)*WAVE + lane), lbs / ubs: bounds,
// c0s: constraint values at the unconstrained minimiser (nullptr: zero),
// softeq: rows that are soft equalities (their Schur block is SPD, so they all start
// active: one solve instead of one iteration each).  On return nu holds the signed
// multipliers of the optimum.  Returns the status (0 optimal, 1 iteration cap,
// 2 infeasible).  EXACT: nc == NC at compile time (no size guards).
// WARM: start from the rows violated at the unconstrained minimiser instead of the empty
// working set (see the block below); n_vars bounds the size of that initial set.
template <int NC, bool EXACT, bool WARM = false>
__device__ __forceinline__ int gi_solve(const double* Qs, const double* lbs, const double* ubs, const double* c0s,
                                        const uint32_t softeq, const int lane, const int nc_rt,
                                        const int max_iter, const bool lane_valid, double (&nu)[NC],
                                        const int n_vars = NC, int32_t* hot = nullptr, const bool use_hot = false)
{
    constexpr int NT = NC * (NC + 1) / 2;
    const int nc = EXACT ? NC : nc_rt;
    // bounds stay in LDS (read once per iteration in the selection scan): keeping
    // them in registers next to the factor spills to scratch
    uint32_t W = softeq, up = 0u, eq = 0u;
    // Violations are measured relative to max(1, |bound|) of the bound in question, not of the row (a
    // one-sided SetConstraint carries the reference's default 1e10 on its other side, constraints.py:199-206:
    // a common scale per row would hide the violation of the real bound).  The scale is recomputed from the
    // bound the scan reads anyway (v_rcp_f64: ranking and a 1e-11 threshold do not need more); keeping the two
    // scales per row in registers instead (CLIK_QP_SCALE_REGS_MAX = rows up to which to do so) measured
    // 0.3-0.5 us slower per tick on config 4 (cold, hot and rollout) and spills beyond 8 rows.
    constexpr double kVtol = 1e-11;
#ifndef CLIK_QP_SCALE_REGS_MAX
#define CLIK_QP_SCALE_REGS_MAX 0
#endif
    constexpr bool kScaleRegs = NC <= CLIK_QP_SCALE_REGS_MAX;
    double isl[kScaleRegs ? NC : 1], ish[kScaleRegs ? NC : 1];
    auto excess = [&](const int i, const double lbi, const double ubi, const double cv, double& xlo,
                      double& xhi) __attribute__((always_inline)) {
        if constexpr (kScaleRegs) {
            xlo = (lbi - cv) * isl[i];
            xhi = (cv - ubi) * ish[i];
        } else {
            xlo = (lbi - cv) * __builtin_amdgcn_rcp(fmax(1.0, fabs(lbi)));
            xhi = (cv - ubi) * __builtin_amdgcn_rcp(fmax(1.0, fabs(ubi)));
        }
    };
    double c[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        nu[i] = 0.0;
        c[i] = 0.0;
        if constexpr (kScaleRegs) isl[i] = ish[i] = 1.0;
        if (EXACT || i < nc) {
            const double lbi = lbs[i * WAVE + lane], ubi = ubs[i * WAVE + lane];
            if (!(ubi - lbi > 0.0)) eq |= 1u << i;
            if constexpr (kScaleRegs) {
                isl[i] = 1.0 / fmax(1.0, fabs(lbi));
                ish[i] = 1.0 / fmax(1.0, fabs(ubi));
            }
        }
    }
    int status = 0;
#ifdef CLIK_QP_DIAG
    int g_qp_diag_cold = 0;
#endif
    bool done = !lane_valid;
    bool need_p = true;
    bool init = softeq != 0u;            // wave-uniform
    int p = 0;
    double sp = 1.0, bp = 0.0;
#ifdef CLIK_QP_DIAG
    int g_qp_diag_warm = 0;
#endif
    if constexpr (WARM) {
        // Warm start.  The dual method may start from any S-pair (W, nu): rows of W linearly
        // independent, the point optimal on {a_i v = b_i, i in W}, multipliers of the right
        // sign.  Start from W0 = rows violated at the unconstrained minimiser (each at the bound
        // it violates) and run a few primal-dual passes: solve  Q_WW nu_W = b_W - c0_W,  drop
        // the rows whose multiplier has the wrong sign AND add the rows the new point violates
        // (kPdPasses times; a pass that changes nothing has found the optimum), then only drop
        // until every multiplier has the right sign.  That is an S-pair, and on saturated
        // instances (most joint-speed rows active) it differs from the optimal working set by
        // 0.3 rows on average instead of 2.7 for the plain "violated at v0" guess - and each
        // pass costs half an active-set iteration.  Lanes whose set gets too large or
        // numerically dependent fall back to the cold start.
#ifndef CLIK_QP_PD_PASSES
#define CLIK_QP_PD_PASSES 4
#endif
        constexpr int kPdPasses = CLIK_QP_PD_PASSES;
        uint32_t W0 = 0u, up0 = 0u;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (EXACT || i < nc) {
                const double lbi = lbs[i * WAVE + lane], ubi = ubs[i * WAVE + lane];
                const double c0 = (c0s != nullptr) ? c0s[i * WAVE + lane] : 0.0;
                double vlo, vhi;
                excess(i, lbi, ubi, c0, vlo, vhi);
                if (fmax(vlo, vhi) > kVtol) {
                    W0 |= 1u << i;
                    if (vhi > vlo) up0 |= 1u << i;
                    ++cnt;
                }
            }
        }
        if (use_hot && hot != nullptr) {
            // hot start: the working set of the previous tick of this instance (bits 0..15 rows,
            // bits 16..31 "at the upper bound"), like the reference's qpOASES hot start
            // (reactive_qp.py:491-513).  Any set is only a guess: the passes below repair it.
            const uint32_t bits = (uint32_t)*hot;
            const uint32_t rowmask = (NC >= 16) ? 0xffffu : ((1u << NC) - 1u);
            W0 = bits & rowmask;
            up0 = (bits >> 16) & W0;
            cnt = __builtin_popcount(W0);
        }
        if (cnt > n_vars || !lane_valid) W0 = 0u;
        int pd_left = kPdPasses;
        for (int pass = 0; pass <= NC + kPdPasses; ++pass) {
            if (__ballot(W0 != 0u) == 0ull) break;
#ifdef CLIK_QP_DIAG
            if (W0 != 0u) ++g_qp_diag_warm;
#endif
            double L[NT], rd[NC], r[NC], a[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const double wi = (double)((W0 >> i) & 1u);
                a[i] = wi - 2.0 * wi * (double)((up0 >> i) & 1u);
                const double bi = ((up0 >> i) & 1u) ? ubs[i * WAVE + lane] : lbs[i * WAVE + lane];
                const double c0 = (c0s != nullptr && (EXACT || i < nc)) ? c0s[i * WAVE + lane] : 0.0;
                r[i] = a[i] * (bi - c0);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double q = (EXACT || i < nc) ? Qs[tri(i, j) * WAVE + lane] : 0.0;
                    L[tri(i, j)] = (a[i] * a[j]) * q;
                }
                L[tri(i, i)] += 1.0 - a[i] * a[i];
            }
            // pivots must stay well above rounding: a (near-)dependent W0 shows up as a tiny pivot
            double dmin = 1e300;
#pragma unroll
            for (int i = 0; i < NC; ++i) dmin = fmin(dmin, L[tri(i, i)]);    // diagonal scale before ...
            const double dscale = dmin;
            qp_ldl_factor<NC, EXACT>(L, rd, nc);
            bool sound = dscale > 0.0;
#pragma unroll
            for (int i = 0; i < NC; ++i) sound = sound & (L[tri(i, i)] > 1e-9 * dscale);                // ... and after
            qp_ldl_solve<NC, EXACT>(L, rd, r, nc);
            uint32_t drop = 0u;
#pragma unroll
            for (int i = 0; i < NC; ++i)
                if (((W0 >> i) & 1u) & !((eq >> i) & 1u) & !(r[i] >= 0.0)) drop |= 1u << i;
            // primal-dual passes also add the rows violated at the point this W0 gives
            uint32_t add = 0u, addup = 0u;
            const bool pd = pd_left > 0 & W0 != 0u & sound;
            if (__ballot(pd) != 0ull) {
                double cc[NC];
#pragma unroll
                for (int i = 0; i < NC; ++i) cc[i] = (c0s != nullptr && (EXACT || i < nc)) ? c0s[i * WAVE + lane] : 0.0;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        const double q = (EXACT || i < nc) ? Qs[tri(i, j) * WAVE + lane] : 0.0;
                        cc[i] = fma(q, a[j] * r[j], cc[i]);
                        if (j != i) cc[j] = fma(q, a[i] * r[i], cc[j]);
                    }
                }
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    if (EXACT || i < nc) {
                        const double lbi = lbs[i * WAVE + lane], ubi = ubs[i * WAVE + lane];
                        double vlo, vhi;
                        excess(i, lbi, ubi, cc[i], vlo, vhi);
                        if (pd & !((W0 >> i) & 1u) & fmax(vlo, vhi) > kVtol) {
                            add |= 1u << i;
                            if (vhi > vlo) addup |= 1u << i;
                        }
                    }
                }
            }
            if (!sound) {
                W0 = 0u;                    // cold start for this lane
#ifdef CLIK_QP_DIAG
                g_qp_diag_cold |= 1;
#endif
            } else if (drop != 0u | add != 0u) {
                W0 = (W0 & ~drop) | add;
                up0 = (up0 & ~drop) | addup;
                pd_left -= 1;
#ifdef CLIK_QP_DIAG
                if (W0 == 0u) g_qp_diag_cold |= 2;
                if (__builtin_popcount(W0) > n_vars) g_qp_diag_cold |= 4;
#endif
                if (__builtin_popcount(W0) > n_vars) W0 = 0u;       // cold start
            } else if (W0 != 0u) {
#pragma unroll
                for (int i = 0; i < NC; ++i) nu[i] = a[i] * r[i];
                W = W0;
                up = up0;
                W0 = 0u;                    // settled (a primal-dual pass that changed nothing: optimal)
            }
        }
    }
#ifdef CLIK_QP_DIAG
    int g_qp_diag_iters = 0;
#endif
    for (int it = 0; it < max_iter; ++it) {
        if (__ballot(!done) == 0ull) break;
#ifdef CLIK_QP_DIAG
        if (!done) ++g_qp_diag_iters;
#endif
        // (1) one pass over Q:  c = c0 + Q nu  and the masked Schur matrix  D Q_WW D.
        //     Masks are applied arithmetically (a_i = +-1 for active rows, 0 otherwise):
        //     per-lane bit tests as control flow would serialise the wave.
        double L[NT], rd[NC], r[NC], rhs[NC], a[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            c[i] = (c0s != nullptr && (EXACT || i < nc)) ? c0s[i * WAVE + lane] : 0.0;
            const double wi = (double)((W >> i) & 1u);
            a[i] = wi - 2.0 * wi * (double)((up >> i) & 1u);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double q = (EXACT || i < nc) ? Qs[tri(i, j) * WAVE + lane] : 0.0;
                c[i] = fma(q, nu[j], c[i]);
                if (j != i) c[j] = fma(q, nu[i], c[j]);
                L[tri(i, j)] = (a[i] * a[j]) * q;
            }
            L[tri(i, i)] += 1.0 - a[i] * a[i];          // identity on inactive rows
        }
        if (init) {
            // block start: all soft equalities active at once, nu_E = Q_EE^-1 b_E
            qp_ldl_factor<NC, EXACT>(L, rd, nc);
#pragma unroll
            for (int i = 0; i < NC; ++i) r[i] = ((W >> i) & 1u) ? lbs[i * WAVE + lane] : 0.0;
            qp_ldl_solve<NC, EXACT>(L, rd, r, nc);
#pragma unroll
            for (int i = 0; i < NC; ++i) nu[i] = ((W >> i) & 1u) ? r[i] : 0.0;
            init = false;
            continue;
        }
        // (2) pick the next constraint to enforce: unsatisfied equalities first,
        //     then the most violated inequality
        //     (written with selects, not per-lane branches: every divergent `if` costs
        //     exec-mask bookkeeping on the scalar unit, which a lone wave pays in full)
        {
            const bool sel = need_p & !done;
            double best = kVtol, bpn = bp;
            int pick = -1;
            bool pick_up = false;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (EXACT || i < nc) {
                    const double lbi = lbs[i * WAVE + lane], ubi = ubs[i * WAVE + lane];
                    double vlo, vhi;
                    excess(i, lbi, ubi, c[i], vlo, vhi);
                    double v = fmax(vlo, vhi);
                    v += (((eq >> i) & 1u) & v > kVtol) ? 1e30 : 0.0;      // equalities take precedence
                    const bool better = !((W >> i) & 1u) & v > best;
                    const bool upper = vhi > vlo;
                    best = better ? v : best;
                    pick = better ? i : pick;
                    pick_up = better ? upper : pick_up;
                    bpn = better ? (upper ? -ubi : lbi) : bpn;
                }
            }
            const bool take = sel & pick >= 0;
            done = done | (sel & pick < 0);
            p = take ? pick : p;
            sp = take ? (pick_up ? -1.0 : 1.0) : sp;
            bp = take ? bpn : bp;
        }
        if (__ballot(!done) == 0ull) break;
        // (3) step direction:  r = S_W^-1 (D Q_Wp sp),   zn = n_p' H^-1 (n_p - N_W r)
        double qpp = 0.0, cp = 0.0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            // column p of Q (dynamic per lane): packed index of (max(i,p), min(i,p))
            const int hi_ = i > p ? i : p, lo_ = i > p ? p : i;
            const double qip = (EXACT || i < nc) ? Qs[(hi_ * (hi_ + 1) / 2 + lo_) * WAVE + lane] : 0.0;
            rhs[i] = (a[i] * sp) * qip;
            const double isp = (i == p) ? 1.0 : 0.0;
            qpp = fma(isp, qip, qpp);
            cp = fma(isp, c[i], cp);
        }
        qp_ldl_factor<NC, EXACT>(L, rd, nc);
#pragma unroll
        for (int i = 0; i < NC; ++i) r[i] = rhs[i];
        qp_ldl_solve<NC, EXACT>(L, rd, r, nc);
        double zn = qpp;
#pragma unroll
        for (int i = 0; i < NC; ++i) zn = fma(-rhs[i], r[i], zn);
        // (4) step lengths.  The ratio test keeps the best candidate as a fraction
        //     (num / den, den > 0; first minimum wins) and divides once.
        double t1n = 1.0, t1d = 0.0;        // den = 0: no candidate yet
        int l = -1;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            // candidate only for active inequality rows with r_i > 0 (a_i = 0 on inactive rows)
            const bool cand_ok = (a[i] != 0.0) & !((eq >> i) & 1u) & r[i] > 1e-14;
            const double num = fmax(a[i] * nu[i], 0.0);
            // num / r_i < t1n / t1d   <=>   num * t1d < t1n * r_i      (both denominators positive)
            const bool better = cand_ok & (t1d == 0.0 | num * t1d < t1n * r[i]);
            t1n = better ? num : t1n;
            t1d = better ? r[i] : t1d;
            l = better ? i : l;
        }
        const double t1 = (l >= 0) ? t1n / t1d : 1e300;
        const double gap = bp - sp * cp;
        // zn = |n_p|^2 sin^2(angle between n_p and the span of the working set) in the H^-1 metric;
        // a row within ~1e-5 rad of that span counts as dependent (then only a dual step is possible).
        // A tighter test lets rounding noise through and the working set goes singular.
        const bool has_primal = zn > 1e-10 * qpp;
        const double t2 = has_primal ? gap / zn : 1e300;
        const double t = fmin(t1, t2);
        {
            const bool live = !done;
            const bool stuck = live & !(t < 1e299);        // constraint p cannot be satisfied
            const bool step = live & !stuck;
            const bool full = step & (t2 <= t1);           // p enters the working set
            const bool part = step & !full;                // l leaves it
            status = stuck ? 2 : status;
            done = done | stuck;
            const double ts = step ? t : 0.0;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                double ni = fma(-ts * a[i], r[i], nu[i]);
                ni = fma((i == p) ? ts : 0.0, sp, ni);
                nu[i] = (part & i == l) ? 0.0 : ni;
            }
            const uint32_t pbit = 1u << (p & 31), lbit = part ? (1u << (l & 31)) : 0u;
            W = (full ? (W | pbit) : W) & ~lbit;
            up = ((full & sp < 0.0) ? (up | pbit) : up) & ~lbit;
            need_p = step ? full : need_p;
        }
    }
    if (!done) status = 1;
    // Safety net (the oracle and qpOASES have the same): the point must satisfy every row.  An
    // infeasible problem whose working set became numerically dependent can leave the loop
    // "optimal" with violated rows; that is status 2, never a silent wrong answer.
#ifndef CLIK_QP_NU_NET
#define CLIK_QP_NU_NET 1
#endif
    if (CLIK_QP_NU_NET && __ballot(status == 0 && lane_valid) != 0ull) {
        double cc[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) cc[i] = (c0s != nullptr && (EXACT || i < nc)) ? c0s[i * WAVE + lane] : 0.0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double q = (EXACT || i < nc) ? Qs[tri(i, j) * WAVE + lane] : 0.0;
                cc[i] = fma(q, nu[j], cc[i]);
                if (j != i) cc[j] = fma(q, nu[i], cc[j]);
            }
        }
        double worst = 0.0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (EXACT || i < nc) {
                const double lbi = lbs[i * WAVE + lane], ubi = ubs[i * WAVE + lane];
                double vlo, vhi;
                excess(i, lbi, ubi, cc[i], vlo, vhi);
                worst = fmax(worst, fmax(vlo, vhi));
            }
        }
        if (status == 0 && !(worst <= 1e-7)) status = 2;      // (a net for garbage, not a precision test)
#ifdef CLIK_QP_DIAG
        if (blockIdx.x == 0) {
            double* o = g_qp_dbg + lane * 40;
#pragma unroll
            for (int i = 0; i < NC && i < 12; ++i) { o[i] = cc[i]; o[12 + i] = nu[i]; }
            o[24] = worst; o[25] = (double)W; o[26] = (double)up; o[27] = (double)status;
        }
        g_qp_diag_cold |= (worst > 1e-7) ? 8 : 0;
        g_qp_diag_cold |= 16;       // the check ran
#endif
    }
    if (hot != nullptr && lane_valid) *hot = (int32_t)((W & 0xffffu) | ((up & 0xffffu) << 16));
#ifdef CLIK_QP_DIAG
    status |= (g_qp_diag_iters << 8) | (g_qp_diag_warm << 16) | (g_qp_diag_cold << 24);
#endif
    return status;
}

// ---- compile-time plan of the reduced QP ---------------------------------------------
constexpr int QPS_MAX_ROWS = 16;       // active-set rows a static QP kernel carries in registers

struct QpPlanS {
    int  nr;                               // rows handed to the active-set solver
    int  row_task[CLIK_MAX_QPROWS];
    int  row_local[CLIK_MAX_QPROWS];
    int  ns;                               // slack variables (= soft rows, in row order)
    int  slack_base[SHAPE_MAX_TASKS];      // first slack of a soft task
    bool folded[SHAPE_MAX_TASKS];          // soft equality: eliminated into P, g
    // active-set row of output i of a task that is not folded, and whether it shares that row with an
    // earlier constraint: two hard joint-space rows on the same state (joint limits  lb <= dq_i <= ub  from a
    // SetConstraint on q and the speed limit  -v <= dq_i <= v  of a VelocitySetConstraint on q - the pair
    // every UR5 notebook stacks, e.g. ur5_dual_quaternion_vs_transformation_matrix.ipynb cell 14) are ONE
    // row  max(lb) <= dq_i <= min(ub): same feasible set, same minimiser, half the active-set size.
    int  row_of[SHAPE_MAX_TASKS][CLIK_MAX_M];
    bool merged[SHAPE_MAX_TASKS][CLIK_MAX_M];
};

constexpr QpPlanS make_qp_plan(const ShapeDesc& sd)
{
    QpPlanS p{};
    for (int ti = 0; ti < sd.n_tasks; ++ti) {
        const int cls = sd.cls[ti];
        const bool soft = sd.soft[ti] != 0;
        p.slack_base[ti] = p.ns;
        if (soft) p.ns += sd.m[ti];
        p.folded[ti] = soft && (cls == CLIK_CLS_EQ || cls == CLIK_CLS_VELEQ);
        if (!p.folded[ti]) {
            const bool box = !soft && shape_unit(sd, ti) && (cls == CLIK_CLS_SET || cls == CLIK_CLS_VELSET);
            for (int i = 0; i < sd.m[ti]; ++i) {
                int same = -1;
                if (box)
                    for (int r = 0; r < p.nr && r < CLIK_MAX_QPROWS && same < 0; ++r) {
                        const int t2 = p.row_task[r];
                        const bool box2 = sd.soft[t2] == 0 && shape_unit(sd, t2) &&
                                          (sd.cls[t2] == CLIK_CLS_SET || sd.cls[t2] == CLIK_CLS_VELSET);
                        if (box2 && sd.ucol[t2][p.row_local[r]] == sd.ucol[ti][i]) same = r;
                    }
                if (same >= 0) {
                    p.row_of[ti][i] = same;
                    p.merged[ti][i] = true;
                    continue;
                }
                p.row_of[ti][i] = p.nr;
                if (p.nr < CLIK_MAX_QPROWS) {
                    p.row_task[p.nr] = ti;
                    p.row_local[p.nr] = i;
                }
                ++p.nr;
            }
        }
    }
    return p;
}

// Box family: after the soft equalities are folded into P and g, every remaining row is a HARD bound on one
// state variable (joint-limit SetConstraint / speed-limit VelocitySetConstraint on q, merged per state):
//     min 1/2 v'P v - g'v   s.t.  lb_c <= v_c <= ub_c  on the bounded states c
// (BASELINE config 4 and the QP stacks of the UR5 notebooks).  Such a QP is solved by a primal active-set
// iteration on the states (qp_box_pas) instead of the dual active-set iteration over rows.
constexpr bool qp_box_family(const ShapeDesc& sd)
{
    const QpPlanS p = make_qp_plan(sd);
    if (p.nr <= 0 || p.nr > CLIK_MAX_DOF) return false;
    for (int r = 0; r < p.nr; ++r) {
        const int ti = p.row_task[r];
        if (sd.soft[ti] != 0 || !shape_unit(sd, ti)) return false;
        if (sd.cls[ti] != CLIK_CLS_SET && sd.cls[ti] != CLIK_CLS_VELSET) return false;
    }
    return true;
}
// Solvers measured for this family on config 4 (16384 instances, cold / hot start per tick, same box):
//   dual active-set iteration over rows (gi_solve, what every other QP shape runs)      38.7 / 10.4 us
//   projected Newton (round 2, tools/experiments/qp_retired.patch)                      38.5 / 25.8 us
//   block principal pivoting (first attempt, history)                                   93   /  7.4 us
//   primal active set from the vertex the linear term points to, qp_box_pas (default)   see DESIGN.md section 5
// The tick is the slowest instance of the batch (every wave has a SIMD to itself), i.e. its pass count times the
// instructions of a pass: the dual iteration needs 11-13 passes of ~1000 instructions on the worst instance, the
// projected Newton 10 of ~1500, the primal active set up to 19 of ~350 (mean 3.3): only 0-3 of the 7 states are
// free at the optimum, so a method that starts from a vertex and frees one state per pass is there quickly, and
// a pass is one masked 7 x 7 factorisation with everything in registers.  -DCLIK_QP_BOX_OFF: this family runs the
// dual iteration like the others (regression switch).
#if defined(CLIK_QP_BOX_OFF)
#define CLIK_QP_BOX_OK(SD) false
#else
#define CLIK_QP_BOX_OK(SD) qp_box_family(SD)
#endif

// Mixed family (round 3): the rows left after folding are hard bounds on single states (the box), a few HARD
// GENERAL rows (a SetConstraint on a task-space expression - the wall sets of ur5_moe2016_example2.ipynb cell 6 -
// reactive_qp.py:221-225; hard equalities), and SOFT inequality rows.  A soft inequality row  lb <= a v - s <= ub
// with cost 1/2 h s^2  is exactly a bounded variable  w = a v - s in [lb, ub]  with cost  1/2 h (a v - w)^2: it is
// LIFTED into the box (z = [v; w]).  The hard general rows enter a primal active set next to the held states
// (qp_mixed_pas).  Row kinds, in plan order:
constexpr int QPK_BOX = 0, QPK_HARD = 1, QPK_LIFT = 2;
#ifndef CLIK_QP_MIXED_MAX_Z
#define CLIK_QP_MIXED_MAX_Z 8          // states + lifted rows carried in registers
#endif
#ifndef CLIK_QP_MIXED_MAX_H
#define CLIK_QP_MIXED_MAX_H 3          // hard general rows
#endif
constexpr int qp_row_kind(const ShapeDesc& sd, const QpPlanS& p, int r)
{
    const int ti = p.row_task[r];
    if (sd.soft[ti] != 0) return QPK_LIFT;
    if (shape_unit(sd, ti) && (sd.cls[ti] == CLIK_CLS_SET || sd.cls[ti] == CLIK_CLS_VELSET)) return QPK_BOX;
    return QPK_HARD;
}
constexpr int qp_kind_count(const ShapeDesc& sd, int kind)
{
    const QpPlanS p = make_qp_plan(sd);
    int n = 0;
    for (int r = 0; r < p.nr && r < CLIK_MAX_QPROWS; ++r) n += qp_row_kind(sd, p, r) == kind;
    return n;
}
// index of row r among the rows of its kind
constexpr int qp_kind_index(const ShapeDesc& sd, int r)
{
    const QpPlanS p = make_qp_plan(sd);
    const int kind = qp_row_kind(sd, p, r);
    int n = 0;
    for (int q = 0; q < r; ++q) n += qp_row_kind(sd, p, q) == kind;
    return n;
}
constexpr bool qp_mixed_family(const ShapeDesc& sd)
{
    const QpPlanS p = make_qp_plan(sd);
    if (p.nr <= 0 || p.nr > CLIK_MAX_QPROWS || qp_box_family(sd)) return false;
    const int nl = qp_kind_count(sd, QPK_LIFT), nh = qp_kind_count(sd, QPK_HARD);
    if (nl + nh == 0) return false;
    // (skills with generated attribute code keep the dual iteration: its row bounds come through another path)
    // (registers: the packed matrix and its factor dominate - without hard rows one more variable fits;
    // measured on the 6-DoF + 3 soft walls skill: 256 VGPRs + 155 AGPRs, no scratch)
    return sd.n + nl <= CLIK_QP_MIXED_MAX_Z + (nh == 0 ? 1 : 0) && nh <= CLIK_QP_MIXED_MAX_H;
}
#if defined(CLIK_QP_MIXED_OFF) || defined(CLIK_QP_BOX_OFF)
#define CLIK_QP_MIXED_OK(SD) false
#else
#define CLIK_QP_MIXED_OK(SD) qp_mixed_family(SD)
#endif

template <const ShapeDesc& SD>
struct QpLayout {
    static constexpr QpPlanS P = make_qp_plan(SD);
    static constexpr bool MIXED = CLIK_QP_MIXED_OK(SD);
    static constexpr int NL = MIXED ? qp_kind_count(SD, QPK_LIFT) : 0;      // lifted soft inequality rows
    static constexpr int NH = MIXED ? qp_kind_count(SD, QPK_HARD) : 0;      // hard general rows
    static constexpr bool BOX = CLIK_QP_BOX_OK(SD);
    static constexpr int N = SD.n;
    static constexpr int NY = SD.n_y > 0 ? SD.n_y : 0;
    static constexpr int NR = P.nr;
    static constexpr int NRA = NR > 0 ? NR : 1;
    static constexpr int NT = NRA * (NRA + 1) / 2;
    static constexpr int NS = P.ns;
    static constexpr int NSA = NS > 0 ? NS : 1;
    static constexpr size_t TAIL_OFF = (sizeof(Img<SD>) + 15) & ~(size_t)15;
    static constexpr int IMG_CHUNKS = (int)((TAIL_OFF + sizeof(QpTail) + 1023) / 1024);
    static constexpr int IMG_DOUBLES = IMG_CHUNKS * 128;
    // 64-double slots behind the image
    static constexpr int O_Z = 0;
    static constexpr int O_Y = O_Z + N;
    static constexpr int O_Q = O_Y + NY;
    static constexpr bool PRIMAL = BOX || MIXED;            // (the primal solvers keep no Q / Y / c0 in LDS)
    static constexpr int O_LB = O_Q + (PRIMAL ? 0 : NT);
    static constexpr int O_UB = O_LB + NRA;
    static constexpr int O_C0 = O_UB + NRA;
    static constexpr int O_YS = O_C0 + (PRIMAL ? 0 : NRA);      // P^-1 a_r'  (NR x N)
    static constexpr int O_SL = O_YS + (PRIMAL ? 0 : NRA * N);  // folded right-hand sides, then the slack output rows
    static constexpr int SLOTS = O_SL + NSA;
    static constexpr size_t LDS_BYTES = ((size_t)IMG_DOUBLES + (size_t)SLOTS * WAVE) * sizeof(double);
};

// bounds  lbA, ubA  of the rows of task TI (reactive_qp.py:191-246)
template <const ShapeDesc& SD, int TI, class TASK>
__device__ __forceinline__ void qp_bounds_s(const TASK& t, const double (&e)[SD.m[TI]],
                                            const double (&Jt)[SD.m[TI]], double (&lo)[SD.m[TI]],
                                            double (&hi)[SD.m[TI]])
{
    constexpr int M = SD.m[TI];
    constexpr int cls = SD.cls[TI];
    if constexpr (cls == CLIK_CLS_EQ) {
        double ke[M];
        gain_apply_s<M, SD.gain_matrix[TI] != 0>(t, e, ke);
#pragma unroll
        for (int i = 0; i < M; ++i) lo[i] = hi[i] = -Jt[i] - ke[i];
    } else if constexpr (cls == CLIK_CLS_SET) {
        double d0[M], g[M];
#pragma unroll
        for (int i = 0; i < M; ++i) d0[i] = t.set_min[i] - e[i];
        gain_apply_s<M, SD.gain_matrix[TI] != 0>(t, d0, g);
#pragma unroll
        for (int i = 0; i < M; ++i) lo[i] = -Jt[i] + g[i];
#pragma unroll
        for (int i = 0; i < M; ++i) d0[i] = t.set_max[i] - e[i];
        gain_apply_s<M, SD.gain_matrix[TI] != 0>(t, d0, g);
#pragma unroll
        for (int i = 0; i < M; ++i) hi[i] = -Jt[i] + g[i];
    } else if constexpr (cls == CLIK_CLS_VELEQ) {
#pragma unroll
        for (int i = 0; i < M; ++i) lo[i] = hi[i] = t.target[i] - Jt[i];
    } else {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            lo[i] = t.set_min[i] - Jt[i];
            hi[i] = t.set_max[i] - Jt[i];
        }
    }
}

// per-task pass 1: bounds; folded tasks accumulate into P and g, the others publish their bounds
// SSTR: stride of the work-area slots - WAVE: the block's LDS area, one column per lane; 1: a private array of the
// lane (the LDS-free value-specialised kernel of the box family; every slot index is a constant, so it dissolves
// into registers)
template <const ShapeDesc& SD, int TI, int SSTR = WAVE>
__device__ __forceinline__ void qp_gather_s(const Img<SD>* __restrict__ S, const QpTail* __restrict__ T,
                                            const TickArgs& tk, const TaskCache<SD>& tc, const double (&z)[SD.n],
                                            const double* ys, const int lane,
                                            double (&Pm)[SD.n * (SD.n + 1) / 2], double (&g)[SD.n], double* slots)
{
    if constexpr (TI < SD.n_tasks) {
        using LY = QpLayout<SD>;
        constexpr int N = SD.n;
        constexpr int M = SD.m[TI];
        constexpr QpPlanS P = LY::P;
        decltype(auto) t = task_consts<SD, TI>(S, tc);
        double e[M], Jt[M], lo[M], hi[M];
        task_values<SD, TI>(S, tk, tc, z, ys, lane, e, Jt);
        qp_bounds_s<SD, TI>(t, e, Jt, lo, hi);
        if constexpr (P.folded[TI]) {
            constexpr int sb = P.slack_base[TI];
            static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const double hs = T->mu + T->slack_w[sb + i];
                const double hb = hs * lo[i];
                slots[(LY::O_SL + sb + i) * SSTR + (SSTR == 1 ? 0 : lane)] = lo[i];
                if constexpr (shape_unit(SD, TI)) {
                    constexpr int col = SD.ucol[TI][i] - 1;
                    Pm[tri(col, col)] += hs;
                    g[col] += hb;
                } else {
                    double hj[N];
#pragma unroll
                    for (int a = 0; a < N; ++a) {
                        const double ja = jac<SD, TI>(S, tc, i, a);
                        hj[a] = hs * ja;
                        g[a] = fma(hb, ja, g[a]);
                    }
#pragma unroll
                    for (int a = 0; a < N; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b) Pm[tri(a, b)] = fma(hj[a], jac<SD, TI>(S, tc, i, b), Pm[tri(a, b)]);
                }
            });
        } else {
            static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int row = P.row_of[TI][i];
                double* lbp = slots + (LY::O_LB + row) * SSTR + (SSTR == 1 ? 0 : lane);
                double* ubp = slots + (LY::O_UB + row) * SSTR + (SSTR == 1 ? 0 : lane);
                if constexpr (P.merged[TI][i]) {
                    // (the row was written by the earlier constraint of the pair: tasks are gathered in order)
                    *lbp = fmax(*lbp, lo[i]);
                    *ubp = fmin(*ubp, hi[i]);
                } else {
                    *lbp = lo[i];
                    *ubp = hi[i];
                }
            });
        }
        qp_gather_s<SD, TI + 1, SSTR>(S, T, tk, tc, z, ys, lane, Pm, g, slots);
    }
}

// element j of active-set row R
template <const ShapeDesc& SD, int R>
__device__ __forceinline__ double qp_row_s(const Img<SD>* __restrict__ S, const TaskCache<SD>& tc, const int j)
{
    constexpr QpPlanS P = QpLayout<SD>::P;
    constexpr int TI = P.row_task[R];
    constexpr int i = P.row_local[R];
    return jac<SD, TI>(S, tc, i, j);
}

// Primal active set for  min f(x) = 1/2 x'P x - g'x,  lb_c <= x_c <= ub_c  (P symmetric positive definite, packed
// lower triangle; unbounded states carry -/+ infinity).  x is feasible throughout, W = the states held on a bound.
// One pass:
//   Newton direction on the free states,  d_F = P_FF^-1 grad_F  (one fixed-size LDL' of P with a 1e30 penalty on
//     the diagonal of the held states: one instruction stream for all lanes, only the masks differ),
//   step  x - alpha d,  alpha = min(1, first bound hit)  (ratio test on fractions, one division);
//   alpha < 1: the states that hit join W (snapped onto their bounds);
//   alpha = 1: x minimises f on the face; the held state whose multiplier has the wrong sign by the largest
//     amount is released (its Newton direction then points inward, so it is not blocked at once: f decreases
//     strictly from face to face and the iteration is finite), none: x is the KKT point.
//   the gradient is recomputed from x (the same 49 multiply-adds an update would cost, and no drift).
// Start: see below (cold: coordinate-wise minimisers, clipped; hot: the partition of the previous tick,
// hot = atL | atU << 16).  Returns 0 (KKT point, checked on the final gradient), 1 (pass cap), 2 (lb > ub).
// (Measured and retired, tools/experiments/qp_retired.patch: the four lanes of a quad on one instance with different
// relaxation factors - QUAD, profiles/r3_qp_portfolio_study.md - and the swept tableau with one rank-one update per
// working-set change instead of a masked refactorisation per pass - profiles/r4_qp_sweep.txt.)

// FOLIO (round 4, small batches, cold start): FOUR WAVES of a block - one per SIMD of the CU - work on the SAME 64
// instances, each with its own START of the active-set passes (number, order and relaxation of the Gauss-Seidel sweeps:
// different instruction streams cost nothing across waves, unlike across the lanes of a wave), and an instance is done as
// soon as ANY of them has reached its KKT point.  The tick of a 16384-instance batch is its slowest instance: with the
// four starts of qp_solve_static_box_folio_values_kernel the slowest instance of a batch of bench inputs needs 4.2 - 5.1 us
// of sweeps + passes instead of 6.25 - 7.2 (numpy model of the iteration, tools/qp_wave_portfolio_study.py; measured:
// profiles/r4_qp_wave_portfolio.txt - the arrangement itself costs 0.9 - 1.2 us at one block per CU, half of that).
// Which wave's answer is taken must not depend on timing: every finish is recorded as a KEY = virtual time (sweeps +
// 8 per pass: a pass costs about eight sweeps) x 4 + strategy, smallest key wins (atomic minimum in LDS); a wave gives an
// instance up only when a key SMALLER than any it could still produce has been recorded, so the strategy that would
// record the smallest key always gets to record it, whatever the interleaving.
struct QpFolio {
    int* key_slot;     // LDS word of this lane's instance (initialised to INT_MAX)
    int sid;           // strategy 0 ... 3 (low bits of the key: ties go to the lower one)
    int sweeps;        // Gauss-Seidel start sweeps
    int kind;          // 0 forward, 1 reverse order, 2 forward with over-relaxation `omega`, 3 reverse with it
    double omega;
};
constexpr int kFolioPassUnits = 8;

template <int N, bool REV, bool RELAX>
__device__ __forceinline__ void qp_box_start_sweep(const double (&Pm)[N * (N + 1) / 2], const double (&ip)[N],
                                                   const double (&lb)[N], const double (&ub)[N], double (&x)[N],
                                                   double (&res)[N], const double omega)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int a = REV ? N - 1 - k : k;
        const double xa = fmin(fmax(RELAX ? x[a] + omega * (res[a] * ip[a]) : fma(res[a], ip[a], x[a]), lb[a]), ub[a]);
        const double dl = xa - x[a];
        x[a] = xa;
#pragma unroll
        for (int b = 0; b < N; ++b) res[b] = fma(-Pm[a >= b ? tri(a, b) : tri(b, a)], dl, res[b]);
    }
}

template <int N, bool FOLIO = false>
__device__ __forceinline__ int qp_box_pas(const double (&Pm)[N * (N + 1) / 2], const double (&g)[N],
                                          const double (&lb)[N], const double (&ub)[N], const int max_pass,
                                          const bool valid, double (&x)[N], int32_t* hot, const bool use_hot,
                                          const QpFolio* fo = nullptr, int* my_key = nullptr)
{
    constexpr int NT = N * (N + 1) / 2;
    CLIK_PHASE("box_setup");
    constexpr int kOne = 0x3ff00000;        // high word of 1.0: the masks below are doubles 1.0 / 0.0 kept as that word
    auto as_mask = [](const int hi) __attribute__((always_inline)) { return __hiloint2double(hi, 0); };
    bool empty = false;
#pragma unroll
    for (int a = 0; a < N; ++a) empty = empty | (lb[a] - ub[a] > 1e-9 * fmax(1.0, fmax(fabs(lb[a]), fabs(ub[a]))));
    uint32_t hl = 0u, hu = 0u;
    if (use_hot && hot != nullptr) {
        const uint32_t h = (uint32_t)*hot;
        hl = h & 0xffffu;
        hu = (h >> 16) & ~hl;
    }
    // held[a]: the state sits on a bound and is not moved (mask word); free_ok[a]: it may be released (lb < ub)
    int held[N], free_ok[N];
    double tol[N];
    // Cold start: projected Gauss-Seidel sweeps from the clipped coordinate-wise minimisers x_a = clip(g_a / P_aa)
    // - each state in turn to the minimiser of its own coordinate, clipped, the residual g - P x updated by one
    // column (77 instructions per sweep against ~460 + seven dependent reciprocals of an active-set pass) - and the
    // states that end on a bound start held.  The sweeps identify most of the optimal partition: on the config-4
    // bench inputs (0-3 of the 7 states free at the optimum) the active-set passes that follow number
    //   sweeps      0      1      2      4      6      8     12
    //   mean     2.95   1.99   1.47   1.23   1.15   1.11   1.07
    //   worst      14     18     11     11      7      7      5        of 16384 instances
    // (tools/qp_pass_study.py; the vertex the linear term points to: 3.35 / 19, the clipped unconstrained
    // minimiser: 5.8 / 15).  Measured per tick at 16384 instances: no sweeps 19.6 us, 6 sweeps 13.9, 12 sweeps 12.9.
    // Hot: the partition of the previous tick, no sweeps.
#ifndef CLIK_QP_BOX_SWEEPS
#define CLIK_QP_BOX_SWEEPS 12
#endif
    CLIK_PHASE("box_cold_sweeps");
    if (!use_hot) {
        double ip[N], res[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            ip[a] = __builtin_amdgcn_rcp(Pm[tri(a, a)]);                  // (a start needs no correct rounding)
            x[a] = fmin(fmax(g[a] * ip[a], lb[a]), ub[a]);
        }
#pragma unroll
        for (int a = 0; a < N; ++a) {
            double sacc = g[a];
#pragma unroll
            for (int b = 0; b < N; ++b) sacc = fma(-Pm[a >= b ? tri(a, b) : tri(b, a)], x[b], sacc);
            res[a] = sacc;
        }
        if constexpr (FOLIO) {
            // (wave-uniform choice: the four waves of the block take different branches)
            if (fo->kind == 1) {
#pragma unroll 1
                for (int sweep = 0; sweep < fo->sweeps; ++sweep) qp_box_start_sweep<N, true, false>(Pm, ip, lb, ub, x, res, 1.0);
            } else if (fo->kind == 2) {
#pragma unroll 1
                for (int sweep = 0; sweep < fo->sweeps; ++sweep) qp_box_start_sweep<N, false, true>(Pm, ip, lb, ub, x, res, fo->omega);
            } else if (fo->kind == 3) {
#pragma unroll 1
                for (int sweep = 0; sweep < fo->sweeps; ++sweep) qp_box_start_sweep<N, true, true>(Pm, ip, lb, ub, x, res, fo->omega);
            } else {
#pragma unroll 1
                for (int sweep = 0; sweep < fo->sweeps; ++sweep) qp_box_start_sweep<N, false, false>(Pm, ip, lb, ub, x, res, 1.0);
            }
        } else {
#pragma unroll 1
        for (int sweep = 0; sweep < CLIK_QP_BOX_SWEEPS; ++sweep) {
#pragma unroll
            for (int a = 0; a < N; ++a) {
                const double xa = fmin(fmax(fma(res[a], ip[a], x[a]), lb[a]), ub[a]);
                const double dl = xa - x[a];
                x[a] = xa;
#pragma unroll
                for (int b = 0; b < N; ++b) res[b] = fma(-Pm[a >= b ? tri(a, b) : tri(b, a)], dl, res[b]);
            }
        }
        }
    }
    CLIK_PHASE("box_partition_gradient");
#pragma unroll
    for (int a = 0; a < N; ++a) {
        const double mid = fmin(fmax(0.0, lb[a]), ub[a]);
        const bool has_l = lb[a] > -1e300, has_u = ub[a] < 1e300;
        const bool on_l = (use_hot ? (((hl >> a) & 1u) != 0u) : (x[a] <= lb[a])) & has_l;
        const bool on_u = (use_hot ? (((hu >> a) & 1u) != 0u) : (x[a] >= ub[a])) & has_u & !on_l;
        const double xs = on_l ? lb[a] : (on_u ? ub[a] : (use_hot ? mid : x[a]));
        x[a] = empty ? 0.0 : xs;
        const bool pin = !(ub[a] > lb[a]);
        held[a] = (on_l | on_u | pin) ? kOne : 0;
        free_ok[a] = pin ? 0 : kOne;
        tol[a] = 1e-9 * fmax(1.0, fabs(g[a]));
    }
    double gr[N];
    auto gradient = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < N; ++a) {
            double sacc = -g[a];
#pragma unroll
            for (int b = 0; b < N; ++b) sacc = fma(Pm[a >= b ? tri(a, b) : tri(b, a)], x[b], sacc);
            gr[a] = sacc;
        }
    };
    gradient();
    bool done = !valid | empty;
    int status = empty ? 2 : 1;
    bool gave_up = false;        // (FOLIO: another strategy has this lane's instance)
    int seen = 0x7fffffff, mine = 0;
    if constexpr (FOLIO) {
        if (valid & empty) {     // (infeasible bounds: every strategy says so at once)
            mine = (fo->sweeps << 2) | fo->sid;
            (void)__hip_atomic_fetch_min(fo->key_slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        seen = *(volatile int*)fo->key_slot;
    }
    // (masks applied arithmetically and selections by min / max: a select of a double costs two instructions and
    // a flag test two more, and a lone wave pays every one of them in full)
    CLIK_PHASE("box_pass");
#pragma unroll 1
    for (int pass = 0; pass < max_pass; ++pass) {
        if constexpr (FOLIO) {
            // the key this pass would record at its end; a smaller one on record: nothing this strategy can still do wins
            // (`seen`: the record as read at the end of the previous pass - the read's latency hides behind that pass's
            // tail; an older value only delays giving up, never the choice of the answer)
            const int kfin = ((fo->sweeps + kFolioPassUnits * (pass + 1)) << 2) | fo->sid;
            const bool beaten = !done & (seen < kfin);
            gave_up = gave_up | beaten;
            done = done | beaten;
        }
        if (__ballot(!done) == 0ull) break;
        const bool done_before = done;
        // Newton direction on the free states
        double d[N];
        double M[NT], rd[N];
#pragma unroll
        for (int a = 0; a < NT; ++a) M[a] = Pm[a];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double hm = as_mask(held[a]);
            M[tri(a, a)] = fma(hm, 1e30, Pm[tri(a, a)]);
            d[a] = fma(-hm, gr[a], gr[a]);
        }
        int seen_mid = 0x7fffffff;
        if constexpr (FOLIO) seen_mid = *(volatile int*)fo->key_slot;       // (consumed after the factorisation)
        ldl_factor_s<N>(M, rd);
        ldl_solve_s<N>(M, rd, d);
        if constexpr (FOLIO) {
            // a second look half-way through the pass: a wave whose last instances have just been finished elsewhere
            // leaves now, not a pass later
            const int kfin = ((fo->sweeps + kFolioPassUnits * (pass + 1)) << 2) | fo->sid;
            const bool beaten = !done & (seen_mid < kfin);
            gave_up = gave_up | beaten;
            done = done | beaten;
            if (__ballot(!done) == 0ull) break;
        } else {
            (void)seen_mid;
        }
        // first bound hit along x - alpha d: alpha = min(1, room_a / d_a).  The quotient needs no correct rounding (a
        // blocked step ends on no face minimum, and the states that land are snapped onto their bounds): hardware
        // reciprocal.  0 * inf = NaN for a held state on its bound, which min ignores.
        double tgt[N], r[N];
        double amin = 1.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            d[a] = fma(-as_mask(held[a]), d[a], d[a]);                    // (exactly zero on the held states)
            tgt[a] = (d[a] > 0.0) ? lb[a] : ub[a];
            r[a] = fabs(x[a] - tgt[a]) * __builtin_amdgcn_rcp(fabs(d[a]));
            amin = fmin(amin, r[a]);
        }
        const bool blocked = amin < 1.0;
        const double alpha = done ? 0.0 : amin;
        const double thr = amin * (1.0 + 1e-7);
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double xn = fma(-alpha, d[a], x[a]);
            const bool lands = blocked & !done & (r[a] <= thr);            // the blocking state (and ties): held there
            x[a] = lands ? tgt[a] : xn;
            held[a] = lands ? kOne : held[a];
        }
        gradient();
        // at a face minimum: release the held state whose multiplier is wrong by the largest amount, or stop
        double c[N];
        double worst = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double push = (x[a] <= lb[a]) ? -gr[a] : gr[a];          // > 0: f decreases by leaving the bound
            c[a] = fma(push, as_mask(held[a] & free_ok[a]), -tol[a]);
            worst = fmax(worst, c[a]);
        }
        const bool release = !blocked & !done & (worst > 0.0);
#pragma unroll
        for (int a = 0; a < N; ++a) held[a] = (release & (c[a] == worst)) ? 0 : held[a];
        done = done | (!blocked & !(worst > 0.0));
        if constexpr (FOLIO) {
            if (done & !done_before) {
                mine = ((fo->sweeps + kFolioPassUnits * (pass + 1)) << 2) | fo->sid;
                (void)__hip_atomic_fetch_min(fo->key_slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            seen = *(volatile int*)fo->key_slot;
        } else {
            (void)done_before;
        }
    }
    if constexpr (FOLIO) {
        if (valid & !done) {      // (pass cap: on record all the same, behind every strategy that converged)
            mine = ((fo->sweeps + kFolioPassUnits * (max_pass + 1)) << 2) | fo->sid;
            (void)__hip_atomic_fetch_min(fo->key_slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        *my_key = mine;
        (void)gave_up;
    } else {
        (void)seen;
        (void)mine;
    }
    CLIK_PHASE("box_kkt_hotword");
    if (!empty) {
        // the KKT conditions of the returned point (gr is the gradient at x)
        bool kkt = true;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const bool fine = ((x[a] <= lb[a]) & (gr[a] >= -tol[a])) | ((x[a] >= ub[a]) & (gr[a] <= tol[a])) | (fabs(gr[a]) <= tol[a]);
            kkt = kkt & fine;
        }
        status = (done & kkt) ? 0 : 1;
    }
    if (hot != nullptr && valid) {
        uint32_t atL = 0u, atU = 0u;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const bool h = held[a] != 0;
            if (h & (x[a] <= lb[a])) atL |= 1u << a;
            else if (h & (x[a] >= ub[a])) atU |= 1u << a;
        }
        *hot = (int32_t)(atL | (atU << 16));
    }
    CLIK_PHASE_END();
    return status;
}

// Primal active set for  min f(z) = 1/2 z'P z - g'z,  lb_c <= z_c <= ub_c,  lbg_r <= G_r z <= ubg_r  (P symmetric
// positive definite, packed lower triangle, NZ variables; G has NH rows that touch the first NV variables only).
// The working set holds states on their bounds (as in qp_box_pas) AND rows on one of theirs (act_r = -1 / +1;
// a row with lbg == ubg is an equality: always active, never released).  One pass:
//   M = P with a 1e30 penalty on the held diagonal (one fixed-size LDL' for all lanes), d0 = M^-1 grad_F,
//   Y_r = M^-1 G_r' (NH solves with the same factor), S = G Y (NH x NH; 1e30 on the diagonal of inactive rows),
//   lambda = S^-1 (res - G d0)  with  res_r = G_r z - bound_r  on the active rows: the Newton step  d = d0 + Y lambda
//     minimises f on the face AND removes the residual of its active rows, so the iteration may start with rows that
//     are violated (they start active at the violated bound; no phase 1),
//   ratio test along z - alpha d over the free states' bounds and the inactive rows (both kinds land and join W),
//   alpha = 1: z is the face minimum, -lambda are the row multipliers there: the held state or active row whose
//     multiplier is wrong by the largest amount is released; none: KKT point - unless an active row is still off its
//     bound (S was singular on this face: its residual cannot be removed with the states that are left, and no
//     multiplier asks for a release): the rows admit no point, status 2.
// Start: Gauss-Seidel sweeps on the box as in qp_box_pas, rows violated there start active; hot: the partition of the
// previous tick (hot = atL | atU << 10 | rows at their lower bound << 20 | rows at their upper bound << 24).
// numpy prototype and its sweep against the oracle: tools/qp_mixed_proto.py (1500 random problems with 1-4
// general rows, 4-8 variables: every minimiser to 3e-10, every infeasible one recognised, no pass cap).
// Returns 0 (KKT point), 1 (pass cap), 2 (no feasible point).
template <int NZ, int NV, int NH>
__device__ __forceinline__ int qp_mixed_pas(const double (&Pm)[NZ * (NZ + 1) / 2], const double (&g)[NZ],
                                            const double (&lb)[NZ], const double (&ub)[NZ],
                                            const double (&G)[NH > 0 ? NH : 1][NV], const double (&lbg)[NH > 0 ? NH : 1],
                                            const double (&ubg)[NH > 0 ? NH : 1], const int max_pass, const bool valid,
                                            double (&x)[NZ], int32_t* hot, const bool use_hot)
{
    constexpr int NT = NZ * (NZ + 1) / 2;
    constexpr int NHA = NH > 0 ? NH : 1;
    constexpr int NHT = NHA * (NHA + 1) / 2;
    constexpr int kOne = 0x3ff00000;
    auto as_mask = [](const int hi) __attribute__((always_inline)) { return __hiloint2double(hi, 0); };
    bool empty = false;
#pragma unroll
    for (int a = 0; a < NZ; ++a) empty = empty | (lb[a] - ub[a] > 1e-9 * fmax(1.0, fmax(fabs(lb[a]), fabs(ub[a]))));
#pragma unroll
    for (int r = 0; r < NH; ++r) {
        empty = empty | (lbg[r] - ubg[r] > 1e-9 * fmax(1.0, fmax(fabs(lbg[r]), fabs(ubg[r]))));
        // a row against the box alone: its range over the box is [sum min(a lb, a ub), sum max(a lb, a ub)] - a wall
        // that asks for more than the joint-speed limits allow (the common infeasible tick of a wall skill, and the
        // slowest one for an active-set iteration to prove) is recognised here in a dozen instructions
        double lo = 0.0, hi = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const double a = G[r][j];
            const double al = (a != 0.0) ? a * lb[j] : 0.0, au = (a != 0.0) ? a * ub[j] : 0.0;
            lo += fmin(al, au);
            hi += fmax(al, au);
        }
        empty = empty | (lbg[r] > hi + 1e-9 * fmax(1.0, fabs(lbg[r]))) | (ubg[r] < lo - 1e-9 * fmax(1.0, fabs(ubg[r])));
    }
    uint32_t hl = 0u, hu = 0u, rl = 0u, ru = 0u;
    if (use_hot && hot != nullptr) {
        const uint32_t h = (uint32_t)*hot;
        hl = h & 0x3ffu;
        hu = (h >> 10) & 0x3ffu & ~hl;
        rl = (h >> 20) & 0xfu;
        ru = (h >> 24) & 0xfu & ~rl;
    }
    int held[NZ], free_ok[NZ];
    double tol[NZ];
    if (!use_hot) {
        double ip[NZ], res[NZ];
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            ip[a] = __builtin_amdgcn_rcp(Pm[tri(a, a)]);
            x[a] = fmin(fmax(g[a] * ip[a], lb[a]), ub[a]);
        }
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            double sacc = g[a];
#pragma unroll
            for (int b = 0; b < NZ; ++b) sacc = fma(-Pm[a >= b ? tri(a, b) : tri(b, a)], x[b], sacc);
            res[a] = sacc;
        }
#pragma unroll 1
        for (int sweep = 0; sweep < CLIK_QP_BOX_SWEEPS; ++sweep) {
#pragma unroll
            for (int a = 0; a < NZ; ++a) {
                const double xa = fmin(fmax(fma(res[a], ip[a], x[a]), lb[a]), ub[a]);
                const double dl = xa - x[a];
                x[a] = xa;
#pragma unroll
                for (int b = 0; b < NZ; ++b) res[b] = fma(-Pm[a >= b ? tri(a, b) : tri(b, a)], dl, res[b]);
            }
        }
        if constexpr (NH > 0) {
            // The sweeps above do not see the rows.  Where the start violates a row, a few more sweeps on the problem
            // with that row pulled in by a penalty, rho (a_r x - bound_r)^2 / 2, let the box partition settle under the
            // row's pull BEFORE the active set starts (which then finds most speed limits already on their bounds):
            // on the wall skill near its walls the slowest of 8192 instances goes from 9 passes to 4, on random
            // problems the mean from 4.3 to 3.8 (tools/qp_mixed_proto.py); same minimisers.  Wave-uniform: skipped
            // when no lane violates a row (walls inactive: the common tick).
            double cf[NH], bv[NH];
            bool anyv = false;
            double pmax = 0.0;
#pragma unroll
            for (int a = 0; a < NZ; ++a) pmax = fmax(pmax, Pm[tri(a, a)]);
#pragma unroll
            for (int r = 0; r < NH; ++r) {
                double gx = 0.0;
#pragma unroll
                for (int jj = 0; jj < NV; ++jj) gx = fma(G[r][jj], x[jj], gx);
                const bool hi = gx > ubg[r] + 1e-12 * fmax(1.0, fabs(ubg[r]));
                const bool lo = gx < lbg[r] - 1e-12 * fmax(1.0, fabs(lbg[r]));
                cf[r] = (hi | lo) ? 1e3 * pmax : 0.0;
                bv[r] = hi ? ubg[r] : (lo ? lbg[r] : 0.0);
                anyv = anyv | hi | lo;
            }
            if (__ballot(anyv & valid & !empty) != 0ull) {
                double P2[NT], g2[NZ];
#pragma unroll
                for (int a = 0; a < NT; ++a) P2[a] = Pm[a];
#pragma unroll
                for (int a = 0; a < NZ; ++a) g2[a] = g[a];
#pragma unroll
                for (int r = 0; r < NH; ++r)
#pragma unroll
                    for (int a = 0; a < NV; ++a) {
                        const double ha = cf[r] * G[r][a];
                        g2[a] = fma(ha, bv[r], g2[a]);
#pragma unroll
                        for (int c = 0; c <= a; ++c) P2[tri(a, c)] = fma(ha, G[r][c], P2[tri(a, c)]);
                    }
#pragma unroll
                for (int a = 0; a < NZ; ++a) {
                    ip[a] = __builtin_amdgcn_rcp(P2[tri(a, a)]);
                    double sacc = g2[a];
#pragma unroll
                    for (int c = 0; c < NZ; ++c) sacc = fma(-P2[a >= c ? tri(a, c) : tri(c, a)], x[c], sacc);
                    res[a] = sacc;
                }
#pragma unroll 1
                for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
                    for (int a = 0; a < NZ; ++a) {
                        const double xa = fmin(fmax(fma(res[a], ip[a], x[a]), lb[a]), ub[a]);
                        const double dl = xa - x[a];
                        x[a] = xa;
#pragma unroll
                        for (int c = 0; c < NZ; ++c) res[c] = fma(-P2[a >= c ? tri(a, c) : tri(c, a)], dl, res[c]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NZ; ++a) {
        const double mid = fmin(fmax(0.0, lb[a]), ub[a]);
        const bool has_l = lb[a] > -1e300, has_u = ub[a] < 1e300;
        const bool on_l = (use_hot ? (((hl >> a) & 1u) != 0u) : (x[a] <= lb[a])) & has_l;
        const bool on_u = (use_hot ? (((hu >> a) & 1u) != 0u) : (x[a] >= ub[a])) & has_u & !on_l;
        const double xs = on_l ? lb[a] : (on_u ? ub[a] : (use_hot ? mid : x[a]));
        x[a] = empty ? 0.0 : xs;
        const bool pin = !(ub[a] > lb[a]);
        held[a] = (on_l | on_u | pin) ? kOne : 0;
        free_ok[a] = pin ? 0 : kOne;
        // (1e-10: a held state with a wrong multiplier below the tolerance stays held, and the velocities feel that error
        // divided by the smallest curvature, mu = 1e-3 along the directions the soft task does not see)
        tol[a] = 1e-10 * fmax(1.0, fabs(g[a]));
    }
    // rows: the hinted ones, the equalities, and whatever the start violates
    int act[NHA];
    bool eqr[NHA];
    double tolr[NHA];
#pragma unroll
    for (int r = 0; r < NHA; ++r) { act[r] = 0; eqr[r] = false; tolr[r] = 0.0; }
#pragma unroll
    for (int r = 0; r < NH; ++r) {
        double gx = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) gx = fma(G[r][j], x[j], gx);
        eqr[r] = !(ubg[r] > lbg[r]);
        const double su = 1e-12 * fmax(1.0, fabs(ubg[r])), sl = 1e-12 * fmax(1.0, fabs(lbg[r]));
        int a0 = (gx > ubg[r] + su) ? 1 : ((gx < lbg[r] - sl) ? -1 : 0);
        if (use_hot) a0 = ((ru >> r) & 1u) ? 1 : (((rl >> r) & 1u) ? -1 : a0);
        act[r] = eqr[r] ? 1 : a0;
    }
    double gr[NZ];
    auto gradient = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            double sacc = -g[a];
#pragma unroll
            for (int b = 0; b < NZ; ++b) sacc = fma(Pm[a >= b ? tri(a, b) : tri(b, a)], x[b], sacc);
            gr[a] = sacc;
        }
    };
    gradient();
    bool done = !valid | empty;
    int status = empty ? 2 : 1;
    double lam[NHA];
#pragma unroll
    for (int r = 0; r < NHA; ++r) lam[r] = 0.0;
    // (every pass changes the working set by one entry or ends; the random sweeps need at most 2 (NZ + NH) passes; an
    // instance still turning after 3 (NZ + NH) + 6 - rows and bounds that contradict each other jointly can make the
    // releases cycle - is cut off and classified below)
    const int pass_cap = max_pass < 3 * (NZ + NH) + 6 ? max_pass : 3 * (NZ + NH) + 6;
#pragma unroll 1
    for (int pass = 0; pass < pass_cap; ++pass) {
        if (__ballot(!done) == 0ull) break;
        double M[NT], rd[NZ], d[NZ];
#pragma unroll
        for (int a = 0; a < NT; ++a) M[a] = Pm[a];
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            const double hm = as_mask(held[a]);
            M[tri(a, a)] = fma(hm, 1e30, Pm[tri(a, a)]);
            d[a] = fma(-hm, gr[a], gr[a]);
        }
        ldl_factor_s<NZ>(M, rd);
        ldl_solve_s<NZ>(M, rd, d);
#pragma unroll
        for (int a = 0; a < NZ; ++a) d[a] = fma(-as_mask(held[a]), d[a], d[a]);      // (exactly zero on the held states)
        if constexpr (NH > 0) {
            double Y[NH][NZ], S[NHT], rhs[NH];
#pragma unroll
            for (int r = 0; r < NH; ++r) {
                // (a row no lane of the wave has in its working set needs no solve: walls are inactive in most ticks)
                const bool used = __ballot((act[r] != 0) & !done) != 0ull;
#pragma unroll
                for (int a = 0; a < NZ; ++a) Y[r][a] = (a < NV && used) ? fma(-as_mask(held[a]), G[r][a < NV ? a : 0], G[r][a < NV ? a : 0]) : 0.0;
                if (used) {
                    ldl_solve_s<NZ>(M, rd, Y[r]);
#pragma unroll
                    for (int a = 0; a < NZ; ++a) Y[r][a] = fma(-as_mask(held[a]), Y[r][a], Y[r][a]);
                }
            }
#pragma unroll
            for (int r = 0; r < NH; ++r) {
                double gx = 0.0, gd = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    gx = fma(G[r][j], x[j], gx);
                    gd = fma(G[r][j], d[j], gd);
                }
                const double bnd = (act[r] > 0) ? ubg[r] : lbg[r];
                rhs[r] = (act[r] != 0) ? (gx - bnd) - gd : 0.0;
#pragma unroll
                for (int q = 0; q <= r; ++q) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) acc = fma(G[r][j], Y[q][j], acc);
                    S[tri(r, q)] = acc;
                }
                // (an active row without any free state under it has S_rr = 0: the small diagonal keeps the factor finite
                // and its huge multiplier then asks for a held state to be released.  Its bias on healthy rows - the
                // step would leave 1e-13 x lambda of the residual - is removed by one step of iterative refinement below)
                S[tri(r, r)] += (act[r] != 0) ? 1e-13 : 1e30;
            }
            double rs[NH];
            ldl_factor_s<NH>(S, rs);
            ldl_solve_s<NH>(S, rs, rhs);
            {
                // S0 lambda = rhs with S = S0 + 1e-13 I on the active rows: lambda += S^-1 (1e-13 lambda)
                double corr[NH];
#pragma unroll
                for (int r = 0; r < NH; ++r) corr[r] = (act[r] != 0) ? 1e-13 * rhs[r] : 0.0;
                ldl_solve_s<NH>(S, rs, corr);
#pragma unroll
                for (int r = 0; r < NH; ++r) rhs[r] += corr[r];
            }
#pragma unroll
            for (int r = 0; r < NH; ++r) {
                // (a lane that is done keeps the multipliers of its final point: the passes other lanes still need
                // must not overwrite them - its rows may not even be solved for any more, see `used`)
                lam[r] = done ? lam[r] : ((act[r] != 0) ? rhs[r] : 0.0);
#pragma unroll
                for (int a = 0; a < NZ; ++a) d[a] = fma(lam[r], Y[r][a], d[a]);
            }
        }
        // first bound hit along x - alpha d: free states, then inactive rows
        double tgt[NZ], rr[NZ];
        double amin = 1.0;
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            tgt[a] = (d[a] > 0.0) ? lb[a] : ub[a];
            rr[a] = fabs(x[a] - tgt[a]) * __builtin_amdgcn_rcp(fabs(d[a]));
            amin = fmin(amin, rr[a]);
        }
        double rq[NHA], slp[NHA];
#pragma unroll
        for (int r = 0; r < NH; ++r) {
            double gx = 0.0, gd = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                gx = fma(G[r][j], x[j], gx);
                gd = fma(G[r][j], d[j], gd);
            }
            slp[r] = gd;
            const double tg = (gd > 0.0) ? lbg[r] : ubg[r];
            const double q = fabs(gx - tg) * __builtin_amdgcn_rcp(fabs(gd));
            rq[r] = (act[r] == 0 && gd != 0.0) ? q : __builtin_inf();
            amin = fmin(amin, rq[r]);
        }
        const bool blocked = amin < 1.0;
        const double alpha = done ? 0.0 : amin;
        const double thr = amin * (1.0 + 1e-7);
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            const double xn = fma(-alpha, d[a], x[a]);
            const bool lands = blocked & !done & (rr[a] <= thr);
            x[a] = lands ? tgt[a] : xn;
            held[a] = lands ? kOne : held[a];
        }
#pragma unroll
        for (int r = 0; r < NH; ++r) {
            const bool lands = blocked & !done & (rq[r] <= thr);
            act[r] = lands ? ((slp[r] > 0.0) ? -1 : 1) : act[r];
        }
        gradient();
        // at a face minimum: release the held state / active row whose multiplier is wrong by the largest amount
        double c[NZ], cr[NHA];
        double worst = 0.0;
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            double gf = gr[a];
            if (a < NV) {
#pragma unroll
                for (int r = 0; r < NH; ++r) gf = fma(lam[r], G[r][a < NV ? a : 0], gf);
            }
            const double push = (x[a] <= lb[a]) ? -gf : gf;
            c[a] = fma(push, as_mask(held[a] & free_ok[a]), -tol[a]);
            worst = fmax(worst, c[a]);
        }
#pragma unroll
        for (int r = 0; r < NH; ++r) {
            const double pr = (act[r] != 0 && !eqr[r]) ? -(double)act[r] * lam[r] : 0.0;
            cr[r] = pr - 1e-10 * fmax(1.0, fabs(lam[r]));
            worst = fmax(worst, cr[r]);
        }
        const bool release = !blocked & !done & (worst > 0.0);
#pragma unroll
        for (int a = 0; a < NZ; ++a) held[a] = (release & (c[a] == worst)) ? 0 : held[a];
#pragma unroll
        for (int r = 0; r < NH; ++r) act[r] = (release & (cr[r] == worst)) ? 0 : act[r];
        const bool fin = !blocked & !(worst > 0.0);
        if (fin & !done) {
            // the active rows must sit on their bounds now (see above)
            bool off = false;
#pragma unroll
            for (int r = 0; r < NH; ++r) {
                double gx = 0.0, mag = 1.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    gx = fma(G[r][j], x[j], gx);
                    mag = fmax(mag, fabs(G[r][j] * x[j]));        // (the rounding of the sum scales with its terms)
                }
                const double bnd = (act[r] > 0) ? ubg[r] : lbg[r];
                off = off | ((act[r] != 0) & (fabs(gx - bnd) > 1e-8 * fmax(mag, fabs(bnd))));
            }
            status = off ? 2 : 0;
        }
        done = done | fin;
    }
    if (!empty) {
        // What is handed back: a KKT point (0), or - cut off, or ended on a face whose conditions do not check out -
        // "no feasible point found" (2) when the point still violates a row, else "pass cap" (1).  The reference's
        // solver raises in both cases.
        bool viol = false;
#pragma unroll
        for (int r = 0; r < NH; ++r) {
            double gx = 0.0, mag = 1.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                gx = fma(G[r][j], x[j], gx);
                mag = fmax(mag, fabs(G[r][j] * x[j]));
            }
            viol = viol | (gx > ubg[r] + 1e-7 * fmax(mag, fabs(ubg[r]))) | (gx < lbg[r] - 1e-7 * fmax(mag, fabs(lbg[r])));
        }
        bool kkt = !viol;
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            double gf = gr[a];
            if (a < NV) {
#pragma unroll
                for (int r = 0; r < NH; ++r) gf = fma(lam[r], G[r][a < NV ? a : 0], gf);
            }
            const double t5 = 100.0 * tol[a];
            const bool fine = ((x[a] <= lb[a]) & (gf >= -t5)) | ((x[a] >= ub[a]) & (gf <= t5)) | (fabs(gf) <= t5);
            kkt = kkt & fine;
        }
        status = ((status == 2) | viol) ? 2 : (((status == 0) & kkt) ? 0 : 1);
    }
    if (hot != nullptr && valid) {
        uint32_t atL = 0u, atU = 0u, rowL = 0u, rowU = 0u;
#pragma unroll
        for (int a = 0; a < NZ; ++a) {
            const bool h = held[a] != 0;
            if (h & (x[a] <= lb[a])) atL |= 1u << a;
            else if (h & (x[a] >= ub[a])) atU |= 1u << a;
        }
#pragma unroll
        for (int r = 0; r < NH; ++r) {
            if (act[r] < 0) rowL |= 1u << r;
            if (act[r] > 0) rowU |= 1u << r;
        }
        *hot = (int32_t)(atL | (atU << 10) | (rowL << 20) | (rowU << 24));
    }
    return status;
}

// One ReactiveQPController tick of the lane's instance: FK, rows, reduced QP, active set.
// v: [robot_vel; virtual_vel], sl: slack values, hot: the lane's working-set word (nullable).
// slots: the block's LDS work area (QpLayout<SD>), reused from tick to tick.
// HAVE_SC: the sines / cosines of the state variables come from the caller (sns / css: evaluated two per lane of a
// quad, or two per wave of a FOLIO block, and exchanged), otherwise this lane evaluates all of them
template <const ShapeDesc& SD, int SSTR = WAVE, bool FOLIO = false, bool HAVE_SC = false>
__device__ __forceinline__ int qp_tick_static(const Img<SD>* __restrict__ S, const QpTail* __restrict__ T,
                                              const TickArgs& tk, const double (&z)[SD.n], const double* ysl,
                                              const int lane, const bool valid, double* slots,
                                              double (&v)[SD.n], double (&sl)[QpLayout<SD>::NSA],
                                              int32_t* hot, const bool use_hot, const QpFolio* fo = nullptr,
                                              int* my_key = nullptr, const double* sns = nullptr,
                                              const double* css = nullptr)
{
    using LY = QpLayout<SD>;
    constexpr int N = SD.n;
    constexpr int NR = LY::NR, NRA = LY::NRA, NS = LY::NS;
    constexpr QpPlanS P = LY::P;
    constexpr int NTN = N * (N + 1) / 2;
    // FK and the state-dependent rows, once
    CLIK_PHASE("qp_fk_rows");
    TaskCache<SD> tc;
    {
        Kin<N> K;
        if constexpr (SD.uses_fk != 0) {
            if constexpr (HAVE_SC) {
                double sn[N], cs[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    sn[j] = sns[j];
                    cs[j] = css[j];
                }
                forward_kinematics_sc<SD>(S, z, sn, cs, K);
            } else {
                forward_kinematics_s<SD>(S, z, K);
            }
            if constexpr (SD.quat_src != 0) orientation_feature_s<SD>(S, ysl, lane, K);
        }
        cache_task<SD, 0, true>(S, tk, K, z, ysl, lane, tc);
    }

    // P = H_v + sum_soft-eq J' diag(h_s) J,  g = sum J' diag(h_s) b;  bounds of the other rows -> LDS
    CLIK_PHASE("qp_gather_P_g");
    double L[NTN], rd[N];
#pragma unroll
    for (int a = 0; a < N; ++a) {
        v[a] = 0.0;
#pragma unroll
        for (int b = 0; b <= a; ++b) L[tri(a, b)] = (a == b) ? T->mu * T->state_w[a] : 0.0;
    }
    static_assert(SSTR == WAVE || QpLayout<SD>::BOX, "private slots: box family only");
    qp_gather_s<SD, 0, SSTR>(S, T, tk, tc, z, ysl, lane, L, v, slots);
    if constexpr (LY::MIXED) {
        // z = [v; w]: the states and one bounded variable per soft inequality row; hard general rows beside the box
        constexpr int NL = LY::NL, NH = LY::NH, NZ = N + NL, NHA = NH > 0 ? NH : 1;
        constexpr int NTZ = NZ * (NZ + 1) / 2;
        double Pz[NTZ], gz[NZ], lbz[NZ], ubz[NZ], zz[NZ];
        double Gh[NHA][N], lbg[NHA], ubg[NHA];
#pragma unroll
        for (int a = 0; a < NTZ; ++a) Pz[a] = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) Pz[tri(a, b)] = L[tri(a, b)];
            gz[a] = v[a];
            lbz[a] = -__builtin_inf();
            ubz[a] = __builtin_inf();
        }
#pragma unroll
        for (int a = N; a < NZ; ++a) gz[a] = 0.0;
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int TI = P.row_task[r];
            constexpr int kind = qp_row_kind(SD, P, r);
            const double lo = slots[(LY::O_LB + r) * SSTR + (SSTR == 1 ? 0 : lane)];
            const double hi = slots[(LY::O_UB + r) * SSTR + (SSTR == 1 ? 0 : lane)];
            if constexpr (kind == QPK_BOX) {
                constexpr int col = SD.ucol[TI][P.row_local[r]] - 1;
                lbz[col] = lo;
                ubz[col] = hi;
            } else {
                double ar[N];
                if constexpr (shape_unit(SD, TI)) {
                    constexpr int col = SD.ucol[TI][P.row_local[r]] - 1;
#pragma unroll
                    for (int j = 0; j < N; ++j) ar[j] = (j == col) ? 1.0 : 0.0;
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) ar[j] = qp_row_s<SD, r>(S, tc, j);
                }
                if constexpr (kind == QPK_LIFT) {
                    constexpr int k = N + qp_kind_index(SD, r);
                    constexpr int sk = P.slack_base[TI] + P.row_local[r];
                    const double hs = T->mu + T->slack_w[sk];
#pragma unroll
                    for (int a = 0; a < N; ++a) {
                        const double ha = hs * ar[a];
#pragma unroll
                        for (int b = 0; b <= a; ++b) Pz[tri(a, b)] = fma(ha, ar[b], Pz[tri(a, b)]);
                        Pz[tri(k, a)] = -ha;
                    }
                    Pz[tri(k, k)] = hs;
                    lbz[k] = lo;
                    ubz[k] = hi;
                } else {
                    constexpr int k = qp_kind_index(SD, r);
#pragma unroll
                    for (int j = 0; j < N; ++j) Gh[k][j] = ar[j];
                    lbg[k] = lo;
                    ubg[k] = hi;
                }
            }
        });
        int status = qp_mixed_pas<NZ, N, NH>(Pz, gz, lbz, ubz, Gh, lbg, ubg, T->max_iter, valid, zz, hot, use_hot);
        if (!valid) status = 0;
#pragma unroll
        for (int a = 0; a < N; ++a) v[a] = zz[a];
        // slack: folded rows s = J v - b; lifted rows s = a v - w
#pragma unroll
        for (int k = 0; k < LY::NSA; ++k) sl[k] = (NS > 0) ? slots[(LY::O_SL + k) * SSTR + (SSTR == 1 ? 0 : lane)] : 0.0;
        static_for<0, SD.n_tasks>([&](auto tc_) __attribute__((always_inline)) {
            constexpr int TI = decltype(tc_)::value;
            if constexpr (P.folded[TI]) {
                constexpr int sb = P.slack_base[TI];
                static_for<0, SD.m[TI]>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    double acc = -sl[sb + i];
                    if constexpr (shape_unit(SD, TI)) {
                        constexpr int col = SD.ucol[TI][i] - 1;
                        acc += v[col];
                    } else {
#pragma unroll
                        for (int j = 0; j < N; ++j) acc = fma(jac<SD, TI>(S, tc, i, j), v[j], acc);
                    }
                    sl[sb + i] = acc;
                });
            }
        });
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int TI = P.row_task[r];
            if constexpr (qp_row_kind(SD, P, r) == QPK_LIFT) {
                constexpr int k = N + qp_kind_index(SD, r);
                constexpr int sk = P.slack_base[TI] + P.row_local[r];
                double acc = -zz[k];
                if constexpr (shape_unit(SD, TI)) {
                    acc += v[SD.ucol[TI][P.row_local[r]] - 1];
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) acc = fma(qp_row_s<SD, r>(S, tc, j), v[j], acc);
                }
                sl[sk] = acc;
            }
        });
        return status;
    } else if constexpr (LY::BOX) {
        // bounds by state (rows of the plan are unit rows on distinct states), then the primal active set
        CLIK_PHASE("qp_bounds");
        double lbc[N], ubc[N], gv[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            lbc[a] = -__builtin_inf();
            ubc[a] = __builtin_inf();
            gv[a] = v[a];
        }
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int col = SD.ucol[P.row_task[r]][P.row_local[r]] - 1;
            lbc[col] = slots[(LY::O_LB + r) * SSTR + (SSTR == 1 ? 0 : lane)];
            ubc[col] = slots[(LY::O_UB + r) * SSTR + (SSTR == 1 ? 0 : lane)];
        });
        int status = qp_box_pas<N, FOLIO>(L, gv, lbc, ubc, T->max_iter, valid, v, hot, use_hot, fo, my_key);
        CLIK_PHASE("qp_slack");
        if (!valid) status = 0;
        // slack of the folded rows: s = J v - b
#pragma unroll
        for (int k = 0; k < LY::NSA; ++k) sl[k] = (NS > 0) ? slots[(LY::O_SL + k) * SSTR + (SSTR == 1 ? 0 : lane)] : 0.0;
        static_for<0, SD.n_tasks>([&](auto tc_) __attribute__((always_inline)) {
            constexpr int TI = decltype(tc_)::value;
            if constexpr (P.folded[TI]) {
                constexpr int sb = P.slack_base[TI];
                static_for<0, SD.m[TI]>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    double acc = -sl[sb + i];
                    if constexpr (shape_unit(SD, TI)) {
                        constexpr int col = SD.ucol[TI][i] - 1;
                        acc += v[col];
                    } else {
#pragma unroll
                        for (int j = 0; j < N; ++j) acc = fma(jac<SD, TI>(S, tc, i, j), v[j], acc);
                    }
                    sl[sb + i] = acc;
                });
            }
        });
        return status;
    } else {
    CLIK_PHASE("qp_general_dual");
    ldl_factor_s<N>(L, rd);
    ldl_solve_s<N>(L, rd, v);          // v0 = P^-1 g

    int status = 0;
    if constexpr (NR > 0) {
        static_assert(NR <= QPS_MAX_ROWS, "too many active-set rows for a static QP kernel");
        double* Qs = slots + LY::O_Q * WAVE;
        double* lbs = slots + LY::O_LB * WAVE;
        double* ubs = slots + LY::O_UB * WAVE;
        double* c0s = slots + LY::O_C0 * WAVE;
        double* Ys = slots + LY::O_YS * WAVE;
        // Y_r = P^-1 a_r',  c0_r = a_r v0,  Q_rs = a_r Y_s (+ slack curvature on the diagonal)
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int TI = P.row_task[r];
            double yr[N];
            double c0 = 0.0;
            if constexpr (shape_unit(SD, TI)) {
                constexpr int col = SD.ucol[TI][P.row_local[r]] - 1;
#pragma unroll
                for (int j = 0; j < N; ++j) yr[j] = (j == col) ? 1.0 : 0.0;
                c0 = v[col];
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    yr[j] = qp_row_s<SD, r>(S, tc, j);
                    c0 = fma(yr[j], v[j], c0);
                }
            }
            ldl_solve_s<N>(L, rd, yr);
            c0s[r * WAVE + lane] = c0;
#pragma unroll
            for (int j = 0; j < N; ++j) Ys[(r * N + j) * WAVE + lane] = yr[j];
            // row r of Q: a_s . Y_r for s >= r  (symmetric; stored at tri(s, r))
            static_for<r, NR>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                constexpr int TS = P.row_task[s];
                double acc;
                if constexpr (shape_unit(SD, TS)) {
                    constexpr int col = SD.ucol[TS][P.row_local[s]] - 1;
                    acc = yr[col];
                } else {
                    acc = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) acc = fma(qp_row_s<SD, s>(S, tc, j), yr[j], acc);
                }
                if constexpr (s == r && SD.soft[TI] != 0) {
                    constexpr int sk = P.slack_base[TI] + P.row_local[r];
                    acc += 1.0 / (T->mu + T->slack_w[sk]);
                }
                Qs[tri(s, r) * WAVE + lane] = acc;
            });
        });
        double nu[NRA];
        status = gi_solve<NRA, true, true>(Qs, lbs, ubs, c0s, 0u, lane, NR, T->max_iter, valid, nu, N, hot, use_hot);
        // v = v0 + Y nu;  slack of soft inequality rows = -nu / h_s
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = fma(nu[r], Ys[(r * N + j) * WAVE + lane], v[j]);
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int TI = P.row_task[r];
            if constexpr (SD.soft[TI] != 0) {
                constexpr int sk = P.slack_base[TI] + P.row_local[r];
                slots[(LY::O_SL + sk) * WAVE + lane] = -nu[r] / (T->mu + T->slack_w[sk]);
            }
        });
        // Safety net in the space of the answer (the oracle and qpOASES check the same): every row
        // must hold for the v that is returned.  (The multiplier-space test inside gi_solve can be
        // fooled when a numerically dependent working set blows the multipliers up.)
        if (status == 0) {
            double worst = 0.0;
            static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                constexpr int TI = P.row_task[r];
                double cv;
                if constexpr (shape_unit(SD, TI)) {
                    cv = v[SD.ucol[TI][P.row_local[r]] - 1];
                } else {
                    cv = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) cv = fma(qp_row_s<SD, r>(S, tc, j), v[j], cv);
                }
                if constexpr (SD.soft[TI] != 0) {
                    constexpr int sk = P.slack_base[TI] + P.row_local[r];
                    cv -= slots[(LY::O_SL + sk) * WAVE + lane];         // row is  a v - s
                }
                const double lbi = lbs[r * WAVE + lane], ubi = ubs[r * WAVE + lane];
                worst = fmax(worst, fmax((lbi - cv) / fmax(1.0, fabs(lbi)), (cv - ubi) / fmax(1.0, fabs(ubi))));
            });
            if (!(worst <= 1e-7)) status = 2;       // (a net for garbage, not a precision test)
        }
    }
    // slack of the folded rows: s = J v - b
#pragma unroll
    for (int k = 0; k < LY::NSA; ++k) sl[k] = (NS > 0) ? slots[(LY::O_SL + k) * WAVE + lane] : 0.0;
    static_for<0, SD.n_tasks>([&](auto tc_) __attribute__((always_inline)) {
        constexpr int TI = decltype(tc_)::value;
        if constexpr (P.folded[TI]) {
            constexpr int sb = P.slack_base[TI];
            static_for<0, SD.m[TI]>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                double acc = -sl[sb + i];
                if constexpr (shape_unit(SD, TI)) {
                    constexpr int col = SD.ucol[TI][i] - 1;
                    acc += v[col];
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) acc = fma(jac<SD, TI>(S, tc, i, j), v[j], acc);
                }
                sl[sb + i] = acc;
            });
        }
    });
    return status;
    }   // (dual active-set path)
}

// PT: one time-slot record per instance (t_inst [B][2 * n_tslots], device; see pinv_solve_static_body)
template <const ShapeDesc& SD> struct QpImg;
template <const ShapeDesc& SD, class IMGV>
constexpr QpImg<SD> qp_values_or_zero();

// skill image + QP options as one object (what the image-reading kernels find at img_g): the type a
// value-specialised kernel's constant has (IMGV::value, see clik_pinv_team.hpp)
template <const ShapeDesc& SD>
struct QpImg {
    Img<SD> img;
    alignas(16) QpTail tail;
};

// IMGV: void - the skill image and the QP options are read from img_g (staged through LDS, copied to registers);
// else IMGV::value is a QpImg<SD> constant and they are compiled in (no image traffic, no image registers)
template <const ShapeDesc& SD, bool PT, class IMGV = void>
__device__ __forceinline__ void qp_solve_static_body(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const int use_hot,
    const TickArgs& tk_uniform, const double* __restrict__ t_inst)
{
    extern __shared__ double lds[];
    using LY = QpLayout<SD>;
    constexpr int N = SD.n;
    constexpr int NR = LY::NR, NRA = LY::NRA, NS = LY::NS;
    constexpr QpPlanS P = LY::P;
    constexpr int NTN = N * (N + 1) / 2;
    const int lane = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    constexpr int NX = SD.n_x, NQ = SD.n - SD.n_x;
    double* slots = lds + LY::IMG_DOUBLES;
    double* zs = slots + LY::O_Z * WAVE;                 // [64][NQ] robot_var, then [64][NX] virtual_var
    double* xs = zs + NQ * WAVE;
    double* ys = slots + LY::O_Y * WAVE;
    // one memory round trip: image + options, joint state and inputs (see pinv_solve_static_kernel)
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr bool VALUES = !std::is_void<IMGV>::value;
    {
        d2 img[LY::IMG_CHUNKS];
        if constexpr (!VALUES) {
            const d2* src = (const d2*)img_g;
#pragma unroll
            for (int k = 0; k < LY::IMG_CHUNKS; ++k) img[k] = src[k * WAVE + lane];
        }
        double qv[NQ], xv[NX > 0 ? NX : 1], yv[SD.n_y > 0 ? SD.n_y : 1];
        stage_load<NQ>(q + b0 * NQ, NQ, rows_valid, lane, qv);
        if constexpr (NX > 0) stage_load<NX>(x + b0 * NX, NX, rows_valid, lane, xv);
        if constexpr (SD.n_y > 0) stage_load<SD.n_y>(y + b0 * SD.n_y, SD.n_y, rows_valid, lane, yv);
        if constexpr (!VALUES) {
            d2* dst = (d2*)lds;
#pragma unroll
            for (int k = 0; k < LY::IMG_CHUNKS; ++k) dst[k * WAVE + lane] = img[k];
        }
        rows_to_lds<NQ>(qv, zs, lane);
        if constexpr (NX > 0) rows_to_lds<NX>(xv, xs, lane);
        if constexpr (SD.n_y > 0) rows_to_lds<SD.n_y>(yv, ys, lane);
    }
    __syncthreads();
    // register copies of the skill image and the QP options (see pinv_solve_static_kernel), or their values
    constexpr QpImg<SD> kValues = qp_values_or_zero<SD, IMGV>();     // (a local constant: its loads fold to immediates)
    const Img<SD> Sreg = VALUES ? kValues.img : *(const Img<SD>*)lds;
    const QpTail Treg = VALUES ? kValues.tail : *(const QpTail*)((const char*)lds + LY::TAIL_OFF);
    const Img<SD>* __restrict__ S = &Sreg;
    const QpTail* __restrict__ T = &Treg;
    const double* ysl = ys + lane * SD.n_y;
    double z[N];
    state_from_lds<NQ, NX>(zs, xs, lane, z);
    __builtin_amdgcn_sched_barrier(0);

    double v[N], sl[LY::NSA];
    const TickArgs& tk = PT ? *reinterpret_cast<const TickArgs*>(
                                  t_inst + (size_t)(b0 + (valid ? lane : rows_valid - 1)) * 2 * Sreg.n_tslots)
                            : tk_uniform;
    const int status = qp_tick_static<SD>(S, T, tk, z, ysl, lane, valid, slots, v, sl,
                                          hot_set != nullptr ? hot_set + (b0 + lane) : nullptr, use_hot != 0);
    const unsigned bad = (status == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
    // outputs through LDS (row-major rows, coalesced stores)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = nan_or(v[j], bad);
    state_to_lds<NQ, NX>(v, zs, xs, lane);
    if constexpr (NS > 0) {
        double* so = slots + LY::O_SL * WAVE;
        if (slack_out != nullptr) {
#pragma unroll
            for (int k = 0; k < NS; ++k) so[lane * NS + k] = nan_or(sl[k], bad);
        }
    }
    __syncthreads();
    rows_from_lds<NQ>(dq + b0 * NQ, rows_valid, zs, lane);
    if constexpr (NX > 0) {
        if (dx != nullptr) rows_from_lds<NX>(dx + b0 * NX, rows_valid, xs, lane);
    }
    if constexpr (NS > 0) {
        if (slack_out != nullptr) rows_from_lds<NS>(slack_out + b0 * NS, rows_valid, slots + LY::O_SL * WAVE, lane);
    }
    if (status_out != nullptr && valid) status_out[b0 + lane] = status;
}

template <const ShapeDesc& SD>
__global__ __launch_bounds__(WAVE) void qp_solve_static_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const int use_hot,
    const TickArgs tk)
{
    qp_solve_static_body<SD, false>(img_g, q, y, dq, slack_out, status_out, B, x, dx, hot_set, use_hot, tk, nullptr);
}

template <const ShapeDesc& SD, class IMGV>
constexpr QpImg<SD> qp_values_or_zero()
{
    if constexpr (std::is_void<IMGV>::value) return QpImg<SD>{};
    else return IMGV::value;
}

// the per-tick kernel with the skill's numbers and the QP options compiled in (clik_qp_attach_value_kernel)
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) void qp_solve_static_values_kernel(
    const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const int use_hot,
    const TickArgs tk)
{
    qp_solve_static_body<SD, false, IMGV>(nullptr, q, y, dq, slack_out, status_out, B, x, dx, hot_set, use_hot, tk, nullptr);
}

// ... for the box family without any LDS: its solver keeps everything in registers, the few work-area slots the row
// gathering fills become a private array, every lane loads its own rows and stores its own results
// HOT: a launch whose every instance is hot-started (use_hot set, working sets given) / cold-started: the launcher
// decides, so the tick is compiled with `use_hot` a LITERAL either way - the hot object has no start sweeps and reads the
// partition straight off the working-set word (69 instructions fewer on the executed path than a body that selects on a
// run-time flag, 263 fewer in the object: 5.29 -> 5.05 us per hot tick), the cold one has no hot partition (122 fewer).
// `use_hot` (the kernel argument) is kept for the signature and not read.
template <const ShapeDesc& SD, class IMGV, bool HOT = false>
__device__ __forceinline__ void qp_box_values_body(
    const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const int use_hot,
    const TickArgs& tk)
{
    using LY = QpLayout<SD>;
    static_assert(LY::BOX, "box family only");
    constexpr int N = SD.n, NX = SD.n_x, NQ = N - NX, NS = LY::NS;
    constexpr QpImg<SD> kValues = IMGV::value;
    CLIK_BODY_BEGIN();
    CLIK_PHASE("rows_in");
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * WAVE + lane;
    const bool valid = inst < B;
    const long long row = valid ? inst : B - 1;
    double z[N];
#pragma unroll
    for (int j = 0; j < NQ; ++j) z[j] = q[row * NQ + j];
    if constexpr (NX > 0) {
#pragma unroll
        for (int j = 0; j < NX; ++j) z[NQ + j] = x[row * NX + j];
    }
    // The input_var row and the working-set word are requested HERE, with the robot_var row: ONE memory round trip per
    // tick.  Left to the compiler they were issued where they are first used - behind the sin / cos evaluations and the
    // cold path's branch -, the working-set word a single instruction before `s_waitcnt vmcnt(0)`: a second, fully
    // exposed round trip in the middle of every tick (round 6, found in the listing: profiles/r6_load_placement.md).
    // The fence below keeps the loads above it.
    constexpr int NY = SD.n_y > 0 ? SD.n_y : 0;
    double ydir[NY > 0 ? NY : 1];
    if constexpr (NY > 0) {
#pragma unroll
        for (int k = 0; k < NY; ++k) ydir[k] = y[row * NY + k];
    }
    int32_t hot_word = 0;
    if (HOT || hot_set != nullptr) hot_word = hot_set[row];
    asm volatile("" ::: "memory");
    const double* ysl = ydir;
    double priv[LY::SLOTS];
    double v[N], sl[LY::NSA];
    // (the working set in a register, its address passed unconditionally: a conditional pointer to it would put it into
    // scratch memory)
    const int status = qp_tick_static<SD, 1>(&kValues.img, &kValues.tail, tk, z, ysl, lane, valid, priv, v, sl,
                                             &hot_word, HOT);
    CLIK_PHASE("store");
    if (valid && (HOT || hot_set != nullptr)) hot_set[inst] = hot_word;
    if (valid) {
        const unsigned bad = (status == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
#pragma unroll
        for (int j = 0; j < NQ; ++j) dq[inst * NQ + j] = nan_or(v[j], bad);
        if constexpr (NX > 0) {
            if (dx != nullptr) {
#pragma unroll
                for (int j = 0; j < NX; ++j) dx[inst * NX + j] = nan_or(v[NQ + j], bad);
            }
        }
        if constexpr (NS > 0) {
            if (slack_out != nullptr) {
#pragma unroll
                for (int k = 0; k < NS; ++k) slack_out[inst * NS + k] = nan_or(sl[k], bad);
            }
        }
        if (status_out != nullptr) status_out[inst] = status;
    }
    CLIK_BODY_END();
}

template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) void qp_solve_static_box_values_kernel(
    const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const int use_hot,
    const TickArgs tk)
{
    qp_box_values_body<SD, IMGV>(q, y, dq, slack_out, status_out, B, x, dx, hot_set, use_hot, tk);
}
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) void qp_solve_static_box_values_hot_kernel(
    const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const TickArgs tk)
{
    qp_box_values_body<SD, IMGV, true>(q, y, dq, slack_out, status_out, B, x, dx, hot_set, 1, tk);
}

// (Measured and retired, tools/experiments/qp_retired.patch: the same body held to two waves per SIMD - it spilled 288 B
// per lane, 30.3 against 25.0 us at 131072 instances, profiles/r4_qp_occ2.txt -; four lanes per instance running the whole
// tick with different relaxation factors - "quad4", profiles/r3_qp_portfolio_study.md -; and launched ticks with four
// lanes per instance that share the sin / cos evaluations - "front4": cold ticks +1.5 - 2 %, hot ticks 0.25 - 0.7 us SLOWER
// although each wave issues 266 instructions fewer, profiles/r5_quad_ab.txt.  The RESIDENT kernel below keeps that
// four-lane front end: there nobody pays for launching four times the waves.)
template <const ShapeDesc& SD>
constexpr bool qp_front4_ok() { return QpLayout<SD>::BOX && SD.uses_fk != 0 && SD.n >= 3 && SD.n <= 8; }

}  // namespace clik
#include "clik_qp_resident.hpp"     // resident ticks of the bound-constrained family (qp_resident_box_front4_kernel)
namespace clik {

// ... with four WAVES per 64 instances, each with its own start of the passes (FOLIO, see qp_box_pas): cold ticks of
// batches up to one block per CU.  Every wave runs the whole tick of its lanes' instances; what it finished it leaves in
// LDS with the key it recorded, and after the block's barrier the first wave stores, per instance, the answer under the
// smallest key.
constexpr int kFolioWaves = 4;
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(kFolioWaves * WAVE) void qp_solve_static_box_folio_values_kernel(
    const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const TickArgs tk,
    const int same_start)
{
    using LY = QpLayout<SD>;
    static_assert(LY::BOX, "box family only");
    constexpr int N = SD.n, NX = SD.n_x, NQ = N - NX, NS = LY::NS, NSA = LY::NSA;
    constexpr QpImg<SD> kValues = IMGV::value;
    __shared__ int key_min[WAVE];
    // the block's four waves work on the SAME 64 instances: wave w evaluates the sines / cosines of state variables
    // 2w and 2w + 1 only and the block shares them through LDS (2 of N evaluations per wave: 200 instructions fewer
    // in each wave's front end)
    constexpr bool SHARE_SC = SD.uses_fk != 0 && N >= 3 && N <= 8;
    __shared__ double sc_lds[SHARE_SC ? 2 * N : 1][WAVE];
    CLIK_BODY_BEGIN();
    const int w = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    if (w == 0) key_min[lane] = 0x7fffffff;
    // ONE memory round trip: this wave's two sin / cos arguments first, then both rows, all requested before anything
    // is waited for (the fence keeps the loads up here).  Round 5's listing had THREE round trips in a row: argument,
    // sin / cos, argument, sin / cos, and only then the rows (round 6: profiles/r6_load_placement.md).
    const long long inst = (long long)blockIdx.x * WAVE + lane;
    const bool valid = inst < B;
    const long long row = valid ? inst : B - 1;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const SinCosK sck = sincos_consts();        // (their scalar loads go out before anything else)
    __builtin_amdgcn_sched_barrier(0);
    double aa[2] = {0.0, 0.0};
    if constexpr (SHARE_SC) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = (2 * wu + k < N) ? 2 * wu + k : N - 1;
            aa[k] = (j < NQ) ? q[row * NQ + j] : x[row * NX + (j - NQ)];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    double z[N];
#pragma unroll
    for (int j = 0; j < NQ; ++j) z[j] = q[row * NQ + j];
    if constexpr (NX > 0) {
#pragma unroll
        for (int j = 0; j < NX; ++j) z[NQ + j] = x[row * NX + j];
    }
    constexpr int NY = SD.n_y > 0 ? SD.n_y : 0;
    double ydir[NY > 0 ? NY : 1];
    if constexpr (NY > 0) {
#pragma unroll
        for (int k = 0; k < NY; ++k) ydir[k] = y[row * NY + k];
    }
    asm volatile("" ::: "memory");
    const double* ysl = ydir;
    if constexpr (SHARE_SC) {
        double sn[2], cs[2];
        sincos_fast(aa[0], sn[0], cs[0], sck);
        sincos_fast(aa[1], sn[1], cs[1], sck);
        const bool huge = (fabs(aa[0]) > kSinCosFastMax) | (fabs(aa[1]) > kSinCosFastMax);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(huge) != 0ull, 0)) {
            if (fabs(aa[0]) > kSinCosFastMax) { const SinCos sc = sincos_slow(aa[0]); sn[0] = sc.s; cs[0] = sc.c; }
            if (fabs(aa[1]) > kSinCosFastMax) { const SinCos sc = sincos_slow(aa[1]); sn[1] = sc.s; cs[1] = sc.c; }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = 2 * wu + k;
            if (j < N) {
                sc_lds[j][lane] = sn[k];
                sc_lds[N + j][lane] = cs[k];
            }
        }
    }
    __syncthreads();
    QpFolio fo;
    fo.key_slot = &key_min[lane];
    fo.sid = w;
    // (same_start: a measuring switch, CLIK_QP_FOLIO_SAME=1 - all four waves start like the lone-wave kernel)
    // the four starts: forward x 3 | reverse x 6 | forward over-relaxed (1.5) x 12 | reverse over-relaxed x 3 - of the
    // sets of four among {forward, reverse} x {plain, relaxed} x {3 ... 24 sweeps} the one with the shortest slowest
    // instance on average over four batches of 16384 bench inputs (numpy model, tools/qp_wave_portfolio_study.py --sets:
    // 4.55 / 4.18 / 5.12 / 4.35 us of sweeps + passes for seeds 0 - 3 against 6.25 / 7.2 / 7.2 / 7.2 of the lone start;
    // round 4's first set - forward x 12 | forward x 6 | reverse x 6 | relaxed x 18 - had 4.55 / 4.55 / 6.45 / 5.3)
    const int strat = w;
    fo.sweeps = same_start ? CLIK_QP_BOX_SWEEPS : ((strat == 0 || strat == 3) ? 3 : ((strat == 1) ? 6 : 12));
    fo.kind = same_start ? 0 : ((strat == 1) ? 1 : ((strat == 2) ? 2 : ((strat == 3) ? 3 : 0)));
    fo.omega = 1.5;
    double priv[LY::SLOTS];
    double v[N], sl[NSA];
    int32_t hot_word = 0;
    int my_key = 0;
    int status;
    if constexpr (SHARE_SC) {
        double sns[N], css[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            sns[j] = sc_lds[j][lane];
            css[j] = sc_lds[N + j][lane];
        }
        status = qp_tick_static<SD, 1, true, true>(&kValues.img, &kValues.tail, tk, z, ysl, lane, valid, priv, v, sl,
                                                   &hot_word, false, &fo, &my_key, sns, css);
    } else {
        status = qp_tick_static<SD, 1, true>(&kValues.img, &kValues.tail, tk, z, ysl, lane, valid, priv, v, sl,
                                             &hot_word, false, &fo, &my_key);
    }
    __syncthreads();             // (every finish of the block is on record)
    // the wave whose key is the smallest on record stores what it holds in its registers
    if (valid && my_key != 0 && key_min[lane] == my_key) {
        const unsigned bad = (status == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
#pragma unroll
        for (int j = 0; j < NQ; ++j) dq[inst * NQ + j] = nan_or(v[j], bad);
        if constexpr (NX > 0) {
            if (dx != nullptr) {
#pragma unroll
                for (int j = 0; j < NX; ++j) dx[inst * NX + j] = nan_or(v[NQ + j], bad);
            }
        }
        if constexpr (NS > 0) {
            if (slack_out != nullptr) {
#pragma unroll
                for (int k = 0; k < NS; ++k) slack_out[inst * NS + k] = nan_or(sl[k], bad);
            }
        }
        if (status_out != nullptr) status_out[inst] = status;
        if (hot_set != nullptr) hot_set[inst] = hot_word;
    }
    CLIK_BODY_END();
}

// ... and its on-device rollout (see qp_rollout_static_kernel): state, working set and Runge-Kutta bookkeeping in
// registers from tick to tick, rows loaded once and stored once by the lane itself
template <const ShapeDesc& SD, class IMGV, bool RK>
__global__ __launch_bounds__(WAVE) void qp_rollout_static_box_values_kernel(
    double* __restrict__ q, const double* __restrict__ y, double* __restrict__ dq, double* __restrict__ slack_out,
    int32_t* __restrict__ status_out, const long long B, const double* __restrict__ tterms, const int n_ticks,
    const double dt, const double max_speed, double* __restrict__ x, double* __restrict__ dx)
{
    using LY = QpLayout<SD>;
    static_assert(LY::BOX, "box family only");
    constexpr int N = SD.n, NX = SD.n_x, NQ = N - NX, NS = LY::NS;
    constexpr QpImg<SD> kValues = IMGV::value;
    constexpr int stages = RK ? 4 : 1;
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * WAVE + lane;
    const bool valid = inst < B;
    const long long row = valid ? inst : B - 1;
    const int nts = kValues.img.n_tslots;
    double z[N];
#pragma unroll
    for (int j = 0; j < NQ; ++j) z[j] = q[row * NQ + j];
    if constexpr (NX > 0) {
#pragma unroll
        for (int j = 0; j < NX; ++j) z[NQ + j] = x[row * NX + j];
    }
    const double* ysl = SD.n_y > 0 ? y + row * SD.n_y : nullptr;
    double v[N], sl[LY::NSA];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = 0.0;
#pragma unroll
    for (int k = 0; k < LY::NSA; ++k) sl[k] = 0.0;
    int32_t hot = 0;
    int worst = 0;
#pragma unroll 1
    for (int tick = 0; tick < n_ticks; ++tick) {
        double z0[N], ks[N];
        bool okl = true;
        if constexpr (RK) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                z0[j] = z[j];
                ks[j] = 0.0;
            }
        }
#pragma unroll 1
        for (int stg = 0; stg < stages; ++stg) {
            const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + ((size_t)tick * stages + stg) * 2 * nts);
            double priv[LY::SLOTS];
            // (one copy of the tick with a run-time `use_hot`: two copies with the flag a literal in each - as the launched and the
            // resident kernels have - measured 85 instructions MORE per tick here, 3.78 against 3.72 us, round 6: the register
            // allocator's doing)
            const int st = qp_tick_static<SD, 1>(&kValues.img, &kValues.tail, tk, z, ysl, lane, valid, priv, v, sl, &hot,
                                                 (tick | stg) > 0);
            worst = st > worst ? st : worst;
            okl = okl & (st != 2);              // an infeasible tick (stage) leaves the state where it was
            if constexpr (!RK) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double d = v[j];
                    if (j < NQ && max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                    v[j] = d;
                    z[j] = okl ? fma(d, dt, z[j]) : z[j];
                }
            } else {
                const double wgt = (stg == 0 || stg == 3) ? 1.0 : 2.0;
                const double cnext = (stg == 2) ? dt : 0.5 * dt;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double d = okl ? v[j] : 0.0;
                    if (j < NQ && max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                    ks[j] = fma(wgt, d, ks[j]);
                    z[j] = fma(d, cnext, z0[j]);
                }
            }
        }
        if constexpr (RK) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double d = ks[j] * (1.0 / 6.0);
                v[j] = okl ? d : v[j];
                z[j] = okl ? fma(d, dt, z0[j]) : z0[j];
            }
        }
    }
    if (valid) {
        const unsigned bad = (worst == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            q[inst * NQ + j] = z[j];
            dq[inst * NQ + j] = nan_or(v[j], bad);
        }
        if constexpr (NX > 0) {
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                x[inst * NX + j] = z[NQ + j];
                dx[inst * NX + j] = nan_or(v[NQ + j], bad);
            }
        }
        if constexpr (NS > 0) {
            if (slack_out != nullptr) {
#pragma unroll
                for (int k = 0; k < NS; ++k) slack_out[inst * NS + k] = nan_or(sl[k], bad);
            }
        }
        if (status_out != nullptr) status_out[inst] = worst;
    }
}

// hipErrorNotSupported outside the box family (the caller then uses the image-reading rollout)
template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_qp_rollout_static_values(const double* d_tterms, int n_ticks, double dt, double max_speed,
                                                  long long B, double* q, const double* y, double* dq, double* slack,
                                                  int32_t* status, double* x, double* dx, hipStream_t stream, int stages)
{
    if constexpr (QpLayout<SD>::BOX) {
        if (SD.n_x != 0 && (x == nullptr || dx == nullptr)) return hipErrorInvalidValue;
        const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
        if (stages == 4)
            hipLaunchKernelGGL((qp_rollout_static_box_values_kernel<SD, IMGV, true>), dim3(grid), dim3(WAVE), 0, stream, q, y,
                               dq, slack, status, B, d_tterms, n_ticks, dt, max_speed, x, dx);
        else
            hipLaunchKernelGGL((qp_rollout_static_box_values_kernel<SD, IMGV, false>), dim3(grid), dim3(WAVE), 0, stream, q, y,
                               dq, slack, status, B, d_tterms, n_ticks, dt, max_speed, x, dx);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

// which value-specialised QP kernel serves a batch (ONE predicate: the launcher below and the label
// clik_jit_qp_value_variant hands to the controllers / bench.py both call it)
enum QpValueKernel { QPV_GENERAL = 0, QPV_LONE, QPV_FOLIO };
inline int current_device_cus()
{
    // (per CURRENT device: a process may drive several, or a partition of one)
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 16 && cached[dev] > 0) return cached[dev];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 16) cached[dev] = cus;
    return cus;
}
template <const ShapeDesc& SD>
inline QpValueKernel qp_values_choice(long long B, int use_hot)
{
    if constexpr (!QpLayout<SD>::BOX) {
        return QPV_GENERAL;
    } else {
        const long long grid = (B + WAVE - 1) / WAVE;
        // (CLIK_QP_FOLIO=0 / 1: the four-waves-per-64-instances kernel for cold ticks of small batches)
        // default: up to ONE block per CU (16384 instances on 256 CUs) - measured per tick against the lone-wave kernel:
        // 10.2 / 11.0 us at 1024 instances, 10.4 / 12.0 at 4096, 11.0 / 12.0 at 8192, 11.2 / 12.1 at 12288, the same at 256
        // and 2048, 11.0 - 11.1 / 11.3 at 16384 (four identical waves cost 12.2 - 12.5 us there - slower waves, not a slower dispatch:
        // tools/stamp_folio.py - and the different starts win 1.3 - 1.6 back, 0.4 of it through the second look half-way
        // through a pass; six input seeds: -7 % on average, every one a gain); CLIK_QP_FOLIO=0 never
        // (profiles/r4_qp_wave_portfolio.txt)
        static const int folio = []() {
            const char* e = getenv("CLIK_QP_FOLIO");
            return e ? ((e[0] == '1') ? 2 : 0) : 1;
        }();
        if (folio != 0 && !use_hot && grid <= (long long)current_device_cus()) return QPV_FOLIO;
        return QPV_LONE;
    }
}
template <const ShapeDesc& SD>
inline const char* qp_values_variant(long long B, int use_hot)
{
    switch (qp_values_choice<SD>(B, use_hot)) {
    case QPV_FOLIO: return "/folio4";
    default: return "";
    }
}

template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_qp_static_values(const TickArgs& tk, long long B, const double* q, const double* x,
                                          const double* y, double* dq, double* dx, double* slack, int32_t* status,
                                          int32_t* hot_set, int use_hot, hipStream_t stream)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    if constexpr (QpLayout<SD>::BOX) {
        static const int folio_same = []() { const char* e = getenv("CLIK_QP_FOLIO_SAME"); return (e && e[0] == '1') ? 1 : 0; }();
        switch (qp_values_choice<SD>(B, use_hot)) {
        case QPV_FOLIO:
            hipLaunchKernelGGL((qp_solve_static_box_folio_values_kernel<SD, IMGV>), dim3(grid), dim3(kFolioWaves * WAVE), 0,
                               stream, q, y, dq, slack, status, B, x, dx, hot_set, tk, folio_same);
            return hipGetLastError();
        default:
            break;
        }
        if (use_hot != 0 && hot_set != nullptr)
            hipLaunchKernelGGL((qp_solve_static_box_values_hot_kernel<SD, IMGV>), dim3(grid), dim3(WAVE), 0, stream, q, y, dq,
                               slack, status, B, x, dx, hot_set, tk);
        else
            hipLaunchKernelGGL((qp_solve_static_box_values_kernel<SD, IMGV>), dim3(grid), dim3(WAVE), 0, stream, q, y, dq,
                               slack, status, B, x, dx, hot_set, use_hot, tk);
        return hipGetLastError();
    }
    constexpr size_t shmem = QpLayout<SD>::LDS_BYTES;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)qp_solve_static_values_kernel<SD, IMGV>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((qp_solve_static_values_kernel<SD, IMGV>), dim3(grid), dim3(WAVE), shmem, stream, q, y, dq, slack,
                       status, B, x, dx, hot_set, use_hot, tk);
    return hipGetLastError();
}

template <const ShapeDesc& SD>
__global__ __launch_bounds__(WAVE) void qp_solve_static_pt_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ x, double* __restrict__ dx, int32_t* __restrict__ hot_set, const int use_hot,
    const double* __restrict__ t_inst)
{
    TickArgs none;      // (never read)
    qp_solve_static_body<SD, true>(img_g, q, y, dq, slack_out, status_out, B, x, dx, hot_set, use_hot, none, t_inst);
}

// n_ticks of (QP tick -> clamp(+-max_speed) -> explicit Euler q += dq dt) in one launch: the host
// loop of the notebooks (ur5_moe2016_example2.ipynb:537-545) for the QP controller.  The working
// set stays in a register from tick to tick (hot start), the skill image and the targets in LDS.
// q is updated in place; dq / slack receive the last tick, status the worst status met.
// RK: classical Runge-Kutta with the controller as the right-hand side (integration_methods.py:17-23; four QP
// solves per tick, the working set hot-started from stage to stage, tterms holds four records per tick) instead of
// explicit Euler; a tick with an infeasible stage leaves the state where it was.
template <const ShapeDesc& SD, bool RK>
__global__ __launch_bounds__(WAVE) void qp_rollout_static_kernel(
    const void* __restrict__ img_g, double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ slack_out, int32_t* __restrict__ status_out, const long long B,
    const double* __restrict__ tterms, const int n_ticks, const double dt, const double max_speed,
    double* __restrict__ x, double* __restrict__ dx)
{
    // x / dx: virtual variables, integrated like the robot variables and never clamped (null without them)
    extern __shared__ double lds[];
    using LY = QpLayout<SD>;
    constexpr int N = SD.n;
    constexpr int NX = SD.n_x, NQ = N - NX;
    constexpr int NS = LY::NS;
    const int lane = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    double* slots = lds + LY::IMG_DOUBLES;
    double* zs = slots + LY::O_Z * WAVE;
    double* ys = slots + LY::O_Y * WAVE;
    typedef double d2 __attribute__((ext_vector_type(2)));
    {
        d2 img[LY::IMG_CHUNKS];
        const d2* src = (const d2*)img_g;
#pragma unroll
        for (int k = 0; k < LY::IMG_CHUNKS; ++k) img[k] = src[k * WAVE + lane];
        double qv[NQ], xv[NX > 0 ? NX : 1], yv[SD.n_y > 0 ? SD.n_y : 1];
        stage_load<NQ>(q + b0 * NQ, NQ, rows_valid, lane, qv);
        if constexpr (NX > 0) stage_load<NX>(x + b0 * NX, NX, rows_valid, lane, xv);
        if constexpr (SD.n_y > 0) stage_load<SD.n_y>(y + b0 * SD.n_y, SD.n_y, rows_valid, lane, yv);
        d2* dst = (d2*)lds;
#pragma unroll
        for (int k = 0; k < LY::IMG_CHUNKS; ++k) dst[k * WAVE + lane] = img[k];
        rows_to_lds<NQ>(qv, zs, lane);
        if constexpr (NX > 0) rows_to_lds<NX>(xv, zs + NQ * WAVE, lane);
        if constexpr (SD.n_y > 0) rows_to_lds<SD.n_y>(yv, ys, lane);
    }
    __syncthreads();
    const Img<SD>* __restrict__ S = (const Img<SD>*)lds;
    const QpTail* __restrict__ T = (const QpTail*)((const char*)lds + LY::TAIL_OFF);
    const int nts = S->n_tslots;
    const double* ysl = ys + lane * SD.n_y;
    double* xs = zs + NQ * WAVE;
    double z[N];
    state_from_lds<NQ, NX>(zs, xs, lane, z);
    double v[N], sl[LY::NSA];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = 0.0;
#pragma unroll
    for (int k = 0; k < LY::NSA; ++k) sl[k] = 0.0;
    int32_t hot = 0;
    int worst = 0;
    if constexpr (!RK) {
#pragma unroll 1
        for (int tick = 0; tick < n_ticks; ++tick) {
            asm volatile("" ::: "memory");      // (keeps the image reads inside the loop, see pinv_rollout_static_kernel)
            const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + (size_t)tick * 2 * nts);
            const int st = qp_tick_static<SD>(S, T, tk, z, ysl, lane, valid, slots, v, sl, &hot, tick > 0);
            worst = st > worst ? st : worst;
            const bool okl = st != 2;           // an infeasible tick leaves the state where it is
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double d = v[j];
                if (j < NQ && max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                v[j] = d;
                z[j] = okl ? fma(d, dt, z[j]) : z[j];
            }
        }
    } else {
        double* z0s = slots + LY::SLOTS * WAVE;      // [N][64] state at the start of the tick, [N][64] sum of w_i k_i
        double* kss = z0s + N * WAVE;
#pragma unroll 1
        for (int tick = 0; tick < n_ticks; ++tick) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                z0s[j * WAVE + lane] = z[j];
                kss[j * WAVE + lane] = 0.0;
            }
            bool okl = true;
#pragma unroll 1
            for (int stg = 0; stg < 4; ++stg) {
                asm volatile("" ::: "memory");
                const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + ((size_t)tick * 4 + stg) * 2 * nts);
                const int st = qp_tick_static<SD>(S, T, tk, z, ysl, lane, valid, slots, v, sl, &hot, (tick | stg) > 0);
                worst = st > worst ? st : worst;
                okl = okl & (st != 2);
                const double wgt = (stg == 0 || stg == 3) ? 1.0 : 2.0;
                const double cnext = (stg == 2) ? dt : 0.5 * dt;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double d = okl ? v[j] : 0.0;
                    if (j < NQ && max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                    kss[j * WAVE + lane] = fma(wgt, d, kss[j * WAVE + lane]);
                    z[j] = fma(d, cnext, z0s[j * WAVE + lane]);
                }
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double d = kss[j * WAVE + lane] * (1.0 / 6.0);
                v[j] = okl ? d : v[j];
                z[j] = okl ? fma(d, dt, z0s[j * WAVE + lane]) : z0s[j * WAVE + lane];
            }
        }
    }
    const unsigned bad = (worst == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
    __syncthreads();
    state_to_lds<NQ, NX>(z, zs, xs, lane);
    __syncthreads();
    rows_from_lds<NQ>(q + b0 * NQ, rows_valid, zs, lane);
    if constexpr (NX > 0) rows_from_lds<NX>(x + b0 * NX, rows_valid, xs, lane);
    __syncthreads();
    {
        double vb[N];
#pragma unroll
        for (int j = 0; j < N; ++j) vb[j] = nan_or(v[j], bad);
        state_to_lds<NQ, NX>(vb, zs, xs, lane);
    }
    if constexpr (NS > 0) {
        double* so = slots + LY::O_SL * WAVE;
        if (slack_out != nullptr) {
#pragma unroll
            for (int k = 0; k < NS; ++k) so[lane * NS + k] = nan_or(sl[k], bad);
        }
    }
    __syncthreads();
    rows_from_lds<NQ>(dq + b0 * NQ, rows_valid, zs, lane);
    if constexpr (NX > 0) rows_from_lds<NX>(dx + b0 * NX, rows_valid, xs, lane);
    if constexpr (NS > 0) {
        if (slack_out != nullptr) rows_from_lds<NS>(slack_out + b0 * NS, rows_valid, slots + LY::O_SL * WAVE, lane);
    }
    if (status_out != nullptr && valid) status_out[b0 + lane] = worst;
}

// stages: controller evaluations per tick (1 explicit Euler, 4 classical Runge-Kutta)
typedef hipError_t (*qp_static_rollout_fn)(const void*, const double*, int, double, double, long long, double*,
                                           const double*, double*, double*, int32_t*, double*, double*, hipStream_t,
                                           int);

template <const ShapeDesc& SD>
inline hipError_t launch_qp_rollout_static(const void* d_img, const double* d_tterms, int n_ticks, double dt,
                                           double max_speed, long long B, double* q, const double* y, double* dq,
                                           double* slack, int32_t* status, double* x, double* dx,
                                           hipStream_t stream, int stages = 1)
{
    if (SD.n_x != 0 && (x == nullptr || dx == nullptr)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    if (stages == 4) {
        constexpr size_t shmem = QpLayout<SD>::LDS_BYTES + (size_t)2 * SD.n * WAVE * sizeof(double);
        static_assert(shmem <= 160u * 1024u, "Runge-Kutta QP rollout needs more LDS than a CU has");
        if (shmem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)qp_rollout_static_kernel<SD, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((qp_rollout_static_kernel<SD, true>), dim3(grid), dim3(WAVE), shmem, stream, d_img, q, y, dq,
                           slack, status, B, d_tterms, n_ticks, dt, max_speed, x, dx);
        return hipGetLastError();
    }
    constexpr size_t shmem = QpLayout<SD>::LDS_BYTES;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)qp_rollout_static_kernel<SD, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((qp_rollout_static_kernel<SD, false>), dim3(grid), dim3(WAVE), shmem, stream, d_img, q, y, dq,
                       slack, status, B, d_tterms, n_ticks, dt, max_speed, x, dx);
    return hipGetLastError();
}

// t_inst: null (tk serves the whole batch) or one time-slot record per instance ([B][2 * n_tslots], device)
typedef hipError_t (*qp_static_fn)(const void*, const TickArgs&, long long, const double*, const double*,
                                   const double*, double*, double*, double*, int32_t*, int32_t*, int, hipStream_t,
                                   const double*);

template <const ShapeDesc& SD>
inline hipError_t launch_qp_static(const void* d_img, const TickArgs& tk, long long B, const double* q,
                                   const double* x, const double* y, double* dq, double* dx, double* slack,
                                   int32_t* status, int32_t* hot_set, int use_hot, hipStream_t stream,
                                   const double* t_inst = nullptr)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    constexpr size_t shmem = QpLayout<SD>::LDS_BYTES;
    static_assert(shmem <= 160u * 1024u, "static QP kernel needs more LDS than a CU has");
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)qp_solve_static_kernel<SD>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    if (t_inst != nullptr) {
        if (shmem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)qp_solve_static_pt_kernel<SD>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((qp_solve_static_pt_kernel<SD>), dim3(grid), dim3(WAVE), shmem, stream, d_img, q, y, dq,
                           slack, status, B, x, dx, hot_set, use_hot, t_inst);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((qp_solve_static_kernel<SD>), dim3(grid), dim3(WAVE), shmem, stream, d_img, q, y, dq, slack,
                       status, B, x, dx, hot_set, use_hot, tk);
    return hipGetLastError();
}

}  // namespace clik
