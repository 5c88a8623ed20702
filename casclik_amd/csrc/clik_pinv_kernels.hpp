// Batched PseudoInverseController tick on gfx950.
//
// Replaces, for B instances per launch, the per-tick body of
//   PseudoInverseController.solve          casclik/controllers/pseudo_inverse.py:512-556
// i.e. the evaluation of the per-mode CasADi functions built by
//   get_problem_expressions                 :259-451
// with damped pseudo-inverses (:92-105), the mode table (:107-130) and the
// tangent-cone tests (:132-257).
//
// Algebra (equal to the reference's up to fp64 rounding):
//   reference:  v += (I - pinv(Ja) rJa) * pinv(Ji) * des          (full matrices)
//   here:       w  = pinv(Ji) des   by ONE solve with (Ji Ji^T + lam I)   [wide]
//                                              or   (Ji^T Ji + lam I)     [tall]
//               w -= pinv(Ja) (rJa w) by ONE solve with the stacked Gram matrix
//   the reference's wide/tall branch choice (cols >= rows) is kept, so the
//   conditioning of every solve is the reference's.  rJa = diag(s) Ja with
//   s in {0,1} (multidim activation S, :289-298) is kept as a bit mask (wide
//   stack) or as the second Gram matrix C = Ja^T diag(s) Ja (tall stack).
//   The first EqualityConstraint is processed twice and stacked twice exactly
//   as :317-326 + :382-396 do.
//
// One source, two instantiation styles (clik_device.hpp, "shape policies"):
//   StaticShape<SD>  sizes / classes / feature flags / options are compile-time:
//                    all guards fold, modes and tasks are unrolled, registers
//                    are statically indexed.  Used when the skill matches one
//                    of the AOT shapes in kShapes[].
//   DynShape         same code with run-time guards: any skill within the
//                    limits of include/clik.h.
#pragma once
#include "clik_device.hpp"
#include "clik_pinv_static.hpp"

namespace clik {

// ---- stacked active Jacobians -------------------------------------------------
// wide form: rows in LDS slots [k*N + j], k < r <= cap, per-lane activation bits
// gram form: G = lam*I + Ja^T Ja (slots [0, NT)), C = Ja^T diag(s) Ja (slots [NT, 2NT))
struct Stack {
    int      r;        // stacked rows (wave-uniform; compile-time in static shapes)
    bool     gram;     // wave-uniform
    uint32_t sbits;    // per lane: activation flag of wide rows
};

template <int N>
__device__ __forceinline__ void stack_push(Stack& st, double* wk, int lane, const int n, const int cap,
                                           const double lam, const double (&J)[N][N], const int m,
                                           const uint32_t srow, const int times)
{
    constexpr int NT = N * (N + 1) / 2;
    if (!st.gram && st.r + times * m <= cap) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (rep < times) {
                const int r0 = st.r;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    if (i < m) {
#pragma unroll
                        for (int j = 0; j < N; ++j)
                            if (j < n) wk[((r0 + i) * N + j) * WAVE + lane] = J[i][j];
                    }
                }
                st.sbits |= (srow & ((1u << m) - 1u)) << r0;
                st.r += m;
            }
        }
        return;
    }
    // (convert to and) accumulate in Gram form.  The row area overlaps both
    // Gram areas, so both matrices are completed in registers before either
    // is stored.
    double Gm[NT], Cm[NT];
    double* gdst = wk + lane;
    double* cdst = wk + (size_t)NT * WAVE + lane;
    if (st.gram) {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            Gm[a] = gdst[a * WAVE];
            Cm[a] = cdst[a * WAVE];
        }
    } else {
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                Gm[tri(a, b)] = (a == b && a < n) ? lam : 0.0;
                Cm[tri(a, b)] = 0.0;
            }
        const int r_old = st.r;
        const uint32_t old_bits = st.sbits;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k < r_old) {
                double row[N];
#pragma unroll
                for (int j = 0; j < N; ++j) row[j] = (j < n) ? wk[(k * N + j) * WAVE + lane] : 0.0;
                const double sk = ((old_bits >> k) & 1u) ? 1.0 : 0.0;
#pragma unroll
                for (int a = 0; a < N; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        const double pr = row[a] * row[b];
                        Gm[tri(a, b)] += pr;
                        Cm[tri(a, b)] = fma(sk, pr, Cm[tri(a, b)]);
                    }
            }
        }
    }
    const double tf = (double)times;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < m) {
            const double si = ((srow >> i) & 1u) ? tf : 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    const double pr = J[i][a] * J[i][b];
                    Gm[tri(a, b)] = fma(tf, pr, Gm[tri(a, b)]);
                    Cm[tri(a, b)] = fma(si, pr, Cm[tri(a, b)]);
                }
        }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        gdst[a * WAVE] = Gm[a];
        cdst[a * WAVE] = Cm[a];
    }
    st.gram = true;
    st.r += times * m;
}

// w <- w - pinv(vstack Ja) * (vstack rJa) * w     (pseudo_inverse.py:387-392)
template <int N>
__device__ __forceinline__ void stack_project(const Stack& st, const double* wk, int lane, const int n,
                                              const double lam, double (&w)[N])
{
    constexpr int NT = N * (N + 1) / 2;
    double L[NT], rd[N], u[N];
    if (st.gram) {
        const double* G = wk + lane;
        const double* Cc = wk + (size_t)NT * WAVE + lane;
#pragma unroll
        for (int a = 0; a < N; ++a) u[a] = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const double c = Cc[tri(a, b) * WAVE];
                u[a] = fma(c, w[b], u[a]);
                if (b != a) u[b] = fma(c, w[a], u[b]);
            }
#pragma unroll
        for (int a = 0; a < NT; ++a) L[a] = G[a * WAVE];
        ldl_factor<N>(L, rd, n);
        ldl_solve<N>(L, rd, u, n);
#pragma unroll
        for (int a = 0; a < N; ++a) w[a] -= u[a];
    } else {
        const int r = st.r;
        // B = Ja Ja^T + lam I (r x r), u = diag(s) Ja w
#pragma unroll
        for (int i = 0; i < N; ++i) {
            u[i] = 0.0;
            if (i < r) {
                double ri[N];
#pragma unroll
                for (int j = 0; j < N; ++j) ri[j] = (j < n) ? wk[(i * N + j) * WAVE + lane] : 0.0;
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) s = fma(ri[j], w[j], s);
                u[i] = ((st.sbits >> i) & 1u) ? s : 0.0;
#pragma unroll
                for (int k = 0; k <= i; ++k) {
                    double acc = (k == i) ? lam : 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (j < n) acc = fma(ri[j], wk[(k * N + j) * WAVE + lane], acc);
                    L[tri(i, k)] = acc;
                }
            }
        }
        ldl_factor<N>(L, rd, r);
        ldl_solve<N>(L, rd, u, r);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k < r) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j < n) w[j] = fma(-u[k], wk[(k * N + j) * WAVE + lane], w[j]);
            }
        }
    }
}

// in-tangent-cone test of an inactive SetConstraint (pseudo_inverse.py:162-185, :222-252)
template <int N>
__device__ __forceinline__ bool in_tangent_cone(const clik_task& t, const int m, const int n,
                                                const double (&e)[N], const double (&J)[N][N],
                                                const double (&Jt)[N], const double (&v)[N])
{
    double de[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
        if (i < m) {
            s = Jt[i];
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j < n) s = fma(J[i][j], v[j], s);
        }
        de[i] = s;
    }
    if (m == 1) {
        const double ev = e[0];
        if (t.set_min[0] - ev < 1e-12) return (ev - t.set_max[0] < 1e-12) ? true : (de[0] < 0.0);
        return de[0] > 0.0;
    }
    bool inside = true, corner = true;
    double od = 0.0, nde = 0.0, nout = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < m) {
            const double le = e[i] - t.set_min[i];
            const double ue = e[i] - t.set_max[i];
            if (!(le >= 1e-12) || !(ue <= 1e-12)) inside = false;
            const double sl = (le > 0.0) - (le < 0.0);
            const double su = (ue > 0.0) - (ue < 0.0);
            if (sl != su) corner = false;
            const double out = 0.5 * (sl + su);
            od = fma(out, de[i], od);
            nde = fma(de[i], de[i], nde);
            nout = fma(out, out, nout);
        }
    }
    bool going_in;
    if (corner) {
        const double dists = (sqrt(nde) + 1e-10) * sqrt(nout);
        going_in = (od < 0.0) ? (fabs(od) / dists < 0.70710678118654757) : false;
    } else {
        going_in = od < 0.0;
    }
    return inside ? true : going_in;
}

// ---- one mode of the controller for the lane's instance --------------------------
// Returns the candidate velocity v and whether every inactive set is in its
// tangent cone.  `act` = activation bit mask of the mode.
template <int N, class SH>
__device__ __forceinline__ bool pinv_mode(const DevSkill* __restrict__ S, const TickArgs& tk,
                                          const uint32_t act, const Kin<N>& K, const double (&z)[N],
                                          const double* ys, double* wk, int lane, double (&v)[N])
{
    constexpr int NT = N * (N + 1) / 2;
    const clik_skill_desc& D = S->d;
    const int n = SH::n(S);
    const int n_tasks = SH::n_tasks(S);
    const bool ff = SH::feedforward(S);
    const bool multidim = SH::multidim(S);
    const bool conv_last = SH::conv_last(S);
    const bool standard = SH::standard(S);
    const double lam = standard ? 0.0 : S->po.damping_factor;
    const int cap = standard ? n - 1 : n;      // rows for which the stacked pinv is "wide"

#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = 0.0;
    Stack st;
    st.r = 0;
    st.gram = false;
    st.sbits = 0u;
    bool ok = true;

    // number of SetConstraints before task ti (compile-time in static shapes)
    auto set_index = [&](int ti) __attribute__((always_inline)) {
        int k = 0;
#pragma unroll
        for (int q = 0; q < CLIK_MAX_TASKS; ++q)
            if (q < ti && SH::cls(S, q) == CLIK_CLS_SET) ++k;
        return k;
    };

    // ---- pass 0: controller algebra over the priority-ordered constraints
    auto algebra = [&](auto tic) __attribute__((always_inline)) {
        const int ti = tic;
        const int cls = SH::cls(S, ti);
        const int m = SH::m(S, ti);
        if (cls == CLIK_CLS_VELSET) return;                 // no branch in the reference (:274-443)
        const bool is_set = cls == CLIK_CLS_SET;
        const bool active = is_set && ((act >> set_index(ti)) & 1u);
        if (is_set && !active) return;                      // tested in pass 1
        const clik_task& t = D.tasks[ti];

        double e[N], J[N][N], Jt[N];
        task_eval<N, N>(S, ti, m, SH::is_static ? SH::flags(S, ti) : -1, SH::all_affine(S), tk, K, z, ys,
                        lane, n, e, J, Jt);

        const bool is_first = (st.r == 0);
        const bool is_last = (ti == n_tasks - 1);
        const bool conv = is_set && is_last && conv_last;
        uint32_t srow = 0xffffffffu;
        if (multidim && is_set) {
            srow = 0u;
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i < m && ((e[i] - t.set_max[i] > 0.0) || (e[i] - t.set_min[i] < 0.0))) srow |= 1u << i;
        }
        const bool contributes = (cls == CLIK_CLS_EQ) || (cls == CLIK_CLS_VELEQ) || conv;
        if (!contributes) {
            // active set (pseudo_inverse.py:398-405): rows only
            stack_push<N>(st, wk, lane, n, cap, lam, J, m, srow, 1);
            return;
        }
        double des[N];
        if (cls == CLIK_CLS_EQ) {
            double ke[N];
            gain_apply<N>(t, m, e, ke);
#pragma unroll
            for (int i = 0; i < N; ++i) des[i] = -ke[i];
        } else if (cls == CLIK_CLS_VELEQ) {
#pragma unroll
            for (int i = 0; i < N; ++i) des[i] = (i < m) ? t.target[i] : 0.0;
        } else {
            double d0[N];
#pragma unroll
            for (int i = 0; i < N; ++i) d0[i] = (i < m) ? t.set_max[i] - e[i] : 0.0;
            gain_apply<N>(t, m, d0, des);
        }
        if (ff) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (i < m) des[i] -= Jt[i];
        }

        // w = pinv(J) des  (pseudo_inverse.py:92-105)
        double w[N], L[NT], rd[N];
        const bool wide = standard ? (m < n) : (n >= m);
        bool have_factor = false;
        if (SH::const_j(S, ti)) {
            // state-independent Jacobian: pinv(J) was formed once on the host
            const double* P = S->cpinv[ti];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = 0.0;
                if (j < n) {
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i < m) s = fma(P[j * CLIK_MAX_M + i], des[i], s);
                }
                w[j] = s;
            }
        } else if (wide) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (i < m) {
#pragma unroll
                    for (int k = 0; k <= i; ++k) {
                        double acc = (k == i) ? lam : 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j)
                            if (j < n) acc = fma(J[i][j], J[k][j], acc);
                        L[tri(i, k)] = acc;
                    }
                }
            }
            ldl_factor<N>(L, rd, m);
            ldl_solve<N>(L, rd, des, m);
            have_factor = true;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i < m) s = fma(J[i][j], des[i], s);
                w[j] = s;
            }
        } else {
#pragma unroll
            for (int a = 0; a < N; ++a) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i < m) s = fma(J[i][a], des[i], s);
                w[a] = s;
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    double acc = (a == b && a < n) ? lam : 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i < m) acc = fma(J[i][a], J[i][b], acc);
                    L[tri(a, b)] = acc;
                }
            }
            ldl_factor<N>(L, rd, n);
            ldl_solve<N>(L, rd, w, n);
        }

        const bool quirk = is_first && cls == CLIK_CLS_EQ;
        if (is_first) {
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] += w[j];
        }
        if (quirk && have_factor) {
            // second processing of the first EqualityConstraint (:382-396): the
            // stack is [J] itself, so the factor of J J^T + lam I is reused:
            //   w2 = w - J^T A^{-1} (J w)
            double u[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
                if (i < m) {
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (j < n) s = fma(J[i][j], w[j], s);
                }
                u[i] = s;
            }
            ldl_solve<N>(L, rd, u, m);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = w[j];
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i < m) s = fma(-J[i][j], u[i], s);
                v[j] += s;
            }
            stack_push<N>(st, wk, lane, n, cap, lam, J, m, 0xffffffffu, 2);
        } else if (quirk) {
            // generic route: push, project, push
            stack_push<N>(st, wk, lane, n, cap, lam, J, m, 0xffffffffu, 1);
            stack_project<N>(st, wk, lane, n, lam, w);
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] += w[j];
            stack_push<N>(st, wk, lane, n, cap, lam, J, m, 0xffffffffu, 1);
        } else {
            if (!is_first) {
                stack_project<N>(st, wk, lane, n, lam, w);
#pragma unroll
                for (int j = 0; j < N; ++j) v[j] += w[j];
            }
            stack_push<N>(st, wk, lane, n, cap, lam, J, m, (conv && multidim) ? srow : 0xffffffffu, 1);
        }
    };

    // ---- pass 1: tangent cones of the inactive sets with the candidate velocity
    auto cones = [&](auto tic) __attribute__((always_inline)) {
        const int ti = tic;
        if (SH::cls(S, ti) != CLIK_CLS_SET) return;
        if ((act >> set_index(ti)) & 1u) return;
        const int m = SH::m(S, ti);
        double e[N], J[N][N], Jt[N];
        task_eval<N, N>(S, ti, m, SH::is_static ? SH::flags(S, ti) : -1, SH::all_affine(S), tk, K, z, ys,
                        lane, n, e, J, Jt);
        ok = ok && in_tangent_cone<N>(D.tasks[ti], m, n, e, J, Jt, v);
    };

    if constexpr (SH::is_static) {
        static_for<0, SH::n_tasks(nullptr)>(algebra);
        static_for<0, SH::n_tasks(nullptr)>(cones);
    } else {
        for (int ti = 0; ti < n_tasks; ++ti) algebra(ti);
        for (int ti = 0; ti < n_tasks; ++ti) cones(ti);
    }
    return ok;
}

constexpr int shape_n_sets(const ShapeDesc& sd)
{
    int k = 0;
    for (int q = 0; q < sd.n_tasks; ++q) k += sd.cls[q] == CLIK_CLS_SET;
    return k;
}

// activation mask of the k-th mode in the reference's scan order (pseudo_inverse.py:107-130:
// sorted by number of active sets, then by value; set 0 = bit 0) - same table as the host's
constexpr unsigned shape_mode_act(const ShapeDesc& sd, int k)
{
    const int ns = shape_n_sets(sd);
    int idx = 0;
    for (int pc = 0; pc <= ns; ++pc)
        for (unsigned v = 0; v < (1u << ns); ++v) {
            int c = 0;
            for (int b = 0; b < ns; ++b) c += (v >> b) & 1u;
            if (c == pc) {
                if (idx == k) return v;
                ++idx;
            }
        }
    return 0u;
}
constexpr int kStaticMaxSets = 6;       // up to 64 mode bodies per kernel (instantiated in scan order)

// ---- one controller tick: FK once, then the mode scan (pseudo_inverse.py:530-555)
template <int N, class SH>
__device__ __forceinline__ void pinv_tick(const DevSkill* __restrict__ S, const TickArgs& tk,
                                          const double (&z)[N], const double* zs, const double* ys,
                                          double* wk, int lane, bool valid, double (&vout)[N],
                                          int& acc_mode)
{
    Kin<N> K;
    if (SH::uses_fk(S)) {
        forward_kinematics<N>(S, zs, wk, lane, K);
        if (SH::quat_src(S) != 0) orientation_feature<N>(S, ys, lane, K);
    }
    bool done = !valid;
    acc_mode = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = 0.0;

    auto one_mode = [&](auto mkc, const uint32_t act) __attribute__((always_inline)) {
        const int mk = mkc;
        double v[N];
        const bool ok = pinv_mode<N, SH>(S, tk, act, K, z, ys, wk, lane, v);
        if (!done && ok) {
            done = true;
            acc_mode = mk;
#pragma unroll
            for (int j = 0; j < N; ++j) vout[j] = v[j];
        }
    };

    {
        const int n_modes = S->n_modes;
        for (int mk = 0; mk < n_modes; ++mk) {
            if (__ballot(!done) == 0ull) break;
            one_mode(mk, S->act[mk]);
        }
    }
}

// LDS slots (doubles per lane) the kernels need for a skill
__host__ __device__ inline int pinv_lds_slots(int N, int ny)
{
    int wk = N * (N + 1);
    if (6 * N > wk) wk = 6 * N;
    return N + ny + wk;
}

template <int N, class SH>
__global__ __launch_bounds__(WAVE) void pinv_solve_kernel(
    const DevSkill* __restrict__ S0, const WarmArgs wa, const TickArgs tk, const long long B,
    const double* __restrict__ q, const double* __restrict__ x, const double* __restrict__ y,
    double* __restrict__ dq, double* __restrict__ dx, int32_t* __restrict__ mode_out)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    // pull the descriptor into the scalar cache (one round trip) before walking it
    const DevSkill* __restrict__ S = warm_descriptor(S0, wa);
    const int n = SH::n(S), nq = S->d.n_q, nx = S->d.n_x, ny = S->d.n_y;
    double* zs = lds;
    double* ys = zs + N * WAVE;
    double* wk = ys + ny * WAVE;

    // zero-fill so tail lanes compute on defined data
    if (rows_valid < WAVE) {
#pragma unroll
        for (int j = 0; j < N; ++j) zs[j * WAVE + lane] = 0.0;
        for (int k = 0; k < ny; ++k) ys[k * WAVE + lane] = 0.0;
        __syncthreads();
    }
    stage_in_dyn(q + b0 * nq, nq, rows_valid, zs, lane);
    if (nx > 0) stage_in_dyn(x + b0 * nx, nx, rows_valid, zs + nq * WAVE, lane);
    if (ny > 0) stage_in_dyn(y + b0 * ny, ny, rows_valid, ys, lane);
    __syncthreads();

    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = (j < n) ? zs[j * WAVE + lane] : 0.0;

    double vout[N];
    int acc_mode;
    pinv_tick<N, SH>(S, tk, z, zs, ys, wk, lane, valid, vout, acc_mode);

    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (j < n) zs[j * WAVE + lane] = vout[j];
    __syncthreads();
    stage_out_dyn(dq + b0 * nq, nq, rows_valid, zs, lane);
    if (nx > 0) stage_out_dyn(dx + b0 * nx, nx, rows_valid, zs + nq * WAVE, lane);
    if (mode_out != nullptr && valid) mode_out[b0 + lane] = acc_mode;
}

// n_ticks of  solve -> clamp -> explicit Euler  in one launch (the loop every
// reference notebook runs on the host, ur5_moe2016_example2.ipynb:537-545).
template <int N, class SH>
__global__ __launch_bounds__(WAVE) void pinv_rollout_kernel(
    const DevSkill* __restrict__ S0, const WarmArgs wa, const double* __restrict__ tterms, const int n_ticks,
    const double dt, const double max_speed, const long long B,
    double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    const DevSkill* __restrict__ S = warm_descriptor(S0, wa);
    const int n = SH::n(S), nq = S->d.n_q, ny = S->d.n_y;
    const int nts = S->d.n_tslots;
    double* zs = lds;
    double* ys = zs + N * WAVE;
    double* wk = ys + ny * WAVE;
#pragma unroll
    for (int j = 0; j < N; ++j) zs[j * WAVE + lane] = 0.0;
    for (int k = 0; k < ny; ++k) ys[k * WAVE + lane] = 0.0;
    __syncthreads();
    stage_in_dyn(q + b0 * nq, nq, rows_valid, zs, lane);
    if (ny > 0) stage_in_dyn(y + b0 * ny, ny, rows_valid, ys, lane);
    __syncthreads();
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = (j < n) ? zs[j * WAVE + lane] : 0.0;
    double vout[N];
    int acc_mode = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = 0.0;
    for (int tick = 0; tick < n_ticks; ++tick) {
        TickArgs tk;
        for (int k = 0; k < 2 * nts; ++k) tk.tv[k] = tterms[(size_t)tick * 2 * nts + k];
        pinv_tick<N, SH>(S, tk, z, zs, ys, wk, lane, valid, vout, acc_mode);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < n) {
                double d = vout[j];
                if (max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                vout[j] = d;
                z[j] = fma(d, dt, z[j]);
                zs[j * WAVE + lane] = z[j];
            }
        }
    }
    __syncthreads();
    stage_out_dyn(q + b0 * nq, nq, rows_valid, zs, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j)
        if (j < n) zs[j * WAVE + lane] = vout[j];
    __syncthreads();
    stage_out_dyn(dq + b0 * nq, nq, rows_valid, zs, lane);
    if (mode_out != nullptr && valid) mode_out[b0 + lane] = acc_mode;
}

// ---- kernel body time (only in builds with -DCLIK_BODY_STAMPS, tools/stamp_body.py; never in the shipped
// library): every wave stamps s_memrealtime (the 100 MHz constant clock all XCDs share) when it starts and after
// its last store has landed; body of a launch = max(end) - min(start) over its waves, read back for the LAST launch
// of a back-to-back sequence.
#ifdef CLIK_BODY_STAMPS
__device__ unsigned long long g_clik_body[2 * 32768];
#define CLIK_BODY_STAMP(slot, drain)                                                            \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  \
        unsigned long long t_;                                                                  \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        const unsigned w_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                        \
        if ((threadIdx.x & 63u) == 0u && w_ < 32768u) g_clik_body[2u * w_ + (slot)] = t_;        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#if CLIK_BODY_STAMPS + 0 == 2
// LIGHT stamps (-DCLIK_BODY_STAMPS=2): only block 0 stamps its start and only every eighth block (and the last one) its
// end - one scalar compare and branch for everybody else - so that the stamped tick is the shipped tick (round 3's
// full stamps cost 0.29 us of a 3.98 us tick, which left the shipped build's body an inference).  Slots not written
// stay zero; tools/stamp_body.py --light takes min(start) / max(end) over the written ones.
#define CLIK_BODY_BEGIN()                                                                       \
    do {                                                                                        \
        if (blockIdx.x == 0u) CLIK_BODY_STAMP(0, 0);                                            \
    } while (0)
#define CLIK_BODY_END()                                                                         \
    do {                                                                                        \
        if ((blockIdx.x & 7u) == 7u || blockIdx.x + 1u == gridDim.x) CLIK_BODY_STAMP(1, 1);     \
    } while (0)
#else
#define CLIK_BODY_BEGIN() CLIK_BODY_STAMP(0, 0)
#define CLIK_BODY_END() CLIK_BODY_STAMP(1, 1)
#endif
#else
#define CLIK_BODY_BEGIN()
#define CLIK_BODY_END()
#endif

// ---- diagnostic time stamps (only in builds with -DCLIK_STAMPS, never in the shipped
// library): s_memtime at phase boundaries of the static kernel, one record per block,
// written to a buffer no other code reads.
#ifdef CLIK_STAMPS
__device__ unsigned long long g_clik_stamps[8 * 4096];
#define CLIK_STAMP(k)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_clik_stamps[blockIdx.x * 8 + (k)] = t_;  \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
// stamp taken by wave w of a multi-wave block (threadIdx.x == 64 w writes)
#define CLIK_STAMP_W(w, k)                                                                    \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        if (threadIdx.x == 64 * (w) && blockIdx.x < 4096) g_clik_stamps[blockIdx.x * 8 + (k)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
// stamp after everything in flight has landed (perturbs the schedule: shares only)
#define CLIK_STAMP_DRAINED(k)                                                                 \
    do {                                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
        CLIK_STAMP(k);                                                                        \
    } while (0)
#else
#define CLIK_STAMP(k)
#define CLIK_STAMP_W(w, k)
#define CLIK_STAMP_DRAINED(k)
#endif

// ---- row-major staging of the shape-specialised kernels --------------------------------
// The wave's [64][W] block keeps its global (row-major) layout in LDS: the coalesced
// chunk i of lane l lands at double i*64 + l (a straight copy, no index arithmetic),
// and lane l then reads its own row at doubles l*W .. l*W+W-1.  For odd W the row
// stride is an odd number of 8-byte words, so the 64-bit row reads are bank-conflict
// free; even W costs a 2- to 8-way conflict on W reads, still far below the
// divide-by-W address arithmetic of a transposing store.
template <int W>
__device__ __forceinline__ void rows_to_lds(const double (&v)[W], double* lds, const int lane)
{
#pragma unroll
    for (int i = 0; i < W; ++i) lds[i * WAVE + lane] = v[i];
}
template <int W>
__device__ __forceinline__ void rows_from_lds(double* __restrict__ g, const int rows_valid, const double* lds,
                                              const int lane)
{
    const int total = rows_valid * W;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int k = i * WAVE + lane;
        if (k < total) g[k] = lds[k];
    }
}

// state of the lane's instance: z = [robot_var (NQ); virtual_var (NX)], each staged row-major
template <int NQ, int NX>
__device__ __forceinline__ void state_from_lds(const double* zs, const double* xs, const int lane, double (&z)[NQ + NX])
{
#pragma unroll
    for (int j = 0; j < NQ; ++j) z[j] = zs[lane * NQ + j];
    if constexpr (NX > 0) {
#pragma unroll
        for (int j = 0; j < NX; ++j) z[NQ + j] = xs[lane * NX + j];
    }
}
template <int NQ, int NX>
__device__ __forceinline__ void state_to_lds(const double (&v)[NQ + NX], double* zs, double* xs, const int lane)
{
#pragma unroll
    for (int j = 0; j < NQ; ++j) zs[lane * NQ + j] = v[j];
    if constexpr (NX > 0) {
#pragma unroll
        for (int j = 0; j < NX; ++j) xs[lane * NX + j] = v[NQ + j];
    }
}

// ---- shape-specialised kernels ---------------------------------------------------
// LDS layout: [skill image | zs (N slots) | ys (ny slots)], slot = 64 doubles.
template <const ShapeDesc& SD>
struct StaticLayout {
    static constexpr int N = SD.n;
    static constexpr int IMG_CHUNKS = (int)((sizeof(Img<SD>) + 1023) / 1024);   // 1 KiB = 64 lanes x 16 B
    static constexpr int IMG_DOUBLES = IMG_CHUNKS * 128;
    static constexpr int n_sets = shape_n_sets(SD);
};

template <const ShapeDesc& SD>
__device__ __forceinline__ const Img<SD>* load_image(const void* __restrict__ img_g, double* lds, const int lane)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* src = (const d2*)img_g;
    d2* dst = (d2*)lds;
#pragma unroll
    for (int k = 0; k < StaticLayout<SD>::IMG_CHUNKS; ++k) dst[k * WAVE + lane] = src[k * WAVE + lane];
    return (const Img<SD>*)lds;
}

// HAVE_SC: the sines / cosines of the state variables come from the caller (sns / css: the four lanes of a quad
// evaluated two each, pinv_solve_static_values_quad_kernel); otherwise every lane evaluates all of them
template <const ShapeDesc& SD, bool HAVE_SC = false>
__device__ __forceinline__ void pinv_tick_static(const Img<SD>* __restrict__ S, const TickArgs& tk,
                                                 const double (&z)[SD.n], const double* ys, const int lane,
                                                 const bool valid, double (&vout)[SD.n], int& acc_mode,
                                                 const double* sns = nullptr, const double* css = nullptr)
{
    constexpr int N = SD.n;
    static_assert(StaticLayout<SD>::n_sets <= kStaticMaxSets, "too many SetConstraints for a static shape");
#ifndef CLIK_NO_SOLO
    if constexpr (shape_team_ok(SD)) {
        // the config-3 family: both modes from shifted copies of one Gram matrix (clik_pinv_static.hpp)
        solo_tick<SD>(S, tk, z, ys, lane, valid, vout, acc_mode);
        return;
    }
#endif
    // FK and the state-dependent task rows once per tick; the FK state dies here
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
            if constexpr (SD.quat_src != 0) orientation_feature_s<SD>(S, ys, lane, K);
        }
        cache_task<SD, 0>(S, tk, K, z, ys, lane, tc);
    }
    CLIK_STAMP(2);
    bool done = !valid;
    acc_mode = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = 0.0;
    // mode scan in the reference's order; a later mode runs only while some lane still has none
    static_for<0, (1 << StaticLayout<SD>::n_sets)>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr unsigned ACT = shape_mode_act(SD, k);
        if (k == 0 || __ballot(!done) != 0ull) {
            double v[N];
            const bool ok = pinv_mode_static<SD, ACT>(S, tk, tc, z, ys, lane, v);
            if (!done && ok) {
                done = true;
                acc_mode = k;
#pragma unroll
                for (int j = 0; j < N; ++j) vout[j] = v[j];
            }
        }
        if constexpr (k == 0) { CLIK_STAMP(3); }
    });
}

// (measuring switch: -DCLIK_OCC2 pins the kernels below to two waves per SIMD; three waves per SIMD and one were measured in
// rounds 3 / 5 and retired, tools/experiments/pinv_retired.patch)
#if defined(CLIK_OCC2)
#define CLIK_OCC_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define CLIK_OCC_ATTR
#endif
// body of the lane-per-instance kernel.  REGIMG: keep a register copy of the skill image (one LDS read burst and
// one wait instead of ~40 separate ~100-cycle LDS stalls: what a lone wave per SIMD wants) or read it from LDS
// where it is used (fewer live registers: what two waves per SIMD want)
// PT: every instance has its own time-slot record (t_inst [B][2 * n_tslots], device): a batch of robots at
// different phases of their trajectories in one launch.  The record is read in place, per lane, where the
// rows use it (the uniform record of the other kernels sits in SGPRs).
template <const ShapeDesc& SD, bool REGIMG, bool PT = false>
__device__ __forceinline__ void pinv_solve_static_body(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const double* __restrict__ x,
    double* __restrict__ dx, const TickArgs& tk_uniform, const double* __restrict__ t_inst = nullptr)
{
    extern __shared__ double lds[];
    CLIK_STAMP(0);
    constexpr int N = SD.n;
    const int lane = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
#ifdef CLIK_STAMPS
    if (rows_valid < 0) return;  // (never taken: makes the next stamp wait for the kernel arguments)
#endif
    CLIK_STAMP_DRAINED(6);      // kernel arguments have arrived
    constexpr int NX = SD.n_x, NQ = SD.n - SD.n_x;
    double* zs = lds + StaticLayout<SD>::IMG_DOUBLES;       // [64][NQ] robot_var, then [64][NX] virtual_var
    double* xs = zs + NQ * WAVE;
    double* ys = zs + N * WAVE;
    // constants, joint state and inputs travel together: every global load is
    // issued (index-clamped, branch-free) before the first LDS write, so the
    // prologue costs one memory round trip (n_x == 0, n_q == N in static shapes)
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 img[StaticLayout<SD>::IMG_CHUNKS];
    {
        const d2* src = (const d2*)img_g;
#pragma unroll
        for (int k = 0; k < StaticLayout<SD>::IMG_CHUNKS; ++k) img[k] = src[k * WAVE + lane];
    }
    double qv[NQ], xv[NX > 0 ? NX : 1], yv[SD.n_y > 0 ? SD.n_y : 1];
    stage_load<NQ>(q + b0 * NQ, NQ, rows_valid, lane, qv);
    if constexpr (NX > 0) stage_load<NX>(x + b0 * NX, NX, rows_valid, lane, xv);
    if constexpr (SD.n_y > 0) stage_load<SD.n_y>(y + b0 * SD.n_y, SD.n_y, rows_valid, lane, yv);
    CLIK_STAMP_DRAINED(7);      // image, q and y are in registers
    {
        d2* dst = (d2*)lds;
#pragma unroll
        for (int k = 0; k < StaticLayout<SD>::IMG_CHUNKS; ++k) dst[k * WAVE + lane] = img[k];
    }
    const Img<SD>* __restrict__ S = (const Img<SD>*)lds;
    // (tail block: the clamped loads filled the rows past rows_valid with copies of
    // the last element - finite values whose results are never stored)
    rows_to_lds<NQ>(qv, zs, lane);
    if constexpr (NX > 0) rows_to_lds<NX>(xv, xs, lane);
    if constexpr (SD.n_y > 0) rows_to_lds<SD.n_y>(yv, ys, lane);
    __syncthreads();
    CLIK_STAMP(1);

    double z[N];
    state_from_lds<NQ, NX>(zs, xs, lane, z);
    double vout[N];
    int acc_mode;
    const TickArgs& tk = PT ? *reinterpret_cast<const TickArgs*>(
                                  t_inst + (size_t)(b0 + (valid ? lane : rows_valid - 1)) * 2 * S->n_tslots)
                            : tk_uniform;
    // Register copy of the skill image: scalar replacement keeps exactly the fields the tick
    // reads, and the scheduling barrier keeps their LDS reads together here (one wait) instead
    // of next to each use (measured: ~40 separate ~100-cycle LDS stalls per tick otherwise).
    if constexpr (!REGIMG) {
        pinv_tick_static<SD>(S, tk, z, ys + lane * SD.n_y, lane, valid, vout, acc_mode);
    } else {
        const Img<SD> Sreg = *S;
        __builtin_amdgcn_sched_barrier(0);
        pinv_tick_static<SD>(&Sreg, tk, z, ys + lane * SD.n_y, lane, valid, vout, acc_mode);
    }
    CLIK_STAMP(4);

    __syncthreads();
    state_to_lds<NQ, NX>(vout, zs, xs, lane);
    __syncthreads();
    rows_from_lds<NQ>(dq + b0 * NQ, rows_valid, zs, lane);
    if constexpr (NX > 0) rows_from_lds<NX>(dx + b0 * NX, rows_valid, xs, lane);
    if (mode_out != nullptr && valid) mode_out[b0 + lane] = acc_mode;
    CLIK_STAMP(5);
}

template <const ShapeDesc& SD>
__global__ __launch_bounds__(WAVE) CLIK_OCC_ATTR void pinv_solve_static_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const double* __restrict__ x,
    double* __restrict__ dx, const TickArgs tk)
{
#ifdef CLIK_NO_REGIMG
    pinv_solve_static_body<SD, false>(img_g, q, y, dq, mode_out, B, x, dx, tk);
#else
    pinv_solve_static_body<SD, true>(img_g, q, y, dq, mode_out, B, x, dx, tk);
#endif
}

// ... with one time-slot record per instance (clik_pinv_solve_batch_t)
template <const ShapeDesc& SD>
__global__ __launch_bounds__(WAVE) CLIK_OCC_ATTR void pinv_solve_static_pt_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const double* __restrict__ x,
    double* __restrict__ dx, const double* __restrict__ t_inst)
{
    TickArgs none;      // (never read)
    pinv_solve_static_body<SD, true, true>(img_g, q, y, dq, mode_out, B, x, dx, none, t_inst);
}

// Large-batch variant (>= kOcc2MinBatch instances: every SIMD has work queued): capped at 256 registers so that
// TWO waves share a SIMD and cover each other's LDS / dependency stalls, skill image read from LDS in place.
// Measured at 1 048 576 instances: 85.7 us against 95.4 us of the kernel above (+11 %); at 131 072 it loses
// (15.6 vs 14.8 us), hence the threshold.  Costs 64 B per lane of scratch (the only static kernel with any).
// Built for the ahead-of-time shapes only (-DCLIK_LARGE_BATCH_VARIANT in their translation unit).
template <const ShapeDesc& SD>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void pinv_solve_static_occ2_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const double* __restrict__ x,
    double* __restrict__ dx, const TickArgs tk)
{
    pinv_solve_static_body<SD, false>(img_g, q, y, dq, mode_out, B, x, dx, tk);
}

// Mode-parallel variant for small batches (fewer wavefronts than SIMDs) and skills with one or
// two SetConstraints: a block = 2 or 4 wavefronts on different SIMDs of one CU working on the
// SAME 64 instances; wave m speculatively evaluates the m-th mode of the reference's scan order.
// The mode scan of pseudo_inverse.py:530-550 becomes a select: the first admissible mode, else
// -1.  The tick costs the longest mode instead of the sum of the modes some lane needs; the
// redundant FK of the extra waves runs on otherwise idle SIMDs.
template <const ShapeDesc& SD>
__global__ __launch_bounds__((1 << shape_n_sets(SD)) * WAVE) CLIK_OCC_ATTR void pinv_solve_static_mp_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const TickArgs tk)
{
    extern __shared__ double lds[];
    constexpr int N = SD.n;
    constexpr int NM = 1 << StaticLayout<SD>::n_sets;       // modes = waves per block (2 or 4)
    static_assert(NM == 2 || NM == 4, "mode-parallel kernel is for shapes with one or two SetConstraints");
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    double* zs = lds + StaticLayout<SD>::IMG_DOUBLES;
    double* ys = zs + N * WAVE;
    double* xs = ys + (SD.n_y > 0 ? SD.n_y : 0) * WAVE;      // exchange: per later mode v (N slots) + ok flag (1 slot)
    typedef double d2 __attribute__((ext_vector_type(2)));
    {
        // the waves share the prologue loads: image chunks round-robin, q by wave 0, y by the last wave
        constexpr int CH = StaticLayout<SD>::IMG_CHUNKS;
        constexpr int PER = (CH + NM - 1) / NM;
        const d2* src = (const d2*)img_g;
        d2* dst = (d2*)lds;
        d2 img[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int ck = k * NM + wave;
            img[k] = src[(ck < CH ? ck : CH - 1) * WAVE + lane];
        }
        double qv[N], yv[SD.n_y > 0 ? SD.n_y : 1];
        if (wave == 0) stage_load<N>(q + b0 * N, N, rows_valid, lane, qv);
        if constexpr (SD.n_y > 0) {
            if (wave == NM - 1) stage_load<SD.n_y>(y + b0 * SD.n_y, SD.n_y, rows_valid, lane, yv);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int ck = k * NM + wave;
            if (ck < CH) dst[ck * WAVE + lane] = img[k];
        }
        if (wave == 0) rows_to_lds<N>(qv, zs, lane);
        if constexpr (SD.n_y > 0) {
            if (wave == NM - 1) rows_to_lds<SD.n_y>(yv, ys, lane);
        }
    }
    __syncthreads();
    // Register copies of the skill image (see pinv_solve_static_kernel), one per phase here (the
    // kinematics, then every task step): each keeps only the fields its phase reads, so later
    // constants are not live (or parked in AGPRs) earlier.  Pays at small batches (-4 % on the
    // config-3 tick); with four busy SIMDs per CU the extra LDS reads cost more than they save,
    // so the one-wave kernel copies once.
#ifdef CLIK_IMG_GLOBAL
    // (experiment: read the skill image through the scalar cache straight from global memory instead of
    // the LDS copy - see DESIGN.md "measured options")
    const Img<SD>* __restrict__ Slds = (const Img<SD>*)img_g;
#else
    const Img<SD>* __restrict__ Slds = (const Img<SD>*)lds;
#endif
    const double* ysl = ys + lane * SD.n_y;
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = zs[lane * N + j];
    TaskCache<SD> tc;
    {
        const Img<SD> Sfk = *Slds;
        __builtin_amdgcn_sched_barrier(0);
        Kin<N> K;
        if constexpr (SD.uses_fk != 0) {
            forward_kinematics_s<SD>(&Sfk, z, K);
            if constexpr (SD.quat_src != 0) orientation_feature_s<SD>(&Sfk, ysl, lane, K);
        }
        cache_task<SD, 0>(&Sfk, tk, K, z, ysl, lane, tc);
    }
    double v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = 0.0;
    bool ok = false;
    // wave m evaluates the m-th mode of the reference's scan order (per-task image copies inside)
    static_for<0, NM>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr unsigned ACT = shape_mode_act(SD, m);
        if (wave == m) {
            ok = pinv_mode_static<SD, ACT>(Slds, tk, tc, z, ysl, lane, v);
            if constexpr (m > 0) {
                double* xm = xs + (m - 1) * (N + 1) * WAVE;
#pragma unroll
                for (int j = 0; j < N; ++j) xm[j * WAVE + lane] = v[j];
                xm[N * WAVE + lane] = ok ? 1.0 : 0.0;
            }
        }
    });
    __syncthreads();
    if (wave == 0) {
        // the scan of pseudo_inverse.py:530-550 as a select: first admissible mode in scan order
        int acc_mode = ok ? 0 : -1;
        bool done = ok;
        static_for<1, NM>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            const double* xm = xs + (m - 1) * (N + 1) * WAVE;
            const bool take = !done && xm[N * WAVE + lane] != 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = take ? xm[j * WAVE + lane] : v[j];
            acc_mode = take ? m : acc_mode;
            done = done || take;
        });
#pragma unroll
        for (int j = 0; j < N; ++j) zs[lane * N + j] = done ? v[j] : 0.0;
        // (single wave from here on: LDS writes above are read back by the same wave)
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        rows_from_lds<N>(dq + b0 * N, rows_valid, zs, lane);
        if (mode_out != nullptr && valid) mode_out[b0 + lane] = acc_mode;
    }
}

// ---- role-split kernel for small batches -----------------------------------------------------
// With fewer wavefronts than SIMDs (<= 16384 instances on 1024 SIMDs) a tick is the serial fp64
// chain of ONE wave.  This variant spends the idle SIMDs of the CU on the same 64 instances:
// per mode a MAIN wave (task solves) and a HELPER wave (builds and factors the Gram-form stack
// lam I + Ja'Ja that the last task projects through, which depends only on the Jacobians), and
// with one SetConstraint both modes speculatively (as pinv_solve_static_mp_kernel): 2 or 4 waves
// per block.  Eligible when every mode has exactly one Gram consumer (ModePlan::helper_ok).
template <const ShapeDesc& SD, int K = 0>
constexpr bool shape_split_ok()
{
    constexpr int ns = shape_n_sets(SD);
    if constexpr (ns > 1) return false;
    else if constexpr (K >= (1 << ns)) return true;
    else return Plan<SD, shape_mode_act(SD, K)>::mode.helper_ok && shape_split_ok<SD, K + 1>();
}

template <const ShapeDesc& SD>
struct SplitLayout {
    static constexpr int N = SD.n;
    static constexpr int NM = 1 << shape_n_sets(SD);          // modes (1 or 2)
    static constexpr int NW = 2 * NM;                          // waves per block
    static constexpr int XCH = N * (N + 1) / 2 + N;            // slots of one published factor
    static constexpr int SLOTS = N + (SD.n_y > 0 ? SD.n_y : 0) + NM * XCH + (NM > 1 ? N + 1 : 0);
    static constexpr size_t LDS_BYTES = ((size_t)StaticLayout<SD>::IMG_DOUBLES + (size_t)SLOTS * WAVE) * sizeof(double);
};

template <const ShapeDesc& SD>
__global__ __launch_bounds__(SplitLayout<SD>::NW * WAVE) void pinv_solve_static_split_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const TickArgs tk)
{
    extern __shared__ double lds[];
    CLIK_STAMP_W(0, 0);
    using LY = SplitLayout<SD>;
    constexpr int N = SD.n;
    constexpr int NM = LY::NM, NW = LY::NW;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    double* zs = lds + StaticLayout<SD>::IMG_DOUBLES;
    double* ys = zs + N * WAVE;
    double* xch = ys + (SD.n_y > 0 ? SD.n_y : 0) * WAVE;      // NM published factors
    double* res = xch + NM * LY::XCH * WAVE;                  // mode 1 result: v (N slots) + ok (1 slot)
    typedef double d2 __attribute__((ext_vector_type(2)));
    {
        // the waves share the prologue loads: image chunks round-robin, q by wave 0, y by the last wave
        constexpr int CH = StaticLayout<SD>::IMG_CHUNKS;
        constexpr int PER = (CH + NW - 1) / NW;
        const d2* src = (const d2*)img_g;
        d2* dst = (d2*)lds;
        d2 img[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int ck = k * NW + wave;
            img[k] = src[(ck < CH ? ck : CH - 1) * WAVE + lane];
        }
        double qv[N], yv[SD.n_y > 0 ? SD.n_y : 1];
        if (wave == 0) stage_load<N>(q + b0 * N, N, rows_valid, lane, qv);
        if constexpr (SD.n_y > 0) {
            if (wave == NW - 1) stage_load<SD.n_y>(y + b0 * SD.n_y, SD.n_y, rows_valid, lane, yv);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int ck = k * NW + wave;
            if (ck < CH) dst[ck * WAVE + lane] = img[k];
        }
        if (wave == 0) rows_to_lds<N>(qv, zs, lane);
        if constexpr (SD.n_y > 0) {
            if (wave == NW - 1) rows_to_lds<SD.n_y>(yv, ys, lane);
        }
    }
    __syncthreads();
    CLIK_STAMP_W(0, 1);
    const Img<SD> Sreg = *(const Img<SD>*)lds;       // register copy, see pinv_solve_static_kernel
    const Img<SD>* __restrict__ S = &Sreg;
    const double* ysl = ys + lane * SD.n_y;
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = zs[lane * N + j];
    __builtin_amdgcn_sched_barrier(0);
    TaskCache<SD> tc;
    {
        Kin<N> K;
        if constexpr (SD.uses_fk != 0) {
            forward_kinematics_s<SD>(S, z, K);
            if constexpr (SD.quat_src != 0) orientation_feature_s<SD>(S, ysl, lane, K);
        }
        cache_task<SD, 0>(S, tk, K, z, ysl, lane, tc);
    }
    CLIK_STAMP_W(0, 2);
    CLIK_STAMP_W(1, 6);
    double v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = 0.0;
    bool ok = false;
    const int my_mode = wave >> 1;
    const bool helper = (wave & 1) != 0;
    static_for<0, NM>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr unsigned ACT = shape_mode_act(SD, m);
        if (my_mode == m) {
            double* mx = xch + m * LY::XCH * WAVE;
            if (helper) {
                helper_mode_static<SD, ACT>(S, tk, tc, z, ysl, lane, mx);
                CLIK_STAMP_W(1, 7);
                __syncthreads();                    // (matches the barrier inside the main wave's projection)
            } else {
                ok = pinv_mode_static<SD, ACT, ROLE_MAIN>(S, tk, tc, z, ysl, lane, v, mx);
            }
        }
    });
    CLIK_STAMP_W(0, 3);
    if constexpr (NM > 1) {
        if (wave == 2) {
#pragma unroll
            for (int j = 0; j < N; ++j) res[j * WAVE + lane] = v[j];
            res[N * WAVE + lane] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
    }
    if (wave == 0) {
        int acc_mode = 0;
        if (!ok) {
            if constexpr (NM > 1) {
                const bool ok1 = res[N * WAVE + lane] != 0.0;
                acc_mode = ok1 ? 1 : -1;
#pragma unroll
                for (int j = 0; j < N; ++j) v[j] = ok1 ? res[j * WAVE + lane] : 0.0;
            } else {
                acc_mode = -1;
#pragma unroll
                for (int j = 0; j < N; ++j) v[j] = 0.0;
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) zs[lane * N + j] = v[j];
        // (single wave from here on: LDS writes above are read back by the same wave)
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        rows_from_lds<N>(dq + b0 * N, rows_valid, zs, lane);
        if (mode_out != nullptr && valid) mode_out[b0 + lane] = acc_mode;
    }
    CLIK_STAMP_W(0, 5);
}

// (one wave per SIMD, stated: the register copy of the skill image lives across the tick loop, and without the
// statement the allocator parks a few of its values in scratch although the wave could use all 512 registers)
#ifndef CLIK_ROLL_ATTR
#define CLIK_ROLL_ATTR __attribute__((amdgpu_waves_per_eu(1, 1)))
#endif
// RK: classical Runge-Kutta (four controller evaluations per tick) instead of explicit Euler.  Two
// instantiations, because the Runge-Kutta bookkeeping (start state and weighted sum of the stage velocities, kept
// in LDS) would otherwise sit in the Euler loop's registers and push the widest stacks into scratch.
template <const ShapeDesc& SD, bool RK>
__global__ __launch_bounds__(WAVE) CLIK_ROLL_ATTR void pinv_rollout_static_kernel(
    const void* __restrict__ img_g, double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B,
    const double* __restrict__ tterms, const int n_ticks, const double dt, const double max_speed,
    double* __restrict__ x, double* __restrict__ dx)
{
    // x / dx: virtual variables (path parameters, cart_on_track_1D...ipynb cells 56-60): integrated like the
    // robot variables, never clamped; unused (null) in skills without them.
    // Euler: the notebooks' loop.  Runge-Kutta: the controller as the right-hand side
    // (integration_methods.py:17-23): k1..k4 at t, t + dt/2, t + dt/2, t + dt, each stage clamped; tterms then
    // holds four time-slot records per tick.
    extern __shared__ double lds[];
    constexpr int N = SD.n;
    constexpr int NX = SD.n_x, NQ = N - NX;
    const int lane = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * WAVE;
    const long long left = B - b0;
    const int rows_valid = left < WAVE ? (int)left : WAVE;
    const bool valid = lane < rows_valid;
    double* zs = lds + StaticLayout<SD>::IMG_DOUBLES;
    double* xs = zs + NQ * WAVE;
    double* ys = zs + N * WAVE;
    const Img<SD>* __restrict__ S = load_image<SD>(img_g, lds, lane);
    {
        double qv[NQ], xv[NX > 0 ? NX : 1], yv[SD.n_y > 0 ? SD.n_y : 1];
        stage_load<NQ>(q + b0 * NQ, NQ, rows_valid, lane, qv);
        if constexpr (NX > 0) stage_load<NX>(x + b0 * NX, NX, rows_valid, lane, xv);
        if constexpr (SD.n_y > 0) stage_load<SD.n_y>(y + b0 * SD.n_y, SD.n_y, rows_valid, lane, yv);
        rows_to_lds<NQ>(qv, zs, lane);
        if constexpr (NX > 0) rows_to_lds<NX>(xv, xs, lane);
        if constexpr (SD.n_y > 0) rows_to_lds<SD.n_y>(yv, ys, lane);
    }
    __syncthreads();
    const int nts = S->n_tslots;
    double z[N];
    state_from_lds<NQ, NX>(zs, xs, lane, z);
    const Img<SD> Sreg = *S;                         // register copy, see pinv_solve_static_kernel
    __builtin_amdgcn_sched_barrier(0);
    double vout[N];
    int acc_mode = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = 0.0;
    if constexpr (!RK) {
#pragma unroll 1
        for (int tick = 0; tick < n_ticks; ++tick) {
            // the skill image is loop invariant: without this fence its LDS reads are all hoisted out of the
            // tick loop and the live constants spill (2.8 KB of scratch per lane)
            asm volatile("" ::: "memory");
            // time terms are read in place ([values | derivatives], 2*nts doubles per tick, never past them)
            const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + (size_t)tick * 2 * nts);
            pinv_tick_static<SD>(&Sreg, tk, z, ys + lane * SD.n_y, lane, valid, vout, acc_mode);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double d = vout[j];
                if (j < NQ && max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                vout[j] = d;
                z[j] = fma(d, dt, z[j]);
            }
        }
    } else {
        double* z0s = ys + SD.n_y * WAVE;       // [N][64] state at the start of the tick, then [N][64] sum of w_i k_i
        double* kss = z0s + N * WAVE;
#pragma unroll 1
        for (int tick = 0; tick < n_ticks; ++tick) {
            int mode0 = -1;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                z0s[j * WAVE + lane] = z[j];
                kss[j * WAVE + lane] = 0.0;
            }
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                asm volatile("" ::: "memory");
                const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + ((size_t)tick * 4 + st) * 2 * nts);
                // (the image is read from LDS in place: with the register copy of the Euler loop the widest stacks
                // spill here)
                pinv_tick_static<SD>(S, tk, z, ys + lane * SD.n_y, lane, valid, vout, acc_mode);
                const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0;
                const double cnext = (st == 2) ? dt : 0.5 * dt;          // offset of the next stage's state
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double d = vout[j];
                    if (j < NQ && max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                    kss[j * WAVE + lane] = fma(wgt, d, kss[j * WAVE + lane]);
                    z[j] = fma(d, cnext, z0s[j * WAVE + lane]);
                }
                mode0 = (st == 0) ? acc_mode : mode0;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                vout[j] = kss[j * WAVE + lane] * (1.0 / 6.0);
                z[j] = fma(vout[j], dt, z0s[j * WAVE + lane]);
            }
            acc_mode = mode0;       // (the mode of the first stage)
        }
    }
    __syncthreads();
    state_to_lds<NQ, NX>(z, zs, xs, lane);
    __syncthreads();
    rows_from_lds<NQ>(q + b0 * NQ, rows_valid, zs, lane);
    if constexpr (NX > 0) rows_from_lds<NX>(x + b0 * NX, rows_valid, xs, lane);
    __syncthreads();
    state_to_lds<NQ, NX>(vout, zs, xs, lane);
    __syncthreads();
    rows_from_lds<NQ>(dq + b0 * NQ, rows_valid, zs, lane);
    if constexpr (NX > 0) rows_from_lds<NX>(dx + b0 * NX, rows_valid, xs, lane);
    if (mode_out != nullptr && valid) mode_out[b0 + lane] = acc_mode;
}

}  // namespace clik
#include "clik_pinv_team.hpp"   // four lanes per instance (needs StaticLayout)
namespace clik {

#ifndef CLIK_DEFER_INPUT_ROWS
#define CLIK_DEFER_INPUT_ROWS 1
#endif
// value-specialised lane kernel (pinv_solve_static_values_kernel): single-mode skills, and the config-3 family
// (whose lane evaluation, solo_tick, beats the one-wave-per-mode kernel once the numbers are compiled in: 5.19 / 5.26 /
// 5.44 us against 5.87 / 5.89 / 5.98 us at 20480 / 24576 / 32768 instances); other skills with up to
// CLIK_VALUE_LANE_MAX_SETS SetConstraints as an experiment switch (plan-driven sequential modes: 0-6 % over mp2)
#ifndef CLIK_VALUE_LANE_MAX_SETS
#define CLIK_VALUE_LANE_MAX_SETS 0
#endif
// ... at every batch size: without an image in LDS or registers the kernel fits two waves per SIMD with no spill,
// and the rows a lane loads / stores itself cost nothing measurable - config 3: 6.18 against 7.19 us at 65536
// instances, 10.2 against 13.3 us at 131072, 67.7 against 87.2 us at 1 M (CLIK_VALUE_LANE_MAX_BATCH caps it)
#ifndef CLIK_VALUE_LANE_MAX_BATCH
#define CLIK_VALUE_LANE_MAX_BATCH (1ll << 40)
#endif

// common launcher signature of the kernel table
struct LaunchArgs {
    const DevSkill* dS;        // dynamic kernels
    const void*     dImg;      // static kernels: device copy of the skill image
    const WarmArgs* warm;
    int nq, nx, ny;
    int mode_parallel;         // small batches: bit 0 two-wave mode scan, bit 1 role-split kernel,
                               // bit 2 team kernel (four lanes per instance) where the shape allows, bit 3 ... at any batch
    double* roll_x;            // rollout of a skill with virtual variables: their state (in/out) and last rates
    double* roll_dx;
    int roll_stages;           // rollout: controller evaluations per tick (0 / 1 explicit Euler, 4 Runge-Kutta)
    const double* t_inst;      // solve: one time-slot record per instance ([B][2 * n_tslots], device) or null
};
typedef hipError_t (*solve_fn)(const LaunchArgs&, const TickArgs&, long long, const double*, const double*,
                               const double*, double*, double*, int32_t*, hipStream_t);
typedef hipError_t (*rollout_fn)(const LaunchArgs&, const double*, int, double, double, long long, double*,
                                 const double*, double*, int32_t*, hipStream_t);

template <int N, class SH>
inline hipError_t launch_solve(const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,
                               const double* x, const double* y, double* dq, double* dx, int32_t* mode,
                               hipStream_t stream)
{
    if (a.t_inst != nullptr) return hipErrorNotSupported;   // (per-instance time: shape-specialised kernels only)
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    const size_t shmem = (size_t)pinv_lds_slots(N, a.ny) * WAVE * sizeof(double);
    hipLaunchKernelGGL((pinv_solve_kernel<N, SH>), dim3(grid), dim3(WAVE), shmem, stream, a.dS, *a.warm, tk, B, q,
                       x, y, dq, dx, mode);
    return hipGetLastError();
}

template <int N, class SH>
inline hipError_t launch_rollout(const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt,
                                 double max_speed, long long B, double* q, const double* y, double* dq,
                                 int32_t* mode, hipStream_t stream)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    const size_t shmem = (size_t)pinv_lds_slots(N, a.ny) * WAVE * sizeof(double);
    hipLaunchKernelGGL((pinv_rollout_kernel<N, SH>), dim3(grid), dim3(WAVE), shmem, stream, a.dS, *a.warm,
                       d_tterms, n_ticks, dt, max_speed, B, q, y, dq, mode);
    return hipGetLastError();
}

template <const ShapeDesc& SD>
inline size_t static_lds_bytes(int ny)
{
    return ((size_t)StaticLayout<SD>::IMG_DOUBLES + (size_t)(SD.n + ny) * WAVE) * sizeof(double);
}

// batches up to this many instances leave SIMDs idle (1024 SIMDs x 64 lanes / 2 waves per block)
constexpr long long kModeParallelMaxBatch = 32768;
// the role-split kernel runs 2-4 waves per 64 instances: up to one block per CU
constexpr long long kRoleSplitMaxBatch = 16384;
// the team kernel runs four lanes per instance: up to 16384 instances its waves have a SIMD each; beyond,
// two of them share a SIMD's fp64 pipe and the tick doubles (measured: 5.1 us at 16384, 9.2 us at 32768 against
// 6.0 us of the two-wave kernel, profiles/r2_lanes_head_to_head.md)
constexpr long long kTeamMaxBatch = 16384;
// from this many instances on every SIMD has several waves queued and the two-waves-per-SIMD build of the
// lane-per-instance kernel wins (pinv_solve_static_occ2_kernel)
constexpr long long kOcc2MinBatch = 524288;

// The skills the value-specialised lane-per-instance kernel serves: single-mode skills without virtual variables (skills
// with SetConstraints keep the one-wave-per-mode kernels at small batches, the config-3 family its four lanes per
// instance) ...
constexpr bool shape_value_lane_ok(const ShapeDesc& sd)
{
    return sd.n_x == 0 && !sd.qp && (shape_n_sets(sd) <= CLIK_VALUE_LANE_MAX_SETS || shape_team_ok(sd));
}
// ... and those whose small batches run four lanes per instance with the sin / cos evaluations split over the quad
// (pinv_solve_static_values_quad_kernel).  ONE predicate for the launcher and for the label.
constexpr bool shape_quad_front_ok(const ShapeDesc& sd)
{
    return shape_value_lane_ok(sd) && !shape_team_ok(sd) && sd.uses_fk != 0 && sd.n >= 3 && sd.n <= 2 * TEAM;
}

// Which kernel variant serves a batch of B instances of a static shape (the label bench.py and the
// tests report): the same conditions launch_solve_static evaluates, on the run-time copy of the shape.
inline const char* static_variant(const ShapeDesc& sd, int mode_parallel, long long B)
{
    if (shape_team_ok(sd) && ((mode_parallel & 8) || ((mode_parallel & 4) && B <= kTeamMaxBatch)))
        return (mode_parallel & 64) ? "team4v" : "team4";       // bit 6: a value-specialised team kernel is attached
    const int ns = shape_n_sets(sd);
    if ((mode_parallel & 64) && shape_value_lane_ok(sd) && B <= CLIK_VALUE_LANE_MAX_BATCH) {
        // value-specialised kernels attached: four lanes per instance (split sin / cos) at small batches of single-mode
        // skills with forward kinematics, one lane per instance otherwise
        if (shape_quad_front_ok(sd) && B <= kTeamMaxBatch && !(mode_parallel & 128))
            return "quadv";
        return "lanev";
    }
    if (sd.n_x == 0 && ns <= 1 && B <= kRoleSplitMaxBatch && (mode_parallel & 2)) {
        bool ok = true;
        for (int k = 0; k < (1 << ns); ++k) ok = ok && make_plan(sd, shape_mode_act(sd, k)).helper_ok;
        if (ok) return "split";
    }
    if ((ns == 1 || ns == 2) && sd.n_x == 0 && B <= kModeParallelMaxBatch / ((1 << ns) / 2) && (mode_parallel & 1))
        return ns == 1 ? "mp2" : "mp4";
    if (B >= kOcc2MinBatch && (mode_parallel & 32) && !(mode_parallel & 16)) return "lane/occ2";
    return "lane";
}

template <const ShapeDesc& SD>
inline hipError_t launch_solve_static(const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,
                                      const double* x, const double* y, double* dq, double* dx, int32_t* mode,
                                      hipStream_t stream)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    if (a.t_inst != nullptr) {
        hipLaunchKernelGGL((pinv_solve_static_pt_kernel<SD>), dim3(grid), dim3(WAVE), static_lds_bytes<SD>(a.ny), stream,
                           a.dImg, q, y, dq, mode, B, x, dx, a.t_inst);
        return hipGetLastError();
    }
    if constexpr (shape_team_ok(SD)) {
        // four lanes per instance, a block of four waves = 64 instances (same grid)
        if ((a.mode_parallel & 8) || ((a.mode_parallel & 4) && B <= kTeamMaxBatch)) {
            hipLaunchKernelGGL((pinv_solve_static_team_kernel<SD>), dim3(grid), dim3(TEAM_WAVES * WAVE),
                               team_lds_bytes<SD>(), stream, a.dImg, q, y, dq, mode, B, tk);
            return hipGetLastError();
        }
    }
    if constexpr (shape_split_ok<SD>() && SD.n_x == 0) {
        // one block per 64 instances, 2 or 4 waves each: worth it while blocks <= CUs-ish
        if (B <= kRoleSplitMaxBatch && (a.mode_parallel & 2)) {
            constexpr size_t shmem = SplitLayout<SD>::LDS_BYTES;
            if (shmem > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)pinv_solve_static_split_kernel<SD>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL((pinv_solve_static_split_kernel<SD>), dim3(grid), dim3(SplitLayout<SD>::NW * WAVE),
                               shmem, stream, a.dImg, q, y, dq, mode, B, tk);
            return hipGetLastError();
        }
    }
    if constexpr ((StaticLayout<SD>::n_sets == 1 || StaticLayout<SD>::n_sets == 2) && SD.n_x == 0) {
        // one wave per mode (2 or 4) on the same 64 instances (the multi-wave kernels stage robot_var only)
        constexpr int NM = 1 << StaticLayout<SD>::n_sets;
        if (B <= kModeParallelMaxBatch / (NM / 2) && (a.mode_parallel & 1)) {
            const size_t shmem = static_lds_bytes<SD>(a.ny) + (size_t)(NM - 1) * (SD.n + 1) * WAVE * sizeof(double);
            hipLaunchKernelGGL((pinv_solve_static_mp_kernel<SD>), dim3(grid), dim3(NM * WAVE), shmem, stream,
                               a.dImg, q, y, dq, mode, B, tk);
            return hipGetLastError();
        }
    }
#ifdef CLIK_LARGE_BATCH_VARIANT
    if (B >= kOcc2MinBatch && !(a.mode_parallel & 16)) {
        hipLaunchKernelGGL((pinv_solve_static_occ2_kernel<SD>), dim3(grid), dim3(WAVE), static_lds_bytes<SD>(a.ny), stream,
                           a.dImg, q, y, dq, mode, B, x, dx, tk);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((pinv_solve_static_kernel<SD>), dim3(grid), dim3(WAVE), static_lds_bytes<SD>(a.ny), stream,
                       a.dImg, q, y, dq, mode, B, x, dx, tk);
    return hipGetLastError();
}

// the team kernel with the skill's numbers compiled in (IMGV::value, see clik_pinv_team.hpp); the caller
// (clik_pinv_solve_batch) uses it for the batches the image-reading team kernel would serve
template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_solve_team_values(const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,
                                           const double* y, double* dq, int32_t* mode, hipStream_t stream)
{
    static_assert(shape_team_ok(SD), "value-specialised kernels exist for the team family only");
    const unsigned grid = (unsigned)((B + TEAM_INST - 1) / TEAM_INST);
    hipLaunchKernelGGL((pinv_solve_static_team_kernel<SD, IMGV>), dim3(grid), dim3(TEAM_WAVES * WAVE),
                       team_lds_bytes<SD>(true), stream, nullptr, q, y, dq, mode, B, tk);
    (void)a;
    return hipGetLastError();
}

constexpr unsigned long long kResidentIntegrateBit = 1ull << 63;
template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_resident_quad_values(const TickArgs& tk, long long B, const double* q, const double* y, double* dq,
                                              int32_t* mode, void* ticket, unsigned* done, int n_ticks,
                                              unsigned long long timeout_ticks, hipStream_t stream);

template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_resident_team_values(const TickArgs& tk, long long B, const double* q, const double* y,
                                              double* dq, int32_t* mode, void* ticket, unsigned* done, int n_ticks,
                                              unsigned long long timeout_ticks, hipStream_t stream)
{
    if constexpr (shape_team_ok(SD)) {
        const unsigned grid = (unsigned)((B + TEAM_INST - 1) / TEAM_INST);
        // (bit 63 of the poll budget: the instantiation that integrates the state itself, clik_pinv_resident_run_state)
        const unsigned long long budget = timeout_ticks & ~kResidentIntegrateBit;
        // Every block of the launch must be resident at once - a block that never starts can never count, and the
        // ones that did would spin until the watchdog fires - AND the ticket feeder must find a SIMD with room for its
        // wave while they spin.  The bound: ONE block per CU of THIS device (its CU count: a partitioned device has
        // fewer), provided this instantiation's occupancy allows a block at all.  Measured on the MI355X (round 4): the
        // occupancy interface reports two blocks per CU, yet launches of 511 and of 512 blocks (32704 / 32768
        // instances) both stall until the watchdog ends them - the second block of a CU takes the registers the
        // feeder needs -, while 256 blocks (16384 instances) run.  Round 3's hard-coded 2 x 256 let those through.
        {
            static int max_blocks[2] = {-1, -1};
            const int which = (timeout_ticks & kResidentIntegrateBit) ? 1 : 0;
            if (max_blocks[which] < 0) {
                int per_cu = 0, dev = 0, cus = 0;
                hipError_t oe = which
                    ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pinv_resident_team_kernel<SD, IMGV, true>,
                                                                   TEAM_WAVES * WAVE, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pinv_resident_team_kernel<SD, IMGV, false>,
                                                                   TEAM_WAVES * WAVE, 0);
                if (oe == hipSuccess) oe = hipGetDevice(&dev);
                if (oe == hipSuccess) oe = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                if (oe != hipSuccess) return oe;
                max_blocks[which] = per_cu >= 1 ? cus : 0;
            }
            if ((long long)grid > (long long)max_blocks[which]) return hipErrorNotSupported;
        }
        if (timeout_ticks & kResidentIntegrateBit)
            hipLaunchKernelGGL((pinv_resident_team_kernel<SD, IMGV, true>), dim3(grid), dim3(TEAM_WAVES * WAVE), 0, stream,
                               q, y, dq, mode, B, tk, (ResidentTicket*)ticket, done, n_ticks, budget);
        else
            hipLaunchKernelGGL((pinv_resident_team_kernel<SD, IMGV, false>), dim3(grid), dim3(TEAM_WAVES * WAVE), 0, stream,
                               q, y, dq, mode, B, tk, (ResidentTicket*)ticket, done, n_ticks, budget);
        return hipGetLastError();
    } else {
        return launch_resident_quad_values<SD, IMGV>(tk, B, q, y, dq, mode, ticket, done, n_ticks, timeout_ticks, stream);
    }
}

template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_rollout_team_values(const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt,
                                             double max_speed, long long B, double* q, const double* y, double* dq,
                                             int32_t* mode, hipStream_t stream)
{
    static_assert(shape_team_ok(SD), "value-specialised kernels exist for the team family only");
    const unsigned grid = (unsigned)((B + TEAM_INST - 1) / TEAM_INST);
    if (a.roll_stages == 4)
        hipLaunchKernelGGL((pinv_rollout_static_team_kernel<SD, IMGV, 4>), dim3(grid), dim3(TEAM_WAVES * WAVE),
                           team_rollout_lds_bytes<SD>(true), stream, nullptr, q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed);
    else
        hipLaunchKernelGGL((pinv_rollout_static_team_kernel<SD, IMGV, 1>), dim3(grid), dim3(TEAM_WAVES * WAVE),
                           team_rollout_lds_bytes<SD>(true), stream, nullptr, q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed);
    return hipGetLastError();
}

// The lane-per-instance kernel with the skill's numbers compiled in (IMGV::value: the skill image as a constant
// expression, see clik_pinv_team.hpp): nothing is staged through LDS - no image copy, no barrier - every lane
// loads its own robot_var / input_var row and stores its own velocity row.  For the small batches where a tick
// is the latency of one wave, and - measured - for the large ones too (see CLIK_VALUE_LANE_MAX_BATCH above).
constexpr long long kValueLaneMaxBatch = CLIK_VALUE_LANE_MAX_BATCH;
// (which skills it serves: shape_value_lane_ok, above static_variant)
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) CLIK_OCC_ATTR void pinv_solve_static_values_kernel(
    const double* __restrict__ q, const double* __restrict__ y, double* __restrict__ dq,
    int32_t* __restrict__ mode_out, const long long B, const TickArgs tk)
{
    static_assert(SD.n_x == 0, "value-specialised lane kernel: robot variables only");
    CLIK_BODY_BEGIN();
    constexpr int N = SD.n;
    constexpr Img<SD> Sval = IMGV::value;        // (a local constant: its loads fold to immediates)
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * WAVE + lane;
    const bool valid = inst < B;
    const long long row = valid ? inst : B - 1;
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = q[row * N + j];
    const double* ys = SD.n_y > 0 ? y + row * SD.n_y : nullptr;
#if CLIK_DEFER_INPUT_ROWS
    // The input_var row is requested only once the robot_var row has ARRIVED (its address is made to depend on z[0]).
    // A batch that fills the device in one generation of waves (65536 - 262144 instances) starts as one burst of loads at
    // HBM bandwidth - 15 MB at 131072 instances, 2.4 us of a 10 us tick - and a wave cannot start before ITS robot_var
    // row is through; with the input rows behind the robot_var rows of EVERY wave the last wave starts after half
    // the burst, and the input rows stream in under the sin / cos and forward-kinematics work that does not need them.
    // (... requested THERE, into registers: left to itself the compiler issues them 75 instructions before their first use)
    double yrow[SD.n_y > 0 ? SD.n_y : 1];
    if constexpr (SD.n_y > 0) {
        double dep = z[0];
        int off = 0;             // (an offset, not the pointer: a laundered pointer loses its address space and its loads become FLAT)
        asm volatile("" : "+v"(off) : "v"(dep));
        const double* yp = ys + off;
#pragma unroll
        for (int k = 0; k < SD.n_y; ++k) yrow[k] = yp[k];
        __builtin_amdgcn_sched_barrier(0);
        ys = yrow;
    }
#endif
    double vout[N];
    int acc_mode;
    pinv_tick_static<SD>(&Sval, tk, z, ys, lane, valid, vout, acc_mode);
    if (valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) dq[inst * N + j] = vout[j];
        if (mode_out != nullptr) mode_out[inst] = acc_mode;
    }
    CLIK_BODY_END();
}

// ... with FOUR LANES PER INSTANCE for the small batches of single-mode skills that use forward kinematics (BASELINE
// config 2: 4096 instances are 64 waves of the kernel above on 1024 SIMDs).  A wave has one instruction stream, so
// the quad can only split work that is the SAME instructions on DIFFERENT data: the sines / cosines of the state
// variables (lane r evaluates variables 2r and 2r + 1, the quad exchanges them by DPP: 2 evaluations + 28 moves per lane
// instead of N evaluations - 170 of the 1240 instructions of the config-2 stream); everything behind is evaluated by
// all four lanes alike (the entries of the <= 8 x 8 matrices cannot be split: clik_pinv_team.hpp) and lane 0 stores.
// Up to kTeamMaxBatch instances (one wave per SIMD).  Same values as the lane kernel to rounding.
// (which skills it serves: shape_quad_front_ok, above static_variant)
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) CLIK_OCC_ATTR void pinv_solve_static_values_quad_kernel(
    const double* __restrict__ q, const double* __restrict__ y, double* __restrict__ dq,
    int32_t* __restrict__ mode_out, const long long B, const TickArgs tk)
{
    static_assert(SD.n_x == 0, "value-specialised kernels: robot variables only");
    CLIK_BODY_BEGIN();
    constexpr int N = SD.n;
    constexpr Img<SD> Sval = IMGV::value;
    const int tid = threadIdx.x;
    const int r = tid & (TEAM - 1);
    // (one wave = 16 instances per block - a compile-time shape: reading blockDim.x costs a scalar load from the dispatch
    // packet and its round trip at the start of every wave, measured as +0.13 us on the config-3 team kernel)
    // (launched for at most kTeamMaxBatch instances: 32-bit row numbers, rows addressed as uniform base + 32-bit lane
    // offset - the 64-bit index arithmetic was a dozen instructions of a lone wave's stream)
    CLIK_PHASE("rows_in");
    const SinCosK sck = sincos_consts();        // (their scalar loads go out before anything else)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned inst = (unsigned)blockIdx.x * (unsigned)(WAVE / TEAM) + ((unsigned)tid >> 2);
    const unsigned last = (unsigned)B - 1u;
    const bool valid = inst <= last;
    const unsigned row = inst < last ? inst : last;
    const unsigned qoff = __umul24(row, (unsigned)(N * sizeof(double)));
    const double* __restrict__ qrow = reinterpret_cast<const double*>(reinterpret_cast<const char*>(q) + qoff);
    // this lane's two sin / cos arguments first (their loads return first), then the whole row
    const int j0 = 2 * r < N ? 2 * r : N - 1, j1 = 2 * r + 1 < N ? 2 * r + 1 : N - 1;
    const double a0 = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(q) + (qoff + (unsigned)j0 * 8u));
    const double a1 = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(q) + (qoff + (unsigned)j1 * 8u));
    __builtin_amdgcn_sched_barrier(0);
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = qrow[j];
    // (the input_var row is requested HERE, with the robot_var row - one memory round trip per tick; left to the compiler
    // it was requested where it is first used, part of it a few instructions before an `s_waitcnt vmcnt(0)` 670
    // instructions into the tick: a second, exposed round trip.  The fence keeps the loads above it.)
    constexpr int NYQ = SD.n_y > 0 ? SD.n_y : 0;
    double ydir[NYQ > 0 ? NYQ : 1];
    if constexpr (NYQ > 0) {
        const double* __restrict__ yrow =
            reinterpret_cast<const double*>(reinterpret_cast<const char*>(y) + __umul24(row, (unsigned)(NYQ * sizeof(double))));
#pragma unroll
        for (int k = 0; k < NYQ; ++k) ydir[k] = yrow[k];
    }
    asm volatile("" ::: "memory");
    const double* ys = NYQ > 0 ? ydir : nullptr;
    CLIK_PHASE("sincos");
    double sn0, cs0, sn1, cs1;
    sincos_fast(a0, sn0, cs0, sck);
    sincos_fast(a1, sn1, cs1, sck);
    const bool huge = (fabs(a0) > kSinCosFastMax) | (fabs(a1) > kSinCosFastMax);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(huge) != 0ull, 0)) {
        if (fabs(a0) > kSinCosFastMax) { const SinCos sc = sincos_slow(a0); sn0 = sc.s; cs0 = sc.c; }
        if (fabs(a1) > kSinCosFastMax) { const SinCos sc = sincos_slow(a1); sn1 = sc.s; cs1 = sc.c; }
    }
    CLIK_PHASE("sincos_exchange");
    double sns[N], css[N];
    static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (shape_state_type(SD, j) == CLIK_JOINT_REVOLUTE) {
            constexpr int CTRL = (j / 2) * 0x55;           // quad_perm:[k,k,k,k], k = the lane that evaluated variable j
            sns[j] = quad_perm_f64<CTRL>((j & 1) ? sn1 : sn0);
            css[j] = quad_perm_f64<CTRL>((j & 1) ? cs1 : cs0);
        } else {
            sns[j] = css[j] = 0.0;
        }
    });
    CLIK_PHASE("tick");
    double vout[N];
    int acc_mode;
    pinv_tick_static<SD, true>(&Sval, tk, z, ys, tid & (WAVE - 1), valid, vout, acc_mode, sns, css);
    CLIK_PHASE("select_store");
    if (valid && r == 0) {
        double* __restrict__ drow = reinterpret_cast<double*>(reinterpret_cast<char*>(dq) + __umul24(inst, (unsigned)(N * sizeof(double))));
#pragma unroll
        for (int j = 0; j < N; ++j) drow[j] = vout[j];
        if (mode_out != nullptr) *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(mode_out) + inst * 4u) = acc_mode;
    }
    CLIK_BODY_END();
    CLIK_PHASE_END();
}

// ... and RESIDENT: one launch that runs tick k whenever ticket k is published (clik_pinv_team.hpp, ResidentTicket: the
// protocol, the watchdog and the ring of input / output slots are pinv_resident_team_kernel's) - the per-tick launch
// boundary of 1.1 us is a third of a config-2 tick.  One wave = 16 instances per block, at most one wave per SIMD
// (launch_resident_team_values); lane r of a quad requests elements 2r and 2r + 1 of its instance's rows - its own sin / cos
// arguments - and stores the same elements of the velocity row.  The kernel has registers to spare (164 of 512), so
// the NEXT tick's rows are requested before this tick's arithmetic starts whenever its ticket is already out.
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) void pinv_resident_quad_kernel(
    const double* q, const double* y, double* dq, int32_t* mode_out, const long long B, const TickArgs tk,
    ResidentTicket* ticket, unsigned* done, const int n_ticks, const unsigned long long max_polls)
{
    static_assert(SD.n_x == 0, "value-specialised kernels: robot variables only");
    constexpr int N = SD.n, NY = SD.n_y > 0 ? SD.n_y : 0;
    constexpr Img<SD> Sval = IMGV::value;
    const int tid = threadIdx.x;
    const int r = tid & (TEAM - 1);
    const long long inst = (long long)blockIdx.x * (WAVE / TEAM) + (tid >> 2);
    const bool valid = inst < B;
    const long long binst = valid ? inst : (B - 1);
    ResidentWave rw;
    rw.init(ticket, done, max_polls, n_ticks, blockIdx.x, gridDim.x, tid);
    bool have_next = false;
    const long long ring = rw.ring_depth();
    constexpr int RQ = (N + 2 * TEAM - 1) / (2 * TEAM), RY = NY > 0 ? (NY + 2 * TEAM - 1) / (2 * TEAM) : 1;
    static_assert(RQ == 1, "resident quad kernel: at most eight state variables");
    double zp[2], yp[2 * RY], zp_next[2], yp_next[2 * RY];
#pragma unroll
    for (int i = 0; i < 2; ++i) zp[i] = zp_next[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 2 * RY; ++i) yp[i] = yp_next[i] = 0.0;
    auto request_rows = [&](const int k, double (&zq)[2], double (&yq)[2 * RY]) __attribute__((always_inline)) {
        const long long row = ((long long)((k - 1) % (int)ring)) * B + binst;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = 2 * r + i;
            zq[i] = __hip_atomic_load(q + row * N + (e < N ? e : N - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if constexpr (NY > 0) {
#pragma unroll
            for (int i = 0; i < 2 * RY; ++i) {
                const int e = 2 * TEAM * (i / 2) + 2 * r + (i & 1);
                yq[i] = __hip_atomic_load(y + row * NY + (e < NY ? e : NY - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    int owed = 0;           // tick whose "done" slot is still to be published (0: none)
    const SinCosK sck = sincos_consts();        // (once per launch)
#pragma unroll 1
    for (int k = 1; k <= n_ticks; ++k) {
        if (!have_next) {
            rw.poll_for((unsigned)k);
            if (rw.leave) break;
            asm volatile("" ::: "memory");
            request_rows(k, zp, yp);
            // (waited for inside this branch: left pending, the compiler - whose wait counts are merged over both ways
            // into the tick - made the fed-ahead path wait for its freshly requested NEXT rows as well)
#pragma unroll
            for (int i = 0; i < 2; ++i) pin_arrived(zp[i]);
            if constexpr (NY > 0) {
#pragma unroll
                for (int i = 0; i < 2 * RY; ++i) pin_arrived(yp[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) zp[i] = zp_next[i];
#pragma unroll
            for (int i = 0; i < 2 * RY; ++i) yp[i] = yp_next[i];
        }
        have_next = false;
        // the next tick's rows, if their ticket is already out: they arrive under this tick's arithmetic
        if (k < n_ticks && rw.seen >= (unsigned)(k + 1)) {
            request_rows(k + 1, zp_next, yp_next);
            have_next = true;
        }
        // ... and the ticket word again, for the NEXT tick's decision: a whole tick for it to arrive.  (Round 5 asked for
        // it at the end of the tick and read it a few instructions later, at the top of the next: an uncached load's
        // latency on the critical path of every tick - the 1300 wait cycles per tick of profiles/r6_counters.json.)
        if (have_next) rw.peek();
        // this lane's own two sin / cos arguments are its share of the row
        double sn0, cs0, sn1, cs1;
        sincos_fast(zp[0], sn0, cs0, sck);
        sincos_fast(zp[1], sn1, cs1, sck);
        const bool huge = (fabs(zp[0]) > kSinCosFastMax) | (fabs(zp[1]) > kSinCosFastMax);
        if (__builtin_expect(__ballot(huge) != 0ull, 0)) {
            if (fabs(zp[0]) > kSinCosFastMax) { const SinCos sc = sincos_slow(zp[0]); sn0 = sc.s; cs0 = sc.c; }
            if (fabs(zp[1]) > kSinCosFastMax) { const SinCos sc = sincos_slow(zp[1]); sn1 = sc.s; cs1 = sc.c; }
        }
        double z[N], sns[N], css[N], yrow[NY > 0 ? NY : 1];
        static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int CTRL = (j / 2) * 0x55;           // quad_perm:[k,k,k,k], k = the lane that holds element j
            z[j] = quad_perm_f64<CTRL>(zp[j & 1]);
            if constexpr (shape_state_type(SD, j) == CLIK_JOINT_REVOLUTE) {
                sns[j] = quad_perm_f64<CTRL>((j & 1) ? sn1 : sn0);
                css[j] = quad_perm_f64<CTRL>((j & 1) ? cs1 : cs0);
            } else {
                sns[j] = css[j] = 0.0;
            }
        });
        if constexpr (NY > 0) {
            static_for<0, NY>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int CTRL = ((j % (2 * TEAM)) / 2) * 0x55;
                yrow[j] = quad_perm_f64<CTRL>(yp[2 * (j / (2 * TEAM)) + (j & 1)]);
            });
        }
        double vout[N];
        int acc_mode;
        pinv_tick_static<SD, true>(&Sval, tk, z, yrow, tid & (WAVE - 1), valid, vout, acc_mode, sns, css);
        // (the next tick's rows and the ticket word, requested at the top of this tick, are waited for HERE - before this
        // tick's stores are issued - not at the top of the next tick, where the same wait would also cover those stores:
        // pin_arrived, clik_device.hpp.  Unconditional: the compiler's wait insertion is not path-sensitive, and with
        // nothing requested there is nothing to wait for)
#pragma unroll
        for (int i = 0; i < 2; ++i) pin_arrived(zp_next[i]);
        if constexpr (NY > 0) {
#pragma unroll
            for (int i = 0; i < 2 * RY; ++i) pin_arrived(yp_next[i]);
        }
        pin_arrived(rw.seen);
        // (a producer that runs exactly ONE tick ahead: the early look was too early - look again, and wait for it inside
        // the branch)
        if (have_next && k + 1 < n_ticks && rw.seen < (unsigned)(k + 2)) {
            rw.peek();
            pin_arrived(rw.seen);
        }
        if (owed != 0) {
            rw.publish_done(owed);         // (the previous tick's stores: issued a whole tick ago, nothing to wait for)
            owed = 0;
        }
        if (valid) {
            const long long orow = ((long long)((k - 1) % (int)ring)) * B + inst;
            double s0 = vout[N - 1], s1 = vout[N - 1];
            static_for<0, TEAM>([&](auto kc) __attribute__((always_inline)) {
                constexpr int kk = decltype(kc)::value;
                if constexpr (2 * kk < N) s0 = (r == kk) ? vout[2 * kk] : s0;
                if constexpr (2 * kk + 1 < N) s1 = (r == kk) ? vout[2 * kk + 1] : s1;
            });
            const int e = 2 * r;
            if (e < N) __hip_atomic_store(dq + orow * N + e, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (e + 1 < N) __hip_atomic_store(dq + orow * N + e + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (mode_out != nullptr && r == 0)
                __hip_atomic_store(mode_out + orow, acc_mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (have_next) owed = k;        // published at the end of the next tick (fed ahead: nobody waits for it)
        else rw.publish_done(k);           // nobody has asked for the next tick yet (a closed loop), or the last one: at once
    }
    if (owed != 0) rw.publish_done(owed);
}

template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_resident_quad_values(const TickArgs& tk, long long B, const double* q, const double* y, double* dq,
                                              int32_t* mode, void* ticket, unsigned* done, int n_ticks,
                                              unsigned long long timeout_ticks, hipStream_t stream)
{
    if constexpr (shape_quad_front_ok(SD)) {
        if (timeout_ticks & kResidentIntegrateBit) return hipErrorNotSupported;     // (the state kept by the kernel: team family only)
        const unsigned grid = (unsigned)((B + WAVE / TEAM - 1) / (WAVE / TEAM));
        // every wave must be resident at once, with room left for the ticket feeder: at most one wave per SIMD
        int dev = 0, cus = 0, per_cu = 0;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pinv_resident_quad_kernel<SD, IMGV>, WAVE, 0);
        if (oe == hipSuccess) oe = hipGetDevice(&dev);
        if (oe == hipSuccess) oe = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (oe != hipSuccess) return oe;
        const long long max_blocks = (long long)cus * (per_cu < 4 ? per_cu : 4);
        if ((long long)grid > max_blocks) return hipErrorNotSupported;
        hipLaunchKernelGGL((pinv_resident_quad_kernel<SD, IMGV>), dim3(grid), dim3(WAVE), 0, stream, q, y, dq, mode, B, tk,
                           (ResidentTicket*)ticket, done, n_ticks, timeout_ticks);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

// ... and its PERSISTENT form for the batches with many waves per SIMD - a round-4 experiment, compiled only with
// -DCLIK_LANE_PERSIST (CLIK_JIT_DEFINES): MEASURED SLOWER, profiles/r4_persistent_lane_variants.txt (1 M instances:
// 67.2 against 65.5 us; 262144: 19.6 against 17.6), bit-identical results (tools/persist_check.py).  The idea: counters
// of the kernel above at 1 M instances (profiles/r3_counters.json) show a wave living 15.7 k cycles, issuing for 7.3 k
// and waiting 4.7 k on its own robot_var row, two waves to a SIMD (248 VGPRs).  Here a launch is exactly the waves the
// device holds at once; every wave walks its chunks of 64 instances (chunk += grid) and, before it computes chunk i,
// starts the copy of chunk i + grid straight into LDS (global_load_lds_dwordx4: 16 B per lane, no destination
// registers - the kernel has none to spare), waits for it after the tick, and only then stores its results (so the one
// `s_waitcnt vmcnt(0)` of an iteration waits for nothing younger than a copy issued a whole tick earlier).  Why it
// does not pay (tools/probe_fp64_peak.hip, profiles/r4_fp64_issue_probe.txt): sixteen waves per SIMD in turn, two
// resident, of 1792 fp64 FMAs each and NO memory at all take 54 - 56 us - the kernel's 65.5 us with its 180 MB of
// traffic is within 16 % of pure issue, and the loop costs what the prefetch saves (256 VGPRs, constants hoisted
// out of the loop into spilled registers unless -mllvm -disable-machine-licm).

// the value-specialised kernel of a skill for one tick: four lanes per instance where the family allows and the
// batch is small, else the lane kernel above (hipErrorNotSupported beyond its batch range: the caller then uses
// the image-reading kernels)
template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_solve_values(const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,
                                      const double* y, double* dq, int32_t* mode, hipStream_t stream)
{
    if constexpr (shape_team_ok(SD)) {
        // (the value-specialised team kernel addresses its rows with 24-bit row numbers: CLIK_LANES=4 at more than 2^24
        // instances falls through to the lane kernel)
        if (((a.mode_parallel & 8) && B <= (1ll << 24)) || ((a.mode_parallel & 4) && B <= kTeamMaxBatch))
            return launch_solve_team_values<SD, IMGV>(a, tk, B, q, y, dq, mode, stream);
    }
    if constexpr (shape_quad_front_ok(SD)) {
        // (bit 7 of mode_parallel: CLIK_QUAD_FRONT=0 keeps one lane per instance - a measuring switch)
        if (B <= kTeamMaxBatch && !(a.mode_parallel & 128)) {
            // one WAVE per block (16 instances): the dispatcher spreads blocks over the CUs, so up to 4096 instances every
            // wave has a CU to itself - waves that share a CU run slower each (measured: 3.02 against 3.17 us at 4096
            // instances with four waves per block, profiles/r5_quad_ab.txt)
            constexpr int block = WAVE;
            constexpr int per_block = block / TEAM;
            const unsigned grid = (unsigned)((B + per_block - 1) / per_block);
            hipLaunchKernelGGL((pinv_solve_static_values_quad_kernel<SD, IMGV>), dim3(grid), dim3(block), 0,
                               stream, q, y, dq, mode, B, tk);
            return hipGetLastError();
        }
    }
    if constexpr (shape_value_lane_ok(SD)) {
        if (B <= kValueLaneMaxBatch) {
            const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
            hipLaunchKernelGGL((pinv_solve_static_values_kernel<SD, IMGV>), dim3(grid), dim3(WAVE), 0, stream, q, y, dq,
                               mode, B, tk);
            return hipGetLastError();
        }
    }
    return hipErrorNotSupported;
}

// ... and its on-device rollout (see pinv_rollout_static_kernel): state in registers from tick to tick, the rows
// loaded once and stored once by the lane itself, no LDS (the Runge-Kutta bookkeeping lives in registers here: without
// the image there is room)
template <const ShapeDesc& SD, class IMGV, bool RK>
__global__ __launch_bounds__(WAVE) CLIK_OCC_ATTR void pinv_rollout_static_values_kernel(
    double* __restrict__ q, const double* __restrict__ y, double* __restrict__ dq, int32_t* __restrict__ mode_out,
    const long long B, const double* __restrict__ tterms, const int n_ticks, const double dt, const double max_speed)
{
    static_assert(SD.n_x == 0, "value-specialised lane kernel: robot variables only");
    constexpr int N = SD.n;
    constexpr Img<SD> Sval = IMGV::value;
    constexpr int stages = RK ? 4 : 1;
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * WAVE + lane;
    const bool valid = inst < B;
    const long long row = valid ? inst : B - 1;
    const int nts = Sval.n_tslots;
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = q[row * N + j];
    const double* ys = SD.n_y > 0 ? y + row * SD.n_y : nullptr;
    double vout[N];
    int acc_mode = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = 0.0;
#pragma unroll 1
    for (int tick = 0; tick < n_ticks; ++tick) {
        if constexpr (!RK) {
            const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + (size_t)tick * 2 * nts);
            pinv_tick_static<SD>(&Sval, tk, z, ys, lane, valid, vout, acc_mode);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double d = vout[j];
                if (max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                vout[j] = d;
                z[j] = fma(d, dt, z[j]);
            }
        } else {
            double z0[N], ks[N];
            int mode0 = -1;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                z0[j] = z[j];
                ks[j] = 0.0;
            }
#pragma unroll 1
            for (int st = 0; st < stages; ++st) {
                const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + ((size_t)tick * stages + st) * 2 * nts);
                pinv_tick_static<SD>(&Sval, tk, z, ys, lane, valid, vout, acc_mode);
                const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0;
                const double cnext = (st == 2) ? dt : 0.5 * dt;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double d = vout[j];
                    if (max_speed > 0.0) d = fmax(fmin(d, max_speed), -max_speed);
                    ks[j] = fma(wgt, d, ks[j]);
                    z[j] = fma(d, cnext, z0[j]);
                }
                mode0 = (st == 0) ? acc_mode : mode0;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                vout[j] = ks[j] * (1.0 / 6.0);
                z[j] = fma(vout[j], dt, z0[j]);
            }
            acc_mode = mode0;
        }
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            q[inst * N + j] = z[j];
            dq[inst * N + j] = vout[j];
        }
        if (mode_out != nullptr) mode_out[inst] = acc_mode;
    }
}

template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_rollout_values(const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt,
                                        double max_speed, long long B, double* q, const double* y, double* dq,
                                        int32_t* mode, hipStream_t stream)
{
    if constexpr (shape_team_ok(SD)) {
        if ((a.mode_parallel & 8) || ((a.mode_parallel & 4) && B <= kTeamMaxBatch))
            return launch_rollout_team_values<SD, IMGV>(a, d_tterms, n_ticks, dt, max_speed, B, q, y, dq, mode, stream);
    }
    if constexpr (shape_value_lane_ok(SD)) {
        if (B <= kValueLaneMaxBatch) {
            const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
            if (a.roll_stages == 4)
                hipLaunchKernelGGL((pinv_rollout_static_values_kernel<SD, IMGV, true>), dim3(grid), dim3(WAVE), 0, stream,
                                   q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed);
            else
                hipLaunchKernelGGL((pinv_rollout_static_values_kernel<SD, IMGV, false>), dim3(grid), dim3(WAVE), 0, stream,
                                   q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed);
            return hipGetLastError();
        }
    }
    return hipErrorNotSupported;
}

template <const ShapeDesc& SD>
inline hipError_t launch_rollout_static(const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt,
                                        double max_speed, long long B, double* q, const double* y, double* dq,
                                        int32_t* mode, hipStream_t stream)
{
    const unsigned grid = (unsigned)((B + WAVE - 1) / WAVE);
    if constexpr (shape_team_ok(SD)) {
        // four lanes per instance (same grid: 64 instances per block of four waves)
        if ((a.mode_parallel & 8) || ((a.mode_parallel & 4) && B <= kTeamMaxBatch)) {
            if (a.roll_stages == 4)
                hipLaunchKernelGGL((pinv_rollout_static_team_kernel<SD, void, 4>), dim3(grid), dim3(TEAM_WAVES * WAVE),
                                   team_rollout_lds_bytes<SD>(), stream, a.dImg, q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed);
            else
                hipLaunchKernelGGL((pinv_rollout_static_team_kernel<SD, void, 1>), dim3(grid), dim3(TEAM_WAVES * WAVE),
                                   team_rollout_lds_bytes<SD>(), stream, a.dImg, q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed);
            return hipGetLastError();
        }
    }
    if (a.roll_stages == 4) {
        const size_t shmem = static_lds_bytes<SD>(a.ny) + (size_t)2 * SD.n * WAVE * sizeof(double);
        if (shmem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)pinv_rollout_static_kernel<SD, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((pinv_rollout_static_kernel<SD, true>), dim3(grid), dim3(WAVE), shmem, stream,
                           a.dImg, q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed, a.roll_x, a.roll_dx);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((pinv_rollout_static_kernel<SD, false>), dim3(grid), dim3(WAVE), static_lds_bytes<SD>(a.ny), stream,
                       a.dImg, q, y, dq, mode, B, d_tterms, n_ticks, dt, max_speed, a.roll_x, a.roll_dx);
    return hipGetLastError();
}

}  // namespace clik
