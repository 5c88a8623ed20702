// Shape-specialised PseudoInverseController mode evaluation.
//
// For a skill whose structure (ShapeDesc) is known at compile time, the whole
// control flow of one mode of reference pseudo_inverse.py:259-451 is decided by
// a constexpr *plan*: which constraints contribute a velocity, which only stack
// rows, when the stacked pseudo-inverse is "wide" (rows kept) or "tall" (Gram
// matrices), where the first-equality double processing happens, and which
// pushes nobody consumes any more.  The device code follows the plan with
// `if constexpr` and exact array sizes, so every index is static, no branch is
// left in the instruction stream and every matrix lives in registers.
#pragma once
#include "clik_device.hpp"

namespace clik {

struct TaskPlan {
    bool skip;          // VelocitySetConstraint (ignored) or inactive set
    bool cone;          // inactive SetConstraint: tangent-cone test after the scan
    bool contributes;   // adds a velocity term
    bool first;         // stack empty before this task
    bool quirk;         // first EqualityConstraint: processed and stacked twice
    bool wide_self;     // pinv(J) uses the wide (J J^T) branch
    bool const_j;
    bool set_rows;      // multidim SetConstraint rows: per-lane activation bits
    bool conv;          // converge_final_set_to_max term
    int  r_before;      // stacked rows before the task
    bool gram_before;   // stack held as Gram matrices before the task
    int  wide_before;   // rows held explicitly before the task (0 when gram)
    int  push_times;    // 0: nothing consumes the stack afterwards
    bool gram_after;
    int  r_after;
    bool c_is_g_before; // all stacked rows so far have activation 1 (C = G - lam I)
    bool c_is_g_after;
    int  wide_const_task; // >= 0: the explicit stack before the task is exactly the rows of this ONE
                          // constant-Jacobian task (pushed once): pinv(Ja) is the host-precomputed cpinv
};

struct ModePlan {
    TaskPlan t[SHAPE_MAX_TASKS];
    int max_wide;       // largest explicit row count the stack reaches
    bool any_cone;
    // explicit (wide-form) stack rows in push order: source task / row, and the slot
    // of the per-lane copy (-1: constant-Jacobian row, read from the skill image)
    int wide_task[CLIK_MAX_DOF];
    int wide_local[CLIK_MAX_DOF];
    int wide_store[CLIK_MAX_DOF];
    int n_store;        // per-lane row slots needed
    // role split (pinv_solve_static_split_kernel): the ONE consumer that projects through a Gram-form
    // stack, or -1.  helper_ok: there is exactly one and it is the last consumer, so a helper wave can
    // build and factor that stack while the main wave runs the solves that precede it.
    int gram_consumer;
    bool helper_ok;
    // Duplicated-row stack: the unique Gram consumer projects through c copies of the rows of ONE
    // state-dependent task (the doubly processed first EqualityConstraint, c = 2) with m < n rows.
    // Then  (c J'J + lam I)^-1 c J'J w  =  c J' (c J J' + lam I)^-1 J w   (push-through identity,
    // exact for the damped inverse): an m x m factorisation of the matrix the task already formed
    // instead of building and factoring the n x n Gram matrix.
    int dup_task;       // the stacked task, -1: not this form
    int dup_times;      // c
};

constexpr ModePlan make_plan(const ShapeDesc& sd, unsigned act)
{
    ModePlan mp{};
    const int n = sd.n;
    const int cap = sd.standard ? n - 1 : n;
    int r = 0, set_idx = 0;
    bool gram = false, c_is_g = true;
    // last task that reads the stack (contributing, not first)
    int last_consumer = -1;
    {
        int rr = 0, si = 0;
        for (int ti = 0; ti < sd.n_tasks; ++ti) {
            const int cls = sd.cls[ti];
            const bool is_set = cls == CLIK_CLS_SET;
            const bool active = is_set && ((act >> si) & 1u);
            if (is_set) ++si;
            if (cls == CLIK_CLS_VELSET || (is_set && !active)) continue;
            const bool conv = is_set && ti == sd.n_tasks - 1 && sd.conv_last;
            const bool contributes = cls == CLIK_CLS_EQ || cls == CLIK_CLS_VELEQ || conv;
            if (contributes && rr > 0) last_consumer = ti;
            rr += sd.m[ti] * ((contributes && rr == 0 && cls == CLIK_CLS_EQ) ? 2 : 1);
        }
    }
    for (int ti = 0; ti < sd.n_tasks; ++ti) {
        TaskPlan& p = mp.t[ti];
        const int cls = sd.cls[ti];
        const int m = sd.m[ti];
        const bool is_set = cls == CLIK_CLS_SET;
        const bool active = is_set && ((act >> set_idx) & 1u);
        if (is_set) ++set_idx;
        p.r_before = r;
        p.gram_before = gram;
        p.wide_before = gram ? 0 : r;
        p.wide_const_task = -1;
        if (!gram && r > 0) {
            const int t0 = mp.wide_task[0];
            bool single = sd.const_j[t0] != 0 && r == sd.m[t0];
            for (int k = 0; k < r && single; ++k) single = mp.wide_task[k] == t0 && mp.wide_local[k] == k;
            if (single) p.wide_const_task = t0;
        }
        p.c_is_g_before = c_is_g;
        p.const_j = sd.const_j[ti] != 0;
        if (cls == CLIK_CLS_VELSET) { p.skip = true; p.r_after = r; p.gram_after = gram; p.c_is_g_after = c_is_g; continue; }
        if (is_set && !active) {
            p.skip = true; p.cone = true; mp.any_cone = true;
            p.r_after = r; p.gram_after = gram; p.c_is_g_after = c_is_g;
            continue;
        }
        p.conv = is_set && ti == sd.n_tasks - 1 && sd.conv_last;
        p.contributes = cls == CLIK_CLS_EQ || cls == CLIK_CLS_VELEQ || p.conv;
        p.first = r == 0;
        p.quirk = p.contributes && p.first && cls == CLIK_CLS_EQ;
        p.wide_self = sd.standard ? (m < n) : (n >= m);
        p.set_rows = is_set && sd.multidim;
        const bool rows_flagged = p.set_rows;           // rows may carry activation 0
        const int times = p.quirk ? 2 : 1;
        const bool consumed = ti < last_consumer || (p.quirk && ti <= last_consumer);
        // a quirk task consumes its own first push even if nothing follows
        p.push_times = (ti < last_consumer) ? times : 0;
        // (the doubly processed first EqualityConstraint evaluates both of its passes in closed form from
        // its own factor - wide or tall, or the host-precomputed inverse of a constant Jacobian - and
        // needs no push of its own, see step_s)
        (void)consumed;
        if (p.push_times > 0) {
            const int r_new = r + p.push_times * m;
            if (!gram && r_new <= cap) {
                if (r_new > mp.max_wide) mp.max_wide = r_new;
                const int first_slot = mp.n_store;
                for (int rep = 0; rep < p.push_times; ++rep)
                    for (int i = 0; i < m; ++i) {
                        const int idx = r + rep * m + i;
                        mp.wide_task[idx] = ti;
                        mp.wide_local[idx] = i;
                        mp.wide_store[idx] = p.const_j ? -1 : first_slot + i;   // a repeated push shares the copy
                    }
                if (!p.const_j) mp.n_store += m;
            } else {
                gram = true;
            }
            r = r_new;
            if (rows_flagged) c_is_g = false;
        }
        p.r_after = r;
        p.gram_after = gram;
        p.c_is_g_after = c_is_g;
    }
    {
        int n_gram = 0;
        mp.gram_consumer = -1;
        for (int ti = 0; ti < sd.n_tasks; ++ti) {
            const TaskPlan& p = mp.t[ti];
            // consumers of the stack: contributing tasks that project (not the first, not the
            // double processing that owns its factor)
            const bool projects = !p.skip && p.contributes && !p.first;
            if (projects && p.gram_before) { ++n_gram; mp.gram_consumer = ti; }
        }
        mp.helper_ok = n_gram == 1 && mp.gram_consumer == last_consumer;
        if (!mp.helper_ok) mp.gram_consumer = -1;
        mp.dup_task = -1;
        mp.dup_times = 0;
        if (mp.helper_ok && !sd.standard) {
            int pushed = 0, src = -1;
            for (int ti = 0; ti < mp.gram_consumer; ++ti)
                if (!mp.t[ti].skip && mp.t[ti].push_times > 0) { ++pushed; src = ti; }
            if (pushed == 1) {
                const TaskPlan& p = mp.t[src];
                // (quirk && wide_self && !const_j: the task forms J J' + lam I itself; no rows with activation 0)
                if (p.quirk && p.wide_self && !p.const_j && !p.set_rows && sd.m[src] < n && mp.t[mp.gram_consumer].c_is_g_before) {
                    mp.dup_task = src;
                    mp.dup_times = p.push_times;
                }
            }
        }
    }
    return mp;
}

template <const ShapeDesc& SD, unsigned ACT>
struct Plan {
    static constexpr ModePlan mode = make_plan(SD, ACT);
};

template <const ShapeDesc& SD>
using Img = SkillImage<SD.nj, SD.n_tasks, shape_rows(SD)>;

// ---- forward kinematics with a compile-time chain --------------------------------
// One joint of  T = prod_j Trans(p_j) R_j [Rot(axis_j, z[q_j]) | Trans(axis_j z[q_j])]
// (URDF convention, what urdf2casadi builds for the reference).  Joint type,
// state index, "origin rotation is identity", "origin translation is zero" and
// axis alignment are compile-time, so the recursion is straight-line code and
// the per-joint axis / origin frames stay in registers.
template <const ShapeDesc& SD, int J>
__device__ __forceinline__ void fk_joint_s(const Img<SD>* __restrict__ S, const double (&z)[SD.n],
                                           const double (&sns)[SD.n], const double (&css)[SD.n],
                                           double (&R)[9], double (&p)[3], double (&ax)[SD.n][3],
                                           double (&org)[SD.n][3])
{
    if constexpr (J < SD.nj) {
        constexpr int jf = SD.jflags[J];
        constexpr int type = SD.jtype[J];
        const clik_joint& jt = S->joints[J];
        if constexpr (!(jf & 2)) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                p[i] = fma(R[3 * i], jt.p[0], fma(R[3 * i + 1], jt.p[1], fma(R[3 * i + 2], jt.p[2], p[i])));
        }
        if constexpr (!(jf & 1)) {
            double T[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    T[3 * i + c] = R[3 * i] * jt.R[c] + R[3 * i + 1] * jt.R[3 + c] + R[3 * i + 2] * jt.R[6 + c];
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = T[i];
        }
        if constexpr (type != CLIK_JOINT_FIXED) {
            constexpr int qi = SD.jq[J];
            constexpr int acode = (jf >> 4) & 7;
            constexpr int ak = (acode & 3) - 1;
            constexpr double asign = (acode & 4) ? -1.0 : 1.0;
            if constexpr (ak >= 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) ax[qi][i] = asign * R[3 * i + ak];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    ax[qi][i] = R[3 * i] * jt.axis[0] + R[3 * i + 1] * jt.axis[1] + R[3 * i + 2] * jt.axis[2];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) org[qi][i] = p[i];
            const double ang = z[qi];
            if constexpr (type == CLIK_JOINT_REVOLUTE) {
                const double sn = sns[qi], cs = css[qi];
                if constexpr (ak >= 0) {
                    // rotation about a local coordinate axis mixes the two other columns
                    const double s = asign * sn;
                    constexpr int c1 = (ak + 1) % 3, c2 = (ak + 2) % 3;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double u = R[3 * i + c1], v = R[3 * i + c2];
                        R[3 * i + c1] = fma(cs, u, s * v);
                        R[3 * i + c2] = fma(cs, v, -s * u);
                    }
                } else {
                    const double C = 1.0 - cs;
                    const double x = jt.axis[0], y = jt.axis[1], zz = jt.axis[2];
                    const double m[9] = {cs + x * x * C, x * y * C - zz * sn, x * zz * C + y * sn,
                                         y * x * C + zz * sn, cs + y * y * C, y * zz * C - x * sn,
                                         zz * x * C - y * sn, zz * y * C + x * sn, cs + zz * zz * C};
                    double T[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc)
                            T[3 * i + cc] = R[3 * i] * m[cc] + R[3 * i + 1] * m[3 + cc] + R[3 * i + 2] * m[6 + cc];
#pragma unroll
                    for (int i = 0; i < 9; ++i) R[i] = T[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) p[i] = fma(ax[qi][i], ang, p[i]);
            }
        }
        fk_joint_s<SD, J + 1>(S, z, sns, css, R, p, ax, org);
    }
}

constexpr int shape_state_type(const ShapeDesc& sd, int qi)
{
    for (int j = 0; j < sd.nj; ++j)
        if (sd.jtype[j] != CLIK_JOINT_FIXED && sd.jq[j] == qi) return sd.jtype[j];
    return CLIK_JOINT_FIXED;     // state variable not driven by the chain
}

// sin / cos of every revolute state variable (the caller may produce them differently: the team kernel
// splits them over the lanes of an instance, clik_pinv_team.hpp)
template <const ShapeDesc& SD>
__device__ __forceinline__ void fk_sincos_all(const double (&z)[SD.n], double (&sns)[SD.n], double (&css)[SD.n])
{
    constexpr int N = SD.n;
    // all sines / cosines first: n independent polynomial chains the scheduler can
    // interleave; inside the chain recursion they would serialise behind the
    // frame products (measured: FK was latency-, not issue-bound)
    bool huge = false;
    static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (shape_state_type(SD, j) == CLIK_JOINT_REVOLUTE) {
            sincos_fast(z[j], sns[j], css[j]);
            huge = huge | (fabs(z[j]) > kSinCosFastMax);
        } else {
            sns[j] = css[j] = 0.0;
        }
    });
    if (__builtin_expect(__ballot(huge) != 0ull, 0)) {
        // some lane has a joint angle beyond the fast path's range: redo those (cold)
        static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr (shape_state_type(SD, j) == CLIK_JOINT_REVOLUTE) {
                if (fabs(z[j]) > kSinCosFastMax) {
                    const SinCos r = sincos_slow(z[j]);
                    sns[j] = r.s;
                    css[j] = r.c;
                }
            }
        });
    }
}

template <const ShapeDesc& SD>
__device__ __forceinline__ void forward_kinematics_sc(const Img<SD>* __restrict__ S, const double (&z)[SD.n],
                                                      const double (&sns)[SD.n], const double (&css)[SD.n],
                                                      Kin<SD.n>& K)
{
    constexpr int N = SD.n;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double p[3] = {0, 0, 0};
    double ax[N][3], org[N][3];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) ax[j][i] = org[j][i] = 0.0;
    fk_joint_s<SD, 0>(S, z, sns, css, R, p, ax, org);
#pragma unroll
    for (int i = 0; i < 9; ++i) K.R[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) K.p[i] = p[i];
    static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int st = shape_state_type(SD, j);
        if constexpr (st == CLIK_JOINT_REVOLUTE) {
            const double r[3] = {p[0] - org[j][0], p[1] - org[j][1], p[2] - org[j][2]};
            double v[3];
            cross3(ax[j], r, v);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                K.Jv[i][j] = v[i];
                K.Jw[i][j] = ax[j][i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                K.Jv[i][j] = ax[j][i];      // prismatic: axis; not in the chain: 0
                K.Jw[i][j] = 0.0;
            }
        }
    });
}

template <const ShapeDesc& SD>
__device__ __forceinline__ void forward_kinematics_s(const Img<SD>* __restrict__ S, const double (&z)[SD.n],
                                                     Kin<SD.n>& K)
{
    double sns[SD.n], css[SD.n];
    fk_sincos_all<SD>(z, sns, css);
    forward_kinematics_sc<SD>(S, z, sns, css, K);
}

template <const ShapeDesc& SD>
__device__ __forceinline__ void orientation_feature_s(const Img<SD>* __restrict__ S, const double* ys,
                                                      const int lane, Kin<SD.n>& K)
{
    double q[4];
    if constexpr (SD.quat_src == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = ys[S->quat_yi[i]];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = S->quat[i];
    }
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double Rd[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                          2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                          2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    const double* R = K.R;
    // M = R Rd^T carries everything:  sum_c r_c x rd_c = vee(M^T - M)  and  tr(Rd^T R) = tr M
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            K.M[3 * i + k] = R[3 * i] * Rd[3 * k] + R[3 * i + 1] * Rd[3 * k + 1] + R[3 * i + 2] * Rd[3 * k + 2];
    K.o[0] = 0.5 * (K.M[5] - K.M[7]);
    K.o[1] = 0.5 * (K.M[6] - K.M[2]);
    K.o[2] = 0.5 * (K.M[1] - K.M[3]);
    K.tr = K.M[0] + K.M[4] + K.M[8];
}

// value, state gradient and time derivative of one affine row; feature flags,
// the number of input_var terms and the sparsity pattern of the feature
// coefficients are compile-time.  NZ / ONE: bit k set = coefficient k of
// [b0..b2 | g0..g8 | h0..h2] is non-zero / exactly 1 (ShapeDesc::row_nz, row_one):
// a pose task's rows pick single components (b = e_i or h = e_i), so most
// products and most coefficient reads disappear.
template <unsigned NZ, unsigned ONE, int K>
__device__ __forceinline__ double row_coef(const double& stored)
{
    if constexpr (((ONE >> K) & 1u) != 0) return 1.0;
    else return stored;
}

template <int N, int FLAGS, int NY, bool HAS_T, unsigned NZ = 0x7fffu, unsigned ONE = 0u>
__device__ __forceinline__ double row_eval_s(const clik_row& r, const int n_tslots, const TickArgs& tk,
                                             const Kin<N>& K, const double (&z)[N], const double* ys,
                                             const int lane, double (&g)[N], double& dt)
{
    double v = r.c;
    dt = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = 0.0;
    if constexpr ((FLAGS & CLIK_ROW_HAS_Q) != 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double a = r.a[j];
            g[j] = a;
            v = fma(a, z[j], v);
        }
    }
    constexpr bool has_p = (FLAGS & CLIK_ROW_HAS_P) != 0 && (NZ & 0x7u) != 0;
    constexpr bool has_r = (FLAGS & CLIK_ROW_HAS_R) != 0 && (NZ & 0xff8u) != 0;
    constexpr bool has_o = (FLAGS & CLIK_ROW_HAS_O) != 0 && (NZ & 0x7000u) != 0;
    double lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
    if constexpr (has_p) {
        static_for<0, 3>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (((NZ >> i) & 1u) != 0) {
                lin[i] = row_coef<NZ, ONE, i>(r.b[i]);
                v = fma(lin[i], K.p[i], v);
            }
        });
    }
    if constexpr (has_r) {
        static_for<0, 3>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            constexpr unsigned colmask = (1u << (3 + c)) | (1u << (6 + c)) | (1u << (9 + c));
            if constexpr ((NZ & colmask) != 0) {
                const double rc[3] = {K.R[c], K.R[3 + c], K.R[6 + c]};
                const double gc[3] = {r.g[c], r.g[3 + c], r.g[6 + c]};
                double u[3];
                cross3(rc, gc, u);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    ang[i] += u[i];
                    v = fma(gc[i], rc[i], v);
                }
            }
        });
    }
    if constexpr (has_o) {
        double hh[3] = {0, 0, 0};
        static_for<0, 3>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (((NZ >> (12 + i)) & 1u) != 0) {
                hh[i] = row_coef<NZ, ONE, 12 + i>(r.h[i]);
                v = fma(hh[i], K.o[i], v);
            }
        });
        static_for<0, 3>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            double mh = 0.0;
            static_for<0, 3>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if constexpr (((NZ >> (12 + k)) & 1u) != 0) mh = fma(K.M[3 * k + i], hh[k], mh);
            });
            if constexpr (((NZ >> (12 + i)) & 1u) != 0) ang[i] += -0.5 * (K.tr * hh[i] - mh);
            else ang[i] += 0.5 * mh;
        });
    }
    if constexpr (has_p || has_r || has_o) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = g[j];
            static_for<0, 3>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (has_p && ((NZ >> i) & 1u) != 0) s = fma(K.Jv[i][j], lin[i], s);
                if constexpr (has_r || has_o) s = fma(K.Jw[i][j], ang[i], s);
            });
            g[j] = s;
        }
    }
    if constexpr ((FLAGS & CLIK_ROW_HAS_Y) != 0) {
        // unused terms carry a zero coefficient and index 0
#pragma unroll
        for (int k = 0; k < NY; ++k) v = fma(r.yc[k], ys[r.yi[k]], v);
    }
    if constexpr (HAS_T) {
        // (only tasks with time terms pay for this data-dependent branch)
        const int slot = r.t_slot;
        if (slot >= 0) {
            v += tk.tv[slot];
            dt = tk.tv[n_tslots + slot];
        }
    }
    return v;
}

// ---- static-size linear algebra ----------------------------------------------------
template <int K>
__device__ __forceinline__ void ldl_factor_s(double (&A)[K * (K + 1) / 2], double (&rd)[K])
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t[K];
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
        for (int i = k + 1; i < K; ++i) {
            double s = A[tri(i, k)];
#pragma unroll
            for (int j = 0; j < k; ++j) s = fma(-A[tri(i, j)], t[j], s);
            A[tri(i, k)] = s * inv;
        }
    }
}

template <int K>
__device__ __forceinline__ void ldl_solve_s(const double (&A)[K * (K + 1) / 2], const double (&rd)[K],
                                            double (&x)[K])
{
#pragma unroll
    for (int i = 1; i < K; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) x[i] = fma(-A[tri(i, j)], x[j], x[i]);
#pragma unroll
    for (int i = 0; i < K; ++i) x[i] *= rd[i];
#pragma unroll
    for (int i = K - 2; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < K; ++j) x[i] = fma(-A[tri(j, i)], x[j], x[i]);
}

// Constraints written outside the affine-in-features family (CLIK_OUT_EXTERN, ShapeDesc::ext): the
// run-time instantiated translation unit specialises ExternTask<TI> with straight-line code generated from
// the caller's expression graph and its symbolic derivatives (casclik_amd/codegen.py) - what the reference
// gets from cs.jacobian + the CasADi JIT (constraints.py:67-73, pseudo_inverse.py:476-483).
// tv: the tick's time-slot values (TickArgs::tv), K: tool frame and its Jacobians.
template <int TI>
struct ExternTask {
    template <int N, int M>
    __device__ static void eval(const double (&)[N], const double*, const double*, const Kin<N>&, double (&)[M],
                                double (&)[M][N], double (&)[M])
    {
        static_assert(TI < 0, "ShapeDesc::ext is set for a task without generated code");
    }
};

// Constraint ATTRIBUTES given as expressions - a gain, set bounds or a velocity target that depend on
// (t, q, virtual, input): the reference multiplies / subtracts them symbolically (constraints.py:35-39, :90-92,
// pseudo_inverse.py:301-318, reactive_qp.py:199-232) and never differentiates them.  ShapeDesc::ext[ti] bits 1..4
// (CLIK_ATTR_* << 1) say which; the generated ExternAttr<TI>::eval writes their per-instance values into the
// task's slice of TaskCache::attr, laid out [gain (1 or m*m) | set_min (m) | set_max (m) | target (m)], present
// parts only.
template <int TI>
struct ExternAttr {
    template <int N>
    __device__ static void eval(const double (&)[N], const double*, const double*, const Kin<N>&, double*)
    {
        static_assert(TI < 0, "ShapeDesc::ext has attribute bits for a task without generated code");
    }
};
constexpr int shape_attr_bits(const ShapeDesc& sd, int ti) { return (sd.ext[ti] >> 1) & 15; }
constexpr int shape_attr_gain_size(const ShapeDesc& sd, int ti) { return sd.gain_matrix[ti] != 0 ? sd.m[ti] * sd.m[ti] : 1; }
// offset of part `bit` (CLIK_ATTR_*) inside the task's slice; its size with bit = 16
constexpr int shape_attr_off(const ShapeDesc& sd, int ti, int bit)
{
    const int b = shape_attr_bits(sd, ti);
    int o = 0;
    if (bit == CLIK_ATTR_GAIN) return o;
    if (b & CLIK_ATTR_GAIN) o += shape_attr_gain_size(sd, ti);
    if (bit == CLIK_ATTR_SET_MIN) return o;
    if (b & CLIK_ATTR_SET_MIN) o += sd.m[ti];
    if (bit == CLIK_ATTR_SET_MAX) return o;
    if (b & CLIK_ATTR_SET_MAX) o += sd.m[ti];
    if (bit == CLIK_ATTR_TARGET) return o;
    if (b & CLIK_ATTR_TARGET) o += sd.m[ti];
    return o;
}
constexpr int shape_attr_base(const ShapeDesc& sd, int ti)
{
    int r = 0;
    for (int i = 0; i < ti; ++i) r += shape_attr_off(sd, i, 16);
    return r;
}
constexpr int shape_attr_total(const ShapeDesc& sd) { return shape_attr_base(sd, sd.n_tasks); }

// e, J, d e/d t of task TI: rows are contiguous and all affine in static shapes
template <const ShapeDesc& SD, int TI>
__device__ __forceinline__ void task_eval_s(const Img<SD>* __restrict__ S, const TickArgs& tk,
                                            const Kin<SD.n>& K, const double (&z)[SD.n], const double* ys,
                                            const int lane, double (&e)[SD.m[TI]], double (&J)[SD.m[TI]][SD.n],
                                            double (&Jt)[SD.m[TI]])
{
    constexpr int N = SD.n;
    constexpr int M = SD.m[TI];
    if constexpr ((SD.ext[TI] & 1) != 0) {
        ExternTask<TI>::template eval<N, M>(z, ys, tk.tv, K, e, J, Jt);
        return;
    }
    const int nts = S->n_tslots;
    static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int r0 = shape_out_row0(SD, TI, i);
        if constexpr (SD.out_nrows[TI][i] == 0) {
            constexpr unsigned NZ = SD.row_nz[r0], ONE = SD.row_one[r0];
            double g[N], dt;
            e[i] = row_eval_s<N, SD.flags[TI], SD.ny_terms[TI], SD.has_t[TI] != 0, NZ, ONE>(S->rows[r0], nts, tk, K, z,
                                                                                             ys, lane, g, dt);
            Jt[i] = dt;
#pragma unroll
            for (int j = 0; j < N; ++j) J[i][j] = g[j];
        } else {
            // 2-norm of a group of affine rows (cs.norm_2 / cs.norm_fro): e = |r|, J = r'G / |r|
            // (0/0 at r = 0, as the reference's symbolic derivative)
            constexpr int NRW = SD.out_nrows[TI][i];
            double ss = 0.0, tacc = 0.0, acc[N];
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] = 0.0;
            static_for<0, NRW>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr unsigned NZ = SD.row_nz[r0 + k], ONE = SD.row_one[r0 + k];
                double g[N], dt;
                const double v = row_eval_s<N, SD.flags[TI], SD.ny_terms[TI], SD.has_t[TI] != 0, NZ, ONE>(
                    S->rows[r0 + k], nts, tk, K, z, ys, lane, g, dt);
                ss = fma(v, v, ss);
                tacc = fma(v, dt, tacc);
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = fma(v, g[j], acc[j]);
            });
            const double nrm = sqrt(ss);
            const double inv = 1.0 / nrm;
            e[i] = nrm;
            Jt[i] = tacc * inv;
#pragma unroll
            for (int j = 0; j < N; ++j) J[i][j] = acc[j] * inv;
        }
    });
}

template <int M, bool MATRIX, class TASK>
__device__ __forceinline__ void gain_apply_s(const TASK& t, const double (&v)[M], double (&out)[M])
{
    if constexpr (!MATRIX) {
        const double g = t.gain[0];
#pragma unroll
        for (int i = 0; i < M; ++i) out[i] = g * v[i];
    } else {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) s = fma(t.gain[i * M + k], v[k], s);
            out[i] = s;
        }
    }
}

// in-tangent-cone test with compile-time row count (pseudo_inverse.py:162-185, :222-252)
template <int N, int M, class TASK>
__device__ __forceinline__ bool in_tangent_cone_s(const TASK& t, const double (&e)[M],
                                                  const double (&J)[M][N], const double (&Jt)[M],
                                                  const double (&v)[N])
{
    double de[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double s = Jt[i];
#pragma unroll
        for (int j = 0; j < N; ++j) s = fma(J[i][j], v[j], s);
        de[i] = s;
    }
    if constexpr (M == 1) {
        const double ev = e[0];
        if (t.set_min[0] - ev < 1e-12) return (ev - t.set_max[0] < 1e-12) ? true : (de[0] < 0.0);
        return de[0] > 0.0;
    } else {
        bool inside = true, corner = true;
        double od = 0.0, nde = 0.0, nout = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
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
        bool going_in;
        if (corner) {
            const double dists = (sqrt(nde) + 1e-10) * sqrt(nout);
            going_in = (od < 0.0) ? (fabs(od) / dists < 0.70710678118654757) : false;
        } else {
            going_in = od < 0.0;
        }
        return inside ? true : going_in;
    }
}

// ---- per-tick cache of the state-dependent task evaluations --------------------------
// e, J, d e/d t of every constraint whose Jacobian depends on the state are
// evaluated once per tick (they do not depend on the mode) so that the FK state
// is dead before the mode scan starts.  Constant-Jacobian constraints are never
// materialised: their coefficients are read from the skill image where used.
constexpr int shape_cache_base(const ShapeDesc& sd, int ti)
{
    int r = 0;
    for (int i = 0; i < ti; ++i)
        if (!sd.const_j[i]) r += sd.m[i];
    return r;
}
constexpr int shape_cache_rows(const ShapeDesc& sd) { return shape_cache_base(sd, sd.n_tasks); }

template <const ShapeDesc& SD>
struct TaskCache {
    static constexpr int ROWS = shape_cache_rows(SD) > 0 ? shape_cache_rows(SD) : 1;
    double e[ROWS];
    double Jt[ROWS];
    double J[ROWS][SD.n];
    // per-instance values of the attributes given as expressions (ExternAttr)
    static constexpr int NATTR = shape_attr_total(SD) > 0 ? shape_attr_total(SD) : 1;
    double attr[NATTR];
};

// the constants of task TI as the tick uses them: the image's task record, or - when some of them are expressions -
// a small per-lane record with those parts replaced by the values ExternAttr computed for this instance (same member
// names: the code below is written against either)
template <const ShapeDesc& SD, int TI>
struct TaskConstsS {
    static constexpr int M = SD.m[TI];
    double gain[shape_attr_gain_size(SD, TI)];
    double set_min[M], set_max[M], target[M];
};
template <const ShapeDesc& SD, int TI>
__device__ __forceinline__ decltype(auto) task_consts(const Img<SD>* __restrict__ S, const TaskCache<SD>& tc)
{
    constexpr int bits = shape_attr_bits(SD, TI);
    if constexpr (bits == 0) {
        return (S->tasks[TI]);          // (a reference into the image)
    } else {
        constexpr int M = SD.m[TI];
        constexpr int base = shape_attr_base(SD, TI);
        constexpr int MG = shape_attr_gain_size(SD, TI);
        // (the offsets must be constant expressions: evaluated at run time they make every access a dynamically
        // indexed one, which pins the whole cache in scratch)
        constexpr int og = base + shape_attr_off(SD, TI, CLIK_ATTR_GAIN);
        constexpr int olo = base + shape_attr_off(SD, TI, CLIK_ATTR_SET_MIN);
        constexpr int ohi = base + shape_attr_off(SD, TI, CLIK_ATTR_SET_MAX);
        constexpr int otg = base + shape_attr_off(SD, TI, CLIK_ATTR_TARGET);
        const clik_task& ti = S->tasks[TI];
        TaskConstsS<SD, TI> t;
        static_for<0, MG>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr ((bits & CLIK_ATTR_GAIN) != 0) t.gain[k] = tc.attr[og + k];
            else t.gain[k] = ti.gain[k];
        });
        static_for<0, M>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr ((bits & CLIK_ATTR_SET_MIN) != 0) t.set_min[k] = tc.attr[olo + k];
            else t.set_min[k] = ti.set_min[k];
            if constexpr ((bits & CLIK_ATTR_SET_MAX) != 0) t.set_max[k] = tc.attr[ohi + k];
            else t.set_max[k] = ti.set_max[k];
            if constexpr ((bits & CLIK_ATTR_TARGET) != 0) t.target[k] = tc.attr[otg + k];
            else t.target[k] = ti.target[k];
        });
        return t;
    }
}

// ALL: also VelocitySetConstraints (the QP controller uses them, the pseudo-inverse one ignores them)
template <const ShapeDesc& SD, int TI, bool ALL = false>
__device__ __forceinline__ void cache_task(const Img<SD>* __restrict__ S, const TickArgs& tk, const Kin<SD.n>& K,
                                           const double (&z)[SD.n], const double* ys, const int lane,
                                           TaskCache<SD>& tc)
{
    if constexpr (TI < SD.n_tasks) {
        if constexpr (shape_attr_bits(SD, TI) != 0 && (ALL || SD.cls[TI] != CLIK_CLS_VELSET)) {
            constexpr int ab = shape_attr_base(SD, TI);
            constexpr int an = shape_attr_off(SD, TI, 16);
            double a[an];
            ExternAttr<TI>::template eval<SD.n>(z, ys, tk.tv, K, a);
            static_for<0, an>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                tc.attr[ab + k] = a[k];
            });
        }
        if constexpr (!SD.const_j[TI] && (ALL || SD.cls[TI] != CLIK_CLS_VELSET)) {
            constexpr int N = SD.n;
            constexpr int M = SD.m[TI];
            constexpr int cb = shape_cache_base(SD, TI);
            double e[M], J[M][N], Jt[M];
            task_eval_s<SD, TI>(S, tk, K, z, ys, lane, e, J, Jt);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                tc.e[cb + i] = e[i];
                tc.Jt[cb + i] = Jt[i];
#pragma unroll
                for (int j = 0; j < N; ++j) tc.J[cb + i][j] = J[i][j];
            }
        }
        cache_task<SD, TI + 1, ALL>(S, tk, K, z, ys, lane, tc);
    }
}

// ---- per-mode state ------------------------------------------------------------------
template <int N, int STORE>
struct StackS {
    double   rows[STORE > 0 ? STORE : 1][N];  // per-lane copies of the state-dependent wide rows
    uint32_t sbits;                           // activation of the explicit rows
    double   G[N * (N + 1) / 2];              // lam I + Ja^T Ja       (gram form)
    double   C[N * (N + 1) / 2];              // Ja^T diag(s) Ja       (gram form, when != G - lam I)
};

// mutable per-mode state: plain arrays only (no pointers / references) so that
// scalar replacement keeps every element in a register
// ROLE: 0 = one wave evaluates the whole mode; ROLE_MAIN = the Gram-form stack is built and
// factored by a helper wave (helper_mode_static) and received through LDS (xch)
constexpr int ROLE_SOLO = 0, ROLE_MAIN = 1, ROLE_HELPER = 2;   // helper: builds the stack only, always in full
template <const ShapeDesc& SD, unsigned ACT, int ROLE = 0>
struct ModeCtx {
    static constexpr int N = SD.n;
    double lam;
    double v[N];
    StackS<N, Plan<SD, ACT>::mode.n_store> st;
    bool ok;
    uint32_t srows[SHAPE_MAX_TASKS];     // ROLE_MAIN: activation bits of the tasks pushed in Gram form
    const double* xch;                   // ROLE_MAIN: factor published by the helper wave ([slot][lane])
    // duplicated-row stack (ModePlan::dup_task): J J' + lam I of that task, before factorisation
    static constexpr int DM = Plan<SD, ACT>::mode.dup_task >= 0 ? SD.m[Plan<SD, ACT>::mode.dup_task > 0 ? Plan<SD, ACT>::mode.dup_task : 0] : 1;
    double dupM[DM * (DM + 1) / 2];
};

// read-only inputs of a mode evaluation, passed as separate parameters
#define CLIK_MODE_IN const Img<SD>* __restrict__ S, const TickArgs& tk, const TaskCache<SD>& tc, \
                     const double (&z)[SD.n], const double* ys, const int lane
#define CLIK_MODE_ARGS S, tk, tc, z, ys, lane
// step_s / cone_s take their own register copy of the image fields they read (S_in may point
// into LDS - the two-wave kernel passes the LDS image so that only the current task's constants
// occupy registers - or to a caller's register copy, in which case this copy is free)
#define CLIK_MODE_IN_RAW const Img<SD>* __restrict__ S_in, const TickArgs& tk, const TaskCache<SD>& tc, \
                         const double (&z)[SD.n], const double* ys, const int lane
#define CLIK_TASK_IMAGE(S, S_in)            \
    const Img<SD> S##_copy = *S_in;         \
    __builtin_amdgcn_sched_barrier(0);      \
    const Img<SD>* __restrict__ S = &S##_copy

// J[i][j] of task TI: cached per-lane value or image coefficient
template <const ShapeDesc& SD, int TI>
__device__ __forceinline__ double jac(const Img<SD>* __restrict__ S, const TaskCache<SD>& tc, const int i,
                                      const int j)
{
    // the bases must be constant expressions here: evaluated at run time they turn every access into a
    // dynamically indexed one, which pins the whole cache in scratch
    if constexpr (SD.const_j[TI] != 0) {
        constexpr int row0 = shape_row_base(SD, TI);
        return S->rows[row0 + i].a[j];
    } else {
        constexpr int cb = shape_cache_base(SD, TI);
        return tc.J[cb + i][j];
    }
}

// element j of explicit stack row R (compile-time R)
template <const ShapeDesc& SD, unsigned ACT, int R, int ROLE = 0>
__device__ __forceinline__ double stack_row(const Img<SD>* __restrict__ S, const ModeCtx<SD, ACT, ROLE>& c, const int j)
{
    constexpr ModePlan MP = Plan<SD, ACT>::mode;
    if constexpr (MP.wide_store[R] >= 0) {
        constexpr int sr = MP.wide_store[R];
        return c.st.rows[sr][j];
    } else {
        constexpr int gr = shape_row_base(SD, MP.wide_task[R]) + MP.wide_local[R];
        return S->rows[gr].a[j];
    }
}

// e (and d e/d t) of task TI without materialising a Jacobian
template <const ShapeDesc& SD, int TI>
__device__ __forceinline__ void task_values(const Img<SD>* __restrict__ S, const TickArgs& tk,
                                            const TaskCache<SD>& tc, const double (&z)[SD.n], const double* ys,
                                            const int lane, double (&e)[SD.m[TI]], double (&Jt)[SD.m[TI]])
{
    constexpr int N = SD.n;
    constexpr int M = SD.m[TI];
    if constexpr (shape_unit(SD, TI)) {
        // joint-space rows: e_i = z[col_i] + c_i (+ input / time terms); the zero coefficients are never read
        constexpr int row0 = shape_row_base(SD, TI);
        const int nts = S->n_tslots;
        static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int col = SD.ucol[TI][i] - 1;
            const clik_row& r = S->rows[row0 + i];
            double v = r.c + z[col], dt = 0.0;
            if constexpr ((SD.flags[TI] & CLIK_ROW_HAS_Y) != 0) {
#pragma unroll
                for (int k = 0; k < SD.ny_terms[TI]; ++k) v = fma(r.yc[k], ys[r.yi[k]], v);
            }
            if constexpr (SD.has_t[TI] != 0) {
                const int slot = r.t_slot;
                if (slot >= 0) {
                    v += tk.tv[slot];
                    dt = tk.tv[nts + slot];
                }
            }
            e[i] = v;
            Jt[i] = dt;
        });
    } else if constexpr (SD.const_j[TI] != 0) {
        constexpr int row0 = shape_row_base(SD, TI);
        const int nts = S->n_tslots;
        Kin<N> nokin;       // constant-Jacobian rows use no kinematic feature (flags exclude P/R/O)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double g[N], dt;
            e[i] = row_eval_s<N, SD.flags[TI] & (CLIK_ROW_HAS_Q | CLIK_ROW_HAS_Y), SD.ny_terms[TI], SD.has_t[TI] != 0>(
                S->rows[row0 + i], nts, tk, nokin, z, ys, lane, g, dt);
            Jt[i] = dt;
        }
    } else {
        constexpr int cb = shape_cache_base(SD, TI);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            e[i] = tc.e[cb + i];
            Jt[i] = tc.Jt[cb + i];
        }
    }
}

// w <- w - pinv(stack) * rJa * w  for the stack state BEFORE task TI
template <const ShapeDesc& SD, unsigned ACT, int TI, int ROLE = 0>
__device__ __forceinline__ void project_s(const Img<SD>* __restrict__ S, const TaskCache<SD>& tc,
                                          ModeCtx<SD, ACT, ROLE>& c, double (&w)[SD.n])
{
    constexpr int N = SD.n;
    constexpr ModePlan MPL = Plan<SD, ACT>::mode;
    constexpr TaskPlan P = MPL.t[TI];
    constexpr int NT = N * (N + 1) / 2;
    if constexpr (P.gram_before && ROLE == ROLE_SOLO && MPL.dup_task >= 0) {
        static_assert(MPL.gram_consumer == TI, "duplicated-row projection belongs to the unique Gram consumer");
        constexpr int T0 = MPL.dup_task;
        constexpr int M0 = SD.m[T0];
        constexpr double cd = (double)MPL.dup_times;
        double L[M0 * (M0 + 1) / 2], rd[M0], t[M0];
        // c (J J' + lam I) - (c - 1) lam I  =  c J J' + lam I
#pragma unroll
        for (int i = 0; i < M0; ++i)
#pragma unroll
            for (int k = 0; k <= i; ++k)
                L[tri(i, k)] = (i == k) ? fma(cd, c.dupM[tri(i, k)], -(cd - 1.0) * c.lam) : cd * c.dupM[tri(i, k)];
#pragma unroll
        for (int i = 0; i < M0; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) sacc = fma(jac<SD, T0>(S, tc, i, j), w[j], sacc);
            t[i] = sacc;
        }
        ldl_factor_s<M0>(L, rd);
        ldl_solve_s<M0>(L, rd, t);
#pragma unroll
        for (int i = 0; i < M0; ++i) t[i] *= cd;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double sacc = w[j];
#pragma unroll
            for (int i = 0; i < M0; ++i) sacc = fma(-jac<SD, T0>(S, tc, i, j), t[i], sacc);
            w[j] = sacc;
        }
    } else if constexpr (P.gram_before && ROLE == ROLE_MAIN) {
        // The Gram matrix lam I + Ja'Ja is built and factored by the helper wave.  Here:
        //   u = Ja' diag(s) Ja w  straight from the Jacobians of the stacked tasks,
        //   then  w -= (lam I + Ja'Ja)^-1 u  with the factor received through LDS.
        static_assert(MPL.helper_ok && MPL.gram_consumer == TI, "role split needs the unique Gram consumer");
        double u[N];
#pragma unroll
        for (int a = 0; a < N; ++a) u[a] = 0.0;
        static_for<0, TI>([&](auto tn) __attribute__((always_inline)) {
            constexpr int T = decltype(tn)::value;
            constexpr TaskPlan PT = MPL.t[T];
            if constexpr (!PT.skip && PT.push_times > 0) {
                constexpr int MT = SD.m[T];
                constexpr double times = (double)PT.push_times;
                const uint32_t sr = c.srows[T];
                static_for<0, MT>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (shape_unit(SD, T)) {
                        constexpr int col = SD.ucol[T][i] - 1;
                        u[col] = fma(times, ((sr >> i) & 1u) ? w[col] : 0.0, u[col]);
                    } else {
                        double sacc = 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) sacc = fma(jac<SD, T>(S, tc, i, j), w[j], sacc);
                        sacc = ((sr >> i) & 1u) ? times * sacc : 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) u[j] = fma(jac<SD, T>(S, tc, i, j), sacc, u[j]);
                    }
                });
            }
        });
        __syncthreads();            // the helper wave has published L and 1/d
        double L[NT], rd[N];
#pragma unroll
        for (int a = 0; a < NT; ++a) L[a] = c.xch[a * WAVE];
#pragma unroll
        for (int a = 0; a < N; ++a) rd[a] = c.xch[(NT + a) * WAVE];
        ldl_solve_s<N>(L, rd, u);
#pragma unroll
        for (int a = 0; a < N; ++a) w[a] -= u[a];
    } else if constexpr (P.gram_before) {
        double u[N], L[NT], rd[N];
#pragma unroll
        for (int a = 0; a < N; ++a) u[a] = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double cc;
                if constexpr (P.c_is_g_before) cc = (a == b) ? c.st.G[tri(a, b)] - c.lam : c.st.G[tri(a, b)];
                else cc = c.st.C[tri(a, b)];
                u[a] = fma(cc, w[b], u[a]);
                if (b != a) u[b] = fma(cc, w[a], u[b]);
            }
#pragma unroll
        for (int a = 0; a < NT; ++a) L[a] = c.st.G[a];
        ldl_factor_s<N>(L, rd);
        ldl_solve_s<N>(L, rd, u);
#pragma unroll
        for (int a = 0; a < N; ++a) w[a] -= u[a];
    } else {
        constexpr int R = P.wide_before;
        static_assert(R > 0, "projection with an empty stack");
        if constexpr (P.wide_const_task >= 0) {
            // stack = rows of one constant-Jacobian task: pinv(Ja) was formed on the host,
            //   w -= pinv(Ja) * (s o (Ja w))      (no Gram matrix, no factorisation)
            constexpr int T0 = P.wide_const_task;
            const double* Pm = S->cpinv[T0];
            if constexpr (shape_unit(SD, T0)) {
                // joint-space stack: Ja w picks components, pinv(Ja) has one entry per row
                static_for<0, R>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int col = SD.ucol[T0][i] - 1;
                    const double u = ((c.st.sbits >> i) & 1u) ? w[col] : 0.0;
                    w[col] = fma(-Pm[col * CLIK_MAX_M + i], u, w[col]);
                });
                return;
            }
            double u[R];
            static_for<0, R>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                double sacc = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) sacc = fma(stack_row<SD, ACT, i, ROLE>(S, c, j), w[j], sacc);
                u[i] = ((c.st.sbits >> i) & 1u) ? sacc : 0.0;
            });
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
                for (int i = 0; i < R; ++i) w[j] = fma(-Pm[j * CLIK_MAX_M + i], u[i], w[j]);
            return;
        }
        double u[R], L[R * (R + 1) / 2], rd[R];
        static_for<0, R>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) s = fma(stack_row<SD, ACT, i, ROLE>(S, c, j), w[j], s);
            u[i] = ((c.st.sbits >> i) & 1u) ? s : 0.0;
            static_for<0, i + 1>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                double acc = (k == i) ? c.lam : 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    acc = fma(stack_row<SD, ACT, i, ROLE>(S, c, j), stack_row<SD, ACT, k, ROLE>(S, c, j), acc);
                L[tri(i, k)] = acc;
            });
        });
        ldl_factor_s<R>(L, rd);
        ldl_solve_s<R>(L, rd, u);
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int j = 0; j < N; ++j) w[j] = fma(-u[k], stack_row<SD, ACT, k, ROLE>(S, c, j), w[j]);
        });
    }
}

// stack the rows of task TI (plan.push_times times) following the plan
template <const ShapeDesc& SD, unsigned ACT, int TI, int ROLE = 0>
__device__ __forceinline__ void push_s(const Img<SD>* __restrict__ S, const TaskCache<SD>& tc, ModeCtx<SD, ACT, ROLE>& c,
                                       const uint32_t srow)
{
    constexpr int N = SD.n;
    constexpr int M = SD.m[TI];
    constexpr ModePlan MP = Plan<SD, ACT>::mode;
    constexpr TaskPlan P = MP.t[TI];
    constexpr int TIMES = P.push_times;
    if constexpr (ROLE == ROLE_MAIN) {
        c.srows[TI] = srow;
        if constexpr (P.gram_after) return;        // Gram form: the helper wave's job
    }
    if constexpr (ROLE == ROLE_SOLO && MP.dup_task >= 0 && P.gram_after) return;   // consumer uses the m x m form
    if constexpr (!P.gram_after) {
        // stays wide: record the activation bits; state-dependent rows get a per-lane copy
        constexpr int r0 = P.r_after - TIMES * M;
        if constexpr (!P.const_j) {
            static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int sr = MP.wide_store[r0 + i];
#pragma unroll
                for (int j = 0; j < N; ++j) c.st.rows[sr][j] = jac<SD, TI>(S, tc, i, j);
            });
        }
        const uint32_t bits = srow & ((1u << M) - 1u);
#pragma unroll
        for (int rep = 0; rep < TIMES; ++rep) c.st.sbits |= bits << (r0 + rep * M);
    } else {
        constexpr int r_prev = P.r_after - TIMES * M;       // rows before this push
        constexpr bool was_gram = P.gram_before;
        // one Gram entry at a time (no temporaries of matrix size): old rows
        // (when converting from the wide form), then the new rows
        static_for<0, N>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            static_for<0, a + 1>([&](auto bc) __attribute__((always_inline)) {
                constexpr int b = decltype(bc)::value;
                double g, cc = 0.0;
                if constexpr (!was_gram) {
                    g = (a == b) ? c.lam : 0.0;
                    if constexpr (P.wide_const_task >= 0 && shape_unit(SD, P.wide_const_task)) {
                        // joint-space rows: J^T J and J^T diag(s) J are diagonal 0/1 matrices
                        constexpr int k = (a == b) ? shape_unit_row(SD, P.wide_const_task, a) : -1;
                        if constexpr (k >= 0) {
                            g += 1.0;
                            if constexpr (!P.c_is_g_before) cc = ((c.st.sbits >> k) & 1u) ? 1.0 : 0.0;
                        }
                    } else if constexpr (P.wide_const_task >= 0) {
                        // the explicit rows are one constant-Jacobian task: its J^T J is host-precomputed;
                        // only the activation-weighted sum needs the rows
                        g += S->cjtj[P.wide_const_task][tri(a, b)];
                        if constexpr (!P.c_is_g_before) {
                            static_for<0, r_prev>([&](auto kc) __attribute__((always_inline)) {
                                constexpr int k = decltype(kc)::value;
                                const double pr = stack_row<SD, ACT, k, ROLE>(S, c, a) * stack_row<SD, ACT, k, ROLE>(S, c, b);
                                cc += ((c.st.sbits >> k) & 1u) ? pr : 0.0;
                            });
                        }
                    } else {
                        static_for<0, r_prev>([&](auto kc) __attribute__((always_inline)) {
                            constexpr int k = decltype(kc)::value;
                            const double pr = stack_row<SD, ACT, k, ROLE>(S, c, a) * stack_row<SD, ACT, k, ROLE>(S, c, b);
                            g += pr;
                            if constexpr (!P.c_is_g_before) cc += ((c.st.sbits >> k) & 1u) ? pr : 0.0;
                        });
                    }
                } else {
                    g = c.st.G[tri(a, b)];
                    if constexpr (!P.c_is_g_before) cc = c.st.C[tri(a, b)];
                }
                if constexpr (P.c_is_g_before && !P.c_is_g_after)
                    cc = (a == b) ? g - c.lam : g;      // C starts to differ from G - lam I here
                double acc, accs = 0.0;
                if constexpr (shape_unit(SD, TI)) {
                    constexpr int k = (a == b) ? shape_unit_row(SD, TI, a) : -1;
                    acc = (k >= 0) ? 1.0 : 0.0;
                    if constexpr (P.set_rows && k >= 0) accs = ((srow >> k) & 1u) ? 1.0 : 0.0;
                } else if constexpr (P.const_j) {
                    acc = S->cjtj[TI][tri(a, b)];       // J^T J of a constant Jacobian: host-precomputed
                    if constexpr (P.set_rows) {
#pragma unroll
                        for (int i = 0; i < M; ++i)
                            accs += ((srow >> i) & 1u) ? jac<SD, TI>(S, tc, i, a) * jac<SD, TI>(S, tc, i, b) : 0.0;
                    }
                } else {
                    acc = 0.0;
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        const double pr = jac<SD, TI>(S, tc, i, a) * jac<SD, TI>(S, tc, i, b);
                        acc += pr;
                        if constexpr (P.set_rows) accs += ((srow >> i) & 1u) ? pr : 0.0;
                    }
                }
                c.st.G[tri(a, b)] = fma((double)TIMES, acc, g);
                if constexpr (!P.c_is_g_after) c.st.C[tri(a, b)] = fma((double)TIMES, P.set_rows ? accs : acc, cc);
            });
        });
    }
}

template <const ShapeDesc& SD, unsigned ACT, int TI, int ROLE = 0>
__device__ __forceinline__ void step_s(CLIK_MODE_IN_RAW, ModeCtx<SD, ACT, ROLE>& c)
{
    constexpr int N = SD.n;
    constexpr int M = SD.m[TI];
    constexpr TaskPlan P = Plan<SD, ACT>::mode.t[TI];
    if constexpr (!P.skip) {
        CLIK_TASK_IMAGE(S, S_in);
        decltype(auto) t = task_consts<SD, TI>(S, tc);
        double e[M], Jt[M];
        task_values<SD, TI>(S, tk, tc, z, ys, lane, e, Jt);
        uint32_t srow = 0xffffffffu;
        if constexpr (P.set_rows) {
            srow = 0u;
#pragma unroll
            for (int i = 0; i < M; ++i)
                srow |= (uint32_t)((e[i] - t.set_max[i] > 0.0) | (e[i] - t.set_min[i] < 0.0)) << i;     // (bitwise: no branch per row)
        }
        if constexpr (!P.contributes) {
            if constexpr (P.push_times > 0) push_s<SD, ACT, TI, ROLE>(S, tc, c, srow);
        } else {
            double des[M];
            if constexpr (SD.cls[TI] == CLIK_CLS_EQ) {
                double ke[M];
                gain_apply_s<M, SD.gain_matrix[TI] != 0>(t, e, ke);
#pragma unroll
                for (int i = 0; i < M; ++i) des[i] = -ke[i];
            } else if constexpr (SD.cls[TI] == CLIK_CLS_VELEQ) {
#pragma unroll
                for (int i = 0; i < M; ++i) des[i] = t.target[i];
            } else {
                double d0[M];
#pragma unroll
                for (int i = 0; i < M; ++i) d0[i] = t.set_max[i] - e[i];
                gain_apply_s<M, SD.gain_matrix[TI] != 0>(t, d0, des);
            }
            if constexpr (SD.feedforward != 0) {
#pragma unroll
                for (int i = 0; i < M; ++i) des[i] -= Jt[i];
            }
            double w[N];
            constexpr bool own_factor = !P.const_j && P.wide_self;
            double L[own_factor ? M * (M + 1) / 2 : 1], rd[own_factor ? M : 1];
            if constexpr (shape_unit(SD, TI)) {
                const double* Pm = S->cpinv[TI];
#pragma unroll
                for (int j = 0; j < N; ++j) w[j] = 0.0;
                static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int col = SD.ucol[TI][i] - 1;
                    w[col] = Pm[col * CLIK_MAX_M + i] * des[i];
                    // (doubly processed first equality, see the constant-Jacobian branch below: J w = w[col])
                    if constexpr (P.quirk) w[col] = fma(-Pm[col * CLIK_MAX_M + i], w[col], 2.0 * w[col]);
                });
            } else if constexpr (P.const_j) {
                const double* Pm = S->cpinv[TI];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < M; ++i) s = fma(Pm[j * CLIK_MAX_M + i], des[i], s);
                    w[j] = s;
                }
                if constexpr (P.quirk) {
                    // Doubly processed first EqualityConstraint with a CONSTANT Jacobian (the path variable's
                    // `300 - x` of cart_on_track_1D...ipynb cell 75, a posture task): pass 1 is w = P d with the
                    // host-precomputed P = pinv(J); pass 2 (:382-396) projects the same term through the stack
                    // [J]: (I - P J) w.  The sum: 2 w - P (J w).
                    double jw[M];
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        double s = 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) s = fma(jac<SD, TI>(S, tc, i, j), w[j], s);
                        jw[i] = s;
                    }
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < M; ++i) s = fma(Pm[j * CLIK_MAX_M + i], jw[i], s);
                        w[j] = fma(2.0, w[j], -s);
                    }
                }
            } else if constexpr (P.wide_self) {
#pragma unroll
                for (int i = 0; i < M; ++i)
#pragma unroll
                    for (int k = 0; k <= i; ++k) {
                        double acc = (k == i) ? c.lam : 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) acc = fma(jac<SD, TI>(S, tc, i, j), jac<SD, TI>(S, tc, k, j), acc);
                        L[tri(i, k)] = acc;
                    }
                if constexpr (ROLE == ROLE_SOLO && Plan<SD, ACT>::mode.dup_task == TI) {
#pragma unroll
                    for (int a = 0; a < M * (M + 1) / 2; ++a) c.dupM[a] = L[a];
                }
                ldl_factor_s<M>(L, rd);
                ldl_solve_s<M>(L, rd, des);
                if constexpr (P.quirk) {
                    // Both passes of the doubly processed first EqualityConstraint at once.  With
                    // y = M^-1 d, M = J J' + lam I:  pass 1 gives w = J'y, pass 2 (stack [J], :382-396)
                    // gives w - J'M^-1 J w = w - J'M^-1 (M - lam I) y = lam J' M^-1 y, so the sum is
                    //     J' (y + lam M^-1 y)
                    // : one more m x m solve, one J' product, and none of the cancellation of the
                    // literal form (same value up to its rounding error).
                    double y2[M];
#pragma unroll
                    for (int i = 0; i < M; ++i) y2[i] = des[i];
                    ldl_solve_s<M>(L, rd, y2);
#pragma unroll
                    for (int i = 0; i < M; ++i) des[i] = fma(c.lam, y2[i], des[i]);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < M; ++i) s = fma(jac<SD, TI>(S, tc, i, j), des[i], s);
                    w[j] = s;
                }
            } else {
                double Lg[N * (N + 1) / 2], rg[N];
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < M; ++i) s = fma(jac<SD, TI>(S, tc, i, a), des[i], s);
                    w[a] = s;
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        double acc = (a == b) ? c.lam : 0.0;
#pragma unroll
                        for (int i = 0; i < M; ++i) acc = fma(jac<SD, TI>(S, tc, i, a), jac<SD, TI>(S, tc, i, b), acc);
                        Lg[tri(a, b)] = acc;
                    }
                }
                ldl_factor_s<N>(Lg, rg);
                ldl_solve_s<N>(Lg, rg, w);
                if constexpr (P.quirk) {
                    // Tall first EqualityConstraint (more rows than states, e.g. an 8-row dual-quaternion
                    // error on a 6-DoF arm): pass 1 gives w = G^-1 J'd with G = J'J + lam I; pass 2 projects
                    // through the stack [J] (:382-396): N = I - G^-1 J'J = lam G^-1, so it adds lam G^-1 w.
                    double w2[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) w2[j] = w[j];
                    ldl_solve_s<N>(Lg, rg, w2);
#pragma unroll
                    for (int j = 0; j < N; ++j) w[j] = fma(c.lam, w2[j], w[j]);
                }
            }
            if constexpr (P.first) {
#pragma unroll
                for (int j = 0; j < N; ++j) c.v[j] += w[j];
            }
            if constexpr (P.quirk) {
                // (w already holds the sum of both passes, see above; it was added as the first task)
                static_assert(P.first, "the doubly processed EqualityConstraint is the first contribution");
                if constexpr (P.push_times > 0) push_s<SD, ACT, TI, ROLE>(S, tc, c, 0xffffffffu);
            } else {
                if constexpr (!P.first) {
                    project_s<SD, ACT, TI, ROLE>(S, tc, c, w);
#pragma unroll
                    for (int j = 0; j < N; ++j) c.v[j] += w[j];
                }
                if constexpr (P.push_times > 0)
                    push_s<SD, ACT, TI, ROLE>(S, tc, c, (P.conv && SD.multidim) ? srow : 0xffffffffu);
            }
        }
    }
}

// in-tangent-cone test of the inactive SetConstraint TI (pseudo_inverse.py:162-185, :222-252)
template <const ShapeDesc& SD, unsigned ACT, int TI, int ROLE = 0>
__device__ __forceinline__ void cone_s(CLIK_MODE_IN_RAW, ModeCtx<SD, ACT, ROLE>& c)
{
    constexpr int N = SD.n;
    constexpr int M = SD.m[TI];
    constexpr TaskPlan P = Plan<SD, ACT>::mode.t[TI];
    if constexpr (P.cone) {
        CLIK_TASK_IMAGE(S, S_in);
        decltype(auto) t = task_consts<SD, TI>(S, tc);
        double e[M], Jt[M], de[M];
        task_values<SD, TI>(S, tk, tc, z, ys, lane, e, Jt);
        if constexpr (shape_unit(SD, TI)) {
            static_for<0, M>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int col = SD.ucol[TI][i] - 1;
                de[i] = Jt[i] + c.v[col];
            });
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double s = Jt[i];
#pragma unroll
                for (int j = 0; j < N; ++j) s = fma(jac<SD, TI>(S, tc, i, j), c.v[j], s);
                de[i] = s;
            }
        }
        bool in_tc;
        if constexpr (M == 1) {
            const double ev = e[0];
            if (t.set_min[0] - ev < 1e-12) in_tc = (ev - t.set_max[0] < 1e-12) ? true : (de[0] < 0.0);
            else in_tc = de[0] > 0.0;
        } else {
            // (flags combined bitwise and half signs by bit operations: the short-circuit / comparison-difference
            // forms compile to a divergent branch per row; everything behind "inside" is skipped when every
            // instance of the wave is inside its limits)
            double le[M], ue[M];
            bool inside = true;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                le[i] = e[i] - t.set_min[i];
                ue[i] = e[i] - t.set_max[i];
                inside = inside & (le[i] >= 1e-12) & (ue[i] <= 1e-12);
            }
            in_tc = true;
            if (__ballot(!inside) != 0ull) {
                bool corner = true;
                double od = 0.0, nde = 0.0, nout = 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const double hl = half_sign(le[i]), hu = half_sign(ue[i]);
                    corner = corner & (hl == hu);
                    const double out = hl + hu;
                    od = fma(out, de[i], od);
                    nde = fma(de[i], de[i], nde);
                    nout = fma(out, out, nout);
                }
                bool going_in = od < 0.0;
                if (__ballot(corner & !inside) != 0ull) {
                    const double dists = (sqrt(nde) + 1e-10) * sqrt(nout);
                    const bool steep = (od < 0.0) & (fabs(od) / dists < 0.70710678118654757);
                    going_in = corner ? steep : going_in;
                }
                in_tc = inside | going_in;
            }
        }
        c.ok = c.ok & in_tc;
    }
}

// Task steps in order.  The image fields of step TI+1 are requested from LDS before step TI
// computes (cur = register copy for step TI), so their latency hides behind that step's
// arithmetic - a lone wave has nothing else to hide it behind.  (When S already points to a
// register copy, as in the one-wave kernel, these copies are free.)
template <const ShapeDesc& SD, unsigned ACT, int TI, int ROLE = 0>
__device__ __forceinline__ void steps_s(CLIK_MODE_IN, ModeCtx<SD, ACT, ROLE>& c, const Img<SD>* cur)
{
    if constexpr (TI < SD.n_tasks) {
        const Img<SD> nxt = *S;
        __builtin_amdgcn_sched_barrier(0);
        step_s<SD, ACT, TI, ROLE>(cur, tk, tc, z, ys, lane, c);
        steps_s<SD, ACT, TI + 1, ROLE>(CLIK_MODE_ARGS, c, &nxt);
    }
}

template <const ShapeDesc& SD, unsigned ACT, int TI, int ROLE = 0>
__device__ __forceinline__ void cones_s(CLIK_MODE_IN, ModeCtx<SD, ACT, ROLE>& c)
{
    if constexpr (TI < SD.n_tasks) {
        cone_s<SD, ACT, TI, ROLE>(CLIK_MODE_ARGS, c);
        cones_s<SD, ACT, TI + 1, ROLE>(CLIK_MODE_ARGS, c);
    }
}

// candidate velocity of the mode with activation mask ACT; returns whether all
// inactive sets are in their tangent cone
template <const ShapeDesc& SD, unsigned ACT, int ROLE = 0>
__device__ __forceinline__ bool pinv_mode_static(const Img<SD>* __restrict__ S, const TickArgs& tk,
                                                 const TaskCache<SD>& tc, const double (&z)[SD.n],
                                                 const double* ys, int lane, double (&v)[SD.n],
                                                 const double* xch = nullptr)
{
    constexpr int N = SD.n;
    ModeCtx<SD, ACT, ROLE> c;
    c.xch = xch + lane;
    c.lam = SD.standard ? 0.0 : S->lam;
#pragma unroll
    for (int j = 0; j < N; ++j) c.v[j] = 0.0;
    c.st.sbits = 0u;
    c.ok = true;
    {
        const Img<SD> first = *S;
        __builtin_amdgcn_sched_barrier(0);
        steps_s<SD, ACT, 0, ROLE>(CLIK_MODE_ARGS, c, &first);
    }
    cones_s<SD, ACT, 0, ROLE>(CLIK_MODE_ARGS, c);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = c.v[j];
    return c.ok;
}

// Helper wave of the role split: builds the stack of mode ACT exactly as the main evaluation
// would (same activation bits, same push order) up to the unique Gram consumer, factors
// lam I + Ja'Ja and publishes L (packed) and 1/d through LDS ([slot][lane]).
template <const ShapeDesc& SD, unsigned ACT>
__device__ __forceinline__ void helper_mode_static(const Img<SD>* __restrict__ S, const TickArgs& tk,
                                                   const TaskCache<SD>& tc, const double (&z)[SD.n],
                                                   const double* ys, int lane, double* xch)
{
    constexpr int N = SD.n;
    constexpr int NT = N * (N + 1) / 2;
    constexpr ModePlan MP = Plan<SD, ACT>::mode;
    static_assert(MP.helper_ok, "mode has no unique Gram consumer");
    ModeCtx<SD, ACT, ROLE_HELPER> c;
    c.lam = SD.standard ? 0.0 : S->lam;
    c.st.sbits = 0u;
    static_for<0, MP.gram_consumer>([&](auto tn) __attribute__((always_inline)) {
        constexpr int TI = decltype(tn)::value;
        constexpr TaskPlan P = MP.t[TI];
        if constexpr (!P.skip && P.push_times > 0) {
            constexpr int M = SD.m[TI];
            uint32_t srow = 0xffffffffu;
            if constexpr (P.set_rows) {
                decltype(auto) t = task_consts<SD, TI>(S, tc);
                double e[M], Jt[M];
                task_values<SD, TI>(S, tk, tc, z, ys, lane, e, Jt);
                srow = 0u;
#pragma unroll
                for (int i = 0; i < M; ++i)
                    srow |= (uint32_t)((e[i] - t.set_max[i] > 0.0) | (e[i] - t.set_min[i] < 0.0)) << i;     // (bitwise: no branch per row)
            }
            // the same argument step_s passes for this kind of task
            if constexpr (!P.contributes) push_s<SD, ACT, TI, ROLE_HELPER>(S, tc, c, srow);
            else if constexpr (P.quirk) push_s<SD, ACT, TI, ROLE_HELPER>(S, tc, c, 0xffffffffu);
            else push_s<SD, ACT, TI, ROLE_HELPER>(S, tc, c, (P.conv && SD.multidim) ? srow : 0xffffffffu);
        }
    });
    double L[NT], rd[N];
#pragma unroll
    for (int a = 0; a < NT; ++a) L[a] = c.st.G[a];
    ldl_factor_s<N>(L, rd);
#pragma unroll
    for (int a = 0; a < NT; ++a) xch[a * WAVE + lane] = L[a];
#pragma unroll
    for (int a = 0; a < N; ++a) xch[(NT + a) * WAVE + lane] = rd[a];
}

// ---- the config-3 family: [joint-limit set on every state; task with m <= n state-dependent rows; joint-space task] -----
// Both modes of such a skill need only shifted copies of ONE Gram matrix Gm = J J' (J: the m x n Jacobian of the
// second constraint) - see clik_pinv_team.hpp for the algebra and the four-lanes-per-instance kernel built on it.
constexpr bool shape_team_ok(const ShapeDesc& sd)
{
    if (sd.qp || sd.n_tasks != 3 || sd.n_x != 0 || sd.standard || sd.conv_last || !sd.multidim) return false;
    if (sd.cls[0] != CLIK_CLS_SET || sd.cls[1] != CLIK_CLS_EQ || sd.cls[2] != CLIK_CLS_EQ) return false;
    if ((sd.ext[0] | sd.ext[1] | sd.ext[2]) & ~1) return false;      // (gains / bounds given as expressions)
    // the set covers every state variable exactly once (then  lam I + Jset'Jset = (1+lam) I)
    if (!shape_unit(sd, 0) || sd.m[0] != sd.n || sd.n < 2) return false;
    for (int c = 0; c < sd.n; ++c)
        if (shape_unit_row(sd, 0, c) < 0) return false;
    if (sd.const_j[1] || sd.m[1] > sd.n || sd.m[1] < 1) return false;
    if (!shape_unit(sd, 2)) return false;
    return true;
}


// The same algebra in ONE lane (the lane-per-instance kernels of this family: large batches, rollouts): the three
// shifted factorisations one after the other, mode 1 only when some lane of the wave needs it.
//   A0 = Gm + lam I:      y = A0^-1 d1, y2 = A0^-1 y;  mode 0's doubly processed first equality sums to J'(y + lam y2),
//                         mode 1's task behind the active set is N_set J'y
//   A1 = 2 Gm + lam I:    mode 0's lower task through [J; J]:   w2 - 2 J' A1^-1 J w2            (push-through)
//   A2 = Gm + (1+lam) I:  mode 1's lower task through [I; J] with activations s:
//                         (x - J' A2^-1 J x) / (1+lam),  x = ((1+lam) - s) o w2                  (Woodbury)
// 1460 instructions when every lane of the wave accepts mode 0 and 1770 with mode 1 (round 5: 72 / 30 fewer), against 1600 / 2800 of the
// plan-driven evaluation (pinv_mode_static) this replaces for the family; same values to rounding (PINV_RTOL).
// -DCLIK_NO_SOLO keeps the plan-driven evaluation (regression switch).
template <const ShapeDesc& SD>
__device__ __forceinline__ void solo_tick(const Img<SD>* __restrict__ S, const TickArgs& tk, const double (&z)[SD.n],
                                          const double* ys, const int lane, const bool valid, double (&vout)[SD.n],
                                          int& acc_mode)
{
    constexpr int N = SD.n, M = SD.m[1], M0 = SD.m[0], M2 = SD.m[2];
    constexpr int NT = M * (M + 1) / 2;
    TaskCache<SD> tc;
    {
        Kin<N> K;
        if constexpr (SD.uses_fk != 0) {
            forward_kinematics_s<SD>(S, z, K);
            if constexpr (SD.quat_src != 0) orientation_feature_s<SD>(S, ys, lane, K);
        }
        cache_task<SD, 0>(S, tk, K, z, ys, lane, tc);
    }
    const double lam = S->lam;
    const double one_lam = 1.0 + lam;
    // desired task velocities  d = -K e - de/dt   (pseudo_inverse.py:318-321, :383-386)
    double des1[M], w2[N];
    {
        double e[M], Jt[M], ke[M];
        task_values<SD, 1>(S, tk, tc, z, ys, lane, e, Jt);
        gain_apply_s<M, SD.gain_matrix[1] != 0>(S->tasks[1], e, ke);
#pragma unroll
        for (int i = 0; i < M; ++i) des1[i] = SD.feedforward != 0 ? -ke[i] - Jt[i] : -ke[i];
    }
    {
        // w2 = pinv(J2) d2 of the joint-space task: one entry per row (host-side pinv of the unit rows)
        double e[M2], Jt[M2], ke[M2];
        task_values<SD, 2>(S, tk, tc, z, ys, lane, e, Jt);
        gain_apply_s<M2, SD.gain_matrix[2] != 0>(S->tasks[2], e, ke);
#pragma unroll
        for (int j = 0; j < N; ++j) w2[j] = 0.0;
        const double* Pm = S->cpinv[2];
        static_for<0, M2>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int col = SD.ucol[2][i] - 1;
            const double d = SD.feedforward != 0 ? -ke[i] - Jt[i] : -ke[i];
            w2[col] = Pm[col * CLIK_MAX_M + i] * d;
        });
    }
    // Gm = J J'
    double Gm[NT];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int k = 0; k <= i; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) acc = fma(jac<SD, 1>(S, tc, i, j), jac<SD, 1>(S, tc, k, j), acc);
            Gm[tri(i, k)] = acc;
        }
    // the factor of  Gm + beta I  (in A / rd).  Every shifted matrix of the family is this form: A1 = 2 Gm + lam I is
    // 2 (Gm + lam/2 I) - its inverse is HALF the inverse of the bracket, exactly (a power of two) - so Gm is never scaled,
    // only its diagonal shifted (round 6; the scaled copy was 21 multiplications per factorisation).
    auto shifted = [&](const double beta, double (&A)[NT], double (&rd)[M]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int k = 0; k <= i; ++k) A[tri(i, k)] = (i == k) ? Gm[tri(i, k)] + beta : Gm[tri(i, k)];
        ldl_factor_s<M>(A, rd);
    };
    auto jt_times = [&](const double (&t)[M], double (&g)[N]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double sacc = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) sacc = fma(jac<SD, 1>(S, tc, i, j), t[i], sacc);
            g[j] = sacc;
        }
    };
    auto j_times = [&](const double (&x)[N], double (&t)[M]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) sacc = fma(jac<SD, 1>(S, tc, i, j), x[j], sacc);
            t[i] = sacc;
        }
    };
    // A0: the first equality, processed twice (mode 0) / once behind the set (mode 1);  A1: mode 0's lower task.
    // Mode 0's velocity is ONE product with J':  J'(y + lam y2) + (w2 - 2 J' t1) = J'(y + lam y2 - 2 t1) + w2
    // (round 5: three products before, 72 instructions more); J'y alone is mode 1's and is formed there.
    double v0[N], ysol[M];
    {
        double u[M];
        {
            double A[NT], rd[M], y2[M];
            shifted(lam, A, rd);
#pragma unroll
            for (int i = 0; i < M; ++i) ysol[i] = des1[i];
            ldl_solve_s<M>(A, rd, ysol);
#pragma unroll
            for (int i = 0; i < M; ++i) y2[i] = ysol[i];
            ldl_solve_s<M>(A, rd, y2);
#pragma unroll
            for (int i = 0; i < M; ++i) u[i] = fma(lam, y2[i], ysol[i]);
        }
        {
            double A[NT], rd[M], t1[M];
            shifted(0.5 * lam, A, rd);          // (Gm + lam/2 I: the solution is twice A1's - the "2 t1" of the formula as it stands)
            j_times(w2, t1);
            ldl_solve_s<M>(A, rd, t1);
#pragma unroll
            for (int i = 0; i < M; ++i) u[i] -= t1[i];
        }
        double gu[N];
        jt_times(u, gu);
#pragma unroll
        for (int j = 0; j < N; ++j) v0[j] = gu[j] + w2[j];
    }
    // the set: tangent-cone test of the mode-0 candidate (:222-252), its activation by state column
    double e0[M0], Jt0[M0];
    task_values<SD, 0>(S, tk, tc, z, ys, lane, e0, Jt0);
    bool in_tc;
    {
        const clik_task& t = S->tasks[0];
        double le[M0], ue[M0];
#pragma unroll
        for (int i = 0; i < M0; ++i) {
            le[i] = e0[i] - t.set_min[i];
            ue[i] = e0[i] - t.set_max[i];
        }
        // (all(le >= 1e-12) & all(ue <= 1e-12) through the smallest le and the largest ue, "no row's half signs differ"
        // as a sum of |hl - hu| that is exactly zero: see team_tick)
        double le_min = le[0], ue_max = ue[0];
#pragma unroll
        for (int i = 1; i < M0; ++i) {
            le_min = fmin(le_min, le[i]);
            ue_max = fmax(ue_max, ue[i]);
        }
        const bool outside = (le_min < 1e-12) | (ue_max > 1e-12);
        in_tc = true;
        if (__builtin_amdgcn_ballot_w64(outside) != 0ull) {
            double od = 0.0, nde = 0.0, nout = 0.0, ndiff = 0.0;
            static_for<0, M0>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int col = SD.ucol[0][i] - 1;
                const double de = Jt0[i] + v0[col];
                const double hl = half_sign(le[i]), hu = half_sign(ue[i]);     // (sign(le) + sign(ue)) / 2 = hl + hu
                ndiff += fabs(hl - hu);
                const double out = hl + hu;
                od = fma(out, de, od);
                nde = fma(de, de, nde);
                nout = fma(out, out, nout);
            });
            const bool corner = ndiff == 0.0;
            bool going_in = od < 0.0;
            if (__builtin_amdgcn_ballot_w64(corner & outside) != 0ull) {
                const double dists = (sqrt(nde) + 1e-10) * sqrt(nout);
                const bool steep = (od < 0.0) & (fabs(od) / dists < 0.70710678118654757);
                going_in = corner ? steep : going_in;
            }
            in_tc = !outside | going_in;
        }
    }
    acc_mode = valid ? (in_tc ? 0 : 1) : -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = v0[j];
    // mode 1 (set active; no inactive set left: always accepted) only when some lane needs it
    if (__ballot(valid & !in_tc) != 0ull) {
        double sact[N], p0[N], x[N];
        static_for<0, M0>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int col = SD.ucol[0][i] - 1;
            sact[col] = ((e0[i] - S->tasks[0].set_max[i] > 0.0) | (e0[i] - S->tasks[0].set_min[i] < 0.0)) ? 1.0 : 0.0;
            p0[col] = S->cpinv[0][col * CLIK_MAX_M + i];
        });
#pragma unroll
        for (int j = 0; j < N; ++j) x[j] = w2[j] * (one_lam - sact[j]);
        double A[NT], rd[M], t2[M], g2[N], g3[N];
        shifted(one_lam, A, rd);
        j_times(x, t2);
        ldl_solve_s<M>(A, rd, t2);
        jt_times(t2, g2);
        jt_times(ysol, g3);
        const double kap = 1.0 / one_lam;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double v1 = fma(g3[j], fma(-p0[j], sact[j], 1.0), kap * (x[j] - g2[j]));
            vout[j] = in_tc ? vout[j] : v1;
        }
    }
    if (!valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) vout[j] = 0.0;
    }
}

}  // namespace clik
