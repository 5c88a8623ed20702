/*
 * CPU ORACLE (C) - TEST INFRASTRUCTURE AND CPU BASELINE ONLY.
 * Never linked into, loaded by, or called from the product (casclik_amd/).
 *
 * Plain-C fp64 restatement of the reference's per-tick hot path, driven by the
 * same flat skill descriptor (include/clik.h) the HIP kernels consume:
 *
 *   PseudoInverseController   /root/reference/casclik/controllers/pseudo_inverse.py
 *     pinv :92-105, activation map :107-130, tangent cones :132-257,
 *     per-mode expressions :259-451 (incl. first-equality double processing
 *     :317-326 + :382-396), mode scan :512-556
 *   ReactiveQPController      /root/reference/casclik/controllers/reactive_qp.py
 *     cost :175-189, constraint rows :191-246
 *
 * PARITY UNPINNED BY THE LETTER (see oracle/clik_oracle.py header): CasADi is not
 * available, so no fixture comes from CasADi arithmetic at 1e-9.  What this file
 * is held to: the numpy / dual-number oracle, the fixtures the reference's own
 * Python produced over a stand-in casadi (tests/golden/ref_pins.npz,
 * tests/test_refpins.py), and - through those - the closed-loop figures the
 * reference's notebooks store of their real CasADi + qpOASES runs, which the
 * numpy oracle, the reference over the stand-in and the HIP path all retrace
 * within a pixel (tests/golden/make_figure_pins.py, tests/test_figure_pins.py).
 *
 * It follows the reference LITERALLY: full pseudo-inverse matrices, full
 * null-space projectors N = I - pinv(vstack Ja) * vstack rJa, products in the
 * reference's order (N * pinv(J)) * des, and `cs.solve` restated as Gaussian
 * elimination with partial pivoting.  Jacobians of rotation features are
 * formed from explicit dR/dq_j matrices (not from the closed forms the HIP
 * kernels use), so the two implementations share the descriptor but not the
 * arithmetic.
 *
 * Build:  gcc -O2 -fopenmp -shared -fPIC -I../include clik_oracle_c.c -o _build/libclik_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "clik.h"

#define NMAX CLIK_MAX_DOF
#define MMAX CLIK_MAX_M
#define RMAX (CLIK_MAX_TASKS * CLIK_MAX_M)   /* stacked rows */

/* ---------------------------------------------------------------- linear algebra */
/* solve A X = B in place (A n x n row-major lda, B n x k row-major ldb);
 * Gaussian elimination with partial pivoting.  Returns 0, or -1 if singular. */
static int ge_solve(int n, int k, double* A, int lda, double* B, int ldb)
{
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(A[c * lda + c]);
        for (int r = c + 1; r < n; ++r) {
            double v = fabs(A[r * lda + c]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) return -1;
        if (piv != c) {
            for (int j = 0; j < n; ++j) { double t = A[c * lda + j]; A[c * lda + j] = A[piv * lda + j]; A[piv * lda + j] = t; }
            for (int j = 0; j < k; ++j) { double t = B[c * ldb + j]; B[c * ldb + j] = B[piv * ldb + j]; B[piv * ldb + j] = t; }
        }
        double inv = 1.0 / A[c * lda + c];
        for (int r = c + 1; r < n; ++r) {
            double f = A[r * lda + c] * inv;
            if (f == 0.0) continue;
            for (int j = c; j < n; ++j) A[r * lda + j] -= f * A[c * lda + j];
            for (int j = 0; j < k; ++j) B[r * ldb + j] -= f * B[c * ldb + j];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double inv = 1.0 / A[c * lda + c];
        for (int j = 0; j < k; ++j) {
            double s = B[c * ldb + j];
            for (int r = c + 1; r < n; ++r) s -= A[c * lda + r] * B[r * ldb + j];
            B[c * ldb + j] = s * inv;
        }
    }
    return 0;
}

/* P (n x r, row-major ld RMAX) = damped / standard pseudo-inverse of J (r x n,
 * row-major ld NMAX) - pseudo_inverse.py:92-105 */
static int dpinv(const clik_pinv_opts* o, int r, int n, const double* J, double* P)
{
    double lam = (o->pinv_method == CLIK_PINV_STANDARD) ? 0.0 : o->damping_factor;
    int wide = (o->pinv_method == CLIK_PINV_STANDARD) ? (r < n) : (n >= r);
    if (wide) {
        /* solve(J J^T + lam I, J)^T */
        static _Thread_local double A[RMAX * RMAX];
        static _Thread_local double X[RMAX * NMAX];
        for (int i = 0; i < r; ++i)
            for (int k = 0; k < r; ++k) {
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += J[i * NMAX + j] * J[k * NMAX + j];
                A[i * r + k] = s + (i == k ? lam : 0.0);
            }
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < n; ++j) X[i * n + j] = J[i * NMAX + j];
        if (ge_solve(r, n, A, r, X, n)) return -1;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < r; ++i) P[j * RMAX + i] = X[i * n + j];
    } else {
        /* solve(J^T J + lam I, J^T) */
        double A[NMAX * NMAX];
        static _Thread_local double X[NMAX * RMAX];
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double s = 0.0;
                for (int i = 0; i < r; ++i) s += J[i * NMAX + a] * J[i * NMAX + b];
                A[a * n + b] = s + (a == b ? lam : 0.0);
            }
        for (int a = 0; a < n; ++a)
            for (int i = 0; i < r; ++i) X[a * r + i] = J[i * NMAX + a];
        if (ge_solve(n, r, A, n, X, r)) return -1;
        for (int a = 0; a < n; ++a)
            for (int i = 0; i < r; ++i) P[a * RMAX + i] = X[a * r + i];
    }
    return 0;
}

/* ---------------------------------------------------------------- kinematics */
typedef struct {
    double R[9], p[3];            /* tool frame                                  */
    double dR[NMAX][9];           /* d R / d z_j (explicit matrices)             */
    double dp[NMAX][3];           /* d p / d z_j                                 */
    double o[3], dO[NMAX][3];     /* orientation error and its derivative        */
} kin_t;

static void mat3mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static void quat_rot(const double* q, double* R)
{
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

static void kinematics(const clik_skill_desc* d, const double* z, const double* y, kin_t* k)
{
    int n = d->n_q + d->n_x;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    double zax[NMAX][3], org[NMAX][3];
    int revolute[NMAX];
    int used[NMAX];
    memset(used, 0, sizeof(used));
    for (int j = 0; j < d->n_joints; ++j) {
        const clik_joint* jt = &d->joints[j];
        double T[9];
        for (int i = 0; i < 3; ++i)
            p[i] += R[i * 3] * jt->p[0] + R[i * 3 + 1] * jt->p[1] + R[i * 3 + 2] * jt->p[2];
        mat3mul(R, jt->R, T);
        memcpy(R, T, sizeof(T));
        if (jt->type == CLIK_JOINT_FIXED) continue;
        int qi = jt->q_index;
        double ax[3];
        for (int i = 0; i < 3; ++i)
            ax[i] = R[i * 3] * jt->axis[0] + R[i * 3 + 1] * jt->axis[1] + R[i * 3 + 2] * jt->axis[2];
        memcpy(zax[qi], ax, sizeof(ax));
        memcpy(org[qi], p, sizeof(p));
        used[qi] = 1;
        revolute[qi] = jt->type == CLIK_JOINT_REVOLUTE;
        if (jt->type == CLIK_JOINT_REVOLUTE) {
            double a = z[qi], c = cos(a), s = sin(a), C = 1.0 - c;
            double x = jt->axis[0], yy = jt->axis[1], zz = jt->axis[2];
            double M[9] = {c + x * x * C, x * yy * C - zz * s, x * zz * C + yy * s,
                           yy * x * C + zz * s, c + yy * yy * C, yy * zz * C - x * s,
                           zz * x * C - yy * s, zz * yy * C + x * s, c + zz * zz * C};
            mat3mul(R, M, T);
            memcpy(R, T, sizeof(T));
        } else {
            for (int i = 0; i < 3; ++i) p[i] += ax[i] * z[qi];
        }
    }
    memcpy(k->R, R, sizeof(R));
    memcpy(k->p, p, sizeof(p));
    for (int j = 0; j < n; ++j) {
        memset(k->dR[j], 0, sizeof(k->dR[j]));
        memset(k->dp[j], 0, sizeof(k->dp[j]));
        if (!used[j]) continue;
        if (revolute[j]) {
            /* dR = [z]x R ; dp = z x (p - o_j) */
            for (int c = 0; c < 3; ++c) {
                double col[3] = {R[c], R[3 + c], R[6 + c]}, o[3];
                cross3(zax[j], col, o);
                k->dR[j][c] = o[0]; k->dR[j][3 + c] = o[1]; k->dR[j][6 + c] = o[2];
            }
            double r[3] = {p[0] - org[j][0], p[1] - org[j][1], p[2] - org[j][2]};
            cross3(zax[j], r, k->dp[j]);
        } else {
            memcpy(k->dp[j], zax[j], sizeof(zax[j]));
        }
    }
    memset(k->o, 0, sizeof(k->o));
    memset(k->dO, 0, sizeof(k->dO));
    if (d->quat_src != 0) {
        double qd[4], Rd[9];
        for (int i = 0; i < 4; ++i) qd[i] = (d->quat_src == 2) ? y[d->quat_yi[i]] : d->quat[i];
        quat_rot(qd, Rd);
        for (int c = 0; c < 3; ++c) {
            double rc[3] = {R[c], R[3 + c], R[6 + c]}, dc[3] = {Rd[c], Rd[3 + c], Rd[6 + c]}, x[3];
            cross3(rc, dc, x);
            for (int i = 0; i < 3; ++i) k->o[i] += 0.5 * x[i];
            for (int j = 0; j < n; ++j) {
                double drc[3] = {k->dR[j][c], k->dR[j][3 + c], k->dR[j][6 + c]};
                cross3(drc, dc, x);
                for (int i = 0; i < 3; ++i) k->dO[j][i] += 0.5 * x[i];
            }
        }
    }
}

/* value and gradient of one affine row */
static double row_eval(const clik_skill_desc* d, const clik_row* r, const kin_t* k,
                       const double* z, const double* y, const double* tterms,
                       double* grad, double* dt)
{
    int n = d->n_q + d->n_x;
    double v = r->c;
    *dt = 0.0;
    for (int j = 0; j < n; ++j) grad[j] = r->a[j];
    for (int j = 0; j < n; ++j) v += r->a[j] * z[j];
    for (int i = 0; i < 3; ++i) {
        v += r->b[i] * k->p[i] + r->h[i] * k->o[i];
        for (int j = 0; j < n; ++j) grad[j] += r->b[i] * k->dp[j][i] + r->h[i] * k->dO[j][i];
    }
    for (int i = 0; i < 9; ++i) {
        v += r->g[i] * k->R[i];
        for (int j = 0; j < n; ++j) grad[j] += r->g[i] * k->dR[j][i];
    }
    for (int i = 0; i < r->n_y; ++i) v += r->yc[i] * y[r->yi[i]];
    if (r->t_slot >= 0) {
        v += tterms[r->t_slot];
        *dt = tterms[d->n_tslots + r->t_slot];
    }
    return v;
}

/* e (m), J (m x n, ld NMAX), Jt (m) of constraint `ti` */
static void task_eval(const clik_skill_desc* d, int ti, const kin_t* k, const double* z,
                      const double* y, const double* tterms, double* e, double* J, double* Jt)
{
    const clik_task* t = &d->tasks[ti];
    int n = d->n_q + d->n_x;
    for (int i = 0; i < t->m; ++i) {
        if (t->out_kind[i] == CLIK_OUT_AFFINE) {
            e[i] = row_eval(d, &d->rows[t->out_row0[i]], k, z, y, tterms, &J[i * NMAX], &Jt[i]);
        } else {
            double ss = 0.0, acc[NMAX], tacc = 0.0;
            memset(acc, 0, sizeof(acc));
            for (int q = 0; q < t->out_nrows[i]; ++q) {
                double g[NMAX], dt;
                double v = row_eval(d, &d->rows[t->out_row0[i] + q], k, z, y, tterms, g, &dt);
                ss += v * v;
                tacc += v * dt;
                for (int j = 0; j < n; ++j) acc[j] += v * g[j];
            }
            double nr = sqrt(ss);
            e[i] = nr;
            Jt[i] = tacc / nr;
            for (int j = 0; j < n; ++j) J[i * NMAX + j] = acc[j] / nr;
        }
    }
}

int orc_task_eval(const clik_skill_desc* d, int ti, const double* tterms, const double* z,
                  const double* y, double* e, double* J /* m x n dense */, double* Jt)
{
    kin_t k;
    double Jl[MMAX * NMAX];
    int n = d->n_q + d->n_x;
    kinematics(d, z, y, &k);
    task_eval(d, ti, &k, z, y, tterms, e, Jl, Jt);
    for (int i = 0; i < d->tasks[ti].m; ++i)
        for (int j = 0; j < n; ++j) J[i * n + j] = Jl[i * NMAX + j];
    return 0;
}

/* ---------------------------------------------------------------- mode table */
static int popcount32(unsigned v) { int c = 0; while (v) { c += v & 1u; v >>= 1; } return c; }

/* pseudo_inverse.py:107-130: bit k of the pattern <-> k-th SetConstraint,
 * patterns stably sorted by popcount.  act[mode] = bit mask. */
static void activation_map(int n_sets, unsigned* act)
{
    int n_modes = 1 << n_sets, k = 0;
    for (int pc = 0; pc <= n_sets; ++pc)
        for (int v = 0; v < n_modes; ++v)
            if (popcount32((unsigned)v) == pc) act[k++] = (unsigned)v;
}

/* ---------------------------------------------------------------- tangent cones */
/* diagnostic (orc_pinv_solve_batch_m): the smallest distance of any tangent-cone decision of the running mode scan
 * from flipping - the quantities the functions below threshold (see clik_oracle.py::tangent_cone_margin) */
static _Thread_local double g_tc_margin;
static void margin_note(double v) { if (v < g_tc_margin) g_tc_margin = v; }

static int in_tc_1d(double e, double lo, double hi, double de)
{
    margin_note(fabs(lo - e - 1e-12));
    margin_note(fabs(e - hi - 1e-12));
    if (!((lo - e < 1e-12) && (e - hi < 1e-12))) margin_note(fabs(de));
    if (lo - e < 1e-12) {
        if (e - hi < 1e-12) return 1;
        return de < 0.0;
    }
    return de > 0.0;
}

static double sgn(double v) { return (v > 0.0) - (v < 0.0); }

static int in_tc_multidim(int m, const double* e, const double* lo, const double* hi, const double* de)
{
    int inside = 1, corner = 1;
    double od = 0.0, nde = 0.0, nout = 0.0;
    for (int i = 0; i < m; ++i) {
        double le = e[i] - lo[i], ue = e[i] - hi[i];
        if (!(le >= 1e-12) || !(ue <= 1e-12)) inside = 0;
        double out = (sgn(le) + sgn(ue)) / 2.0;
        if (sgn(le) != sgn(ue)) corner = 0;
        od += out * de[i];
        nde += de[i] * de[i];
        nout += out * out;
    }
    for (int i = 0; i < m; ++i) {
        margin_note(fabs(e[i] - lo[i] - 1e-12));
        margin_note(fabs(e[i] - hi[i] - 1e-12));
        if (!inside) { margin_note(fabs(e[i] - lo[i])); margin_note(fabs(e[i] - hi[i])); }
    }
    if (inside) return 1;
    {
        double scale = (sqrt(nde) + 1e-10) * (nout > 0.0 ? sqrt(nout) : 1e-300);
        margin_note(fabs(od) / scale);
        if (corner && od < 0.0) margin_note(fabs(fabs(od) / scale - cos(M_PI / 4)));
    }
    if (corner) {
        if (od < 0.0) {
            double dists = (sqrt(nde) + 1e-10) * sqrt(nout);
            return fabs(-od) / dists < cos(M_PI / 4);
        }
        return 0;
    }
    return od < 0.0;
}

/* ---------------------------------------------------------------- pinv controller */
static void gain_apply(const clik_task* t, const double* v, double* out)
{
    int m = t->m;
    if (!t->gain_is_matrix) {
        for (int i = 0; i < m; ++i) out[i] = t->gain[0] * v[i];
    } else {
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += t->gain[i * m + k] * v[k];
            out[i] = s;
        }
    }
}

typedef struct {
    double e[CLIK_MAX_TASKS][MMAX];
    double Jt[CLIK_MAX_TASKS][MMAX];
    double J[CLIK_MAX_TASKS][MMAX * NMAX];
} tasks_t;

/* v += (N * pinv(Ji)) * des with N = I - pinv(vstack Ja) * vstack rJa
 * (pseudo_inverse.py:387-394); nstack == 0 -> N = I */
static int add_projected(const clik_pinv_opts* o, int n, int nstack, const double* Ja,
                         const double* rJa, int m, const double* Ji, const double* des, double* v)
{
    static _Thread_local double P[NMAX * RMAX], Pi[NMAX * RMAX];
    double N[NMAX * NMAX], NJ[NMAX * MMAX];
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) N[a * n + b] = (a == b) ? 1.0 : 0.0;
    if (nstack > 0) {
        if (dpinv(o, nstack, n, Ja, P)) return -1;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double s = 0.0;
                for (int r = 0; r < nstack; ++r) s += P[a * RMAX + r] * rJa[r * NMAX + b];
                N[a * n + b] -= s;
            }
    }
    if (dpinv(o, m, n, Ji, Pi)) return -1;
    for (int a = 0; a < n; ++a)
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int b = 0; b < n; ++b) s += N[a * n + b] * Pi[b * RMAX + i];
            NJ[a * MMAX + i] = s;
        }
    for (int a = 0; a < n; ++a) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += NJ[a * MMAX + i] * des[i];
        v[a] += s;
    }
    return 0;
}

static int pinv_one(const clik_skill_desc* d, const clik_pinv_opts* o, const unsigned* act,
                    int n_sets, const double* tterms, const double* z, const double* y,
                    double* vout, int32_t* mode_out)
{
    int n = d->n_q + d->n_x;
    int n_modes = 1 << n_sets;
    kin_t kin;
    static _Thread_local tasks_t T;
    static _Thread_local double Ja[RMAX * NMAX], rJa[RMAX * NMAX], Pi[NMAX * RMAX];
    kinematics(d, z, y, &kin);
    for (int ti = 0; ti < d->n_tasks; ++ti)
        task_eval(d, ti, &kin, z, y, tterms, T.e[ti], T.J[ti], T.Jt[ti]);
    for (int mode = 0; mode < n_modes; ++mode) {
        double v[NMAX];
        int nstack = 0, set_idx = 0, ntc = 0, tc[CLIK_MAX_TASKS];
        memset(v, 0, sizeof(v));
        for (int ti = 0; ti < d->n_tasks; ++ti) {
            const clik_task* t = &d->tasks[ti];
            int m = t->m;
            const double* e = T.e[ti];
            const double* Jt = T.Jt[ti];
            const double* Ji = T.J[ti];
            int is_first = (nstack == 0);
            int is_last = (ti == d->n_tasks - 1);
            int is_set = t->cls == CLIK_CLS_SET, is_eq = t->cls == CLIK_CLS_EQ;
            int is_veleq = t->cls == CLIK_CLS_VELEQ;
            double S[MMAX], des[MMAX], tmp[MMAX];
            for (int i = 0; i < m; ++i) S[i] = 1.0;
            if (o->multidim_sets && is_set)
                for (int i = 0; i < m; ++i)
                    S[i] = ((e[i] - t->set_max[i] > 0.0) || (e[i] - t->set_min[i] < 0.0)) ? 1.0 : 0.0;
            if (!o->multidim_sets && is_set && m > 1) return -2;
#define PUSH(scale_rows)                                                        \
            do {                                                                \
                for (int i = 0; i < m; ++i)                                     \
                    for (int j = 0; j < n; ++j) {                               \
                        Ja[(nstack + i) * NMAX + j] = Ji[i * NMAX + j];         \
                        rJa[(nstack + i) * NMAX + j] =                          \
                            ((scale_rows) ? S[i] : 1.0) * Ji[i * NMAX + j];     \
                    }                                                           \
                nstack += m;                                                    \
            } while (0)
            /* chain 1 (:317-326) */
            if (is_first && is_eq) {
                gain_apply(t, e, tmp);
                for (int i = 0; i < m; ++i) des[i] = -tmp[i] - (o->feedforward ? Jt[i] : 0.0);
                if (dpinv(o, m, n, Ji, Pi)) return -1;
                for (int a = 0; a < n; ++a)
                    for (int i = 0; i < m; ++i) v[a] += Pi[a * RMAX + i] * des[i];
                PUSH(0);
            }
            /* chain 2 (:327-443) */
            if (is_first && is_veleq) {
                for (int i = 0; i < m; ++i) des[i] = t->target[i] - (o->feedforward ? Jt[i] : 0.0);
                if (dpinv(o, m, n, Ji, Pi)) return -1;
                for (int a = 0; a < n; ++a)
                    for (int i = 0; i < m; ++i) v[a] += Pi[a * RMAX + i] * des[i];
                PUSH(0);
            } else if (is_set && is_last && o->converge_final_set_to_max) {
                if ((act[mode] >> set_idx) & 1u) {
                    for (int i = 0; i < m; ++i) tmp[i] = t->set_max[i] - e[i];
                    gain_apply(t, tmp, des);
                    if (o->feedforward) for (int i = 0; i < m; ++i) des[i] -= Jt[i];
                    if (add_projected(o, n, nstack, Ja, rJa, m, Ji, des, v)) return -1;
                    PUSH(o->multidim_sets);
                } else {
                    tc[ntc++] = ti;
                }
                ++set_idx;
            } else if (is_eq) {
                gain_apply(t, e, tmp);
                for (int i = 0; i < m; ++i) des[i] = -tmp[i] - (o->feedforward ? Jt[i] : 0.0);
                if (add_projected(o, n, nstack, Ja, rJa, m, Ji, des, v)) return -1;
                PUSH(0);
            } else if (is_set) {
                if ((act[mode] >> set_idx) & 1u) {
                    PUSH(o->multidim_sets);
                } else {
                    tc[ntc++] = ti;
                }
                ++set_idx;
            } else if (is_veleq) {
                for (int i = 0; i < m; ++i) des[i] = t->target[i] - (o->feedforward ? Jt[i] : 0.0);
                if (add_projected(o, n, nstack, Ja, rJa, m, Ji, des, v)) return -1;
                PUSH(0);
            }
#undef PUSH
        }
        /* mode scan (:530-550) */
        int ok = 1;
        for (int k = 0; k < ntc && ok; ++k) {
            const clik_task* t = &d->tasks[tc[k]];
            double de[MMAX];
            for (int i = 0; i < t->m; ++i) {
                double s = T.Jt[tc[k]][i];
                for (int j = 0; j < n; ++j) s += T.J[tc[k]][i * NMAX + j] * v[j];
                de[i] = s;
            }
            if (t->m == 1) ok = in_tc_1d(T.e[tc[k]][0], t->set_min[0], t->set_max[0], de[0]);
            else ok = in_tc_multidim(t->m, T.e[tc[k]], t->set_min, t->set_max, de);
        }
        if (ok) {
            for (int j = 0; j < n; ++j) vout[j] = v[j];
            *mode_out = mode;
            return 0;
        }
    }
    for (int j = 0; j < n; ++j) vout[j] = 0.0;
    *mode_out = -1;
    return 0;
}

int orc_pinv_solve_batch_m(const clik_skill_desc* d, const clik_pinv_opts* o, int64_t B,
                           const double* tterms, const double* q, const double* x,
                           const double* y, double* dq, double* dx, int32_t* mode, int nthreads, double* margin);

int orc_pinv_solve_batch(const clik_skill_desc* d, const clik_pinv_opts* o, int64_t B,
                         const double* tterms, const double* q, const double* x,
                         const double* y, double* dq, double* dx, int32_t* mode, int nthreads)
{
    return orc_pinv_solve_batch_m(d, o, B, tterms, q, x, y, dq, dx, mode, nthreads, 0);
}

/* ... margin [B] (nullable): per instance, the smallest tangent-cone decision margin of its mode scan */
int orc_pinv_solve_batch_m(const clik_skill_desc* d, const clik_pinv_opts* o, int64_t B,
                           const double* tterms, const double* q, const double* x,
                           const double* y, double* dq, double* dx, int32_t* mode, int nthreads, double* margin)
{
    int n_sets = 0, nq = d->n_q, nx = d->n_x, ny = d->n_y;
    unsigned act[1 << CLIK_MAX_SETS];
    for (int ti = 0; ti < d->n_tasks; ++ti) n_sets += d->tasks[ti].cls == CLIK_CLS_SET;
    if (n_sets > CLIK_MAX_SETS) return -2;
    act[0] = 0;
    if (n_sets) activation_map(n_sets, act);
    int err = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(| : err)
#endif
    for (int64_t b = 0; b < B; ++b) {
        double z[NMAX], v[NMAX];
        int32_t md = -1;
        for (int j = 0; j < nq; ++j) z[j] = q[b * nq + j];
        for (int j = 0; j < nx; ++j) z[nq + j] = x[b * nx + j];
        g_tc_margin = INFINITY;
        int rc = pinv_one(d, o, act, n_sets, tterms, z, y ? y + b * ny : 0, v, &md);
        if (margin) margin[b] = g_tc_margin;
        if (rc) err |= 1;
        for (int j = 0; j < nq; ++j) dq[b * nq + j] = v[j];
        for (int j = 0; j < nx; ++j) dx[b * nx + j] = v[nq + j];
        if (mode) mode[b] = md;
    }
    return err ? -1 : 0;
}

/* ---------------------------------------------------------------- QP data
 * reactive_qp.py:175-246.  Hdiag [B][nv], A [B][nc][nv], lb/ub [B][nc] */
int orc_qp_data_batch(const clik_skill_desc* d, const clik_qp_opts* o, int64_t B,
                      const double* tterms, const double* q, const double* x, const double* y,
                      double* Hd, double* A, double* lb, double* ub)
{
    int nq = d->n_q, nx = d->n_x, ny = d->n_y, n = nq + nx, ns = 0, nc = 0;
    for (int ti = 0; ti < d->n_tasks; ++ti) {
        nc += d->tasks[ti].m;
        if (d->tasks[ti].soft) ns += d->tasks[ti].m;
    }
    int nv = n + ns;
    for (int64_t b = 0; b < B; ++b) {
        double z[NMAX];
        kin_t kin;
        for (int j = 0; j < nq; ++j) z[j] = q[b * nq + j];
        for (int j = 0; j < nx; ++j) z[nq + j] = x[b * nx + j];
        const double* yb = y ? y + b * ny : 0;
        kinematics(d, z, yb, &kin);
        double* h = Hd + b * nv;
        for (int j = 0; j < n; ++j) h[j] = o->weight_shifter * o->state_weights[j];
        for (int j = 0; j < ns; ++j) h[n + j] = o->weight_shifter + o->slack_weights[j];
        int row = 0, slack = 0;
        for (int ti = 0; ti < d->n_tasks; ++ti) {
            const clik_task* t = &d->tasks[ti];
            double e[MMAX], J[MMAX * NMAX], Jt[MMAX], tmp[MMAX], g[MMAX];
            task_eval(d, ti, &kin, z, yb, tterms, e, J, Jt);
            for (int i = 0; i < t->m; ++i) {
                double* a = A + (b * nc + row + i) * nv;
                for (int j = 0; j < nv; ++j) a[j] = 0.0;
                for (int j = 0; j < n; ++j) a[j] = J[i * NMAX + j];
                if (t->soft) a[n + slack + i] = -1.0;
                lb[b * nc + row + i] = -Jt[i];
                ub[b * nc + row + i] = -Jt[i];
            }
            if (t->cls == CLIK_CLS_EQ) {
                gain_apply(t, e, g);
                for (int i = 0; i < t->m; ++i) { lb[b * nc + row + i] -= g[i]; ub[b * nc + row + i] -= g[i]; }
            } else if (t->cls == CLIK_CLS_SET) {
                for (int i = 0; i < t->m; ++i) tmp[i] = t->set_min[i] - e[i];
                gain_apply(t, tmp, g);
                for (int i = 0; i < t->m; ++i) lb[b * nc + row + i] += g[i];
                for (int i = 0; i < t->m; ++i) tmp[i] = t->set_max[i] - e[i];
                gain_apply(t, tmp, g);
                for (int i = 0; i < t->m; ++i) ub[b * nc + row + i] += g[i];
            } else if (t->cls == CLIK_CLS_VELEQ) {
                for (int i = 0; i < t->m; ++i) { lb[b * nc + row + i] += t->target[i]; ub[b * nc + row + i] += t->target[i]; }
            } else {
                for (int i = 0; i < t->m; ++i) { lb[b * nc + row + i] += t->set_min[i]; ub[b * nc + row + i] += t->set_max[i]; }
            }
            if (t->soft) slack += t->m;
            row += t->m;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- dense QP (CPU baseline of the QP path)
 * min 1/2 x'Hx  s.t. lb <= A x <= ub,  H = diag(hd) > 0: the problem reactive_qp.py:491-513 hands to
 * cs.conic / qpOASES (third-party, not in the container).  Same method as oracle/clik_oracle.py::qp_solve_dense
 * (Goldfarb & Idnani 1983, explicit dense solves per step), restated in C so that the QP has a compiled CPU
 * baseline too.  Returns 0, or 2 when no feasible point exists / the iteration gives up. */
#define QV_MAX 24
#define QC_MAX 32
typedef struct { double n[QV_MAX]; double rhs; int is_eq; int row; int sgn; } qp_act_t;

static int qp_add_constraint(int nv, int nc, const double* Ginv, double* x, qp_act_t* act, int* nact,
                             double* u, const double* nvec_in, double rhs, int is_eq, int row, int sgn)
{
    double nvec[QV_MAX], up[2 * QC_MAX + 1];
    for (int j = 0; j < nv; ++j) nvec[j] = nvec_in[j];
    for (int j = 0; j < *nact; ++j) up[j] = u[j];
    up[*nact] = 0.0;
    for (int guard = 0; guard < 4 * (nc + 2); ++guard) {
        int q = *nact;
        double s = -rhs, z[QV_MAX], rvec[2 * QC_MAX];
        for (int j = 0; j < nv; ++j) s += nvec[j] * x[j];
        if (q > 0) {
            double M[2 * QC_MAX * 2 * QC_MAX], r[2 * QC_MAX];
            for (int a = 0; a < q; ++a) {
                for (int b = 0; b < q; ++b) {
                    double acc = 0.0;
                    for (int j = 0; j < nv; ++j) acc += act[a].n[j] * Ginv[j] * act[b].n[j];
                    M[a * q + b] = acc;
                }
                double acc = 0.0;
                for (int j = 0; j < nv; ++j) acc += act[a].n[j] * Ginv[j] * nvec[j];
                r[a] = acc;
            }
            if (ge_solve(q, 1, M, q, r, 1)) return 2;
            for (int a = 0; a < q; ++a) rvec[a] = r[a];
            for (int j = 0; j < nv; ++j) {
                double acc = nvec[j];
                for (int a = 0; a < q; ++a) acc -= act[a].n[j] * rvec[a];
                z[j] = Ginv[j] * acc;
            }
        } else {
            for (int j = 0; j < nv; ++j) z[j] = Ginv[j] * nvec[j];
        }
        double zn = 0.0, nmax = 0.0, gmax = 0.0;
        for (int j = 0; j < nv; ++j) {
            zn += z[j] * nvec[j];
            if (fabs(nvec[j]) > nmax) nmax = fabs(nvec[j]);
            if (Ginv[j] > gmax) gmax = Ginv[j];
        }
        double t1 = INFINITY, t2 = INFINITY;
        int drop = -1;
        for (int a = 0; a < q; ++a)
            if (!act[a].is_eq && rvec[a] > 1e-14) {
                double cand = up[a] / rvec[a];
                if (cand < t1) { t1 = cand; drop = a; }
            }
        double scale = nmax * nmax * gmax;
        if (scale < 1.0) scale = 1.0;
        if (zn > 1e-13 * scale) t2 = -s / zn;
        if (is_eq && s > 0) {
            for (int j = 0; j < nv; ++j) nvec[j] = -nvec[j];
            rhs = -rhs;
            continue;
        }
        double tstep = t1 < t2 ? t1 : t2;
        if (tstep == INFINITY) return 2;
        if (t2 != INFINITY)
            for (int j = 0; j < nv; ++j) x[j] += tstep * z[j];
        for (int a = 0; a < q; ++a) up[a] -= tstep * rvec[a];
        up[q] += tstep;
        if (t2 != INFINITY && tstep == t2) {
            for (int j = 0; j < nv; ++j) act[q].n[j] = nvec[j];
            act[q].rhs = rhs; act[q].is_eq = is_eq; act[q].row = row; act[q].sgn = sgn;
            *nact = q + 1;
            for (int a = 0; a <= q; ++a) u[a] = up[a];
            return 0;
        }
        for (int a = drop; a < q - 1; ++a) act[a] = act[a + 1];
        for (int a = drop; a < q; ++a) up[a] = up[a + 1];
        *nact = q - 1;
    }
    return 2;
}

static int qp_solve_dense_c(int nv, int nc, const double* hd, const double* A, const double* lb,
                            const double* ub, double* x, int max_iter)
{
    double Ginv[QV_MAX], u[2 * QC_MAX + 1];
    qp_act_t act[2 * QC_MAX + 1];
    int nact = 0;
    if (nv > QV_MAX || nc > QC_MAX) return 2;
    for (int j = 0; j < nv; ++j) { Ginv[j] = 1.0 / hd[j]; x[j] = 0.0; }
    for (int i = 0; i < nc; ++i)
        if (ub[i] - lb[i] <= 0.0) {
            double rhs = 0.5 * (lb[i] + ub[i]), nvec[QV_MAX], s = -rhs;
            for (int j = 0; j < nv; ++j) { nvec[j] = A[i * nv + j]; s += nvec[j] * x[j]; }
            if (s > 0) { for (int j = 0; j < nv; ++j) nvec[j] = -nvec[j]; rhs = -rhs; }
            if (qp_add_constraint(nv, nc, Ginv, x, act, &nact, u, nvec, rhs, 1, i, 0)) return 2;
        }
    for (int it = 0; it < max_iter; ++it) {
        double worst = -1e-11, prhs = 0.0;
        int pick = -1, psgn = 0;
        for (int i = 0; i < nc; ++i) {
            if (ub[i] - lb[i] <= 0.0) continue;
            for (int side = 0; side < 2; ++side) {
                int sgn = side == 0 ? +1 : -1;
                double bound = side == 0 ? lb[i] : ub[i];
                if (!isfinite(bound)) continue;
                double rhs = sgn * bound, ax = 0.0;
                int in_act = 0;
                for (int a = 0; a < nact; ++a) in_act |= (act[a].row == i && act[a].sgn == sgn);
                if (in_act) continue;
                for (int j = 0; j < nv; ++j) ax += A[i * nv + j] * x[j];
                double nrm = fabs(rhs) > 1.0 ? fabs(rhs) : 1.0;
                double viol = (sgn * ax - rhs) / nrm;
                if (viol < worst) { worst = viol; pick = i; psgn = sgn; prhs = rhs; }
            }
        }
        if (pick < 0) {
            for (int i = 0; i < nc; ++i) {
                double ax = 0.0;
                for (int j = 0; j < nv; ++j) ax += A[i * nv + j] * x[j];
                if (isfinite(lb[i]) && (lb[i] - ax) / (fabs(lb[i]) > 1.0 ? fabs(lb[i]) : 1.0) > 1e-8) return 2;
                if (isfinite(ub[i]) && (ax - ub[i]) / (fabs(ub[i]) > 1.0 ? fabs(ub[i]) : 1.0) > 1e-8) return 2;
            }
            return 0;
        }
        double nvec[QV_MAX];
        for (int j = 0; j < nv; ++j) nvec[j] = psgn * A[pick * nv + j];
        if (qp_add_constraint(nv, nc, Ginv, x, act, &nact, u, nvec, prhs, 0, pick, psgn)) return 2;
    }
    return 2;
}

/* literal ReactiveQPController.solve (reactive_qp.py:461-528) for a batch: xs [B][nv], status [B] */
int orc_qp_solve_batch(const clik_skill_desc* d, const clik_qp_opts* o, int64_t B,
                       const double* tterms, const double* q, const double* x, const double* y,
                       double* xs, int32_t* status, int nthreads)
{
    int nq = d->n_q, nx = d->n_x, ny = d->n_y, n = nq + nx, ns = 0, nc = 0;
    for (int ti = 0; ti < d->n_tasks; ++ti) {
        nc += d->tasks[ti].m;
        if (d->tasks[ti].soft) ns += d->tasks[ti].m;
    }
    int nv = n + ns;
    if (nv > QV_MAX || nc > QC_MAX) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        double Hd[QV_MAX], A[QC_MAX * QV_MAX], lb[QC_MAX], ub[QC_MAX];
        orc_qp_data_batch(d, o, 1, tterms, q + b * nq, x ? x + b * nx : 0, y ? y + b * ny : 0, Hd, A, lb, ub);
        int st = qp_solve_dense_c(nv, nc, Hd, A, lb, ub, xs + b * nv, 200);
        status[b] = st;
        if (st) for (int j = 0; j < nv; ++j) xs[b * nv + j] = NAN;
    }
    return 0;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
