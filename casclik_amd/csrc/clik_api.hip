// C ABI (include/clik.h) of the MI355X batched CLIK hot path: handle
// management, descriptor validation, mode table, kernel dispatch.
//
// Every entry point replaces a piece of the reference's controller setup/solve:
//   clik_pinv_create       <- PseudoInverseController.setup_problem_functions
//                             (pseudo_inverse.py:453-483) + create_activation_map (:107-130)
//   clik_pinv_solve_batch  <- PseudoInverseController.solve (:512-556)
//   clik_qp_create         <- ReactiveQPController.setup_problem_functions / setup_solver
//                             (reactive_qp.py:248-298)
//   clik_qp_solve_batch    <- ReactiveQPController.solve (:461-528)
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>
#include "clik_device.hpp"
#include "clik_workspace.hpp"

namespace clik {
struct LaunchArgs {
    const DevSkill* dS;
    const void*     dImg;
    const WarmArgs* warm;
    int nq, nx, ny;
    int mode_parallel;
    double* roll_x;         // (mirror of clik_pinv_kernels.hpp)
    double* roll_dx;
    int roll_stages;
    const double* t_inst;
};
int pinv_pick_kernel(const DevSkill& S, int allow_static);
const char* pinv_kernel_name(int k);
const char* pinv_static_variant(const ShapeDesc& sd, int mode_parallel, long long B);
bool shape_team_ok_rt(const ShapeDesc& sd);
bool shape_quad_front_ok_rt(const ShapeDesc& sd);
long long pinv_team_max_batch();
bool shape_value_lane_ok_rt(const ShapeDesc& sd);
long long pinv_value_lane_max_batch();
int pinv_kernel_width(int k);
int pinv_kernel_is_static(int k);
hipError_t pinv_launch_solve(int k, const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,
                             const double* x, const double* y, double* dq, double* dx, int32_t* mode,
                             hipStream_t stream);
hipError_t pinv_launch_rollout(int k, const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt,
                               double max_speed, long long B, double* q, const double* y, double* dq,
                               int32_t* mode, hipStream_t stream);
int pinv_lds_slots_host(int N, int ny);
hipError_t qp_launch_solve(int k, const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                           const double* q, const double* x, const double* y, double* dq, double* dx,
                           double* slack, int32_t* status, hipStream_t stream, GwsOwner* owner);
hipError_t qp_launch_data(int k, const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                          const double* q, const double* x, const double* y, double* Hd, double* A, double* lb,
                          double* ub, hipStream_t stream, GwsOwner* owner);
bool qp_variant_uses_workspace(int k);
int qp_pick_variant(int n, int nv, int nc);
int qp_variant_width(int k);
size_t qp_variant_lds(int k, int ny);
int qp_pick_static(const ShapeDesc& sd);
const char* qp_static_name(int k);
bool qp_box_family_rt(const ShapeDesc& sd);
int qp_plan_rows_rt(const ShapeDesc& sd);
int qp_layout_slots_rt(const ShapeDesc& sd);
int team_waves_rt(long long B);                 // (clik_pinv.hip)
hipError_t launch_ticket_feed(void* ticket, const unsigned* done, int n_ticks, int closed_loop, unsigned waves_per_tick,
                              unsigned long long timeout_ticks, hipStream_t stream);
hipError_t qp_launch_static(int k, const void* d_img, const TickArgs& tk, long long B, const double* q,
                            const double* x, const double* y, double* dq, double* dx, double* slack,
                            int32_t* status, int32_t* hot_set, int use_hot, hipStream_t stream,
                            const double* t_inst);
hipError_t qp_launch_rollout_static(int k, const void* d_img, const double* d_tterms, int n_ticks, double dt,
                                    double max_speed, long long B, double* q, const double* y, double* dq,
                                    double* slack, int32_t* status, double* x, double* dx, hipStream_t stream,
                                    int stages);
}  // namespace clik

using clik::DevSkill;
using clik::TickArgs;
using clik::WarmArgs;

typedef hipError_t (*clik_jit_solve_fn)(const clik::LaunchArgs*, const TickArgs*, long long, const double*,
                                        const double*, const double*, double*, double*, int32_t*, hipStream_t);
typedef hipError_t (*clik_jit_rollout_fn)(const clik::LaunchArgs*, const double*, int, double, double, long long,
                                          double*, const double*, double*, int32_t*, hipStream_t);

typedef hipError_t (*clik_jit_value_fn)(const clik::LaunchArgs*, const TickArgs*, long long, const double*,
                                        const double*, double*, int32_t*, hipStream_t);

struct clik_pinv {
    DevSkill  host;
    DevSkill* dev;
    clik::WarmArgs warm;
    void*     d_img;        // static kernels: device copy of the compact skill image
    int       mode_parallel; // speculative two-wave kernel for small batches (CLIK_MODE_PARALLEL=0 disables)
    // shape-specialised kernel attached at run time (clik_pinv_attach_kernel)
    clik_jit_solve_fn   jit_solve;
    clik_jit_rollout_fn jit_rollout;
    char      jit_name[64];
    int       kernel;       // index into the kernel table (static shape or dynamic)
    // team kernel with this skill's numbers compiled in (clik_pinv_attach_value_kernel), used for the batches the
    // image-reading team kernel would serve
    clik_jit_value_fn   val_solve;
    clik_jit_rollout_fn val_rollout;
    // ... and its resident form (clik_pinv_attach_resident_kernel)
    hipError_t (*val_resident)(const TickArgs*, long long, const double*, const double*, double*, int32_t*, void*,
                               unsigned*, int, unsigned long long, hipStream_t);
};

typedef hipError_t (*clik_jit_qp_fn)(const void*, const TickArgs*, long long, const double*, const double*,
                                     const double*, double*, double*, double*, int32_t*, int32_t*, int, hipStream_t,
                                     const double*);

typedef hipError_t (*clik_jit_qp_rollout_fn)(const void*, const double*, int, double, double, long long, double*,
                                             const double*, double*, double*, int32_t*, double*, double*,
                                             hipStream_t, int);

typedef hipError_t (*clik_jit_qp_value_fn)(const TickArgs*, long long, const double*, const double*, const double*,
                                           double*, double*, double*, int32_t*, int32_t*, int, hipStream_t);

typedef hipError_t (*clik_jit_qp_value_rollout_fn)(const double*, int, double, double, long long, double*, const double*,
                                                   double*, double*, int32_t*, double*, double*, hipStream_t, int);

struct clik_qp {
    DevSkill  host;
    DevSkill* dev;
    clik::WarmArgs warm;
    int       variant;      // dynamic kernel variant (qp_data and the fallback use it); -1: the skill only fits
                            // the shape-specialised kernels
    void*     d_img;        // shape-specialised kernels: skill image + QP options
    int       static_k;     // AOT shape-specialised kernel, -1 none
    clik_jit_qp_fn jit_solve;
    clik_jit_qp_rollout_fn jit_rollout;
    clik_jit_qp_value_fn val_solve;     // per-tick kernel with this skill's numbers and QP options compiled in
    clik_jit_qp_value_rollout_fn val_rollout;   // ... and its on-device rollout (box family), or null
    // ... and its resident form (clik_qp_attach_resident_kernel), or null
    hipError_t (*val_resident)(const TickArgs*, long long, const double*, const double*, double*, double*, int32_t*, void*,
                               unsigned*, int, unsigned long long, hipStream_t);
    char      jit_name[64];
    // work area of the global-workspace kernels (clik_workspace.hpp): belongs to this handle, released by clik_qp_destroy
    clik::GwsOwner* gws;
    ~clik_qp()
    {
        if (gws) {
            gws->release();
            delete gws;
        }
    }
};

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hipfail(hipError_t e, const char* what)
{
    return fail(CLIK_EHIP, "%s: %s", what, hipGetErrorString(e));
}

extern "C" const char* clik_last_error(void) { return g_err; }
extern "C" int32_t clik_abi_version(void) { return CLIK_ABI_VERSION; }

static int popcount32(unsigned v)
{
    int c = 0;
    while (v) { c += v & 1u; v >>= 1; }
    return c;
}

// Structural validation shared by both controllers; fills the derived fields.
static int validate_and_derive(const clik_skill_desc* d, DevSkill* S)
{
    if (!d) return fail(CLIK_EINVAL, "null skill descriptor");
    if (d->abi_version != CLIK_ABI_VERSION)
        return fail(CLIK_EINVAL, "descriptor ABI version %d, library %d", d->abi_version, CLIK_ABI_VERSION);
    const int n = d->n_q + d->n_x;
    if (d->n_q < 1 || d->n_x < 0 || n > CLIK_MAX_DOF)
        return fail(CLIK_EUNSUPPORTED, "n_robot_var + n_virtual_var = %d outside [1, %d]", n, CLIK_MAX_DOF);
    if (d->n_y < 0) return fail(CLIK_EINVAL, "negative n_y");
    if (d->n_joints < 0 || d->n_joints > CLIK_MAX_JOINTS)
        return fail(CLIK_EUNSUPPORTED, "chain has %d joints (limit %d)", d->n_joints, CLIK_MAX_JOINTS);
    if (d->n_tasks < 0 || d->n_tasks > CLIK_MAX_TASKS)
        return fail(CLIK_EUNSUPPORTED, "%d constraints (limit %d)", d->n_tasks, CLIK_MAX_TASKS);
    if (d->n_rows < 0 || d->n_rows > CLIK_MAX_ROWS) return fail(CLIK_EUNSUPPORTED, "too many affine rows");
    if (d->n_tslots < 0 || d->n_tslots > CLIK_MAX_TSLOTS) return fail(CLIK_EUNSUPPORTED, "too many time slots");
    if (d->quat_src < 0 || d->quat_src > 2) return fail(CLIK_EINVAL, "bad quat_src");
    if (d->quat_src == 2)
        for (int i = 0; i < 4; ++i)
            if (d->quat_yi[i] < 0 || d->quat_yi[i] >= d->n_y)
                return fail(CLIK_EINVAL, "orientation target index outside input_var");
    memset(S, 0, sizeof(*S));
    S->d = *d;
    S->n = n;
    bool any_fk = false, any_o = false;
    for (int j = 0; j < d->n_joints; ++j) {
        const clik_joint& jt = d->joints[j];
        int jf = 0;
        static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        bool r_ident = true;
        for (int k = 0; k < 9; ++k) r_ident = r_ident && (jt.R[k] == I3[k]);    // -0.0 == 0.0
        if (r_ident) jf |= 1;
        if (jt.p[0] == 0.0 && jt.p[1] == 0.0 && jt.p[2] == 0.0) jf |= 2;
        if (jt.type != CLIK_JOINT_FIXED)
            for (int k = 0; k < 3; ++k) {
                const double a = jt.axis[k], b = jt.axis[(k + 1) % 3], c = jt.axis[(k + 2) % 3];
                if (b == 0.0 && c == 0.0 && (a == 1.0 || a == -1.0)) jf |= ((k + 1) | (a < 0 ? 4 : 0)) << 4;
            }
        S->jflags[j] = jf;
        S->shape.jtype[j] = jt.type;
        S->shape.jq[j] = (jt.type == CLIK_JOINT_FIXED) ? -1 : jt.q_index;
        S->shape.jflags[j] = jf;
        if (jt.type == CLIK_JOINT_FIXED) continue;
        if (jt.type != CLIK_JOINT_REVOLUTE && jt.type != CLIK_JOINT_PRISMATIC)
            return fail(CLIK_EINVAL, "joint %d: unknown type %d", j, jt.type);
        if (jt.q_index < 0 || jt.q_index >= n) return fail(CLIK_EINVAL, "joint %d: q_index out of range", j);
        if (S->used_mask & (1u << jt.q_index))
            return fail(CLIK_EUNSUPPORTED, "state variable %d drives two joints", jt.q_index);
        S->used_mask |= 1u << jt.q_index;
        if (jt.type == CLIK_JOINT_REVOLUTE) S->rev_mask |= 1u << jt.q_index;
    }
    int n_sets = 0, n_slack = 0, n_rows_qp = 0;
    for (int ti = 0; ti < d->n_tasks; ++ti) {
        const clik_task& t = d->tasks[ti];
        if (t.cls < CLIK_CLS_EQ || t.cls > CLIK_CLS_VELSET) return fail(CLIK_EINVAL, "task %d: bad class", ti);
        if (t.attr_ext & ~(CLIK_ATTR_GAIN | CLIK_ATTR_SET_MIN | CLIK_ATTR_SET_MAX | CLIK_ATTR_TARGET))
            return fail(CLIK_EINVAL, "task %d: bad attr_ext", ti);
        if (t.attr_ext != 0 && ti >= clik::SHAPE_MAX_TASKS)
            return fail(CLIK_EUNSUPPORTED, "task %d: attributes given as expressions need a shape-specialised kernel "
                                           "(at most %d constraints)", ti, clik::SHAPE_MAX_TASKS);
        if (t.m < 1 || t.m > CLIK_MAX_M)
            return fail(CLIK_EUNSUPPORTED, "task %d: %d rows (limit %d)", ti, t.m, CLIK_MAX_M);
        for (int i = 0; i < t.m; ++i) {
            const int r0 = t.out_row0[i];
            const int nr = (t.out_kind[i] == CLIK_OUT_NORM2) ? t.out_nrows[i] : 1;
            if (t.out_kind[i] != CLIK_OUT_AFFINE && t.out_kind[i] != CLIK_OUT_NORM2 &&
                t.out_kind[i] != CLIK_OUT_EXTERN)
                return fail(CLIK_EINVAL, "task %d row %d: bad out_kind", ti, i);
            if ((t.out_kind[i] == CLIK_OUT_EXTERN) != (t.out_kind[0] == CLIK_OUT_EXTERN))
                return fail(CLIK_EINVAL, "task %d: EXTERN outputs cannot be mixed with table rows", ti);
            if (t.out_kind[i] == CLIK_OUT_EXTERN && ti >= clik::SHAPE_MAX_TASKS)
                return fail(CLIK_EUNSUPPORTED, "code-generated constraints need a shape-specialised kernel "
                                               "(at most %d constraints)", clik::SHAPE_MAX_TASKS);
            if (nr < 0 || r0 < 0 || r0 + nr > d->n_rows) return fail(CLIK_EINVAL, "task %d row %d: row range", ti, i);
            for (int k = r0; k < r0 + nr; ++k) {
                const clik_row& r = d->rows[k];
                if (r.flags & (CLIK_ROW_HAS_P | CLIK_ROW_HAS_R | CLIK_ROW_HAS_O)) any_fk = true;
                if (r.flags & CLIK_ROW_HAS_O) any_o = true;
                if ((r.flags & CLIK_ROW_HAS_T) && (r.t_slot < 0 || r.t_slot >= d->n_tslots))
                    return fail(CLIK_EINVAL, "row %d: t_slot out of range", k);
                if (r.n_y < 0 || r.n_y > CLIK_MAX_YTERMS) return fail(CLIK_EINVAL, "row %d: n_y", k);
                for (int q = 0; q < r.n_y; ++q)
                    if (r.yi[q] < 0 || r.yi[q] >= d->n_y) return fail(CLIK_EINVAL, "row %d: input index", k);
            }
        }
        if (t.cls == CLIK_CLS_SET) ++n_sets;
        if (t.soft) n_slack += t.m;
        n_rows_qp += t.m;
    }
    // per-task feature flags, constant-Jacobian detection, shape descriptor
    S->shape.n = n;
    S->shape.n_y = d->n_y;
    S->shape.n_x = d->n_x;
    S->shape.nj = d->n_joints;
    S->shape.n_tasks = d->n_tasks;
    S->shape.all_affine = 1;
    int last_row = 0;
    for (int ti = 0; ti < d->n_tasks; ++ti) {
        const clik_task& t = d->tasks[ti];
        int fl = 0, affine = 1;
        for (int i = 0; i < t.m; ++i) {
            const int nr = (t.out_kind[i] == CLIK_OUT_NORM2) ? t.out_nrows[i] : 1;
            if (t.out_kind[i] != CLIK_OUT_AFFINE) affine = 0;
            for (int k = t.out_row0[i]; k < t.out_row0[i] + nr; ++k) fl |= d->rows[k].flags;
            if (t.out_row0[i] + nr > last_row) last_row = t.out_row0[i] + nr;
        }
        S->task_flags[ti] = fl;
        S->task_const_j[ti] =
            (affine && !(fl & (CLIK_ROW_HAS_P | CLIK_ROW_HAS_R | CLIK_ROW_HAS_O))) ? 1 : 0;
        if (!affine) S->shape.all_affine = 0;
        if (ti < clik::SHAPE_MAX_TASKS) {
            S->shape.cls[ti] = t.cls;
            S->shape.m[ti] = t.m;
            // time terms do not change the code path of a row: drop the flag from the shape
            S->shape.flags[ti] = fl & ~CLIK_ROW_HAS_T;
            S->shape.const_j[ti] = S->task_const_j[ti];
            S->shape.gain_matrix[ti] = t.gain_is_matrix ? 1 : 0;
            S->shape.ext[ti] = ((t.out_kind[0] == CLIK_OUT_EXTERN) ? 1 : 0) | ((t.attr_ext & 15) << 1);
            int nyt = 0;
            for (int i = 0; i < t.m; ++i) {
                const int nrw = (t.out_kind[i] == CLIK_OUT_NORM2) ? t.out_nrows[i] : 1;
                for (int k = 0; k < nrw; ++k)
                    if (d->rows[t.out_row0[i] + k].n_y > nyt) nyt = d->rows[t.out_row0[i] + k].n_y;
            }
            S->shape.ny_terms[ti] = nyt;
            S->shape.has_t[ti] = (fl & CLIK_ROW_HAS_T) ? 1 : 0;
            // joint-space task: rows are distinct unit vectors (clik_device.hpp, ShapeDesc::ucol)
            bool unit = S->task_const_j[ti] != 0 && t.m > 0 && t.m <= CLIK_MAX_M;
            int cols[CLIK_MAX_M] = {0};
            unsigned seen = 0;
            for (int i = 0; i < t.m && unit; ++i) {
                const clik_row& r = d->rows[t.out_row0[i]];
                int col = -1;
                for (int j = 0; j < n; ++j) {
                    if (r.a[j] == 0.0) continue;
                    if (r.a[j] != 1.0 || col >= 0) { unit = false; break; }
                    col = j;
                }
                if (col < 0 || (seen >> col) & 1u) unit = false;
                else { seen |= 1u << col; cols[i] = col + 1; }
            }
            for (int i = 0; i < CLIK_MAX_M; ++i) S->shape.ucol[ti][i] = (unit && i < t.m) ? cols[i] : 0;
        }
        if (fl & CLIK_ROW_HAS_T) S->shape.all_affine = S->shape.all_affine;  // (time slots are run-time data)
    }
    // feature-coefficient patterns of the rows in task order (what the static kernels index:
    // rows of static shapes are contiguous); skills with other row layouts never match a static shape
    for (int r = 0; r < clik::SHAPE_MAX_ROWS; ++r) S->shape.row_nz[r] = S->shape.row_one[r] = 0u;
    {
        int r = 0;
        for (int ti = 0; ti < d->n_tasks && ti < clik::SHAPE_MAX_TASKS; ++ti) {
            const clik_task& t = d->tasks[ti];
            for (int i = 0; i < t.m; ++i) {
                const int nrw = (t.out_kind[i] == CLIK_OUT_NORM2) ? t.out_nrows[i] : 1;
                S->shape.out_nrows[ti][i] = (t.out_kind[i] == CLIK_OUT_NORM2) ? t.out_nrows[i] : 0;
                for (int k2 = 0; k2 < nrw; ++k2, ++r) {
                    if (r >= clik::SHAPE_MAX_ROWS) continue;
                    const clik_row& row = d->rows[t.out_row0[i] + k2];
                    unsigned nz = 0u, one = 0u;
                    auto mark = [&](double v, int bit) {
                        if (v != 0.0) nz |= 1u << bit;
                        if (v == 1.0) one |= 1u << bit;
                    };
                    if (row.flags & CLIK_ROW_HAS_P) for (int k = 0; k < 3; ++k) mark(row.b[k], k);
                    if (row.flags & CLIK_ROW_HAS_R) for (int k = 0; k < 9; ++k) mark(row.g[k], 3 + k);
                    if (row.flags & CLIK_ROW_HAS_O) for (int k = 0; k < 3; ++k) mark(row.h[k], 12 + k);
                    S->shape.row_nz[r] = nz;
                    S->shape.row_one[r] = one;
                }
            }
        }
    }
    S->shape.uses_fk = any_fk ? 1 : 0;
    S->shape.quat_src = any_o ? d->quat_src : 0;
    S->lds_slots = last_row;   // (temporarily) number of used affine rows, for the warm-up ranges
    if (any_fk && d->n_joints == 0) return fail(CLIK_EINVAL, "rows use the tool frame but the chain is empty");
    if (any_o && d->quat_src == 0) return fail(CLIK_EINVAL, "rows use the orientation error but no target is set");
    S->d.uses_fk = any_fk ? 1 : 0;
    S->n_sets = n_sets;
    S->n_slack = n_slack;
    S->n_qp_rows = n_rows_qp;
    S->n_qp_vars = n + n_slack;
    return CLIK_OK;
}

// dense solve A X = B (A k x k, B k x c, row-major), partial pivoting; host only
static bool host_solve(int k, int c, double* A, double* B)
{
    for (int col = 0; col < k; ++col) {
        int piv = col;
        for (int r = col + 1; r < k; ++r)
            if (std::fabs(A[r * k + col]) > std::fabs(A[piv * k + col])) piv = r;
        if (A[piv * k + col] == 0.0) return false;
        if (piv != col) {
            for (int j = 0; j < k; ++j) std::swap(A[col * k + j], A[piv * k + j]);
            for (int j = 0; j < c; ++j) std::swap(B[col * c + j], B[piv * c + j]);
        }
        for (int r = col + 1; r < k; ++r) {
            const double f = A[r * k + col] / A[col * k + col];
            for (int j = col; j < k; ++j) A[r * k + j] -= f * A[col * k + j];
            for (int j = 0; j < c; ++j) B[r * c + j] -= f * B[col * c + j];
        }
    }
    for (int col = k - 1; col >= 0; --col)
        for (int j = 0; j < c; ++j) {
            double s = B[col * c + j];
            for (int r = col + 1; r < k; ++r) s -= A[col * k + r] * B[r * c + j];
            B[col * c + j] = s / A[col * k + col];
        }
    return true;
}

// P (n x m, ld CLIK_MAX_M) = pinv(J) with the controller's method
// (pseudo_inverse.py:92-105) for a state-independent Jacobian J (m x n)
static bool host_const_pinv(const clik_pinv_opts& o, int m, int n, const double* J, double* P)
{
    const bool standard = o.pinv_method == CLIK_PINV_STANDARD;
    const double lam = standard ? 0.0 : o.damping_factor;
    const bool wide = standard ? (m < n) : (n >= m);
    double A[CLIK_MAX_M * CLIK_MAX_M], X[CLIK_MAX_M * CLIK_MAX_M];
    if (wide) {
        for (int i = 0; i < m; ++i)
            for (int k = 0; k < m; ++k) {
                double sacc = (i == k) ? lam : 0.0;
                for (int j = 0; j < n; ++j) sacc += J[i * n + j] * J[k * n + j];
                A[i * m + k] = sacc;
            }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) X[i * n + j] = J[i * n + j];
        if (!host_solve(m, n, A, X)) return false;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i) P[j * CLIK_MAX_M + i] = X[i * n + j];
    } else {
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double sacc = (a == b) ? lam : 0.0;
                for (int i = 0; i < m; ++i) sacc += J[i * n + a] * J[i * n + b];
                A[a * n + b] = sacc;
            }
        for (int a = 0; a < n; ++a)
            for (int i = 0; i < m; ++i) X[a * m + i] = J[i * n + a];
        if (!host_solve(n, m, A, X)) return false;
        for (int a = 0; a < n; ++a)
            for (int i = 0; i < m; ++i) P[a * CLIK_MAX_M + i] = X[a * m + i];
    }
    return true;
}

// Compact skill image for the shape-specialised kernels; layout =
// clik::SkillImage<nj, nt, nr> (clik_device.hpp), padded to a multiple of 1 KiB.
// Static shapes require the rows contiguous in task / output order.
static bool build_skill_image(const DevSkill& S, std::vector<char>& out, size_t* image_bytes = nullptr,
                              size_t extra = 0)
{
    const int nj = S.d.n_joints, nt = S.d.n_tasks;
    int nr = 0;
    for (int ti = 0; ti < nt; ++ti) {
        const clik_task& t = S.d.tasks[ti];
        for (int i = 0; i < t.m; ++i) {
            // rows in task order, output order; a 2-norm output owns a group of consecutive rows
            if (t.out_row0[i] != nr) return false;
            nr += (t.out_kind[i] == CLIK_OUT_NORM2) ? t.out_nrows[i] : 1;
        }
    }
    if (nr > clik::SHAPE_MAX_ROWS) return false;
    const size_t o_j = 0;
    const size_t o_t = o_j + sizeof(clik_joint) * (size_t)(nj > 0 ? nj : 1);
    const size_t o_r = o_t + sizeof(clik_task) * (size_t)nt;
    const size_t o_c = o_r + sizeof(clik_row) * (size_t)nr;
    const size_t o_g = o_c + sizeof(double) * CLIK_MAX_DOF * CLIK_MAX_M * (size_t)nt;
    constexpr size_t GT = CLIK_MAX_DOF * (CLIK_MAX_DOF + 1) / 2;
    const size_t o_tail = o_g + sizeof(double) * GT * (size_t)nt;
    const size_t total = o_tail + sizeof(double) * 5 + sizeof(int32_t) * 6;
    if (image_bytes) *image_bytes = total;
    out.assign((((total + 15) & ~(size_t)15) + extra + 1023) / 1024 * 1024, 0);
    memcpy(&out[o_j], S.d.joints, sizeof(clik_joint) * (size_t)nj);
    memcpy(&out[o_t], S.d.tasks, sizeof(clik_task) * (size_t)nt);
    memcpy(&out[o_r], S.d.rows, sizeof(clik_row) * (size_t)nr);
    for (int ti = 0; ti < nt; ++ti)
        memcpy(&out[o_c + sizeof(double) * CLIK_MAX_DOF * CLIK_MAX_M * (size_t)ti], S.cpinv[ti],
               sizeof(double) * CLIK_MAX_DOF * CLIK_MAX_M);
    // J^T J of the constant-Jacobian tasks, packed lower triangle tri(a,b) = a(a+1)/2 + b
    for (int ti = 0; ti < nt; ++ti) {
        if (!S.task_const_j[ti]) continue;
        const clik_task& t = S.d.tasks[ti];
        double* g = reinterpret_cast<double*>(&out[o_g + sizeof(double) * GT * (size_t)ti]);
        for (int a = 0; a < S.n; ++a)
            for (int b = 0; b <= a; ++b) {
                double acc = 0.0;
                for (int i = 0; i < t.m; ++i)
                    acc += S.d.rows[t.out_row0[i]].a[a] * S.d.rows[t.out_row0[i]].a[b];
                g[a * (a + 1) / 2 + b] = acc;
            }
    }
    // unused input_var terms must read index 0 with coefficient 0
    for (int r = 0; r < nr; ++r) {
        clik_row* row = reinterpret_cast<clik_row*>(&out[o_r + sizeof(clik_row) * (size_t)r]);
        for (int k = row->n_y; k < CLIK_MAX_YTERMS; ++k) { row->yc[k] = 0.0; row->yi[k] = 0; }
    }
    double tail_d[5] = {S.po.pinv_method == CLIK_PINV_STANDARD ? 0.0 : S.po.damping_factor,
                        S.d.quat[0], S.d.quat[1], S.d.quat[2], S.d.quat[3]};
    int32_t tail_i[6] = {S.d.quat_yi[0], S.d.quat_yi[1], S.d.quat_yi[2], S.d.quat_yi[3], S.d.n_tslots, 0};
    memcpy(&out[o_tail], tail_d, sizeof(tail_d));
    memcpy(&out[o_tail + sizeof(tail_d)], tail_i, sizeof(tail_i));
    return true;
}

// scalar-cache warm-up ranges (clik_device.hpp, warm_descriptor): the used
// prefixes of the descriptor arrays and the derived tables
static void compute_warm(const DevSkill& S, int used_rows, clik::WarmArgs& w)
{
    auto lo = [](size_t off) { return (int32_t)(off & ~(size_t)63); };
    auto hi = [](size_t off) { return (int32_t)((off + 63) & ~(size_t)63); };
    const size_t o_d = offsetof(DevSkill, d);
    w.off[0] = lo(o_d);
    w.end[0] = hi(o_d + offsetof(clik_skill_desc, joints) + (size_t)S.d.n_joints * sizeof(clik_joint));
    w.off[1] = lo(o_d + offsetof(clik_skill_desc, tasks));
    w.end[1] = hi(o_d + offsetof(clik_skill_desc, tasks) + (size_t)S.d.n_tasks * sizeof(clik_task));
    w.off[2] = lo(o_d + offsetof(clik_skill_desc, rows));
    w.end[2] = hi(o_d + offsetof(clik_skill_desc, rows) + (size_t)used_rows * sizeof(clik_row));
    w.off[3] = lo(offsetof(DevSkill, shape));
    w.end[3] = hi(offsetof(DevSkill, cpinv) + (size_t)S.d.n_tasks * sizeof(S.cpinv[0]));
    w.off[4] = lo(offsetof(DevSkill, po));
    w.end[4] = hi(sizeof(DevSkill));
    w.token_off = (int32_t)offsetof(DevSkill, zero_token);
}

static void finish_pinv_shape(DevSkill& S, const clik_pinv_opts* opts)
{
    S.po = *opts;
    S.shape.feedforward = opts->feedforward ? 1 : 0;
    S.shape.multidim = opts->multidim_sets ? 1 : 0;
    S.shape.conv_last = opts->converge_final_set_to_max ? 1 : 0;
    S.shape.standard = opts->pinv_method == CLIK_PINV_STANDARD ? 1 : 0;
    S.shape.qp = 0;
    for (int ti = 0; ti < clik::SHAPE_MAX_TASKS; ++ti) S.shape.soft[ti] = 0;
}

// shape of a ReactiveQPController: no pinv options, per-task soft flags
static void finish_qp_shape(DevSkill& S)
{
    S.shape.feedforward = S.shape.multidim = S.shape.conv_last = S.shape.standard = 0;
    S.shape.qp = 1;
    for (int ti = 0; ti < clik::SHAPE_MAX_TASKS; ++ti)
        S.shape.soft[ti] = (ti < S.d.n_tasks && S.d.tasks[ti].soft) ? 1 : 0;
}

// rows handed to the active-set solver by the shape-specialised QP kernel (clik_qp_static.hpp,
// make_qp_plan): everything but the soft equalities
static int qp_static_rows(const DevSkill& S)
{
    int nr = 0;
    for (int ti = 0; ti < S.d.n_tasks; ++ti) {
        const clik_task& t = S.d.tasks[ti];
        const bool folded = t.soft && (t.cls == CLIK_CLS_EQ || t.cls == CLIK_CLS_VELEQ);
        if (!folded) nr += t.m;
    }
    return nr;
}

// can a shape-specialised QP kernel serve the skill?  (image layout, row budget, LDS)
static bool qp_static_eligible(const DevSkill& S)
{
    if (S.d.n_tasks > clik::SHAPE_MAX_TASKS) return false;
    std::vector<char> img;
    size_t image_bytes = 0;
    if (!build_skill_image(S, img, &image_bytes, sizeof(clik::QpTail))) return false;
    // (the kernels' own count: a joint-limit row and a speed-limit row on the same state are ONE active-set row -
    // a skill with walls, joint limits and speed limits on every joint of a 7-DoF arm has 10 such rows, not 17)
    const int nr = clik::qp_plan_rows_rt(S.shape);
    if (nr > 16) return false;
    // (the layout's own slot count: the primal families - bound-constrained, mixed - keep no dual Hessian in LDS, so a
    // two-arm skill with 14 merged box rows fits where the dual form's 14 x 14 + 14 x 14 slots would not)
    const size_t slots = (size_t)clik::qp_layout_slots_rt(S.shape);
    return img.size() + slots * 64 * sizeof(double) <= 160u * 1024u;
}

static bool build_qp_image(const DevSkill& S, std::vector<char>& out)
{
    size_t image_bytes = 0;
    if (!build_skill_image(S, out, &image_bytes, sizeof(clik::QpTail))) return false;
    clik::QpTail t;
    memset(&t, 0, sizeof(t));
    t.mu = S.qo.weight_shifter;
    for (int j = 0; j < CLIK_MAX_DOF; ++j) t.state_w[j] = S.qo.state_weights[j];
    for (int k = 0; k < CLIK_MAX_QPROWS; ++k) t.slack_w[k] = S.qo.slack_weights[k];
    t.max_iter = S.qo.max_iter;
    memcpy(&out[(image_bytes + 15) & ~(size_t)15], &t, sizeof(t));
    return true;
}

// constraints whose rows are code generated from the caller's expression graph (CLIK_OUT_EXTERN)
// exist only inside the kernel instantiated for them
static bool skill_has_extern(const DevSkill& S)
{
    for (int ti = 0; ti < S.d.n_tasks; ++ti)
        if (S.d.tasks[ti].out_kind[0] == CLIK_OUT_EXTERN || S.d.tasks[ti].attr_ext != 0) return true;
    return false;
}
// ... and so do constraints with more rows than the built-in kernels are wide (CLIK_DYN_MAX_M)
static bool skill_has_wide_task(const DevSkill& S)
{
    for (int ti = 0; ti < S.d.n_tasks; ++ti)
        if (S.d.tasks[ti].m > CLIK_DYN_MAX_M) return true;
    return false;
}
static bool skill_needs_static(const DevSkill& S) { return skill_has_extern(S) || skill_has_wide_task(S); }
static int extern_needs_kernel(const char* what)
{
    return fail(CLIK_EUNSUPPORTED, "%s: the skill has code-generated constraint rows / attributes or a constraint with more "
                                   "than %d rows, and no kernel instantiated for it is attached "
                                   "(casclik_amd.jit needs hipcc)", what, CLIK_DYN_MAX_M);
}

// a handle created without a built-in kernel (more state variables than the dynamic-shape kernels carry) solves only
// through an instantiated kernel: say so with the documented code instead of failing inside the launch
static int no_kernel_for_wide_state(const DevSkill& S, const char* what)
{
    return fail(CLIK_EUNSUPPORTED, "%s: the skill has %d state variables (the built-in kernels carry 8) and no kernel "
                                   "instantiated for it is attached (casclik_amd.jit needs hipcc)", what, S.n);
}

// C++ aggregate initialiser of a ShapeDesc (field order of clik_device.hpp)
static std::string shape_to_string(const clik::ShapeDesc& h)
{
    std::string o = "{";
    auto num = [&](int v) { o += std::to_string(v); o += ", "; };
    auto arr = [&](const int* a, int nn, int used) {
        o += "{";
        for (int i = 0; i < nn; ++i) { o += std::to_string(i < used ? a[i] : 0); if (i + 1 < nn) o += ", "; }
        o += "}, ";
    };
    const int nt = h.n_tasks < clik::SHAPE_MAX_TASKS ? h.n_tasks : clik::SHAPE_MAX_TASKS;
    num(h.n); num(h.n_tasks);
    arr(h.cls, clik::SHAPE_MAX_TASKS, nt); arr(h.m, clik::SHAPE_MAX_TASKS, nt);
    arr(h.flags, clik::SHAPE_MAX_TASKS, nt); arr(h.const_j, clik::SHAPE_MAX_TASKS, nt);
    num(h.all_affine); num(h.uses_fk); num(h.quat_src);
    num(h.feedforward); num(h.multidim); num(h.conv_last); num(h.standard);
    num(h.nj);
    arr(h.jtype, CLIK_MAX_JOINTS, h.nj); arr(h.jq, CLIK_MAX_JOINTS, h.nj); arr(h.jflags, CLIK_MAX_JOINTS, h.nj);
    arr(h.gain_matrix, clik::SHAPE_MAX_TASKS, nt);
    arr(h.ny_terms, clik::SHAPE_MAX_TASKS, nt);
    num(h.n_y);
    arr(h.has_t, clik::SHAPE_MAX_TASKS, nt);
    o += "{";
    for (int i = 0; i < clik::SHAPE_MAX_TASKS; ++i) {
        o += "{";
        for (int k = 0; k < CLIK_MAX_M; ++k) { o += std::to_string(i < nt ? h.ucol[i][k] : 0); if (k + 1 < CLIK_MAX_M) o += ", "; }
        o += (i + 1 < clik::SHAPE_MAX_TASKS) ? "}, " : "}";
    }
    o += "}, ";
    num(h.qp);
    arr(h.soft, clik::SHAPE_MAX_TASKS, nt);
    auto uarr = [&](const unsigned* a, int nn, bool last) {
        o += "{";
        for (int i = 0; i < nn; ++i) { o += std::to_string(a[i]) + "u"; if (i + 1 < nn) o += ", "; }
        o += last ? "}" : "}, ";
    };
    uarr(h.row_nz, clik::SHAPE_MAX_ROWS, false);
    uarr(h.row_one, clik::SHAPE_MAX_ROWS, false);
    o += "{";
    for (int i = 0; i < clik::SHAPE_MAX_TASKS; ++i) {
        o += "{";
        for (int k = 0; k < CLIK_MAX_M; ++k) { o += std::to_string(i < nt ? h.out_nrows[i][k] : 0); if (k + 1 < CLIK_MAX_M) o += ", "; }
        o += (i + 1 < clik::SHAPE_MAX_TASKS) ? "}, " : "}";
    }
    o += "}, " + std::to_string(h.n_x) + ", ";
    o += "{";
    for (int i = 0; i < clik::SHAPE_MAX_TASKS; ++i) { o += std::to_string(i < nt ? h.ext[i] : 0); if (i + 1 < clik::SHAPE_MAX_TASKS) o += ", "; }
    o += "}}";
    return o;
}

// C++ initialiser of the ShapeDesc a skill + options map to (host only, no HIP
// call): the input of tools/gen_shapes.py, which writes clik_shapes_gen.hpp.
extern "C" int clik_shape_describe(const clik_skill_desc* desc, const clik_pinv_opts* opts, char* buf, int cap)
{
    if (!opts || !buf || cap <= 0) return fail(CLIK_EINVAL, "bad arguments");
    DevSkill* S = new (std::nothrow) DevSkill();
    if (!S) return fail(CLIK_ENOMEM, "out of host memory");
    int rc = validate_and_derive(desc, S);
    if (rc) { delete S; return rc; }
    finish_pinv_shape(*S, opts);
    const clik::ShapeDesc& h = S->shape;
    const std::string o = shape_to_string(h);
    // (kStaticMaxSets of clik_pinv_kernels.hpp: up to 2^6 mode bodies per kernel)
    bool eligible = S->d.n_tasks <= clik::SHAPE_MAX_TASKS && S->n_sets <= 6;
    {
        std::vector<char> img;
        if (!build_skill_image(*S, img)) eligible = false;      // rows not in task order / too many rows
    }
    // the static plan evaluates the doubly processed first EqualityConstraint in closed form from its own
    // factor or from the host-precomputed inverse of a constant Jacobian (clik_pinv_static.hpp); "standard"
    // keeps the wide-only rule (its tall branch has no damping to carry the closed form)
    for (unsigned act = 0; eligible && act < (1u << S->n_sets); ++act) {
        int r = 0, set_idx = 0;
        for (int ti = 0; ti < h.n_tasks; ++ti) {
            const int cls = h.cls[ti];
            if (cls == CLIK_CLS_VELSET) continue;
            if (cls == CLIK_CLS_SET) {
                const bool active = (act >> set_idx) & 1u;
                ++set_idx;
                if (!active) continue;
            }
            const bool conv = cls == CLIK_CLS_SET && ti == h.n_tasks - 1 && h.conv_last;
            const bool contributes = cls == CLIK_CLS_EQ || cls == CLIK_CLS_VELEQ || conv;
            if (contributes && r == 0 && cls == CLIK_CLS_EQ) {
                const bool wide = h.standard ? (h.m[ti] < h.n) : (h.n >= h.m[ti]);
                if (!wide && h.standard) eligible = false;
            }
            r += h.m[ti];
        }
    }
    delete S;
    if ((int)o.size() + 1 > cap) return fail(CLIK_EINVAL, "buffer too small");
    memcpy(buf, o.c_str(), o.size() + 1);
    return eligible ? 1 : 0;
}

// CLIK_HOST_ONLY=1: handles are created WITHOUT touching a device (no allocation, no upload) - for the host-side
// queries only (clik_*_image_words, kernel names): casclik_amd/jit.py uses it to instantiate value-specialised kernels
// ahead of time on a machine without a GPU.  Every solve / rollout / data entry point refuses such a handle.
// (clik_*_create_host ask for it explicitly, per call and per thread; the environment variable remains for whole processes)
static thread_local bool g_host_only_call = false;
static bool host_only_mode()
{
    if (g_host_only_call) return true;
    const char* e = getenv("CLIK_HOST_ONLY");
    return e && e[0] == '1';
}
#define CLIK_NEEDS_DEVICE_HANDLE(h)                                                                        \
    do {                                                                                                   \
        if ((h) && (h)->dev == nullptr)                                                                    \
            return fail(CLIK_EINVAL, "this handle was created host-only (clik_*_create_host / CLIK_HOST_ONLY=1): host-side queries only"); \
    } while (0)

extern "C" int clik_pinv_create(const clik_skill_desc* desc, const clik_pinv_opts* opts, clik_pinv** out)
{
    if (!out) return fail(CLIK_EINVAL, "null out pointer");
    *out = nullptr;
    if (!opts) return fail(CLIK_EINVAL, "null options");
    clik_pinv* h = new (std::nothrow) clik_pinv();
    if (!h) return fail(CLIK_ENOMEM, "out of host memory");
    int rc = validate_and_derive(desc, &h->host);
    if (rc) { delete h; return rc; }
    DevSkill& S = h->host;
    h->d_img = nullptr;
    h->val_solve = nullptr;
    h->val_rollout = nullptr;
    h->val_resident = nullptr;
    h->jit_solve = nullptr;
    h->jit_rollout = nullptr;
    h->val_solve = nullptr;
    h->val_rollout = nullptr;
    h->val_resident = nullptr;
    h->jit_name[0] = 0;
    finish_pinv_shape(S, opts);
    if (opts->pinv_method != CLIK_PINV_DAMPED && opts->pinv_method != CLIK_PINV_STANDARD) {
        delete h;
        return fail(CLIK_EINVAL, "pinv_method must be damped or standard");
    }
    if (S.n_sets > CLIK_MAX_SETS) {
        delete h;
        return fail(CLIK_EUNSUPPORTED, "%d SetConstraints give 2^%d modes (limit %d sets)", S.n_sets, S.n_sets,
                    CLIK_MAX_SETS);
    }
    for (int ti = 0; ti < S.d.n_tasks; ++ti) {
        const clik_task& t = S.d.tasks[ti];
        if (t.cls == CLIK_CLS_SET && t.m > 1 && !opts->multidim_sets) {
            // same refusal as pseudo_inverse.py:299-312
            delete h;
            return fail(CLIK_EUNSUPPORTED,
                        "PseudoInverseController does not yet have guaranteed stable support for "
                        "multidimensional SetConstraints (task %d has %d rows). Set the multidim_sets "
                        "field in options to True for experimental support.", ti, t.m);
        }
    }
    // mode table (pseudo_inverse.py:107-130): bit k <-> k-th SetConstraint in
    // priority order; patterns stably sorted by the number of active sets
    S.n_modes = 1 << S.n_sets;
    {
        int k = 0;
        for (int pc = 0; pc <= S.n_sets; ++pc)
            for (int v = 0; v < S.n_modes; ++v)
                if (popcount32((unsigned)v) == pc) S.act[k++] = (uint32_t)v;
    }
    S.last_set_converges = (S.d.n_tasks > 0 && S.d.tasks[S.d.n_tasks - 1].cls == CLIK_CLS_SET &&
                            opts->converge_final_set_to_max) ? 1 : 0;
    for (int ti = 0; ti < S.d.n_tasks; ++ti) {
        if (!S.task_const_j[ti]) continue;
        const clik_task& t = S.d.tasks[ti];
        double J[CLIK_MAX_M * CLIK_MAX_DOF];
        for (int i = 0; i < t.m; ++i)
            for (int j = 0; j < S.n; ++j) J[i * S.n + j] = S.d.rows[t.out_row0[i]].a[j];
        if (!host_const_pinv(*opts, t.m, S.n, J, S.cpinv[ti])) {
            // singular Gram matrix (pinv_method "standard" on a rank-deficient
            // task): let the device run the generic solve like the reference would
            S.task_const_j[ti] = 0;
            if (ti < clik::SHAPE_MAX_TASKS) S.shape.const_j[ti] = 0;
        }
    }
    {
        // CLIK_FORCE_DYNAMIC=1: dynamic kernel only; CLIK_NO_AOT=1: skip the AOT table
        // (the Python layer may still attach a run-time instantiated kernel)
        const char* force = getenv("CLIK_FORCE_DYNAMIC");
        const char* noaot = getenv("CLIK_NO_AOT");
        h->kernel = clik::pinv_pick_kernel(S, ((force && force[0] == '1') || (noaot && noaot[0] == '1')) ? 0 : 1);
    }
    if (h->kernel < 0 && !skill_has_wide_task(S) && S.n <= 8) {
        delete h;
        return fail(CLIK_EUNSUPPORTED, "no kernel variant for n = %d", S.n);
    }
    // (kernel < 0 with a wide constraint or with more than eight state variables - a 7-DoF arm with two or three
    // virtual variables, CLIK_MAX_DOF = 10: no built-in kernel is wide enough; the handle solves once a kernel
    // instantiated for the skill is attached, and answers CLIK_EUNSUPPORTED until then)
    compute_warm(S, S.lds_slots, h->warm);
    S.zero_token = 0;
    S.lds_slots = clik::pinv_lds_slots_host(clik::pinv_kernel_width(h->kernel), S.d.n_y);
    if ((size_t)S.lds_slots * clik::WAVE * sizeof(double) > 160u * 1024u) {
        delete h;
        return fail(CLIK_EUNSUPPORTED, "skill needs %d LDS slots per lane (input_var too large)", S.lds_slots);
    }
    h->dev = nullptr;
    const bool host_only = host_only_mode();
    hipError_t e = hipSuccess;
    if (!host_only) {
        e = hipMalloc((void**)&h->dev, sizeof(DevSkill));
        if (e != hipSuccess) { delete h; return hipfail(e, "hipMalloc(skill)"); }
        e = hipMemcpy(h->dev, &S, sizeof(DevSkill), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return hipfail(e, "hipMemcpy(skill)"); }
    }
    if (!host_only && clik::pinv_kernel_is_static(h->kernel)) {
        std::vector<char> img;
        if (!build_skill_image(S, img)) {
            // rows not laid out contiguously: serve the skill with the dynamic kernel
            h->kernel = clik::pinv_pick_kernel(S, 0);
        } else {
            e = hipMalloc(&h->d_img, img.size());
            if (e == hipSuccess) e = hipMemcpy(h->d_img, img.data(), img.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (h->d_img) (void)hipFree(h->d_img);
                (void)hipFree(h->dev);
                delete h;
                return hipfail(e, "skill image upload");
            }
        }
    }
    {
        // Small batches (fewer wavefronts than SIMDs) put more than one wave on the same 64
        // instances.  mode_parallel bit 0: speculative two-wave evaluation of both modes
        // (pinv_solve_static_mp_kernel; measured 8.0 -> 7.2 us on the config-3 stack, "mixed").
        // Bit 1: role split (pinv_solve_static_split_kernel: main + helper wave per mode, 4 waves);
        // it shortens the critical wave by 11 % in cycles but measures the same wall time as the
        // two-wave kernel (6.42 vs 6.47 us), so it is opt-in: CLIK_ROLE_SPLIT=1.
        // CLIK_MODE_PARALLEL=0 disables both.
        const char* mp = getenv("CLIK_MODE_PARALLEL");
        const char* rs = getenv("CLIK_ROLE_SPLIT");
        h->mode_parallel = (mp && mp[0] == '0') ? 0 : 1;
        if (h->mode_parallel && rs && rs[0] == '1') h->mode_parallel |= 2;
        // Bit 2: team kernel, four lanes per instance (pinv_solve_static_team_kernel) for the config-3
        // family up to kTeamMaxBatch instances; bit 3: at any batch size.  CLIK_LANES=1 keeps the
        // lane-per-instance kernels, CLIK_LANES=4 forces the team kernel (head-to-head runs), unset / 0 =
        // the library's choice.
        // Bit 4: CLIK_LARGE_BATCH=0 keeps the one-wave-per-SIMD lane kernel at every batch size; bit 5 marks
        // handles served by the ahead-of-time table (the only ones that carry the large-batch build)
        const char* lb = getenv("CLIK_LARGE_BATCH");
        if (lb && lb[0] == '0') h->mode_parallel |= 16;
        if (clik::pinv_kernel_is_static(h->kernel)) h->mode_parallel |= 32;
        {
            const char* qf = getenv("CLIK_QUAD_FRONT");      // 0: single-mode skills keep one lane per instance at small batches
            if (qf && qf[0] == '0') h->mode_parallel |= 128;
        }
        const char* ln = getenv("CLIK_LANES");
        if (!ln || ln[0] == '0' || ln[0] == '\0') h->mode_parallel |= 4;
        else if (ln[0] == '4') h->mode_parallel |= 4 | 8;
    }
    *out = h;
    return CLIK_OK;
}

// Attach a shape-specialised kernel instantiated at run time for this handle's
// skill (casclik_amd/jit.py compiles clik_pinv_kernels.hpp for the ShapeDesc that
// clik_shape_describe printed).  Mirrors the reference's JIT at
// setup_problem_functions (pseudo_inverse.py:476-483).
extern "C" int clik_pinv_attach_kernel(clik_pinv* h, void* solve_fn, void* rollout_fn, const char* name)
{
    // (rollout_fn may be NULL: the rollout then keeps the kernel chosen at creation)
    if (!h || !solve_fn) return fail(CLIK_EINVAL, "null argument");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (!h->d_img) {
        std::vector<char> img;
        if (!build_skill_image(h->host, img))
            return fail(CLIK_EUNSUPPORTED, "skill rows are not contiguous: no static kernel possible");
        hipError_t e = hipMalloc(&h->d_img, img.size());
        if (e == hipSuccess) e = hipMemcpy(h->d_img, img.data(), img.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (h->d_img) (void)hipFree(h->d_img);
            h->d_img = nullptr;
            return hipfail(e, "skill image upload");
        }
    }
    h->jit_solve = (clik_jit_solve_fn)solve_fn;
    h->jit_rollout = (clik_jit_rollout_fn)rollout_fn;
    snprintf(h->jit_name, sizeof(h->jit_name), "%s", name ? name : "jit");
    return CLIK_OK;
}

extern "C" int clik_pinv_attach_value_kernel(clik_pinv* h, void* solve_fn, void* rollout_fn)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    if (solve_fn && !clik::shape_team_ok_rt(h->host.shape) && !clik::shape_value_lane_ok_rt(h->host.shape))
        return fail(CLIK_EUNSUPPORTED, "value-specialised kernels exist for the four-lanes-per-instance family and for "
                                       "single-mode skills without virtual variables");
    h->val_solve = (clik_jit_value_fn)solve_fn;
    h->val_rollout = solve_fn ? (clik_jit_rollout_fn)rollout_fn : nullptr;
    if (solve_fn) h->mode_parallel |= 64;
    else h->mode_parallel &= ~64;
    return CLIK_OK;
}

static int fill_tick(const DevSkill& S, const double* tterms, TickArgs* tk);

extern "C" int clik_pinv_attach_resident_kernel(clik_pinv* h, void* resident_fn)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    if (resident_fn && !clik::shape_team_ok_rt(h->host.shape) && !clik::shape_quad_front_ok_rt(h->host.shape))
        return fail(CLIK_EUNSUPPORTED, "resident ticks exist for the four-lanes-per-instance kernels only (the config-3 family, "
                                       "single-mode skills with forward kinematics)");
    h->val_resident = (decltype(h->val_resident))resident_fn;
    return CLIK_OK;
}

extern "C" int clik_pinv_resident_waves(const clik_pinv* h, int64_t B)
{
    if (!h || B <= 0) return fail(CLIK_EINVAL, "bad arguments");
    // (the config-3 family runs four waves per 64 instances, the single-mode skills one wave per 16: one slot per wave)
    if (!clik::shape_team_ok_rt(h->host.shape) && clik::shape_quad_front_ok_rt(h->host.shape)) return (int)((B + 15) / 16);
    return clik::team_waves_rt((long long)B);
}

static int resident_run_common(const clik_pinv* h, int64_t B, int32_t n_ticks, const double* tterms, const double* q,
                               const double* y, double* dq, int32_t* mode, clik_ticket* ticket, uint32_t* done,
                               double integrate_dt, double max_speed, double timeout_s, void* stream);

extern "C" int clik_pinv_resident_run(const clik_pinv* h, int64_t B, int32_t n_ticks, const double* tterms,
                                      const double* q, const double* y, double* dq, int32_t* mode,
                                      clik_ticket* ticket, uint32_t* done, double timeout_s, void* stream)
{
    return resident_run_common(h, B, n_ticks, tterms, q, y, dq, mode, ticket, done, 0.0, 0.0, timeout_s, stream);
}

extern "C" int clik_pinv_resident_run_state(const clik_pinv* h, int64_t B, int32_t n_ticks, const double* tterms,
                                            const double* q, const double* y, double* dq, int32_t* mode,
                                            clik_ticket* ticket, uint32_t* done, double integrate_dt, double max_speed,
                                            double timeout_s, void* stream)
{
    if (!(integrate_dt > 0.0)) return fail(CLIK_EINVAL, "integrate_dt must be positive");
    if (max_speed < 0.0) return fail(CLIK_EINVAL, "max_speed must not be negative (0: no clamp)");
    return resident_run_common(h, B, n_ticks, tterms, q, y, dq, mode, ticket, done, integrate_dt, max_speed, timeout_s,
                               stream);
}

static int resident_run_common(const clik_pinv* h, int64_t B, int32_t n_ticks, const double* tterms, const double* q,
                               const double* y, double* dq, int32_t* mode, clik_ticket* ticket, uint32_t* done,
                               double integrate_dt, double max_speed, double timeout_s, void* stream)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (!h->val_resident)
        return fail(CLIK_EUNSUPPORTED, "resident ticks need the value-specialised kernel of the four-lanes-per-instance "
                                       "family attached to this handle (none is)");
    if (B <= 0 || n_ticks <= 0) return fail(CLIK_EINVAL, "B and n_ticks must be positive");
    if (!q || !dq || !ticket || !done) return fail(CLIK_EINVAL, "q, dq, ticket and done must be device pointers");
    const DevSkill& S = h->host;
    if (S.d.n_y > 0 && !y) return fail(CLIK_EINVAL, "skill has input_var: y required");
    if (!(timeout_s > 0.0) || timeout_s > 60.0) return fail(CLIK_EINVAL, "timeout_s must lie in (0, 60]");
    // (every wave of the launch must be resident at once: the launch wrapper of the instantiated kernel checks the grid
    // against that kernel's occupancy on this device and answers hipErrorNotSupported beyond it)
    TickArgs tk;
    int rc = fill_tick(S, tterms, &tk);
    if (rc) return rc;
    // (watchdog budget: polls at a nominal 0.2 us each - a load that hits the L2 plus s_sleep 1; round 6 found the
    // budget of the earlier 2.5 us figure used up in a tenth of timeout_s)
    unsigned long long budget = (unsigned long long)(timeout_s * 5e6);
    if (integrate_dt > 0.0) {
        // the step and the clamp travel in the ticket (stream-ordered in front of the kernel); bit 63 of the budget
        // picks the instantiation that keeps the state
        const double pair[2] = {integrate_dt, max_speed};
        hipError_t ce = hipMemcpyAsync(&ticket->integrate_dt, pair, sizeof(pair), hipMemcpyHostToDevice, (hipStream_t)stream);
        if (ce != hipSuccess) return hipfail(ce, "resident ticks: writing the integration step into the ticket");
        budget |= 1ull << 63;
    }
    hipError_t e = h->val_resident(&tk, (long long)B, q, y, dq, mode, (void*)ticket, (unsigned*)done, n_ticks, budget,
                                   (hipStream_t)stream);
    if (e == hipErrorNotSupported)
        return fail(CLIK_EUNSUPPORTED, "resident ticks: %lld instances need more blocks than this kernel can keep resident "
                                       "at once on this device (every wave must be running for a tick to complete)",
                    (long long)B);
    if (e != hipSuccess) return hipfail(e, "resident kernel launch");
    return CLIK_OK;
}

extern "C" int clik_ticket_feed(clik_ticket* ticket, const uint32_t* done, int32_t n_ticks, int32_t closed_loop,
                                int32_t waves_per_tick, double timeout_s, void* stream)
{
    if (!ticket || !done || n_ticks <= 0 || waves_per_tick <= 0) return fail(CLIK_EINVAL, "bad arguments");
    if (!(timeout_s > 0.0) || timeout_s > 60.0) return fail(CLIK_EINVAL, "timeout_s must lie in (0, 60]");
    hipError_t e = clik::launch_ticket_feed((void*)ticket, (const unsigned*)done, n_ticks, closed_loop, (unsigned)waves_per_tick,
                                            (unsigned long long)(timeout_s * 5e6), (hipStream_t)stream);
    if (e != hipSuccess) return hipfail(e, "ticket feeder launch");
    return CLIK_OK;
}

// the skill image of this handle (what the static kernels read from memory) as 64-bit words, for
// casclik_amd/jit.py to compile into a value-specialised kernel; returns the number of words
extern "C" int clik_pinv_image_words(const clik_pinv* h, uint64_t* buf, int cap)
{
    if (!h || !buf || cap <= 0) return fail(CLIK_EINVAL, "bad arguments");
    std::vector<char> img;
    size_t bytes = 0;
    if (!build_skill_image(h->host, img, &bytes)) return fail(CLIK_EUNSUPPORTED, "skill rows are not contiguous");
    const int words = (int)((bytes + 7) / 8);
    if (words > cap) return fail(CLIK_EINVAL, "image needs %d words, buffer holds %d", words, cap);
    memcpy(buf, img.data(), (size_t)words * 8);
    return words;
}

extern "C" int clik_pinv_create_host(const clik_skill_desc* desc, const clik_pinv_opts* opts, clik_pinv** out)
{
    g_host_only_call = true;
    const int rc = clik_pinv_create(desc, opts, out);
    g_host_only_call = false;
    return rc;
}

extern "C" int clik_qp_create_host(const clik_skill_desc* desc, const clik_qp_opts* opts, clik_qp** out)
{
    g_host_only_call = true;
    const int rc = clik_qp_create(desc, opts, out);
    g_host_only_call = false;
    return rc;
}

extern "C" int clik_pinv_destroy(clik_pinv* h)
{
    if (!h) return CLIK_OK;
    if (h->d_img) (void)hipFree(h->d_img);
    if (h->dev) (void)hipFree(h->dev);
    delete h;
    return CLIK_OK;
}

extern "C" int clik_pinv_n_modes(const clik_pinv* h) { return h ? h->host.n_modes : 0; }
extern "C" const char* clik_pinv_kernel_name(const clik_pinv* h)
{
    if (h && h->jit_solve) return h->jit_name;
    return h ? clik::pinv_kernel_name(h->kernel) : "none";
}

extern "C" const char* clik_pinv_kernel_variant(const clik_pinv* h, int64_t B)
{
    if (!h) return "none";
    if (!h->jit_solve && !(h->kernel >= 0 && clik::pinv_kernel_is_static(h->kernel))) return "dynamic";
    return clik::pinv_static_variant(h->host.shape, h->mode_parallel, (long long)B);
}

static int fill_tick(const DevSkill& S, const double* tterms, TickArgs* tk)
{
    memset(tk, 0, sizeof(*tk));
    const int nts = S.d.n_tslots;
    if (nts > 0) {
        if (!tterms) return fail(CLIK_EINVAL, "skill has %d time slots but tterms is NULL", nts);
        memcpy(tk->tv, tterms, sizeof(double) * 2 * nts);
    }
    return CLIK_OK;
}

static int pinv_solve_common(const clik_pinv* h, int64_t B, const double* tterms, const double* t_inst,
                             const double* q, const double* x, const double* y, double* dq, double* dx,
                             int32_t* mode, void* stream)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (B < 0) return fail(CLIK_EINVAL, "negative batch size");
    if (B == 0) return CLIK_OK;
    const DevSkill& S = h->host;
    if (!q || !dq) return fail(CLIK_EINVAL, "q and dq must be device pointers");
    if (S.d.n_x > 0 && (!x || !dx)) return fail(CLIK_EINVAL, "skill has virtual_var: x and dx required");
    if (S.d.n_y > 0 && !y) return fail(CLIK_EINVAL, "skill has input_var: y required");
    const bool is_static = h->jit_solve || (h->kernel >= 0 && clik::pinv_kernel_is_static(h->kernel));
    if (!is_static && skill_needs_static(S)) return extern_needs_kernel("clik_pinv_solve_batch");
    if (h->kernel < 0 && !h->jit_solve) return no_kernel_for_wide_state(S, "clik_pinv_solve_batch");
    TickArgs tk;
    if (t_inst == nullptr) {
        int rc = fill_tick(S, tterms, &tk);
        if (rc) return rc;
    } else if (!is_static) {
        return fail(CLIK_EUNSUPPORTED, "clik_pinv_solve_batch_t: per-instance time needs a shape-specialised kernel "
                                       "for the skill (none built in, none attached)");
    }
    const clik::LaunchArgs la = {h->dev, h->d_img, &h->warm, S.d.n_q, S.d.n_x, S.d.n_y, h->mode_parallel, nullptr, nullptr, 1,
                                 t_inst};
    // a value-specialised kernel serves the batches the image-reading team kernel would serve (config-3 family), or
    // the small batches of a single-mode skill
    const bool lane_values = clik::shape_value_lane_ok_rt(S.shape) && B <= clik::pinv_value_lane_max_batch();
    const bool team_batch = clik::shape_team_ok_rt(S.shape)
                                ? ((h->mode_parallel & 8) || ((h->mode_parallel & 4) && B <= clik::pinv_team_max_batch()) || lane_values)
                                : lane_values;
    // (the value-specialised library evaluates its own batch limits - it may have been built with other
    // CLIK_VALUE_LANE_* settings than this one: hipErrorNotSupported means "not mine", and the image-reading
    // kernels take the call)
    hipError_t e = hipErrorNotSupported;
    if (h->val_solve && team_batch && t_inst == nullptr)
        e = h->val_solve(&la, &tk, (long long)B, q, y, dq, mode, (hipStream_t)stream);
    if (e == hipErrorNotSupported)
        e = h->jit_solve
                ? h->jit_solve(&la, &tk, (long long)B, q, x, y, dq, dx, mode, (hipStream_t)stream)
                : clik::pinv_launch_solve(h->kernel, la, tk, (long long)B, q, x, y, dq, dx, mode,
                                          (hipStream_t)stream);
    if (e != hipSuccess) return hipfail(e, "pinv_solve_kernel launch");
    return CLIK_OK;
}

extern "C" int clik_pinv_solve_batch(const clik_pinv* h, int64_t B, const double* tterms, const double* q,
                                     const double* x, const double* y, double* dq, double* dx,
                                     int32_t* mode, void* stream)
{
    return pinv_solve_common(h, B, tterms, nullptr, q, x, y, dq, dx, mode, stream);
}

extern "C" int clik_pinv_solve_batch_t(const clik_pinv* h, int64_t B, const double* t_inst, const double* q,
                                       const double* x, const double* y, double* dq, double* dx,
                                       int32_t* mode, void* stream)
{
    if (h && h->host.d.n_tslots > 0 && !t_inst)
        return fail(CLIK_EINVAL, "clik_pinv_solve_batch_t: t_inst (device, [B][2 * n_tslots]) required");
    if (h && h->host.d.n_tslots == 0)       // (nothing depends on time)
        return pinv_solve_common(h, B, nullptr, nullptr, q, x, y, dq, dx, mode, stream);
    return pinv_solve_common(h, B, nullptr, t_inst, q, x, y, dq, dx, mode, stream);
}

// Time-slot records of a rollout (host) -> a device buffer that lives for this call only: allocated, filled and
// released in stream order (hipMallocAsync / hipFreeAsync), so handles stay immutable and two rollouts on two
// streams never share it.  The host array is consumed before the call returns BY CONSTRUCTION: it is copied into a
// pinned staging slot of a process-wide pool here (a plain memcpy), and the asynchronous host-to-device copy reads
// that slot, so nothing depends on how the runtime treats a pageable source and the call never waits for the
// stream.  A slot is reused once the event recorded behind its copy has completed.  (A rollout with time slots is
// therefore not graph-capturable - its records are consumed at call time; a skill without time slots stages
// nothing and captures like a tick.)
namespace {
struct StageSlot { void* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; int dev = -1; };
std::mutex g_stage_mu;
std::vector<StageSlot> g_stage_slots;
}

static int stage_tterms(const double* tterms, size_t count, hipStream_t stream, double** out)
{
    *out = nullptr;
    if (count == 0) return CLIK_OK;
    if (!tterms) return fail(CLIK_EINVAL, "tterms required");
    const size_t bytes = count * sizeof(double);
    int dev = 0;
    (void)hipGetDevice(&dev);
    StageSlot* slot = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        for (auto& sl : g_stage_slots) {
            if (sl.dev != dev || sl.cap < bytes) continue;
            if (sl.busy && hipEventQuery(sl.done) != hipSuccess) continue;
            slot = &sl;
            break;
        }
        if (!slot) {
            StageSlot sl;
            sl.cap = bytes < 4096 ? 4096 : bytes;
            sl.dev = dev;
            hipError_t e = hipHostMalloc(&sl.host, sl.cap, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
            if (e != hipSuccess) {
                if (sl.host) (void)hipHostFree(sl.host);
                return hipfail(e, "pinned staging slot (tterms)");
            }
            g_stage_slots.reserve(64);          // (pointers into the vector are taken under the lock only)
            g_stage_slots.push_back(sl);
            slot = &g_stage_slots.back();
        }
        slot->busy = true;
        std::memcpy(slot->host, tterms, bytes);
        hipError_t e = hipMallocAsync((void**)out, bytes, stream);
        if (e != hipSuccess) { *out = nullptr; slot->busy = false; return hipfail(e, "hipMallocAsync(tterms)"); }
        e = hipMemcpyAsync(*out, slot->host, bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(slot->done, stream);
        if (e != hipSuccess) {
            (void)hipFreeAsync(*out, stream);
            *out = nullptr;
            return hipfail(e, "hipMemcpyAsync(tterms)");
        }
    }
    return CLIK_OK;
}

extern "C" int clik_pinv_rollout_batch(const clik_pinv* hc, int64_t B, int32_t n_ticks, double dt,
                                       double max_speed, const double* tterms, double* q, const double* y,
                                       double* dq, int32_t* mode, void* stream)
{
    return clik_pinv_rollout_batch_x(hc, B, n_ticks, dt, max_speed, tterms, q, nullptr, y, dq, nullptr, mode, stream);
}

extern "C" int clik_pinv_rollout_batch_x(const clik_pinv* hc, int64_t B, int32_t n_ticks, double dt,
                                         double max_speed, const double* tterms, double* q, double* x,
                                         const double* y, double* dq, double* dx, int32_t* mode, void* stream)
{
    return clik_pinv_rollout_batch_m(hc, B, n_ticks, CLIK_INTEGRATE_EULER, dt, max_speed, tterms, q, x, y, dq, dx, mode,
                                     stream);
}

extern "C" int clik_pinv_rollout_batch_m(const clik_pinv* h, int64_t B, int32_t n_ticks, int32_t method, double dt,
                                         double max_speed, const double* tterms, double* q, double* x,
                                         const double* y, double* dq, double* dx, int32_t* mode, void* stream)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (B < 0 || n_ticks < 0) return fail(CLIK_EINVAL, "negative size");
    if (method != CLIK_INTEGRATE_EULER && method != CLIK_INTEGRATE_RK4) return fail(CLIK_EINVAL, "unknown integration method %d", method);
    if (B == 0 || n_ticks == 0) return CLIK_OK;
    const DevSkill& S = h->host;
    const bool has_static = h->jit_rollout || (h->kernel >= 0 && clik::pinv_kernel_is_static(h->kernel));
    if (S.d.n_x > 0) {
        if (!x || !dx) return fail(CLIK_EINVAL, "skill has virtual_var: x and dx required (clik_pinv_rollout_batch_x)");
        if (!has_static)
            return fail(CLIK_EUNSUPPORTED, "the rollout of a skill with virtual_var needs a shape-specialised kernel "
                                           "(none attached for this skill)");
    }
    if (method == CLIK_INTEGRATE_RK4 && !has_static)
        return fail(CLIK_EUNSUPPORTED, "the Runge-Kutta rollout needs a shape-specialised kernel (none attached for this skill)");
    if (!h->jit_rollout && (h->kernel < 0 || !clik::pinv_kernel_is_static(h->kernel)) && skill_needs_static(S))
        return extern_needs_kernel("clik_pinv_rollout_batch");
    if (h->kernel < 0 && !h->jit_rollout) return no_kernel_for_wide_state(S, "clik_pinv_rollout_batch");
    if (!q || !dq) return fail(CLIK_EINVAL, "q and dq must be device pointers");
    if (S.d.n_y > 0 && !y) return fail(CLIK_EINVAL, "skill has input_var: y required");
    const int stages = method == CLIK_INTEGRATE_RK4 ? 4 : 1;
    double* d_tt = nullptr;
    int rc = stage_tterms(tterms, (size_t)n_ticks * stages * 2 * (size_t)S.d.n_tslots, (hipStream_t)stream, &d_tt);
    if (rc) return rc;
    const clik::LaunchArgs la = {h->dev, h->d_img, &h->warm, S.d.n_q, S.d.n_x, S.d.n_y, h->mode_parallel, x, dx, stages, nullptr};
    // the value-specialised rollout serves the batches its per-tick kernel serves (see pinv_solve_common)
    const bool lane_values = clik::shape_value_lane_ok_rt(S.shape) && B <= clik::pinv_value_lane_max_batch();
    const bool team_batch = clik::shape_team_ok_rt(S.shape)
                                ? ((h->mode_parallel & 8) || ((h->mode_parallel & 4) && B <= clik::pinv_team_max_batch()) || lane_values)
                                : lane_values;
    hipError_t e = hipErrorNotSupported;
    if (h->val_rollout && team_batch && S.d.n_x == 0)
        e = h->val_rollout(&la, d_tt, n_ticks, dt, max_speed, (long long)B, q, y, dq, mode, (hipStream_t)stream);
    if (e == hipErrorNotSupported)          // (see pinv_solve_common)
        e = h->jit_rollout
                ? h->jit_rollout(&la, d_tt, n_ticks, dt, max_speed, (long long)B, q, y, dq, mode,
                                 (hipStream_t)stream)
                : clik::pinv_launch_rollout(h->kernel, la, d_tt, n_ticks, dt, max_speed, (long long)B,
                                            q, y, dq, mode, (hipStream_t)stream);
    if (d_tt) (void)hipFreeAsync(d_tt, (hipStream_t)stream);
    if (e != hipSuccess) return hipfail(e, "pinv_rollout_kernel launch");
    return CLIK_OK;
}

// ---------------------------------------------------------------------------- QP
static int qp_upload_image(clik_qp* h)
{
    if (h->d_img) return CLIK_OK;
    std::vector<char> img;
    if (!build_qp_image(h->host, img))
        return fail(CLIK_EUNSUPPORTED, "skill rows are not contiguous: no static kernel possible");
    hipError_t e = hipMalloc(&h->d_img, img.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_img, img.data(), img.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->d_img) (void)hipFree(h->d_img);
        h->d_img = nullptr;
        return hipfail(e, "QP skill image upload");
    }
    return CLIK_OK;
}

extern "C" int clik_qp_create(const clik_skill_desc* desc, const clik_qp_opts* opts, clik_qp** out)
{
    if (!out) return fail(CLIK_EINVAL, "null out pointer");
    *out = nullptr;
    if (!opts) return fail(CLIK_EINVAL, "null options");
    clik_qp* h = new (std::nothrow) clik_qp();
    if (!h) return fail(CLIK_ENOMEM, "out of host memory");
    h->gws = new (std::nothrow) clik::GwsOwner();
    if (!h->gws) { delete h; return fail(CLIK_ENOMEM, "out of host memory"); }
    int rc = validate_and_derive(desc, &h->host);
    if (rc) { delete h; return rc; }
    DevSkill& S = h->host;
    S.qo = *opts;
    if (S.qo.max_iter <= 0) S.qo.max_iter = 4 * (S.n_qp_rows + S.n_qp_vars) + 16;
    for (int k = 0; k < S.n_slack; ++k)
        if (!(opts->weight_shifter + opts->slack_weights[k] > 0.0)) {
            delete h;
            return fail(CLIK_EINVAL, "QP slack weight %d makes the cost non-convex", k);
        }
    if (S.n_qp_vars > CLIK_MAX_QPVARS || S.n_qp_rows > CLIK_MAX_QPROWS) {
        delete h;
        return fail(CLIK_EUNSUPPORTED, "QP with %d variables x %d rows exceeds the device limits (%d x %d)",
                    S.n_qp_vars, S.n_qp_rows, CLIK_MAX_QPVARS, CLIK_MAX_QPROWS);
    }
    for (int j = 0; j < S.n; ++j)
        if (!(opts->weight_shifter * opts->state_weights[j] > 0.0)) {
            delete h;
            return fail(CLIK_EINVAL, "QP cost weight %d is not positive", j);
        }
    {
        int need = S.n;
        for (int ti = 0; ti < S.d.n_tasks; ++ti)
            if (S.d.tasks[ti].m > need) need = S.d.tasks[ti].m;
        h->variant = clik::qp_pick_variant(need, S.n_qp_vars, S.n_qp_rows);
    }
    finish_qp_shape(S);
    {
        // The built-in (dynamic) kernel keeps every row, soft equalities included, in its active set;
        // the shape-specialised kernels eliminate those and fit larger skills.  A skill only they can
        // serve gets a handle without a built-in kernel (variant -1): it solves once a kernel is
        // attached (clik_qp_attach_kernel) or the AOT table has one.
        const bool dyn_rows = h->variant >= 0;
        const bool dyn_lds = dyn_rows && clik::qp_variant_lds(h->variant, S.d.n_y) <= 160u * 1024u;
        if (!dyn_lds) {
            if (!qp_static_eligible(S)) {
                delete h;
                if (!dyn_rows)
                    return fail(CLIK_EUNSUPPORTED,
                                "no QP kernel variant for %d variables x %d rows (device limit: %d rows)",
                                S.n_qp_vars, S.n_qp_rows, CLIK_MAX_QPROWS);
                return fail(CLIK_EUNSUPPORTED, "QP needs more LDS than a CU has (input_var too large)");
            }
            h->variant = -1;
        }
    }
    compute_warm(S, S.lds_slots, h->warm);
    S.zero_token = 0;
    S.lds_slots = 0;
    h->d_img = nullptr;
    h->static_k = -1;
    h->jit_solve = nullptr;
    h->jit_rollout = nullptr;
    h->val_resident = nullptr;
    h->jit_name[0] = 0;
    h->dev = nullptr;
    if (host_only_mode()) {
        *out = h;
        return CLIK_OK;
    }
    hipError_t e = hipMalloc((void**)&h->dev, sizeof(DevSkill));
    if (e != hipSuccess) { delete h; return hipfail(e, "hipMalloc(skill)"); }
    e = hipMemcpy(h->dev, &S, sizeof(DevSkill), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return hipfail(e, "hipMemcpy(skill)"); }
    {
        // shape-specialised kernel from the AOT table (CLIK_FORCE_DYNAMIC=1 / CLIK_NO_AOT=1 skip it)
        const char* force = getenv("CLIK_FORCE_DYNAMIC");
        const char* noaot = getenv("CLIK_NO_AOT");
        const bool allow = !((force && force[0] == '1') || (noaot && noaot[0] == '1'));
        if (allow && qp_static_eligible(S)) {
            const int k = clik::qp_pick_static(S.shape);
            if (k >= 0) {
                int rc2 = qp_upload_image(h);
                if (rc2) { (void)hipFree(h->dev); delete h; return rc2; }
                h->static_k = k;
            }
        }
    }
    *out = h;
    return CLIK_OK;
}

// ShapeDesc initialiser of a QP skill (see clik_shape_describe); returns 1 when a
// shape-specialised QP kernel can serve it, 0 when not, negative on error.
extern "C" int clik_qp_shape_describe(const clik_skill_desc* desc, char* buf, int cap)
{
    if (!buf || cap <= 0) return fail(CLIK_EINVAL, "bad arguments");
    DevSkill* S = new (std::nothrow) DevSkill();
    if (!S) return fail(CLIK_ENOMEM, "out of host memory");
    int rc = validate_and_derive(desc, S);
    if (rc) { delete S; return rc; }
    finish_qp_shape(*S);
    const std::string o = shape_to_string(S->shape);
    const bool eligible = qp_static_eligible(*S);
    delete S;
    if ((int)o.size() + 1 > cap) return fail(CLIK_EINVAL, "buffer too small");
    memcpy(buf, o.c_str(), o.size() + 1);
    return eligible ? 1 : 0;
}

// Attach a shape-specialised QP kernel compiled at run time (casclik_amd/jit.py instantiates
// clik_qp_static.hpp for the ShapeDesc that clik_qp_shape_describe printed).
extern "C" int clik_qp_attach_kernel(clik_qp* h, void* solve_fn, void* rollout_fn, const char* name)
{
    // (rollout_fn may be NULL: clik_qp_rollout_batch then needs an AOT kernel)
    if (!h || !solve_fn) return fail(CLIK_EINVAL, "null argument");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (!qp_static_eligible(h->host))
        return fail(CLIK_EUNSUPPORTED, "skill is outside the shape-specialised QP family");
    int rc = qp_upload_image(h);
    if (rc) return rc;
    h->jit_solve = (clik_jit_qp_fn)solve_fn;
    h->jit_rollout = (clik_jit_qp_rollout_fn)rollout_fn;
    snprintf(h->jit_name, sizeof(h->jit_name), "%s", name ? name : "jit");
    return CLIK_OK;
}

// n_ticks of QP solve -> clamp -> Euler in one launch (SURVEY.md 8(f).1 for the QP controller); the
// working set is carried from tick to tick inside the kernel.  Needs a shape-specialised kernel.
extern "C" int clik_qp_rollout_batch(const clik_qp* hc, int64_t B, int32_t n_ticks, double dt, double max_speed,
                                     const double* tterms, double* q, const double* y, double* dq,
                                     double* slack, int32_t* status, void* stream)
{
    return clik_qp_rollout_batch_x(hc, B, n_ticks, dt, max_speed, tterms, q, nullptr, y, dq, nullptr, slack, status,
                                   stream);
}

extern "C" int clik_qp_rollout_batch_x(const clik_qp* hc, int64_t B, int32_t n_ticks, double dt, double max_speed,
                                       const double* tterms, double* q, double* x, const double* y, double* dq,
                                       double* dx, double* slack, int32_t* status, void* stream)
{
    return clik_qp_rollout_batch_m(hc, B, n_ticks, CLIK_INTEGRATE_EULER, dt, max_speed, tterms, q, x, y, dq, dx, slack,
                                   status, stream);
}

extern "C" int clik_qp_rollout_batch_m(const clik_qp* hc, int64_t B, int32_t n_ticks, int32_t method, double dt,
                                       double max_speed, const double* tterms, double* q, double* x, const double* y,
                                       double* dq, double* dx, double* slack, int32_t* status, void* stream)
{
    const clik_qp* h = hc;
    if (method != CLIK_INTEGRATE_EULER && method != CLIK_INTEGRATE_RK4)
        return fail(CLIK_EINVAL, "clik_qp_rollout_batch_m: unknown integration method %d", method);
    const int stages = method == CLIK_INTEGRATE_RK4 ? 4 : 1;
    if (!h) return fail(CLIK_EINVAL, "null handle");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (B < 0 || n_ticks < 0) return fail(CLIK_EINVAL, "negative size");
    if (B == 0 || n_ticks == 0) return CLIK_OK;
    const DevSkill& S = h->host;
    if (S.d.n_x > 0 && (!x || !dx))
        return fail(CLIK_EINVAL, "skill has virtual_var: x and dx required (clik_qp_rollout_batch_x)");
    if (!h->jit_rollout && h->static_k < 0)
        return fail(CLIK_EUNSUPPORTED, "the QP rollout needs a shape-specialised kernel (none attached for this skill)");
    if (!q || !dq) return fail(CLIK_EINVAL, "q and dq must be device pointers");
    if (S.d.n_y > 0 && !y) return fail(CLIK_EINVAL, "skill has input_var: y required");
    double* d_tt = nullptr;
    int rc = stage_tterms(tterms, (size_t)n_ticks * stages * 2 * (size_t)S.d.n_tslots, (hipStream_t)stream, &d_tt);
    if (rc) return rc;
    hipError_t e = h->val_rollout
                       ? h->val_rollout(d_tt, n_ticks, dt, max_speed, (long long)B, q, y, dq, slack, status, x, dx,
                                        (hipStream_t)stream, stages)
                   : h->jit_rollout
                       ? h->jit_rollout(h->d_img, d_tt, n_ticks, dt, max_speed, (long long)B, q, y, dq, slack,
                                        status, x, dx, (hipStream_t)stream, stages)
                       : clik::qp_launch_rollout_static(h->static_k, h->d_img, d_tt, n_ticks, dt, max_speed,
                                                        (long long)B, q, y, dq, slack, status, x, dx,
                                                        (hipStream_t)stream, stages);
    if (d_tt) (void)hipFreeAsync(d_tt, (hipStream_t)stream);
    if (e != hipSuccess) return hipfail(e, "qp_rollout_kernel launch");
    return CLIK_OK;
}

// the skill image + QP options of this handle (what the shape-specialised QP kernels read from memory, laid out
// as clik::QpImg) as 64-bit words, for casclik_amd/jit.py to compile into a value-specialised kernel
extern "C" int clik_qp_image_words(const clik_qp* h, uint64_t* buf, int cap)
{
    if (!h || !buf || cap <= 0) return fail(CLIK_EINVAL, "bad arguments");
    std::vector<char> img;
    if (!build_qp_image(h->host, img)) return fail(CLIK_EUNSUPPORTED, "skill rows are not contiguous");
    size_t image_bytes = 0;
    {
        std::vector<char> plain;
        if (!build_skill_image(h->host, plain, &image_bytes, sizeof(clik::QpTail))) return fail(CLIK_EUNSUPPORTED, "no image");
    }
    const size_t bytes = ((((image_bytes + 15) & ~(size_t)15) + sizeof(clik::QpTail)) + 15) & ~(size_t)15;
    const int words = (int)(bytes / 8);
    if (words > cap) return fail(CLIK_EINVAL, "image needs %d words, buffer holds %d", words, cap);
    memset(buf, 0, (size_t)words * 8);
    memcpy(buf, img.data(), img.size() < bytes ? img.size() : bytes);
    return words;
}

// 1: after folding the soft equalities every remaining row of the skill's QP is a hard bound on one state (the
// family solved by Gauss-Seidel sweeps + the primal active set, and the one whose value-specialised kernel needs no LDS)
extern "C" int clik_qp_is_box_family(const clik_qp* h)
{
    if (!h) return 0;
    if (!h->jit_solve && h->static_k < 0) return 0;
    return clik::qp_box_family_rt(h->host.shape) ? 1 : 0;
}

extern "C" int clik_qp_attach_value_kernel(clik_qp* h, void* solve_fn, void* rollout_fn)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    if (solve_fn && !h->jit_solve && h->static_k < 0)
        return fail(CLIK_EUNSUPPORTED, "value-specialised QP kernels exist for skills a shape-specialised kernel serves");
    h->val_solve = (clik_jit_qp_value_fn)solve_fn;
    // (the value-specialised rollout exists for the box family only)
    h->val_rollout = (solve_fn && clik::qp_box_family_rt(h->host.shape)) ? (clik_jit_qp_value_rollout_fn)rollout_fn : nullptr;
    return CLIK_OK;
}

extern "C" int clik_qp_attach_resident_kernel(clik_qp* h, void* resident_fn)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    if (resident_fn && !(clik::qp_box_family_rt(h->host.shape) && h->host.d.n_x == 0 && h->host.shape.uses_fk != 0 &&
                         h->host.n >= 3 && h->host.n <= 8))
        return fail(CLIK_EUNSUPPORTED, "resident QP ticks exist for bound-constrained skills with forward kinematics and "
                                       "without virtual variables");
    h->val_resident = (decltype(h->val_resident))resident_fn;
    return CLIK_OK;
}

extern "C" int clik_qp_resident_waves(const clik_qp* h, int64_t B)
{
    if (!h || B <= 0) return fail(CLIK_EINVAL, "bad arguments");
    return (int)((B + 15) / 16);         // (four lanes per instance: one wave per 16 instances, one `done` slot per wave)
}

extern "C" int clik_qp_resident_run(const clik_qp* h, int64_t B, int32_t n_ticks, const double* tterms, const double* q,
                                    const double* y, double* dq, double* slack, int32_t* status, clik_ticket* ticket,
                                    uint32_t* done, double timeout_s, void* stream)
{
    if (!h) return fail(CLIK_EINVAL, "null handle");
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (!h->val_resident)
        return fail(CLIK_EUNSUPPORTED, "resident QP ticks need the value-specialised kernel of a bound-constrained skill "
                                       "attached to this handle (none is)");
    if (B <= 0 || n_ticks <= 0) return fail(CLIK_EINVAL, "B and n_ticks must be positive");
    if (!q || !dq || !ticket || !done) return fail(CLIK_EINVAL, "q, dq, ticket and done must be device pointers");
    const DevSkill& S = h->host;
    if (S.d.n_y > 0 && !y) return fail(CLIK_EINVAL, "skill has input_var: y required");
    if (!(timeout_s > 0.0) || timeout_s > 60.0) return fail(CLIK_EINVAL, "timeout_s must lie in (0, 60]");
    TickArgs tk;
    int rc = fill_tick(S, tterms, &tk);
    if (rc) return rc;
    const unsigned long long budget = (unsigned long long)(timeout_s * 5e6);       // (polls, at a nominal 0.2 us each)
    hipError_t e = h->val_resident(&tk, (long long)B, q, y, dq, slack, status, (void*)ticket, (unsigned*)done, n_ticks,
                                   budget, (hipStream_t)stream);
    if (e == hipErrorNotSupported)
        return fail(CLIK_EUNSUPPORTED, "resident QP ticks: %lld instances need more waves than this kernel can keep resident "
                                       "at once on this device", (long long)B);
    if (e != hipSuccess) return hipfail(e, "resident QP kernel launch");
    return CLIK_OK;
}

extern "C" const char* clik_qp_kernel_name(const clik_qp* h)
{
    if (!h) return "none";
    if (h->jit_solve) return h->jit_name;
    if (h->static_k >= 0) return clik::qp_static_name(h->static_k);
    return h->variant >= 0 ? "dynamic" : "none";
}

extern "C" int clik_qp_destroy(clik_qp* h)
{
    if (!h) return CLIK_OK;
    if (h->d_img) (void)hipFree(h->d_img);
    if (h->dev) (void)hipFree(h->dev);
    delete h;           // (~clik_qp releases the work area of the global-workspace kernels)
    return CLIK_OK;
}

extern "C" int64_t clik_qp_workspace_bytes(const clik_qp* h) { return (h && h->gws) ? (int64_t)h->gws->footprint() : 0; }

extern "C" int clik_qp_n_vars(const clik_qp* h) { return h ? h->host.n_qp_vars : 0; }
extern "C" int clik_qp_n_rows(const clik_qp* h) { return h ? h->host.n_qp_rows : 0; }

static int qp_check_args(const clik_qp* h, int64_t B, const double* q, const double* x, const double* y)
{
    CLIK_NEEDS_DEVICE_HANDLE(h);
    if (!h) return fail(CLIK_EINVAL, "null handle");
    if (B < 0) return fail(CLIK_EINVAL, "negative batch size");
    const DevSkill& S = h->host;
    if (B > 0 && !q) return fail(CLIK_EINVAL, "q must be a device pointer");
    // (an empty batch has no rows to point at: a zero-row device tensor's data pointer is null)
    if (B > 0 && S.d.n_x > 0 && !x) return fail(CLIK_EINVAL, "skill has virtual_var: x required");
    if (B > 0 && S.d.n_y > 0 && !y) return fail(CLIK_EINVAL, "skill has input_var: y required");
    return CLIK_OK;
}

extern "C" int clik_qp_solve_batch(const clik_qp* h, int64_t B, const double* tterms, const double* q,
                                   const double* x, const double* y, double* dq, double* dx, double* slack,
                                   int32_t* status, void* stream)
{
    return clik_qp_solve_batch_hot(h, B, tterms, q, x, y, dq, dx, slack, status, nullptr, 0, stream);
}

static int qp_solve_common(const clik_qp* h, int64_t B, const double* tterms, const double* t_inst, const double* q,
                           const double* x, const double* y, double* dq, double* dx, double* slack,
                           int32_t* status, int32_t* hot_set, int32_t use_hot, void* stream)
{
    int rc = qp_check_args(h, B, q, x, y);
    if (rc) return rc;
    if (B == 0) return CLIK_OK;
    if (!dq) return fail(CLIK_EINVAL, "dq must be a device pointer");
    // (a constraint with more than eight rows is fine for the dynamic QP kernels twelve rows wide - the variant picked at
    // create time covers the widest constraint; only generated constraint code needs an instantiated kernel)
    if (!h->jit_solve && h->static_k < 0 &&
        (skill_has_extern(h->host) || (skill_has_wide_task(h->host) && (h->variant < 0 || clik::qp_variant_width(h->variant) <= CLIK_DYN_MAX_M))))
        return extern_needs_kernel("clik_qp_solve_batch");
    TickArgs tk;
    if (t_inst == nullptr) {
        rc = fill_tick(h->host, tterms, &tk);
        if (rc) return rc;
    } else if (!h->jit_solve && h->static_k < 0) {
        return fail(CLIK_EUNSUPPORTED, "clik_qp_solve_batch_t: per-instance time needs a shape-specialised kernel "
                                       "for the skill (none built in, none attached)");
    }
    hipError_t e;
    if (h->val_solve && t_inst == nullptr)
        e = h->val_solve(&tk, (long long)B, q, x, y, dq, dx, slack, status, hot_set, use_hot, (hipStream_t)stream);
    else if (h->jit_solve)
        e = h->jit_solve(h->d_img, &tk, (long long)B, q, x, y, dq, dx, slack, status, hot_set, use_hot,
                         (hipStream_t)stream, t_inst);
    else if (h->static_k >= 0)
        e = clik::qp_launch_static(h->static_k, h->d_img, tk, (long long)B, q, x, y, dq, dx, slack, status, hot_set,
                                   use_hot, (hipStream_t)stream, t_inst);
    else if (h->variant < 0)
        return fail(CLIK_EUNSUPPORTED, "clik_qp_solve_batch: this skill needs a shape-specialised kernel and none "
                                       "is attached (casclik_amd.jit needs hipcc)");
    else
    {
        if (clik::qp_variant_uses_workspace(h->variant) && (hipStream_t)stream == hipStreamPerThread)
            return fail(CLIK_EUNSUPPORTED, "clik_qp_solve_batch: this skill's kernel keeps its work area in global memory, "
                                           "owned by the handle and ordered per stream - hipStreamPerThread is not supported "
                                           "for it (pass an explicit stream)");
        e = clik::qp_launch_solve(h->variant, h->dev, h->warm, tk, (long long)B, h->host.d.n_y, q, x, y, dq, dx,
                                  slack, status, (hipStream_t)stream, h->gws);
        if (e == hipErrorStreamCaptureUnsupported)
            return fail(CLIK_EHIP, "qp_solve_kernel launch: the work area of this skill's kernel must grow for this batch "
                                   "size, which cannot be captured - run one tick of this batch size before capturing");
    }
    if (e != hipSuccess) return hipfail(e, "qp_solve_kernel launch");
    return CLIK_OK;
}

extern "C" int clik_qp_solve_batch_hot(const clik_qp* h, int64_t B, const double* tterms, const double* q,
                                       const double* x, const double* y, double* dq, double* dx, double* slack,
                                       int32_t* status, int32_t* hot_set, int32_t use_hot, void* stream)
{
    return qp_solve_common(h, B, tterms, nullptr, q, x, y, dq, dx, slack, status, hot_set, use_hot, stream);
}

extern "C" int clik_qp_solve_batch_t(const clik_qp* h, int64_t B, const double* t_inst, const double* q,
                                     const double* x, const double* y, double* dq, double* dx, double* slack,
                                     int32_t* status, int32_t* hot_set, int32_t use_hot, void* stream)
{
    if (h && h->host.d.n_tslots > 0 && !t_inst)
        return fail(CLIK_EINVAL, "clik_qp_solve_batch_t: t_inst (device, [B][2 * n_tslots]) required");
    if (h && h->host.d.n_tslots == 0)
        return qp_solve_common(h, B, nullptr, nullptr, q, x, y, dq, dx, slack, status, hot_set, use_hot, stream);
    return qp_solve_common(h, B, nullptr, t_inst, q, x, y, dq, dx, slack, status, hot_set, use_hot, stream);
}

extern "C" int clik_qp_data_batch(const clik_qp* h, int64_t B, const double* tterms, const double* q,
                                  const double* x, const double* y, double* Hdiag, double* A, double* lbA,
                                  double* ubA, void* stream)
{
    int rc = qp_check_args(h, B, q, x, y);
    if (rc) return rc;
    if (B == 0) return CLIK_OK;
    if (!Hdiag || !A || !lbA || !ubA) return fail(CLIK_EINVAL, "output pointers required");
    if (skill_has_extern(h->host) ||
        (skill_has_wide_task(h->host) && (h->variant < 0 || clik::qp_variant_width(h->variant) <= CLIK_DYN_MAX_M)))
        return fail(CLIK_EUNSUPPORTED, "clik_qp_data_batch: not available for skills that only the "
                                       "shape-specialised kernels serve (generated rows, > 8 rows per constraint)");
    if (h->variant < 0)
        return fail(CLIK_EUNSUPPORTED, "clik_qp_data_batch: the skill exceeds the built-in kernel's limits");
    TickArgs tk;
    rc = fill_tick(h->host, tterms, &tk);
    if (rc) return rc;
    if (clik::qp_variant_uses_workspace(h->variant) && (hipStream_t)stream == hipStreamPerThread)
        return fail(CLIK_EUNSUPPORTED, "clik_qp_data_batch: hipStreamPerThread is not supported for skills whose kernel keeps its "
                                       "work area in global memory (pass an explicit stream)");
    hipError_t e = clik::qp_launch_data(h->variant, h->dev, h->warm, tk, (long long)B, h->host.d.n_y, q, x, y, Hdiag, A,
                                        lbA, ubA, (hipStream_t)stream, h->gws);
    if (e != hipSuccess) return hipfail(e, "qp_data_kernel launch");
    return CLIK_OK;
}
