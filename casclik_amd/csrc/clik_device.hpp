// Device-side building blocks of the batched CLIK hot path (gfx950 / wave64).
//
// Execution model: ONE ROBOT INSTANCE PER LANE, 64 instances per wavefront.
// Everything an instance needs per tick (tool frame, geometric Jacobian, the
// current task Jacobian, Gram factors) lives in that lane's VGPRs with static
// indexing; the skill descriptor is wave-uniform and is read through scalar
// loads; LDS holds (a) the coalesced-load staging of the joint state / inputs
// (transposed so lane l owns instance l) and (b) the dynamically indexed,
// cold part of the per-instance state (stacked Jacobian rows, Gram matrices).
// LDS slots are laid out [slot][lane] so every access is bank-conflict free.
//
// These are tiny fp64 solves (<= 8x8): no MFMA.  See DESIGN.md for why the
// lane-per-instance layout is used instead of wave-per-instance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "clik.h"

// Source annotation only (expands to nothing): phase boundaries of a tick.  tools/phase_budget.py compiles a kernel with
// -g, reads the inlining chain the listing gives for every instruction (`.loc` comments) and books the instruction
// on the last CLIK_PHASE mark above the innermost frame that lies between a function's first mark and its
// CLIK_PHASE_END - the shipped object itself is what gets counted, no fences, no marker instructions.
#define CLIK_PHASE(name)
#define CLIK_PHASE_END()

namespace clik {

constexpr int WAVE = 64;

// Compile-time "shape" of a skill: everything that steers control flow or
// register indexing.  A kernel instantiated for a ShapeDesc has no size guards
// left; numeric values (gains, bounds, chain constants, row coefficients)
// still come from the descriptor.  The dynamic fallback reads the same fields
// from DevSkill at run time.
constexpr int SHAPE_MAX_TASKS = 8;
constexpr int SHAPE_MAX_ROWS = SHAPE_MAX_TASKS * CLIK_MAX_M;
struct ShapeDesc {
    int n;                          // n_q + n_x
    int n_tasks;
    int cls[SHAPE_MAX_TASKS];
    int m[SHAPE_MAX_TASKS];
    int flags[SHAPE_MAX_TASKS];     // OR of the CLIK_ROW_HAS_* of the task's rows
    int const_j[SHAPE_MAX_TASKS];   // Jacobian independent of the state (rows use HAS_Q only)
    int all_affine;                 // no NORM2 output rows
    int uses_fk, quat_src;
    int feedforward, multidim, conv_last, standard;
    // chain structure (static shapes are per robot structure x skill structure)
    int nj;                         // chain joints, fixed ones included
    int jtype[CLIK_MAX_JOINTS];     // CLIK_JOINT_*
    int jq[CLIK_MAX_JOINTS];        // state index driven by the joint (-1 fixed)
    int jflags[CLIK_MAX_JOINTS];    // DevSkill::jflags encoding
    int gain_matrix[SHAPE_MAX_TASKS];
    int ny_terms[SHAPE_MAX_TASKS];  // max input_var terms of a row of the task
    int n_y;                        // input_var width
    int has_t[SHAPE_MAX_TASKS];     // some row of the task carries a time slot
    // joint-space tasks (joint limits, joint centering): every row of the constant Jacobian is a unit
    // vector e_col with distinct columns.  ucol[ti][i] = col + 1 of row i, 0 when the task is not of
    // that form (then all of its entries are 0).
    int ucol[SHAPE_MAX_TASKS][CLIK_MAX_M];
    // ReactiveQPController shapes (0 in pseudo-inverse shapes)
    int qp;                         // 1: shape of a QP controller (the pinv option fields are 0)
    int soft[SHAPE_MAX_TASKS];      // constraint_type "soft": the rows carry slack variables
    // sparsity pattern of the tool-frame feature coefficients [b0..b2 | g0..g8 | h0..h2] of every row
    // (rows in task order): bit k set = coefficient k is non-zero / is exactly 1.0
    unsigned row_nz[SHAPE_MAX_ROWS];
    unsigned row_one[SHAPE_MAX_ROWS];
    // output i of task ti: 0 = one affine row; k >= 1 = 2-norm of k consecutive affine rows
    // (cs.norm_2 / cs.norm_fro constraint expressions)
    int out_nrows[SHAPE_MAX_TASKS][CLIK_MAX_M];
    int n_x;                        // virtual variables: state = [robot_var (n - n_x); virtual_var (n_x)]
    // 1: e, J, d e/d t of the task come from ExternTask<ti> (code generated from the caller's expression
    // graph, CLIK_OUT_EXTERN); only run-time instantiated kernels can carry such a task
    int ext[SHAPE_MAX_TASKS];
};

constexpr bool shape_unit(const ShapeDesc& sd, int ti) { return sd.const_j[ti] != 0 && sd.m[ti] > 0 && sd.ucol[ti][0] > 0; }
// row of unit task ti whose 1 sits in column col, or -1
constexpr int shape_unit_row(const ShapeDesc& sd, int ti, int col)
{
    for (int i = 0; i < sd.m[ti]; ++i)
        if (sd.ucol[ti][i] == col + 1) return i;
    return -1;
}

// affine rows behind output i of task ti, and the row layout of static shapes (rows are
// stored in task order, output order, group order)
constexpr int shape_out_rows(const ShapeDesc& sd, int ti, int i) { return sd.out_nrows[ti][i] > 0 ? sd.out_nrows[ti][i] : 1; }
constexpr int shape_task_rows(const ShapeDesc& sd, int ti)
{
    int r = 0;
    for (int i = 0; i < sd.m[ti]; ++i) r += shape_out_rows(sd, ti, i);
    return r;
}
constexpr int shape_rows(const ShapeDesc& sd)
{
    int r = 0;
    for (int i = 0; i < sd.n_tasks; ++i) r += shape_task_rows(sd, i);
    return r;
}
constexpr int shape_row_base(const ShapeDesc& sd, int ti)
{
    int r = 0;
    for (int i = 0; i < ti; ++i) r += shape_task_rows(sd, i);
    return r;
}
constexpr int shape_out_row0(const ShapeDesc& sd, int ti, int i)
{
    int r = shape_row_base(sd, ti);
    for (int k = 0; k < i; ++k) r += shape_out_rows(sd, ti, k);
    return r;
}

// Compact image of the numeric content of a skill, copied to LDS at kernel
// start by shape-specialised kernels (one coalesced pass) and read from there
// with static offsets: in-order, pipelined ds_reads into VGPRs instead of
// out-of-order scalar loads that serialise a lone wavefront.  The host builds
// the same layout in clik_api.hip (build_skill_image).
template <int NJ, int NT, int NR>
struct SkillImage {
    clik_joint joints[NJ > 0 ? NJ : 1];
    clik_task  tasks[NT];
    clik_row   rows[NR];
    double     cpinv[NT][CLIK_MAX_DOF * CLIK_MAX_M];
    double     cjtj[NT][CLIK_MAX_DOF * (CLIK_MAX_DOF + 1) / 2];   // J^T J of constant-Jacobian tasks (packed, stride n)
    double     lam;
    double     quat[4];
    int32_t    quat_yi[4];
    int32_t    n_tslots;
    int32_t    pad;
};

// options of the QP controller, appended to the skill image of QP shapes at the next
// 16-byte boundary (host: build_qp_image in clik_api.hip, device: clik_qp_static.hpp)
struct QpTail {
    double  mu;                                   // weight_shifter
    double  state_w[CLIK_MAX_DOF];
    double  slack_w[CLIK_MAX_QPROWS];
    int32_t max_iter;
    int32_t pad;
};

inline bool shape_equal(const ShapeDesc& a, const ShapeDesc& b)
{
    if (a.n != b.n || a.n_tasks != b.n_tasks || a.all_affine != b.all_affine || a.uses_fk != b.uses_fk ||
        a.quat_src != b.quat_src || a.feedforward != b.feedforward || a.multidim != b.multidim ||
        a.conv_last != b.conv_last || a.standard != b.standard || a.nj != b.nj || a.n_y != b.n_y || a.qp != b.qp ||
        a.n_x != b.n_x)
        return false;
    for (int i = 0; i < a.n_tasks; ++i)
        if (a.cls[i] != b.cls[i] || a.m[i] != b.m[i] || a.flags[i] != b.flags[i] || a.const_j[i] != b.const_j[i] ||
            a.gain_matrix[i] != b.gain_matrix[i] || a.ny_terms[i] != b.ny_terms[i] || a.has_t[i] != b.has_t[i] ||
            a.soft[i] != b.soft[i] || a.ext[i] != b.ext[i])
            return false;
    for (int i = 0; i < a.n_tasks; ++i)
        for (int k = 0; k < CLIK_MAX_M; ++k)
            if (a.ucol[i][k] != b.ucol[i][k]) return false;
    for (int j = 0; j < a.nj; ++j)
        if (a.jtype[j] != b.jtype[j] || a.jq[j] != b.jq[j] || a.jflags[j] != b.jflags[j]) return false;
    for (int r = 0; r < SHAPE_MAX_ROWS; ++r)
        if (a.row_nz[r] != b.row_nz[r] || a.row_one[r] != b.row_one[r]) return false;
    for (int i = 0; i < a.n_tasks; ++i)
        for (int k = 0; k < CLIK_MAX_M; ++k)
            if (a.out_nrows[i][k] != b.out_nrows[i][k]) return false;
    return true;
}

// Device-resident skill: descriptor + controller options + derived tables.
struct DevSkill {
    clik_skill_desc d;
    ShapeDesc shape;
    int32_t  task_flags[CLIK_MAX_TASKS];   // OR of row flags per task
    int32_t  task_const_j[CLIK_MAX_TASKS]; // 1: rows depend on the state through a.z only
    // for constant-Jacobian tasks: P = pinv(J) (n x m, row-major ld CLIK_MAX_M),
    // computed once on the host with the controller's pinv options
    double   cpinv[CLIK_MAX_TASKS][CLIK_MAX_DOF * CLIK_MAX_M];
    clik_pinv_opts  po;
    clik_qp_opts    qo;
    int32_t  n;                       // n_q + n_x
    int32_t  n_sets;
    int32_t  n_modes;
    int32_t  last_set_converges;      // last task is a SetConstraint && converge_final_set_to_max
    uint32_t act[1 << CLIK_MAX_SETS]; // activation bit masks in mode order (pseudo_inverse.py:107-130)
    // per chain joint: bit0 origin rotation is identity, bit1 origin translation
    // is zero, bits 4..6 joint axis code (0 generic, 1/2/3 = +x/+y/+z, 5/6/7 = -x/-y/-z)
    int32_t  jflags[CLIK_MAX_JOINTS];
    uint32_t used_mask;               // state indices driven by a chain joint
    uint32_t rev_mask;                // ... that are revolute
    int32_t  n_slack;
    int32_t  n_qp_rows;
    int32_t  n_qp_vars;
    int32_t  lds_slots;               // doubles per lane of dynamic LDS
    int32_t  zero_token;              // always 0 (see warm_descriptor)
};

// ---- shape policies ---------------------------------------------------------
struct DynShape {
    static constexpr bool is_static = false;
    __device__ static int n(const DevSkill* S) { return S->n; }
    __device__ static int n_tasks(const DevSkill* S) { return S->d.n_tasks; }
    __device__ static int cls(const DevSkill* S, int ti) { return S->d.tasks[ti].cls; }
    __device__ static int m(const DevSkill* S, int ti) { return S->d.tasks[ti].m; }
    __device__ static int flags(const DevSkill* S, int ti) { return S->task_flags[ti]; }
    __device__ static bool const_j(const DevSkill* S, int ti) { return S->task_const_j[ti] != 0; }
    __device__ static bool all_affine(const DevSkill* S) { return S->shape.all_affine != 0; }
    __device__ static bool uses_fk(const DevSkill* S) { return S->d.uses_fk != 0; }
    __device__ static int quat_src(const DevSkill* S) { return S->d.quat_src; }
    __device__ static bool feedforward(const DevSkill* S) { return S->po.feedforward != 0; }
    __device__ static bool multidim(const DevSkill* S) { return S->po.multidim_sets != 0; }
    __device__ static bool conv_last(const DevSkill* S) { return S->po.converge_final_set_to_max != 0; }
    __device__ static bool standard(const DevSkill* S) { return S->po.pinv_method == CLIK_PINV_STANDARD; }
};

template <const ShapeDesc& SD>
struct StaticShape {
    static constexpr bool is_static = true;
    static constexpr const ShapeDesc& desc = SD;
    __device__ static constexpr int n(const DevSkill*) { return SD.n; }
    __device__ static constexpr int n_tasks(const DevSkill*) { return SD.n_tasks; }
    __device__ static constexpr int cls(const DevSkill*, int ti) { return SD.cls[ti]; }
    __device__ static constexpr int m(const DevSkill*, int ti) { return SD.m[ti]; }
    __device__ static constexpr int flags(const DevSkill*, int ti) { return SD.flags[ti]; }
    __device__ static constexpr bool const_j(const DevSkill*, int ti) { return SD.const_j[ti] != 0; }
    __device__ static constexpr bool all_affine(const DevSkill*) { return SD.all_affine != 0; }
    __device__ static constexpr bool uses_fk(const DevSkill*) { return SD.uses_fk != 0; }
    __device__ static constexpr int quat_src(const DevSkill*) { return SD.quat_src; }
    __device__ static constexpr bool feedforward(const DevSkill*) { return SD.feedforward != 0; }
    __device__ static constexpr bool multidim(const DevSkill*) { return SD.multidim != 0; }
    __device__ static constexpr bool conv_last(const DevSkill*) { return SD.conv_last != 0; }
    __device__ static constexpr bool standard(const DevSkill*) { return SD.standard != 0; }
};

// With one wavefront per CU per launch every scalar load of the descriptor is
// a cold miss (~0.3-1 us each, serialised by the data-dependent walk over tasks
// and rows).  warm_descriptor() touches every 64-B line of the used descriptor
// ranges with independent s_load_dword's and waits ONCE, so the walk afterwards
// hits the scalar cache.
//
// It is one non-volatile asm without a memory clobber (a volatile or clobbering
// asm would make the compiler give up scalar loads for the whole kernel).  Its
// output is the always-zero `zero_token` of the skill; adding that to the skill
// pointer creates the data dependence that keeps the warm-up ahead of every
// descriptor access without changing any address.
constexpr int WARM_RANGES = 5;
struct WarmArgs {
    int32_t off[WARM_RANGES];     // byte offsets into DevSkill, multiples of 64
    int32_t end[WARM_RANGES];     // exclusive ends (off >= end: empty)
    int32_t token_off;            // offset of DevSkill::zero_token
};

__device__ __forceinline__ const struct DevSkill* warm_descriptor(const struct DevSkill* S, const WarmArgs& wa)
{
    int tok;
#define CLIK_WARM_RANGE(K)                                   \
    "s_mov_b32 s96, %[o" #K "]\n"                            \
    "1:\n"                                                   \
    "s_cmp_ge_u32 s96, %[e" #K "]\n"                         \
    "s_cbranch_scc1 2f\n"                                    \
    "s_load_dword s97, %[base], s96\n"                       \
    "s_add_u32 s96, s96, 64\n"                               \
    "s_branch 1b\n"                                          \
    "2:\n"
    asm(CLIK_WARM_RANGE(0) CLIK_WARM_RANGE(1) CLIK_WARM_RANGE(2) CLIK_WARM_RANGE(3) CLIK_WARM_RANGE(4)
        "s_load_dword %[tok], %[base], %[toff]\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [tok] "=s"(tok)
        : [base] "s"(S), [toff] "s"(wa.token_off),
          [o0] "s"(wa.off[0]), [e0] "s"(wa.end[0]), [o1] "s"(wa.off[1]), [e1] "s"(wa.end[1]),
          [o2] "s"(wa.off[2]), [e2] "s"(wa.end[2]), [o3] "s"(wa.off[3]), [e3] "s"(wa.end[3]),
          [o4] "s"(wa.off[4]), [e4] "s"(wa.end[4])
        : "s96", "s97", "scc");
#undef CLIK_WARM_RANGE
    return (const struct DevSkill*)((const char*)S + tok);
}

// per-tick values / time derivatives of the time-only sub-expressions
struct TickArgs {
    double tv[2 * CLIK_MAX_TSLOTS];
};

__device__ __forceinline__ int tri(int i, int k) { return i * (i + 1) / 2 + k; }  // k <= i

// uniform (wave-invariant) value -> SGPR
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------------------
// LDL^T of a packed symmetric positive definite matrix of runtime size r <= N.
// In place: strictly lower part <- unit lower factor, diagonal <- d_k;
// rd[k] = 1/d_k.  Loops are fully unrolled with wave-uniform guards so every
// index is static (registers, not scratch).
template <int N>
__device__ __forceinline__ void ldl_factor(double (&A)[N * (N + 1) / 2], double (&rd)[N], int r)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < r) {
            double t[N];
            double d = A[tri(k, k)];
#pragma unroll
            for (int j = 0; j < k; ++j) {
                t[j] = A[tri(k, j)] * A[tri(j, j)];
                d = fma(-A[tri(k, j)], t[j], d);
            }
            A[tri(k, k)] = d;
            const double inv = 1.0 / d;
            rd[k] = inv;
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                if (i < r) {
                    double s = A[tri(i, k)];
#pragma unroll
                    for (int j = 0; j < k; ++j) s = fma(-A[tri(i, j)], t[j], s);
                    A[tri(i, k)] = s * inv;
                }
            }
        }
    }
}

// x <- A^{-1} x with the factor above.
template <int N>
__device__ __forceinline__ void ldl_solve(const double (&A)[N * (N + 1) / 2], const double (&rd)[N],
                                          double (&x)[N], int r)
{
#pragma unroll
    for (int i = 1; i < N; ++i) {
        if (i < r) {
#pragma unroll
            for (int j = 0; j < i; ++j) x[i] = fma(-A[tri(i, j)], x[j], x[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < r) x[i] *= rd[i];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) {
        if (i < r) {
#pragma unroll
            for (int j = i + 1; j < N; ++j)
                if (j < r) x[i] = fma(-A[tri(j, i)], x[j], x[i]);
        }
    }
}

// ---------------------------------------------------------------------------
// sin and cos of a joint angle.  Joint angles are O(1): a three-term FMA
// Cody-Waite reduction by pi/2 (exact to ~1e-33 relative to k) plus the fdlibm
// kernel polynomials on |r| <= pi/4 gives <= 1 ulp in ~35 VALU instructions,
// against ~150 for the generic library routine with its huge-argument path.
// Arguments beyond 1e5 rad take the library routine.
// out-of-line library path for huge arguments; results are RETURNED (an out-pointer
// would force the caller's sin/cos variables into scratch memory on the fast path too)
struct SinCos {
    double s, c;
};
__device__ __attribute__((noinline)) SinCos sincos_slow(const double x)
{
    SinCos r;
    sincos(x, &r.s, &r.c);
    return r;
}

// 1/d to <= 1 ulp: hardware estimate + two Newton steps (5 instructions; the
// IEEE division sequence is ~12).  d is a Gram / LDL pivot: positive, finite.
__device__ __forceinline__ double recip(const double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// "This value has ARRIVED": an empty statement that reads a load's destination register, so the compiler places the
// load's wait HERE.  A wave's memory operations complete in order and the compiler waits by count, so where a loaded value
// is first read decides what ELSE that wait covers: read at the top of a resident kernel's next tick, a row requested a
// whole tick earlier still cost a wait for the stores issued just before the back edge (round 6, from the listings).
__device__ __forceinline__ void pin_arrived(double& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_arrived(unsigned& x) { asm volatile("" : "+v"(x)); }

// x, or a quiet NaN where `bad_hi` is 0x7ff80000 (0: x as it is).  As BITS: device code is compiled with
// -fno-honor-nans (build.py DEVICE_FP), under which arithmetic with a NaN constant is undefined and "x + NaN" was folded
// to x - the QP kernels mark an infeasible instance's outputs this way.
__device__ __forceinline__ double nan_or(const double x, const unsigned bad_hi)
{
    return __hiloint2double((int)((unsigned)__double2hiint(x) | bad_hi), __double2loint(x));
}

// sign(x) / 2 as a double: -0.5, 0 or +0.5 (three instructions: the sign bit onto the pattern of 0.5, zero test)
__device__ __forceinline__ double half_sign(const double x)
{
    // x 2^3000 is +-inf for every x != 0 (denormals included) and keeps a zero: clamped to +-1/2 (both inline constants)
    return fmax(fmin(ldexp(x, 3000), 0.5), -0.5);
}

// argument range of the fast path (3-term Cody-Waite reduction stays exact)
constexpr double kSinCosFastMax = 1.0e5;

// sin/cos for |x| <= kSinCosFastMax: straight-line, no call.  Callers that evaluate several
// angles use this for all of them and handle larger arguments afterwards in ONE cold block
// (sincos_slow): a call between two evaluations makes the compiler re-materialise the ~26
// literal polynomial constants after every call site (measured: ~200 extra moves per tick).
// The 17 full-width constants of sincos_fast live in constant memory: an fp64 instruction takes a 32-bit literal only (the
// HIGH word of the double), so every other constant costs the wave two s_mov_b32 - 34 scalar instructions per tick, each
// a full issue slot of a lone wave - while scalar loads fetch eight doubles per instruction, at kernel start, beside the
// wave's row loads.  (__constant__ without const: the compiler must not fold the loads back into literals.)
#ifndef CLIK_SINCOS_POOL
#define CLIK_SINCOS_POOL 1
#endif
inline __constant__ double kSinCosPool[20] = {
    0.6366197723675814, -1.5707963267948966, -6.123233995736766e-17, 1.4973849048591698e-33,
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
    -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
    2.08757232129817482790e-09, -1.13596475577881948265e-11, 0.0, 0.0, 0.0, 0.0};

// the constants as a value: a kernel that wants their scalar loads issued at a place of its choosing - before its row
// loads, before its tick loop - takes them there (sincos_consts) and hands them to the evaluations
struct SinCosK {
    double c[16];
};
__device__ __forceinline__ SinCosK sincos_consts()
{
    SinCosK k;
#if CLIK_SINCOS_POOL
#pragma unroll
    for (int i = 0; i < 16; ++i) k.c[i] = kSinCosPool[i];
#else
    // (-DCLIK_SINCOS_POOL=0: the same constants as literals - two s_mov_b32 each, no scalar loads at kernel start)
    constexpr double L[16] = {
        0.6366197723675814, -1.5707963267948966, -6.123233995736766e-17, 1.4973849048591698e-33,
        -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
        -2.50507602534068634195e-08, 1.58969099521155010221e-10,
        4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
        2.08757232129817482790e-09, -1.13596475577881948265e-11};
#pragma unroll
    for (int i = 0; i < 16; ++i) k.c[i] = L[i];
#endif
    return k;
}

__device__ __forceinline__ void sincos_fast(const double x, double& sn, double& cs, const SinCosK& K)
{
    const double (&P)[16] = K.c;
    const double k = rint(x * P[0]);
    double r = fma(k, P[1], x);
    r = fma(k, P[2], r);
    r = fma(k, P[3], r);
    const double z = r * r;
    // __kernel_sin
    const double S1 = P[4], S2 = P[5], S3 = P[6], S4 = P[7], S5 = P[8], S6 = P[9];
    const double ps = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double sr = fma(z * r, fma(z, ps, S1), r);
    // __kernel_cos
    const double C1 = P[10], C2 = P[11], C3 = P[12], C4 = P[13], C5 = P[14], C6 = P[15];
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * pc);
    // quadrant: swap on odd k, signs from bits 1 of k and k + 1.  (As selects.  The same through bit operations -
    // v_bfi_b32 / v_xor_b32 on the halves, no VCC - was measured in round 4 after tools/gen_probe_banks.py showed two
    // v_cndmask_b32 in a row on one VCC costing a lone wave 48 cycles instead of 12: bit-identical results, 24
    // instructions fewer per team wave, headline unchanged at 3.96 us, lane kernel 2 % SLOWER (10.5 against 10.25 us at
    // 131072, 69 against 68 us at 1 M) - the selects stay.)
    const int q = (int)k & 3;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void sincos_fast(const double x, double& sn, double& cs)
{
    sincos_fast(x, sn, cs, sincos_consts());
}

__device__ __forceinline__ void sincos_joint(const double x, double& sn, double& cs)
{
    if (__builtin_expect(fabs(x) > kSinCosFastMax, 0)) {
        const SinCos r = sincos_slow(x);
        sn = r.s;
        cs = r.c;
        return;
    }
    sincos_fast(x, sn, cs);
}

// ---------------------------------------------------------------------------
// Per-lane kinematic state of the skill's serial chain.
template <int N>
struct Kin {
    double p[3];       // tool position
    double R[9];       // tool rotation, row-major
    double Jv[3][N];   // d p / d z_j
    double Jw[3][N];   // angular Jacobian columns (world joint axes)
    double o[3];       // orientation error 1/2 sum_c r_c x rd_c
    double M[9];       // R * Rd^T
    double tr;         // trace(Rd^T R)
};

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// Forward kinematics + geometric Jacobian.  URDF convention
//   T = prod_j Trans(p_j) R_j [Rot(axis_j, z[q_j]) | Trans(axis_j z[q_j])]
// (what urdf2casadi's converter.from_file builds for the reference,
// ur5_moe2016_example2.ipynb:47).  `zs` is the LDS copy of the state
// ([slot][lane]) used for the dynamically indexed joint value; `fr` is LDS
// scratch (6*N slots) for per-joint axis/origin, also dynamically indexed.
template <int N>
__device__ __forceinline__ void forward_kinematics(const DevSkill* __restrict__ S, const double* zs,
                                                   double* fr, int lane, Kin<N>& K)
{
    const clik_skill_desc& D = S->d;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double p[3] = {0, 0, 0};
    const int nj = D.n_joints;
    for (int j = 0; j < nj; ++j) {
        const clik_joint& jt = D.joints[j];
        const int jf = S->jflags[j];
        double T[9];
        if (!(jf & 2)) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                p[i] = fma(R[3 * i], jt.p[0], fma(R[3 * i + 1], jt.p[1], fma(R[3 * i + 2], jt.p[2], p[i])));
        }
        if (!(jf & 1)) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    T[3 * i + c] = R[3 * i] * jt.R[c] + R[3 * i + 1] * jt.R[3 + c] + R[3 * i + 2] * jt.R[6 + c];
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = T[i];
        }
        const int type = jt.type;
        if (type != CLIK_JOINT_FIXED) {
            const int qi = jt.q_index;
            const int acode = (jf >> 4) & 7;
            const int ak = (acode & 3) - 1;            // local coordinate axis, -1: generic
            const double asign = (acode & 4) ? -1.0 : 1.0;
            double ax[3];
            if (ak >= 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    ax[i] = asign * (ak == 0 ? R[3 * i] : (ak == 1 ? R[3 * i + 1] : R[3 * i + 2]));
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    ax[i] = R[3 * i] * jt.axis[0] + R[3 * i + 1] * jt.axis[1] + R[3 * i + 2] * jt.axis[2];
            }
            double* f = fr + (size_t)(6 * qi) * WAVE + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f[i * WAVE] = ax[i];
                f[(3 + i) * WAVE] = p[i];
            }
            const double ang = zs[qi * WAVE + lane];
            if (type == CLIK_JOINT_REVOLUTE) {
                double s, c;
                sincos_joint(ang, s, c);
                if (ak >= 0) {
                    // rotation about a local coordinate axis mixes the two other columns
                    s *= asign;
                    const int c1 = (ak + 1) % 3, c2 = (ak + 2) % 3;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double u = (c1 == 0 ? R[3 * i] : (c1 == 1 ? R[3 * i + 1] : R[3 * i + 2]));
                        const double v = (c2 == 0 ? R[3 * i] : (c2 == 1 ? R[3 * i + 1] : R[3 * i + 2]));
                        const double un = fma(c, u, s * v);
                        const double vn = fma(c, v, -s * u);
                        if (c1 == 0) R[3 * i] = un; else if (c1 == 1) R[3 * i + 1] = un; else R[3 * i + 2] = un;
                        if (c2 == 0) R[3 * i] = vn; else if (c2 == 1) R[3 * i + 1] = vn; else R[3 * i + 2] = vn;
                    }
                } else {
                    const double C = 1.0 - c;
                    const double x = jt.axis[0], y = jt.axis[1], z = jt.axis[2];
                    const double m[9] = {c + x * x * C, x * y * C - z * s, x * z * C + y * s,
                                         y * x * C + z * s, c + y * y * C, y * z * C - x * s,
                                         z * x * C - y * s, z * y * C + x * s, c + z * z * C};
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
                for (int i = 0; i < 3; ++i) p[i] = fma(ax[i], ang, p[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) K.R[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) K.p[i] = p[i];
    const uint32_t used = S->used_mask, rev = S->rev_mask;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double ax[3] = {0, 0, 0}, org[3] = {0, 0, 0};
        const bool is_used = (used >> j) & 1u;
        const bool is_rev = (rev >> j) & 1u;
        if (is_used) {
            const double* f = fr + (size_t)(6 * j) * WAVE + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ax[i] = f[i * WAVE];
                org[i] = f[(3 + i) * WAVE];
            }
        }
        if (is_used && is_rev) {
            const double r[3] = {p[0] - org[0], p[1] - org[1], p[2] - org[2]};
            double v[3];
            cross3(ax, r, v);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                K.Jv[i][j] = v[i];
                K.Jw[i][j] = ax[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                K.Jv[i][j] = ax[i];     // prismatic: axis; unused: 0
                K.Jw[i][j] = 0.0;
            }
        }
    }
}

// Orientation feature  o = 1/2 sum_c r_c x rd_c  w.r.t. the skill's target
// quaternion (constant or from input_var); M = R Rd^T and tr = trace(Rd^T R)
// give its Jacobian in closed form:  d o / d z_j = -1/2 (tr I - M) Jw[:,j].
template <int N>
__device__ __forceinline__ void orientation_feature(const DevSkill* __restrict__ S, const double* ys,
                                                    int lane, Kin<N>& K)
{
    const clik_skill_desc& D = S->d;
    double q[4];
    if (D.quat_src == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = ys[D.quat_yi[i] * WAVE + lane];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = D.quat[i];
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

// Value, state gradient and time derivative of one affine row (clik_row).
template <int N>
__device__ __forceinline__ double row_eval(const DevSkill* __restrict__ S, const clik_row& r, const int flags,
                                           const TickArgs& tk, const Kin<N>& K, const double (&z)[N],
                                           const double* ys, int lane, int n, double (&g)[N], double& dt)
{
    double v = r.c;
    dt = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = 0.0;
    if (flags & CLIK_ROW_HAS_Q) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < n) {
                const double a = r.a[j];
                g[j] = a;
                v = fma(a, z[j], v);
            }
        }
    }
    double lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
    bool fk = false;
    if (flags & CLIK_ROW_HAS_P) {
        fk = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lin[i] = r.b[i];
            v = fma(r.b[i], K.p[i], v);
        }
    }
    if (flags & CLIK_ROW_HAS_R) {
        fk = true;
        // d(sum_ic g_ic R_ic)/dz_j = Jw[:,j] . sum_c (r_c x g_c)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
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
    }
    if (flags & CLIK_ROW_HAS_O) {
        fk = true;
        // d(h.o)/dz_j = Jw[:,j] . ( -1/2 (tr h - M^T h) )
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double mh = K.M[i] * r.h[0] + K.M[3 + i] * r.h[1] + K.M[6 + i] * r.h[2];
            ang[i] += -0.5 * (K.tr * r.h[i] - mh);
            v = fma(r.h[i], K.o[i], v);
        }
    }
    if (fk) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < n) {
                double s = g[j];
#pragma unroll
                for (int i = 0; i < 3; ++i) s = fma(K.Jv[i][j], lin[i], fma(K.Jw[i][j], ang[i], s));
                g[j] = s;
            }
        }
    }
    if (flags & CLIK_ROW_HAS_Y) {
        const int ny = r.n_y;
        for (int k = 0; k < ny; ++k) v = fma(r.yc[k], ys[r.yi[k] * WAVE + lane], v);
    }
    {
        // time slots are run-time data even in shape-specialised kernels
        const int slot = r.t_slot;
        if (slot >= 0) {
            v += tk.tv[slot];
            dt = tk.tv[S->d.n_tslots + slot];
        }
    }
    return v;
}

// e (m), J (m x n) and d e/d t of constraint `ti`: the numeric content of the
// reference's cnstr.expression / cnstr.jacobian(state) / cnstr.jacobian(time)
// (pseudo_inverse.py:285-286, reactive_qp.py:210-216).  `m`, `tflags` (>= 0:
// use these flags for every row) and `all_affine` are compile-time constants
// in shape-specialised kernels.
template <int N, int M>
__device__ __forceinline__ void task_eval(const DevSkill* __restrict__ S, int ti, const int m,
                                          const int tflags, const bool all_affine, const TickArgs& tk,
                                          const Kin<N>& K, const double (&z)[N], const double* ys,
                                          int lane, int n, double (&e)[M], double (&J)[M][N],
                                          double (&Jt)[M])
{
    const clik_task& t = S->d.tasks[ti];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        if (i < m) {
            const int row0 = t.out_row0[i];
            if (all_affine || t.out_kind[i] == CLIK_OUT_AFFINE) {
                const clik_row& r = S->d.rows[row0];
                double g[N], dt;
                e[i] = row_eval<N>(S, r, tflags >= 0 ? tflags : r.flags, tk, K, z, ys, lane, n, g, dt);
                Jt[i] = dt;
#pragma unroll
                for (int j = 0; j < N; ++j) J[i][j] = g[j];
            } else {
                double ss = 0.0, tacc = 0.0, acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = 0.0;
                const int nr = t.out_nrows[i];
                for (int k = 0; k < nr; ++k) {
                    const clik_row& r = S->d.rows[row0 + k];
                    double g[N], dt;
                    const double v = row_eval<N>(S, r, r.flags, tk, K, z, ys, lane, n, g, dt);
                    ss = fma(v, v, ss);
                    tacc = fma(v, dt, tacc);
#pragma unroll
                    for (int j = 0; j < N; ++j) acc[j] = fma(v, g[j], acc[j]);
                }
                const double nrm = sqrt(ss);
                const double inv = 1.0 / nrm;
                e[i] = nrm;
                Jt[i] = tacc * inv;
#pragma unroll
                for (int j = 0; j < N; ++j) J[i][j] = acc[j] * inv;
            }
        }
    }
}

// compile-time loop with the index as an integral constant
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// gain * v  (float or m x m matrix gain, constraints.py:32-65)
template <int M>
__device__ __forceinline__ void gain_apply(const clik_task& t, const int m, const double (&v)[M], double (&out)[M])
{
    if (!t.gain_is_matrix) {
        const double g = t.gain[0];
#pragma unroll
        for (int i = 0; i < M; ++i) out[i] = (i < m) ? g * v[i] : 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double s = 0.0;
            if (i < m) {
#pragma unroll
                for (int k = 0; k < M; ++k)
                    if (k < m) s = fma(t.gain[i * m + k], v[k], s);
            }
            out[i] = s;
        }
    }
}

// Cooperative, fully coalesced load of the wave's [64][w] row-major block into registers
// (chunk i of lane l = element i*64 + l of the block).  rows_valid <= 64; the loads are
// unconditional with clamped indices, so a caller can put several blocks in flight before
// the first LDS write (one memory round trip).  WMAX bounds the unrolled issue (w <= WMAX).
// The shape-specialised kernels then copy the chunks to LDS unchanged (rows_to_lds).
template <int WMAX>
__device__ __forceinline__ void stage_load(const double* __restrict__ g, const int w, const int rows_valid,
                                           const int lane, double (&v)[WMAX])
{
    const int last = rows_valid * w - 1;
#pragma unroll
    for (int i = 0; i < WMAX; ++i) {
        int k = (i < w ? i : 0) * WAVE + lane;
        k = k < last ? k : last;
        v[i] = g[k];
    }
}

// run-time width (dynamic kernels): chunks of 8 columns
__device__ __forceinline__ void stage_in_dyn(const double* __restrict__ g, const int w, const int rows_valid,
                                             double* lds, const int lane)
{
    const int total = rows_valid * w;
    for (int base = 0; base < w; base += 8) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = (base + i) * WAVE + lane;
            v[i] = (base + i < w && k < total) ? g[k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = (base + i) * WAVE + lane;
            if (base + i < w && k < total) {
                const int r = k / w, c = k - r * w;
                lds[c * WAVE + r] = v[i];
            }
        }
    }
}

__device__ __forceinline__ void stage_out_dyn(double* __restrict__ g, const int w, const int rows_valid,
                                              const double* lds, const int lane)
{
    const int total = rows_valid * w;
    for (int k = lane; k < total; k += WAVE) {
        const int r = k / w, c = k - r * w;
        g[k] = lds[c * WAVE + r];
    }
}

}  // namespace clik
