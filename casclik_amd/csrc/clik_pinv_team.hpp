// Team kernel: FOUR LANES PER ROBOT INSTANCE for the priority-stack family of BASELINE config 3
//
//     [ SetConstraint on every joint (multidim, unit rows) ;  state-dependent EqualityConstraint
//       with m <= n rows (the pose) ;  joint-space EqualityConstraint (unit rows, e.g. centering) ]
//
// (reference: pseudo_inverse.py:259-451 evaluated for both modes of the one-set table :107-130,
// mode scan :530-550).  At <= 32768 instances the lane-per-instance kernels leave 3/4 of the SIMDs
// idle and a tick is the serial instruction stream of one wave; here the four lanes of a DPP quad
// share one instance, so 16384 instances are 1024 waves (one per SIMD) and each wave issues fewer
// instructions.
//
// What the lanes split is NOT the entries of the <= 8x8 matrices (moving an fp64 between lanes costs
// 1-2 instructions, a Gram entry costs 7 FMAs: entry-level splits are communication-bound, see
// DESIGN.md) but the INDEPENDENT SOLVES of the two speculated modes.  Both modes of the family need
// only damped solves with shifted copies of ONE Gram matrix  Gm = J J'  of the pose task:
//
//   A0 = Gm + lam I        y  = A0^-1 d1,  y2 = A0^-1 y                       (both modes)
//        mode 0: the doubly processed first equality (:317-326 + :382-396) sums to J'(y + lam y2)
//        mode 1: the pose behind the active set:  N_set J' y
//   A1 = 2 Gm + lam I      mode 0: the centering task projects through [J; J]:
//                              (2J'J + lam I)^-1 2J'J w  =  2 J' A1^-1 (J w)       (push-through)
//   A2 = Gm + (1+lam) I    mode 1: the centering task projects through [I; J] with activations S:
//                              G = (1+lam) I + J'J,  C = S + J'J,
//                              G^-1 u = (u - J' A2^-1 (J u)) / (1+lam),  u = C w   (Woodbury)
//
// so lane r of the quad factors  A_r = alpha_r Gm + beta_r I  and back-substitutes ITS right-hand
// side with the SAME instruction stream (only data differ): three 6x6 LDL' factorisations, four
// solves and four J' products cost one of each.  Lane 3 repeats lane 0's first solve without the
// second pass (mode 1 needs J'y alone).  The results meet through DPP quad_perm moves (no LDS
// round trip); FK, the task rows and Gm are evaluated redundantly by the four lanes.
//
// The identities are exact for the damped inverse; the values differ from the literal Gram-form
// evaluation by rounding only (A1, A2 are the better conditioned forms), inside PINV_RTOL.
#pragma once
#include "clik_pinv_static.hpp"

namespace clik {

// (What each phase of the tick costs: tools/phase_budget.py reads it off the compiled kernel - profiles/r6_phase_budget.md.
// Round 2's timing ablation (-DCLIK_TEAM_ABLATE, wrong results by design) and the "front end once per quad + broadcast"
// experiment (-DCLIK_TEAM_FRONT_ONCE, 4.25 against 3.97 us) are retired: tools/experiments/pinv_retired.patch.)

constexpr int TEAM = 4;                     // lanes per instance = one DPP quad
constexpr int TEAM_WAVES = 4;               // waves per block: 64 instances, one wave per SIMD of a CU
constexpr int TEAM_INST = TEAM_WAVES * WAVE / TEAM;

// value of lane SRC_EVEN / SRC_ODD of each lane pair of the quad (pairs (0,1) and (2,3))
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(const double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int QUAD_LANE0 = 0x00;    // quad_perm:[0,0,0,0]

template <const ShapeDesc& SD>
inline size_t team_lds_bytes(bool values = false)
{
    if (values) return 0;       // (the value-specialised instantiation reads and writes global memory directly)
    return ((size_t)StaticLayout<SD>::IMG_DOUBLES + (size_t)(SD.n + (SD.n_y > 0 ? SD.n_y : 0)) * TEAM_INST) * sizeof(double);
}

// One tick of the lane's instance in the quad: candidate velocities of both modes and the tangent-cone verdict.
// Slds: skill image (LDS copy, or the address of a local constexpr object whose loads fold to literals);
// z: the instance's state in every lane of the quad; a0 / a1: state variables 2r and 2r+1 (clamped to N-1) of lane
// r; on return lane 0 holds the mode-0 velocity in v, lane 3 the mode-1 one, in_tc the cone test of v.
// -DCLIK_RESIDENT_PIPELINE=0: the resident tick kernel without its software pipeline (every tick: poll the ticket, load
// the rows, compute, store, wait for the acknowledgements, publish) - for tools/resident_probe.py's comparison only
#ifndef CLIK_RESIDENT_PIPELINE
#define CLIK_RESIDENT_PIPELINE 1
#endif

// What a lane's role in the quad means in numbers (functions of r = lane & 3 and of the damping only):
//   lane          0          1           2          3
//   beta         lam       lam / 2     1 + lam     lam        shift of the Gram matrix the lane factors
//   hsel          1          0           0          1         solo lanes: right-hand side = the pose task's velocity
//   nsolo         0          1           1          0         lanes 1 / 2: right-hand side = J X
//   cax, cx2     1, 0       1, 0      1+lam, 1   1+lam, 1     X = w2 o (cax - cx2 s)
//   lam0         lam         0           0          0         second pass of the doubly processed equality
//   kap, hp      1, 0       1, 0     1/(1+lam),1 1/(1+lam),1  the pair's closing combination
// The value-specialised kernels LOAD them (a 4 x 8 table in constant memory, requested with the rows: 4 load
// instructions) - built from r with selects they were ~40 instructions of every tick.
struct RoleConsts {
    double beta, hsel, nsolo, cax, cx2, lam0, kap, hp;
};
__device__ __forceinline__ RoleConsts role_consts_computed(const int r, const double lam)
{
    const double one_lam = 1.0 + lam;
    const bool hi = (r & 2) != 0, solo = (r == 0) || (r == 3);
    RoleConsts rc;
    rc.beta = (r == 1) ? 0.5 * lam : ((r == 2) ? one_lam : lam);
    rc.hsel = solo ? 1.0 : 0.0;
    rc.nsolo = solo ? 0.0 : 1.0;
    rc.cax = hi ? one_lam : 1.0;
    rc.cx2 = hi ? 1.0 : 0.0;
    rc.lam0 = (r == 0) ? lam : 0.0;
    rc.kap = hi ? 1.0 / one_lam : 1.0;
    rc.hp = hi ? 1.0 : 0.0;
    return rc;
}
struct RoleTable {
    double v[TEAM][8];
};
constexpr RoleTable role_table(const double lam)
{
    const double one_lam = 1.0 + lam;
    return RoleTable{{{lam, 1.0, 0.0, 1.0, 0.0, lam, 1.0, 0.0},
                      {0.5 * lam, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 0.0},
                      {one_lam, 0.0, 1.0, one_lam, 1.0, 0.0, 1.0 / one_lam, 1.0},
                      {lam, 1.0, 0.0, one_lam, 1.0, 0.0, 1.0 / one_lam, 1.0}}};
}
// (not const: the loads must stay loads)
template <class IMGV>
inline __constant__ RoleTable kRoleTable = role_table(IMGV::value.lam);
#ifndef CLIK_ROLE_TABLE
#define CLIK_ROLE_TABLE 1       // (0: the value-specialised kernels build the role constants from r with selects, as the others do)
#endif
template <class IMGV>
__device__ __forceinline__ RoleConsts role_consts_loaded(const int r)
{
#if !CLIK_ROLE_TABLE
    return role_consts_computed(r, IMGV::value.lam);
#endif
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p = reinterpret_cast<const d2*>(&kRoleTable<IMGV>.v[r][0]);
    const d2 a = p[0], b = p[1], c = p[2], d = p[3];
    return RoleConsts{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

struct NoMidTick {
    __device__ __forceinline__ void operator()(const double) const {}
};
// MID: called once behind the forward kinematics / task rows - the resident tick kernel requests the NEXT tick's rows
// there (pinv_resident_team_kernel), everyone else nothing
template <const ShapeDesc& SD, class MID = NoMidTick>
__device__ __forceinline__ void team_tick(const Img<SD>* __restrict__ Slds, const TickArgs& tk,
                                          const double (&z)[SD.n], const double* ysl, const double a0, const double a1,
                                          const int r, const int inst, const RoleConsts& rc, const SinCosK& sck,
                                          double (&v)[SD.n], bool& in_tc, MID&& mid = MID())
{
    constexpr int N = SD.n, M = SD.m[1], M0 = SD.m[0], M2 = SD.m[2];
    constexpr int NT = M * (M + 1) / 2;
    // ---- front end: sin / cos split over the quad, then FK and the task rows in every lane ----
    TaskCache<SD> tc;
    CLIK_PHASE("sincos");
    {
        double sns[N], css[N];
        if constexpr (SD.uses_fk != 0) {
            // lane r evaluated state variables 2r and 2r+1 (a0, a1), the quad exchanges by DPP
            double sn0, cs0, sn1, cs1;
            sincos_fast(a0, sn0, cs0, sck);
            sincos_fast(a1, sn1, cs1, sck);
            const bool huge = (fabs(a0) > kSinCosFastMax) | (fabs(a1) > kSinCosFastMax);
            if (__builtin_expect(__ballot(huge) != 0ull, 0)) {
                if (fabs(a0) > kSinCosFastMax) { const SinCos sc = sincos_slow(a0); sn0 = sc.s; cs0 = sc.c; }
                if (fabs(a1) > kSinCosFastMax) { const SinCos sc = sincos_slow(a1); sn1 = sc.s; cs1 = sc.c; }
            }
            CLIK_PHASE("sincos_exchange");
            static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if constexpr (shape_state_type(SD, j) == CLIK_JOINT_REVOLUTE) {
                    constexpr int CTRL = (j / 2) * 0x55;           // quad_perm:[k,k,k,k], k = owner lane
                    sns[j] = quad_perm_f64<CTRL>((j & 1) ? sn1 : sn0);
                    css[j] = quad_perm_f64<CTRL>((j & 1) ? cs1 : cs0);
                } else {
                    sns[j] = css[j] = 0.0;
                }
            });
        }
        const Img<SD> Sfk = *Slds;
        __builtin_amdgcn_sched_barrier(0);
        CLIK_PHASE("fk");
        Kin<N> K;
        if constexpr (SD.uses_fk != 0) {
            forward_kinematics_sc<SD>(&Sfk, z, sns, css, K);
            if constexpr (SD.quat_src != 0) orientation_feature_s<SD>(&Sfk, ysl, inst, K);
        }
        CLIK_PHASE("task_rows");
        cache_task<SD, 0>(&Sfk, tk, K, z, ysl, inst, tc);
    }

    const Img<SD> Sb = *Slds;
    __builtin_amdgcn_sched_barrier(0);
    CLIK_PHASE("desired");
    const Img<SD>* __restrict__ S = &Sb;
    const double lam = S->lam;
    const double one_lam = 1.0 + lam;

    // desired task velocities  d = -K e - de/dt   (pseudo_inverse.py:318-321, :383-386)
    double des1[M], w2[N];
    {
        double e[M], Jt[M], ke[M];
        task_values<SD, 1>(S, tk, tc, z, ysl, inst, e, Jt);
        gain_apply_s<M, SD.gain_matrix[1] != 0>(S->tasks[1], e, ke);
#pragma unroll
        for (int i = 0; i < M; ++i) des1[i] = SD.feedforward != 0 ? -ke[i] - Jt[i] : -ke[i];
    }
    {
        // w2 = pinv(J2) d2 of the joint-space task: one entry per row (host-side pinv of the unit rows)
        double e[M2], Jt[M2], ke[M2];
        task_values<SD, 2>(S, tk, tc, z, ysl, inst, e, Jt);
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
    // the set: violated rows (activation S of the multidim set, :289-298) by state column, and the
    // values its tangent-cone test needs
    double e0[M0], Jt0[M0];
    task_values<SD, 0>(S, tk, tc, z, ysl, inst, e0, Jt0);
    double sact[N], p0[N];          // sact[col] = 1.0 when the set row on that state is violated
    static_for<0, M0>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int col = SD.ucol[0][i] - 1;
        sact[col] = ((e0[i] - S->tasks[0].set_max[i] > 0.0) | (e0[i] - S->tasks[0].set_min[i] < 0.0)) ? 1.0 : 0.0;
        p0[col] = S->cpinv[0][col * CLIK_MAX_M + i];
    });

    // Gm = J J'
    CLIK_PHASE("gram");
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

    // ---- per-lane role ------------------------------------------------------------------------
    // The lower-priority task's projected contribution, with the stack matrix G = D + c J'J of the mode
    // (D = lam I in mode 0 with c = 2; D = (1+lam) I in mode 1 with c = 1) and C = G - (D - S):
    //     w2 - G^-1 C w2  =  G^-1 (D - S) w2  =  D^-1 (x - c J' A^-1 J x),   x = (D - S) w2,  A = c J J' + D
    // (Woodbury; S = 0 in mode 0).  So lane 1 solves A1 t = J w2 (then x/D = w2, c = 2), lane 2 solves
    // A2 t = J x with x = ((1+lam) - s) o w2 (c = 1), lanes 0 / 3 solve A0 y = d1.
    if constexpr (!std::is_same<std::decay_t<MID>, NoMidTick>::value) {
        // The hook gets the first Gram entry as an ANCHOR: what it does has to depend on it (the resident kernel ties its
        // decision to it through an empty asm), or the compiler - free to sink pure arithmetic below the hook's branch -
        // puts the hook wherever it likes (round 5: 140 instructions into the tick).  Gm[0] needs the task rows: the hook
        // runs behind the forward kinematics, ~300 instructions in, ~800 before the tick's end.
        mid(Gm[0]);
    }
    CLIK_PHASE("role_rhs");
    // Lane 1's matrix is 2 Gm + lam I = 2 (Gm + lam/2 I): it factors Gm + lam/2 I and its solution is TWICE the one the
    // formula above names - exactly (a power of two), so "2 g1" below is its J' product as it stands.  Every lane
    // then factors Gm + beta I with its own beta: Gm is used as it is, only the diagonal is touched.
    // X = w2 o (cax - cx2 s): w2 in the mode-0 pair, w2 ((1+lam) - s) in the mode-1 pair - the right-hand side J X of
    // lanes 1 / 2 and the term the solo lanes 0 / 3 add to their velocity at the end, one evaluation for both uses.
    double X[N], rhs[M];
#pragma unroll
    for (int j = 0; j < N; ++j) X[j] = w2[j] * fma(-rc.cx2, sact[j], rc.cax);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) sacc = fma(jac<SD, 1>(S, tc, i, j), X[j], sacc);
        rhs[i] = fma(rc.nsolo, sacc, rc.hsel * des1[i]);
    }
    double A[NT], rd[M], s2[M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int k = 0; k <= i; ++k) A[tri(i, k)] = (i == k) ? Gm[tri(i, k)] + rc.beta : Gm[tri(i, k)];
    CLIK_PHASE("factor");
    ldl_factor_s<M>(A, rd);
    CLIK_PHASE("solve1");
    ldl_solve_s<M>(A, rd, rhs);
    CLIK_PHASE("solve2");
#pragma unroll
    for (int i = 0; i < M; ++i) s2[i] = rhs[i];
    ldl_solve_s<M>(A, rd, s2);      // (second pass of the doubly processed equality: lane 0 only)
#pragma unroll
    for (int i = 0; i < M; ++i) rhs[i] = fma(rc.lam0, s2[i], rhs[i]);
    CLIK_PHASE("jt_product");
    double g[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double sacc = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) sacc = fma(jac<SD, 1>(S, tc, i, j), rhs[i], sacc);
        g[j] = sacc;
    }

    
    // ---- the quad's results meet: lane 0 forms the mode-0 velocity, lane 3 the mode-1 one --
    //   mode 0:  v = J'(y + lam y2)  +  (w2 - 2 J' A1^-1 J w2)                      = g0 + (w2 - g1), g1 of the halved matrix   (lane 0)
    //   mode 1:  v = N_set J'y       +  (x - J' A2^-1 J x) / (1+lam)                 = (1 - p0 s) o g3 + (x - g2)/(1+lam)   (lane 3)
    CLIK_PHASE("meet");
    // (one exchange per entry: lanes swap inside their pair, so lane 0 holds g0 and receives g1, lane 3 holds g3 and
    // receives g2 - the mode-0 velocity forms in lane 0, the mode-1 one in lane 3; lanes 1 / 2 compute unused values)
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double other = quad_perm_f64<0xB1>(g[j]);      // quad_perm:[1,0,3,2]
        const double nmul = fma(-p0[j] * rc.hp, sact[j], 1.0);   // 1 in the mode-0 pair, 1 - p0 s in the mode-1 pair
        v[j] = fma(g[j], nmul, rc.kap * (X[j] - other));
    }
    // tangent-cone test of the inactive set on the mode-0 candidate (:222-252; the other lanes evaluate it
    // on the other candidate, unused)
    CLIK_PHASE("cone");
    {
        // (pseudo_inverse.py:222-252, same values; written for few instructions: half signs by bit operations,
        // flags combined bitwise - the short-circuit forms compile to a branch per row -, and everything behind
        // "inside" skipped when every instance of the wave is inside its limits, the normal state of a control loop)
        const clik_task& t = S->tasks[0];
        double le[M0], ue[M0];
#pragma unroll
        for (int i = 0; i < M0; ++i) {
            le[i] = e0[i] - t.set_min[i];
            ue[i] = e0[i] - t.set_max[i];
        }
        // all(le >= 1e-12) & all(ue <= 1e-12) through the smallest le and the largest ue: a compare writes a lane mask
        // into scalar registers and every "and" of two masks is a scalar instruction of its own - 14 compares + 13 ands
        // against 12 min / max + 2 compares + 1 and
        double le_min = le[0], ue_max = ue[0];
#pragma unroll
        for (int i = 1; i < M0; ++i) {
            le_min = fmin(le_min, le[i]);
            ue_max = fmax(ue_max, ue[i]);
        }
        // (as "outside" and through the builtin: the lane mask of two compares or-ed is tested as it stands)
        const bool outside = (le_min < 1e-12) | (ue_max > 1e-12);
        in_tc = true;
        if (__builtin_amdgcn_ballot_w64(outside) != 0ull) {
            double od = 0.0, nde = 0.0, nout = 0.0, ndiff = 0.0;
            static_for<0, M0>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int col = SD.ucol[0][i] - 1;
                const double de = Jt0[i] + v[col];
                const double hl = half_sign(le[i]), hu = half_sign(ue[i]);     // (sign(le) + sign(ue)) / 2 = hl + hu
                ndiff += fabs(hl - hu);
                const double out = hl + hu;
                od = fma(out, de, od);
                nde = fma(de, de, nde);
                nout = fma(out, out, nout);
            });
            // corner = all(sign(le) == sign(ue)): the half signs are multiples of 1/2, so "no row differs" is "the sum
            // of |hl - hu| is exactly zero" - arithmetic instead of seven compares and the masks' ands
            const bool corner = ndiff == 0.0;
            bool going_in = od < 0.0;
            if (__builtin_amdgcn_ballot_w64(corner & outside) != 0ull) {
                // every joint beyond a limit (a corner of the box): inward only within 45 degrees of the diagonal
                const double dists = (sqrt(nde) + 1e-10) * sqrt(nout);
                const bool steep = (od < 0.0) & (fabs(od) / dists < 0.70710678118654757);
                going_in = corner ? steep : going_in;
            }
            in_tc = !outside | going_in;
        }
    }
    CLIK_PHASE_END();
}

// raw words of a skill image as a literal (value-specialised kernels, see below)
template <int K>
struct RawImage {
    unsigned long long w[K];
};

// IMGV = void: the skill's numbers (chain, gains, bounds, row coefficients) come from the skill image in memory
// (global -> LDS -> registers).  IMGV = a type with `static constexpr Img<SD> value`: they are COMPILED IN - the
// instantiation belongs to one skill, as the functions CasADi generates for the reference do
// (pseudo_inverse.py:476-483 compiles the expression graph with its constants).  No image traffic, no LDS reads of
// constants, and every product with a structural 0 / 1 disappears at compile time.  casclik_amd/jit.py builds
// such an instantiation when the controller is set up (clik_pinv_attach_value_kernel).
template <const ShapeDesc& SD, class IMGV = void>
__global__ __launch_bounds__(TEAM_WAVES * WAVE) void pinv_solve_static_team_kernel(
    const void* __restrict__ img_g, const double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B, const TickArgs tk)
{
    static_assert(shape_team_ok(SD), "shape outside the team kernel's family");
    constexpr bool VALUES = !std::is_void<IMGV>::value;
    extern __shared__ double lds[];
    constexpr int N = SD.n, M = SD.m[1], M0 = SD.m[0], M2 = SD.m[2], NY = SD.n_y > 0 ? SD.n_y : 0;
    constexpr int NT = M * (M + 1) / 2;
    constexpr int NTHREADS = TEAM_WAVES * WAVE;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = tid & (TEAM - 1);                 // role in the team
    const int inst = tid >> 2;                      // instance within the block
    const long long b0 = (long long)blockIdx.x * TEAM_INST;
    const long long left = B - b0;
    const int rows_valid = left < TEAM_INST ? (int)left : TEAM_INST;
    double* zs = lds + (VALUES ? 0 : StaticLayout<SD>::IMG_DOUBLES);       // [64][N] joint state, later the velocities
    double* ys = zs + N * TEAM_INST;                        // [64][NY]
    typedef double d2 __attribute__((ext_vector_type(2)));
    CLIK_STAMP_W(0, 0);
    CLIK_BODY_BEGIN();
    CLIK_PHASE("rows_in");
    // Value-specialised instantiation: every index into the state / input rows is a literal, so each lane reads its
    // instance's rows straight from global memory into registers (the four lanes of a quad hit the same addresses,
    // a wave's 16 rows are contiguous) and the selected lane stores the velocities itself: no LDS, no barrier.
    // (this kernel serves batches of at most 32768 instances - launch_solve_values -: the row index is a 32-bit number,
    // and the rows are addressed as uniform base + 32-bit lane offset: the 64-bit index arithmetic was 17 instructions)
    const unsigned uinst = (unsigned)blockIdx.x * (unsigned)TEAM_INST + (unsigned)inst;
    const unsigned ulast = (unsigned)B - 1u;
    const unsigned urow = uinst < ulast ? uinst : ulast;
    const long long binst = (long long)urow;
    double zdir[N], ydir[NY > 0 ? NY : 1];
    RoleConsts rc;
    const int j0 = 2 * r < N ? 2 * r : N - 1, j1 = 2 * r + 1 < N ? 2 * r + 1 : N - 1;
    double a0 = 0.0, a1 = 0.0;
    // (the sin / cos constants: their scalar loads go out before anything else)
    const SinCosK sck = sincos_consts();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VALUES) {
        // The loads of a wave return IN ORDER: this lane's two sin / cos arguments are requested FIRST (the tick starts
        // with them), then the robot_var row, the input_var row, and last the role constants (needed 500 instructions
        // in).  The compiler's scheduler must not reorder them - it had put the two arguments BEHIND both rows, so
        // the first sine waited for nine loads instead of one (0.2 us per tick: the round-6 diet of 150 instructions
        // bought nothing until this was seen) - hence the fences.
        const unsigned qoff = __umul24(urow, (unsigned)(N * sizeof(double)));
        a0 = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(q) + (qoff + (unsigned)j0 * 8u));
        a1 = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(q) + (qoff + (unsigned)j1 * 8u));
        __builtin_amdgcn_sched_barrier(0);
        const double* __restrict__ qrow = reinterpret_cast<const double*>(reinterpret_cast<const char*>(q) + qoff);
#pragma unroll
        for (int j = 0; j < N; ++j) zdir[j] = qrow[j];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NY > 0) {
            const double* __restrict__ yrow = reinterpret_cast<const double*>(reinterpret_cast<const char*>(y) + __umul24(urow, (unsigned)(NY * sizeof(double))));
#pragma unroll
            for (int k = 0; k < NY; ++k) ydir[k] = yrow[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        rc = role_consts_loaded<IMGV>(r);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!VALUES) {
        // one memory round trip: image chunks round-robin over the waves, the block's q / y rows
        // cooperatively (coalesced, index-clamped), all issued before the first LDS write
        constexpr int CH = VALUES ? 0 : StaticLayout<SD>::IMG_CHUNKS;
        constexpr int PER = VALUES ? 1 : (CH + TEAM_WAVES - 1) / TEAM_WAVES;
        constexpr int QR = (N * TEAM_INST + NTHREADS - 1) / NTHREADS;
        constexpr int YR = (NY * TEAM_INST + NTHREADS - 1) / NTHREADS;
        const d2* src = (const d2*)img_g;
        d2* dst = (d2*)lds;
        const int lane = tid & (WAVE - 1);
        d2 img[PER];
        if constexpr (!VALUES) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int ck = k * TEAM_WAVES + wave;
                img[k] = src[(ck < CH ? ck : CH - 1) * WAVE + lane];
            }
        }
        double qv[QR], yv[YR > 0 ? YR : 1];
        const double* qg = q + b0 * N;
        const int qlast = rows_valid * N - 1;
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            const int k = i * NTHREADS + tid;
            qv[i] = qg[k < qlast ? k : qlast];
        }
        if constexpr (NY > 0) {
            const double* yg = y + b0 * NY;
            const int ylast = rows_valid * NY - 1;
#pragma unroll
            for (int i = 0; i < YR; ++i) {
                const int k = i * NTHREADS + tid;
                yv[i] = yg[k < ylast ? k : ylast];
            }
        }
        if constexpr (!VALUES) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int ck = k * TEAM_WAVES + wave;
                if (ck < CH) dst[ck * WAVE + lane] = img[k];
            }
        }
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            const int k = i * NTHREADS + tid;
            if (k < N * TEAM_INST) zs[k] = qv[i];
        }
        if constexpr (NY > 0) {
#pragma unroll
            for (int i = 0; i < YR; ++i) {
                const int k = i * NTHREADS + tid;
                if (k < NY * TEAM_INST) ys[k] = yv[i];
            }
        }
        __syncthreads();
    }
    CLIK_STAMP_W(0, 1);
    // (a LOCAL constexpr copy: loads from it fold to immediates; the namespace-scope object itself is emitted
    // "externally_initialized" on the device and would be read from memory)
    constexpr Img<SD> Sval = []() constexpr { if constexpr (VALUES) return IMGV::value; else return Img<SD>{}; }();
    const Img<SD>* __restrict__ Slds = VALUES ? &Sval : (const Img<SD>*)lds;
    const double* ysl = VALUES ? ydir : ys + inst * NY;
    if constexpr (!VALUES) rc = role_consts_computed(r, Slds->lam);
    double z[N];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = VALUES ? zdir[j] : zs[inst * N + j];

    if constexpr (!VALUES) {
        a0 = zs[inst * N + j0];
        a1 = zs[inst * N + j1];
    }
    double v[N];
    bool in_tc;
    team_tick<SD>(Slds, tk, z, ysl, a0, a1, r, inst, rc, sck, v, in_tc);
    // the scan of :530-550 as a select: mode 0 if its cone test passes, else mode 1 (the active set
    // has no cone test, so mode 1 is always admissible)
    CLIK_STAMP_W(0, 6);
    CLIK_PHASE("select_store");
    const bool ok0 = __builtin_amdgcn_mov_dpp((int)in_tc, QUAD_LANE0, 0xf, 0xf, true) != 0;
    if constexpr (VALUES) {
        if (r == (ok0 ? 0 : 3) && uinst <= ulast) {
            double* __restrict__ drow = reinterpret_cast<double*>(reinterpret_cast<char*>(dq) + __umul24(uinst, (unsigned)(N * sizeof(double))));
#pragma unroll
            for (int j = 0; j < N; ++j) drow[j] = v[j];
            if (mode_out != nullptr)
                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(mode_out) + uinst * 4u) = ok0 ? 0 : 1;
        }
    } else {
        if (r == (ok0 ? 0 : 3)) {
            // (each team reads and writes only its own row, and a team is inside one wave: no barrier needed)
#pragma unroll
            for (int j = 0; j < N; ++j) zs[inst * N + j] = v[j];
            if (mode_out != nullptr && inst < rows_valid) mode_out[b0 + inst] = ok0 ? 0 : 1;
        }
        __syncthreads();
        constexpr int QR = (N * TEAM_INST + NTHREADS - 1) / NTHREADS;
        double* dg = dq + b0 * N;
        const int total = rows_valid * N;
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            const int k = i * NTHREADS + tid;
            if (k < total) dg[k] = zs[k];
        }
    }
    CLIK_STAMP_W(0, 5);
    CLIK_BODY_END();
    CLIK_PHASE_END();
}

// n_ticks of (tick -> clamp(+-max_speed) -> integrate) in one launch with four lanes per instance: the host loop
// of the notebooks (ur5_moe2016_example2.ipynb:537-545) without the per-tick launch and HBM round trip, the state
// in registers (replicated over the quad).  stages = 1: explicit Euler; 4: classical Runge-Kutta with the
// controller as the right-hand side (see pinv_rollout_static_kernel).  q is updated in place; dq / mode receive
// the last tick (Runge-Kutta: the combined rate and the mode of the first stage).
template <const ShapeDesc& SD, class IMGV = void, int STAGES = 1>
__global__ __launch_bounds__(TEAM_WAVES * WAVE) void pinv_rollout_static_team_kernel(
    const void* __restrict__ img_g, double* __restrict__ q, const double* __restrict__ y,
    double* __restrict__ dq, int32_t* __restrict__ mode_out, const long long B,
    const double* __restrict__ tterms, const int n_ticks, const double dt, const double max_speed)
{
    static_assert(STAGES == 1 || STAGES == 4, "explicit Euler or classical Runge-Kutta");
    constexpr int stages = STAGES;
    static_assert(shape_team_ok(SD), "shape outside the team kernel's family");
    constexpr bool VALUES = !std::is_void<IMGV>::value;
    extern __shared__ double lds[];
    constexpr int N = SD.n, NY = SD.n_y > 0 ? SD.n_y : 0;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = tid & (TEAM - 1);
    const int inst = tid >> 2;
    const long long b0 = (long long)blockIdx.x * TEAM_INST;
    const bool valid = b0 + inst < B;
    const long long binst = valid ? (b0 + inst) : (B - 1);
    double* ys = lds + (VALUES ? 0 : StaticLayout<SD>::IMG_DOUBLES);        // [64][NY] (image-reading build only)
    double z[N], ydir[NY > 0 ? NY : 1];
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = q[binst * N + j];
    if constexpr (NY > 0) {
#pragma unroll
        for (int k = 0; k < NY; ++k) ydir[k] = y[binst * NY + k];
    }
    if constexpr (!VALUES) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        constexpr int CH = StaticLayout<SD>::IMG_CHUNKS;
        const d2* src = (const d2*)img_g;
        d2* dst = (d2*)lds;
        const int lane = tid & (WAVE - 1);
        for (int ck = wave; ck < CH; ck += TEAM_WAVES) dst[ck * WAVE + lane] = src[ck * WAVE + lane];
        if constexpr (NY > 0) {
            // (run-time input indices: the rows of the image-reading build live in LDS; the quad writes the same values)
#pragma unroll
            for (int k = 0; k < NY; ++k) ys[inst * NY + k] = ydir[k];
        }
        __syncthreads();
    }
    constexpr Img<SD> Sval = []() constexpr { if constexpr (VALUES) return IMGV::value; else return Img<SD>{}; }();
    const Img<SD>* __restrict__ Slds = VALUES ? &Sval : (const Img<SD>*)lds;
    const double* ysl = VALUES ? ydir : ys + inst * NY;
    RoleConsts rc;
    if constexpr (VALUES) rc = role_consts_loaded<IMGV>(r);
    else rc = role_consts_computed(r, Slds->lam);
    const SinCosK sck = sincos_consts();        // (once per launch)
    const int nts = Slds->n_tslots;
    double vout[N];
    int acc_mode = -1;
#pragma unroll
    for (int j = 0; j < N; ++j) vout[j] = 0.0;
    // (no clamp = a bound no finite velocity reaches: the same two instructions, no test per element)
    const double vmax = max_speed > 0.0 ? max_speed : 1.7976931348623157e308;
    // the accepted candidate in every lane of the quad: mode 0 lives in lane 0, mode 1 in lane 3 - lane 0's value is
    // broadcast, and only quads that rejected mode 0 (a uniform decision inside a quad) fetch lane 3's on top
    auto accepted = [&](const double (&v)[N], const bool ok0, double (&d)[N]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < N; ++j) d[j] = quad_perm_f64<0x00>(v[j]);
        if (__builtin_amdgcn_ballot_w64(!ok0) != 0ull) {
            if (!ok0) {
#pragma unroll
                for (int j = 0; j < N; ++j) d[j] = quad_perm_f64<0xFF>(v[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) d[j] = fmax(fmin(d[j], vmax), -vmax);
    };
    auto sincos_args = [&](const double (&zz)[N], double& a0, double& a1) __attribute__((always_inline)) {
        // lane r's two sin / cos arguments out of the replicated state (register selects)
        a0 = zz[N - 1];
        a1 = zz[N - 1];
        static_for<0, TEAM>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr (2 * k < N) a0 = (r == k) ? zz[2 * k] : a0;
            if constexpr (2 * k + 1 < N) a1 = (r == k) ? zz[2 * k + 1] : a1;
        });
    };
#pragma unroll 1
    for (int tick = 0; tick < n_ticks; ++tick) {
        if constexpr (STAGES == 1) {
            // explicit Euler (the notebooks' loop, ur5_moe2016_example2.ipynb:537-545): nothing of the Runge-Kutta
            // staging - no saved state, no stage sums, no weights
            asm volatile("" ::: "memory");      // (keeps the image reads inside the loop, see pinv_rollout_static_kernel)
            const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + (size_t)tick * 2 * nts);
            double a0, a1;
            sincos_args(z, a0, a1);
            double v[N];
            bool in_tc;
            team_tick<SD>(Slds, tk, z, ysl, a0, a1, r, inst, rc, sck, v, in_tc);
            const bool ok0 = __builtin_amdgcn_mov_dpp((int)in_tc, QUAD_LANE0, 0xf, 0xf, true) != 0;
            accepted(v, ok0, vout);
#pragma unroll
            for (int j = 0; j < N; ++j) z[j] = fma(vout[j], dt, z[j]);
            acc_mode = ok0 ? 0 : 1;
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
            asm volatile("" ::: "memory");
            const TickArgs& tk = *reinterpret_cast<const TickArgs*>(tterms + ((size_t)tick * stages + st) * 2 * nts);
            double a0, a1;
            sincos_args(z, a0, a1);
            double v[N], d[N];
            bool in_tc;
            team_tick<SD>(Slds, tk, z, ysl, a0, a1, r, inst, rc, sck, v, in_tc);
            const bool ok0 = __builtin_amdgcn_mov_dpp((int)in_tc, QUAD_LANE0, 0xf, 0xf, true) != 0;
            accepted(v, ok0, d);
            const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0;
            const double cnext = (st == 2) ? dt : 0.5 * dt;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                ks[j] = fma(wgt, d[j], ks[j]);
                z[j] = fma(d[j], cnext, z0[j]);
            }
            mode0 = (st == 0) ? (ok0 ? 0 : 1) : mode0;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            vout[j] = ks[j] * (1.0 / 6.0);
            z[j] = fma(vout[j], dt, z0[j]);
        }
        acc_mode = mode0;
        }
    }
    if (r == 0 && valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            q[(b0 + inst) * N + j] = z[j];
            dq[(b0 + inst) * N + j] = vout[j];
        }
        if (mode_out != nullptr) mode_out[b0 + inst] = acc_mode;
    }
}

// ---- resident ticks -----------------------------------------------------------------------------------------
// ONE launch of the value-specialised team kernel that stays on the device and runs tick k whenever the producer of
// the inputs has published ticket k (clik_ticket::in_seq >= k, written with release semantics after q / y): a closed
// loop with fresh targets every tick then costs a device-side hand-off instead of a kernel launch.  Every wave polls
// the ticket itself (the four waves of a block share nothing in this instantiation), then reads its rows, runs the
// tick, stores, and writes k into ITS OWN slot of the `done` array once its stores are acknowledged; tick k is complete
// when every slot holds k.  (One shared counter was measured first: 1024 waves adding to one address serialise at
// the memory side - 24.6 us per closed-loop tick against 4.0 us for a launch per tick.)  The kernel leaves when n_ticks are done, when anyone sets `stop`, or when its watchdog (the
// 100 MHz s_memrealtime clock against the timeout given at launch) expires - it then writes stop = 2 so that every
// other wave and the producer leave too.  It never spins without that check.  (The watchdog counts polls, see below.)
struct ResidentTicket {
    unsigned in_seq, p0[15];
    unsigned ring_depth, p1a;         // input / output slots (0 or 1: one buffer)
    double integrate_dt, max_speed;   // > 0: the state is integrated in the kernel (q read at tick 1 only)
    unsigned p1[10];
    unsigned stop, p2[15];
    unsigned waves, ticks_done, p3[14];
};
static_assert(sizeof(ResidentTicket) == 256, "clik_ticket layout (include/clik.h)");

__device__ __forceinline__ unsigned long long realtime_100mhz()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// The rows of a resident wave's SIXTEEN instances (four lanes each), copied memory -> LDS WITHOUT passing through a
// register (global_load_lds_dwordx4 / _dword: lane l's piece lands at the LDS address in M0 + l x size; sc0 sc1: past the
// caches, as the resident kernels' other loads).  Round 6: a resident wave has its SIMD to itself, so the next tick's rows
// must be requested a whole tick ahead - and a load's destination registers then stay reserved for a whole tick in kernels
// that have none to spare: the compiler parked them in AGPRs, and the copy into the AGPR WAITS for the load (in the QP
// kernel 150 instructions after it was issued; the config-3 kernel therefore requested its rows only 300 instructions into
// the tick and still waited 0.4 us per tick at its end - profiles/r6_boundary_free_counters.md).  Through LDS the request
// costs no register at all.  The wave's rows are ONE contiguous block of 16 x NE doubles; `row0` its first row, the array
// `total_rows` long: a block that would run past the end of the array is read from `total_rows - 16` instead and the
// instances find their rows `shift` further down (stage_row).  `al16`: base address and row pitch allow 16-byte pieces
// (else 4-byte pieces: four times the requests) - the caller's `al16` must also imply that the ARRAY's byte size is a
// multiple of 16 (the resident kernels: base aligned and B x NE even), because a piece that would run past the end of
// the array is moved back to end there: for a size of 8 mod 16 that piece would land shifted by 8 bytes (the retired
// lane-kernel experiment tools/experiments/lane_rows_lds.patch got exactly that wrong).  Returns the shift.
template <int NE>
struct RowStage {
    static constexpr int BLOCK = 16 * NE * 8;                        // bytes of a wave's rows
    static constexpr int PIECES = (BLOCK + 1023) / 1024;             // 64 lanes x 16 bytes each
    static constexpr int DOUBLES = NE > 0 ? PIECES * 128 : 0;        // LDS reserved: the last piece whole
};
template <int NE>
__device__ __forceinline__ int stage_rows(const double* base, const long long row0, const long long total_rows,
                                          double* lds, const int lane, const bool al16)
{
    constexpr int BLOCK = RowStage<NE>::BLOCK;
    long long r0 = row0 < total_rows - 16 ? row0 : total_rows - 16;
    r0 = r0 > 0 ? r0 : 0;
    const char* src = reinterpret_cast<const char*>(base) + r0 * (NE * 8);
    const long long room = (total_rows - r0) * (NE * 8);             // bytes from src to the end of the array
    if (al16 && room >= 16) {
#pragma unroll
        for (int p = 0; p < RowStage<NE>::PIECES; ++p) {
            int off = p * 1024 + lane * 16;
            off = off < BLOCK - 16 ? off : BLOCK - 16;               // (the piece's upper lanes re-read the block's last bytes)
            off = off < room - 16 ? off : (int)(room - 16);          // (an array shorter than sixteen rows)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                             (__attribute__((address_space(3))) void*)(lds + p * 128), 16, 0, 17);
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4 * RowStage<NE>::PIECES; ++p) {
            int off = p * 256 + lane * 4;
            off = off < BLOCK - 4 ? off : BLOCK - 4;
            off = off < room - 4 ? off : (int)(room - 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                             (__attribute__((address_space(3))) void*)(lds + p * 32), 4, 0, 17);
        }
    }
    return (int)(row0 - r0);
}
// instance i (0 ... 15) of the wave: its row inside the staged block
__device__ __forceinline__ int stage_row(const int i, const int shift) { return (i + shift) < 15 ? (i + shift) : 15; }

// One wave's side of the ticket protocol (round 5: shared by the resident kernels that run one wave per block -
// pinv_resident_quad_kernel, qp_resident_box_front4_kernel; pinv_resident_team_kernel above keeps its own copy, with
// four waves per block).  `slot`: this wave's index into the `done` array.
struct ResidentWave {
    ResidentTicket* ticket;
    unsigned* done;
    unsigned long long polls, max_polls;
    unsigned seen, slot;
    bool leave, lane0, first;

    __device__ __forceinline__ void init(ResidentTicket* t, unsigned* d, const unsigned long long budget, const int n_ticks,
                                         const unsigned wave_slot, const unsigned n_waves, const int tid)
    {
        ticket = t;
        done = d;
        polls = 0;
        max_polls = budget;
        seen = 0u;
        slot = wave_slot;
        leave = false;
        lane0 = (tid & (WAVE - 1)) == 0;
        first = wave_slot == 0u && tid == 0;
        if (first) {
            ticket->waves = n_waves;
            ticket->p3[0] = (unsigned)budget;            // (diagnostics: what the launch handed over)
            ticket->p3[1] = (unsigned)(budget >> 32);
            ticket->p3[2] = (unsigned)n_ticks;
        }
    }
    // wait until ticket `want` is out (-> `seen`), someone sets `stop`, or the poll budget is used up (-> `leave`)
    __device__ __forceinline__ void poll_for(const unsigned want)
    {
#pragma unroll 1
        for (;;) {
            // (relaxed, system scope = a load that bypasses the caches; the rows are read the same way)
            seen = __hip_atomic_load(&ticket->in_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (seen >= want) {
                __atomic_signal_fence(__ATOMIC_ACQUIRE);     // (keeps the COMPILER from hoisting a row load above it)
                return;
            }
            if (__hip_atomic_load(&ticket->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                leave = true;
                return;
            }
            if (++polls > max_polls) {
                __hip_atomic_store(&ticket->stop, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (lane0) {
                    ticket->p3[3] = (unsigned)polls;        // (diagnostics: the wave that gave up, and after how many polls)
                    ticket->p3[4] = slot;
                }
                leave = true;
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __device__ __forceinline__ void peek() { seen = __hip_atomic_load(&ticket->in_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    // (the same load into a register of the caller's: read a tick later, it has a whole tick to arrive)
    __device__ __forceinline__ unsigned look() const { return __hip_atomic_load(&ticket->in_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    // every lane's stores acknowledged, then the wave's own slot (no shared counter)
    __device__ __forceinline__ void publish_done(const int k)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane0) __hip_atomic_store(done + slot, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (first) ticket->ticks_done = (unsigned)k;
    }
    __device__ __forceinline__ long long ring_depth() const
    {
        const unsigned raw = __hip_atomic_load(&ticket->ring_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return raw > 1u ? (long long)raw : 1ll;
    }
};

template <const ShapeDesc& SD, class IMGV, bool INTEGRATE = false>
__global__ __launch_bounds__(TEAM_WAVES * WAVE) void pinv_resident_team_kernel(
    const double* q, const double* y, double* dq, int32_t* mode_out, const long long B, const TickArgs tk,
    ResidentTicket* ticket, unsigned* done, const int n_ticks, const unsigned long long max_polls)
{
    static_assert(shape_team_ok(SD), "shape outside the team kernel's family");
    static_assert(!std::is_void<IMGV>::value, "resident ticks: value-specialised instantiation only");
    constexpr int N = SD.n, NY = SD.n_y > 0 ? SD.n_y : 0;
    const int tid = threadIdx.x;
    const int r = tid & (TEAM - 1);
    const int inst = tid >> 2;
    const long long b0 = (long long)blockIdx.x * TEAM_INST;
    const bool valid = b0 + inst < B;
    const long long binst = valid ? (b0 + inst) : (B - 1);
    constexpr Img<SD> Sval = IMGV::value;
    CLIK_PHASE("res_setup");
    const RoleConsts rc = role_consts_loaded<IMGV>(r);
    const SinCosK sck = sincos_consts();        // (once per launch)
    // watchdog: a budget of POLLS over the kernel's whole life (a poll is an L2-hitting load plus s_sleep 1: ~0.2 us measured, the nominal figure of clik_api.hip):
    // deterministic, unlike a clock - the first version compared s_memrealtime readings and misfired at the first tick
    // inside the long test run (never alone), leaving the kernel at once with stop = 2
    unsigned long long polls = 0;
    if (blockIdx.x == 0 && tid == 0) {
        ticket->waves = gridDim.x * TEAM_WAVES;
        ticket->p3[0] = (unsigned)max_polls;            // (diagnostics: what the launch handed over)
        ticket->p3[1] = (unsigned)(max_polls >> 32);
        ticket->p3[2] = (unsigned)n_ticks;
    }
    // Software pipeline over the ticks (a wave has its SIMD to itself, nothing else hides memory latency).  In the
    // order the wave issues them:
    //   start of tick k     request the ticket word (read in tick k + 1, see the loop)
    //   ~300 instructions   (team_tick's hook) if the ticket word requested in tick k - 1 says the producer has
    //   into tick k         published tick k + 1, request its rows: they arrive while the rest of tick k runs
    //   end of tick k       publish tick k - 1's "done" slot - its stores were issued a whole tick ago -, store dq of
    //                       tick k
    // so a wave that is being fed ahead never waits for memory.  If the producer has NOT published the next tick (a
    // closed loop: it waits for "done"), the slot is published at once and the wave polls, as before.
    // Loads of one wave return in order and a row is requested only after a ticket value that covers it has been SEEN
    // by this wave, so a row is never read before its producer wrote it.
    double zn[N], yn[NY > 0 ? NY : 1];
    bool have_next = false, leave = false;
    unsigned seen = 0u;             // latest ticket value this wave has requested (monotone at the producer)
    auto poll_for = [&](const unsigned want) __attribute__((always_inline)) {
#pragma unroll 1
        for (;;) {
            // (relaxed, system scope = a load that bypasses the caches; no L2 invalidate: the rows are read the same way)
            seen = __hip_atomic_load(&ticket->in_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (seen >= want) {
                // the rows are requested only after this value has been consumed (loads of a wave issue in program
                // order and bypass the caches); the fence keeps the COMPILER from hoisting a row load above it
                __atomic_signal_fence(__ATOMIC_ACQUIRE);
                return;
            }
            if (__hip_atomic_load(&ticket->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                leave = true;
                return;
            }
            if (++polls > max_polls) {
                __hip_atomic_store(&ticket->stop, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((tid & (WAVE - 1)) == 0) {
                    ticket->p3[3] = (unsigned)polls;        // (diagnostics: the wave that gave up, and after how many polls)
                    ticket->p3[4] = blockIdx.x * TEAM_WAVES + (tid >> 6);
                }
                leave = true;
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // A RING of `ring` input / output slots (ticket->ring_depth, 0 or 1: one buffer): tick k reads q / y from slot
    // (k - 1) % ring and writes dq / mode into the same slot of the output arrays ([ring][B][.] each).  With one
    // buffer a producer can only write the next inputs after every wave has finished the current tick (a closed loop by
    // construction); with three or more slots it writes tick k + 1's rows while tick k runs and publishes its ticket
    // ahead - the mode the software pipeline below is for.
    const unsigned ring_raw = __hip_atomic_load(&ticket->ring_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long ring = ring_raw > 1u ? (long long)ring_raw : 1ll;
    // integrate_dt > 0: the state stays in the kernel (read at tick 1, then q += clamp(dq) dt after every tick): only
    // the targets come from outside
    // (its own instantiation, INTEGRATE: the plain kernel carries none of this - a run-time switch cost it 0.2 us a tick)
    const double step_dt = INTEGRATE ? ticket->integrate_dt : 0.0, step_clamp = INTEGRATE ? ticket->max_speed : 0.0;
    constexpr bool integrate = INTEGRATE;
    double zstate[INTEGRATE ? N : 1];
#pragma unroll
    for (int j = 0; j < (INTEGRATE ? N : 1); ++j) zstate[j] = 0.0;
    // The rows come through LDS (stage_rows above): the wave's sixteen instances' rows are one contiguous block, copied
    // without a destination register, so the NEXT tick's rows are requested at the very top of a tick - a whole tick for
    // them to arrive - and read out of LDS at its end (round 5 / early round 6: each lane loaded two elements of its
    // instance's rows into registers 300 instructions into the tick and the quad handed them round by DPP).
    constexpr int RQ = (N + 2 * TEAM - 1) / (2 * TEAM);          // (rounds of the stores: lane r writes elements 2r, 2r + 1)
    using SQ = RowStage<N>;
    using SY = RowStage<NY>;
    __shared__ double stage[TEAM_WAVES][SQ::DOUBLES + SY::DOUBLES + 1];
    const int wv = tid >> 6, lane = tid & (WAVE - 1), iw = inst & 15;
    double* const zs = &stage[wv][0];
    double* const ys = &stage[wv][SQ::DOUBLES];
    const long long total_rows = ring * B;
    const bool al16 = (((unsigned long long)(uintptr_t)q | (unsigned long long)(uintptr_t)y) & 15ull) == 0ull
                      && ((B * N) & 1ll) == 0ll && ((B * NY) & 1ll) == 0ll;
    int shift_q = 0, shift_y = 0;
    auto request_rows = [&](const int k) __attribute__((always_inline)) {
        const long long row0 = ((long long)((k - 1) % (int)ring)) * B + b0 + 16 * wv;
        // (the rows read out last tick are in registers: nothing of the stage is still to be read)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!integrate || k == 1) shift_q = stage_rows<N>(q, row0, total_rows, zs, lane, al16);
        if constexpr (NY > 0) shift_y = stage_rows<NY>(y, row0, total_rows, ys, lane, al16);
    };
    auto read_rows = [&]() __attribute__((always_inline)) {      // (after the copies have landed: s_waitcnt vmcnt(0))
        const int iq = stage_row(iw, shift_q) * N;
#pragma unroll
        for (int j = 0; j < N; ++j) zn[j] = zs[iq + j];
        if constexpr (NY > 0) {
            const int iy = stage_row(iw, shift_y) * NY;
#pragma unroll
            for (int j = 0; j < NY; ++j) yn[j] = ys[iy + j];
        }
    };
    auto publish_done = [&](const int k) __attribute__((always_inline)) {
        // every lane's stores acknowledged, then the wave's own slot (no shared counter)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((tid & (WAVE - 1)) == 0)
            __hip_atomic_store(done + (blockIdx.x * TEAM_WAVES + (tid >> 6)), (unsigned)k, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        if (blockIdx.x == 0 && tid == 0) ticket->ticks_done = (unsigned)k;
    };
    int owed = 0;           // tick whose "done" slot is still to be published (0: none)
#pragma unroll 1
    for (int k = 1; k <= n_ticks; ++k) {
        CLIK_PHASE("res_poll_request");
        if (!have_next) {
            poll_for((unsigned)k);
            if (leave) break;
            asm volatile("" ::: "memory");
            request_rows(k);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_rows();
        }
        CLIK_PHASE("res_spread");
        double z[N], ydir[NY > 0 ? NY : 1];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if constexpr (INTEGRATE) z[j] = (k > 1) ? zstate[j] : zn[j];
            else z[j] = zn[j];
        }
#pragma unroll
        for (int j = 0; j < (NY > 0 ? NY : 1); ++j) ydir[j] = yn[j];
        have_next = false;
        // The next tick's rows are requested HERE, at the top of tick k, if the ticket word says the producer has
        // published tick k + 1; and the ticket word itself is requested again, to be READ in tick k + 1 (`seen`, copied at
        // the loop's end).  Round 5 requested it at the END of the tick into the loop-carried variable: the copy into that
        // variable's register at the loop's back edge made the compiler wait for the load at once - an uncached load's
        // whole latency on the critical path of every tick.  The price: what the wave knows about the
        // producer is one tick old - it runs fed ahead when the producer is TWO tickets ahead (rings of three or more
        // slots), else it falls back to publishing at once and polling.
        if (CLIK_RESIDENT_PIPELINE && k < n_ticks && seen >= (unsigned)(k + 1)) {
            __atomic_signal_fence(__ATOMIC_ACQUIRE);     // (the rows are requested only after the ticket value is in)
            request_rows(k + 1);
            have_next = true;
        }
        unsigned seen_next = seen;
        if (CLIK_RESIDENT_PIPELINE && k < n_ticks)
            seen_next = __hip_atomic_load(&ticket->in_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        double a0 = z[N - 1], a1 = z[N - 1];
        static_for<0, TEAM>([&](auto kc) __attribute__((always_inline)) {
            constexpr int kk = decltype(kc)::value;
            if constexpr (2 * kk < N) a0 = (r == kk) ? z[2 * kk] : a0;
            if constexpr (2 * kk + 1 < N) a1 = (r == kk) ? z[2 * kk + 1] : a1;
        });
        double v[N];
        bool in_tc;
        team_tick<SD>(&Sval, tk, z, ydir, a0, a1, r, inst, rc, sck, v, in_tc);
        CLIK_PHASE("res_accept");
        const bool ok0 = __builtin_amdgcn_mov_dpp((int)in_tc, QUAD_LANE0, 0xf, 0xf, true) != 0;
        // the accepted candidate in every lane of the quad (mode 0 lives in lane 0, mode 1 in lane 3): lane r stores
        // elements 2r and 2r + 1 of the row (write-through stores of one lane per element cost 32 bytes each at the
        // memory); INTEGRATE: clamped, and the state stepped - exactly pinv_rollout_static_team_kernel's Euler step
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double c0 = quad_perm_f64<0x00>(v[j]), c1 = quad_perm_f64<0xFF>(v[j]);
            double d = ok0 ? c0 : c1;
            if constexpr (INTEGRATE) {
                if (step_clamp > 0.0) d = fmax(fmin(d, step_clamp), -step_clamp);
                zstate[j] = fma(d, step_dt, z[j]);
            }
            v[j] = d;
        }
        CLIK_PHASE("res_publish");
        // The next tick's rows are read out of LDS HERE, before this tick's stores go out: a wave's memory operations
        // complete in order and are waited for by count, so "the copies have landed" waits for everything older - at this
        // point the previous tick's stores, a whole tick old.  (Round 5 took the rows at the top of the next tick: that
        // wait also covered THIS tick's write-through stores, issued fifty instructions earlier - a store's whole round
        // trip at the top of every tick; found in the listing, round 6.)
        if (have_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_rows();
        }
        pin_arrived(seen_next);
        if (owed != 0) {
            publish_done(owed);         // (the previous tick's stores: issued a whole tick ago)
            owed = 0;
        }
        CLIK_PHASE("res_store");
        if (valid) {
            // (write-through stores: visible to every agent once acknowledged)
            const long long orow = ((long long)((k - 1) % (int)ring)) * B + b0 + inst;
            static_for<0, RQ>([&](auto rc) __attribute__((always_inline)) {
                constexpr int rho = decltype(rc)::value;
                double s0 = v[N - 1], s1 = v[N - 1];
                static_for<0, TEAM>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int kk = decltype(kc)::value;
                    constexpr int j0 = 2 * TEAM * rho + 2 * kk;
                    if constexpr (j0 < N) s0 = (r == kk) ? v[j0] : s0;
                    if constexpr (j0 + 1 < N) s1 = (r == kk) ? v[j0 + 1] : s1;
                });
                const int e = 2 * TEAM * rho + 2 * r;
                if (e < N) __hip_atomic_store(dq + orow * N + e, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (e + 1 < N) __hip_atomic_store(dq + orow * N + e + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            });
            if (mode_out != nullptr && r == 0)
                __hip_atomic_store(mode_out + orow, ok0 ? 0 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (have_next) owed = k;        // published at the end of the next tick
        else publish_done(k);           // nobody has asked for the next tick yet (or this was the last): at once
        seen = seen_next;               // (requested at the top of this tick)
    }
    if (owed != 0) publish_done(owed);
    CLIK_PHASE_END();
}

// reference producer / test harness of the resident ticks: publishes tickets 1 .. n_ticks from the device, either as
// fast as it can (closed_loop = 0: the kernel never waits - its own per-tick cost) or each one only after every wave
// has finished the previous tick (closed_loop = 1: the hand-off both ways is on the critical path)
template <int UNIQUE = 0>        // (a template only so that the header may be included by several translation units)
__global__ __launch_bounds__(1024) void resident_feed_kernel(ResidentTicket* ticket, const unsigned* done, const int n_ticks,
                                                             const int closed_loop, const unsigned waves_per_tick,
                                                             const unsigned long long max_polls)
{
    // one thread per wave slot (strided when there are more slots than threads); thread 0 publishes
    __shared__ int s_leave;
    unsigned long long polls = 0;
    if (threadIdx.x == 0) s_leave = 0;
    __syncthreads();
    if (!closed_loop) {
        // every ticket at once: the resident kernel never waits - what is measured is its own per-tick cost (publishing
        // them one by one costs a store acknowledgement each, about 4 us: the kernel would be measured waiting for THAT)
        if (threadIdx.x == 0)
            __hip_atomic_store(&ticket->in_seq, (unsigned)n_ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#pragma unroll 1
    for (int k = 1; k <= n_ticks; ++k) {
        if (closed_loop) {
            for (unsigned w = threadIdx.x; w < waves_per_tick; w += blockDim.x) {
#pragma unroll 1
                for (;;) {
                    if (__hip_atomic_load(done + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned)(k - 1)) break;
                    if (__hip_atomic_load(&ticket->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                        s_leave = 1;
                        break;
                    }
                    if (++polls > max_polls) {
                        __hip_atomic_store(&ticket->stop, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        s_leave = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (s_leave) return;
        }
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&ticket->in_seq, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <const ShapeDesc& SD>
inline size_t team_rollout_lds_bytes(bool values = false)
{
    if (values) return 0;
    return ((size_t)StaticLayout<SD>::IMG_DOUBLES + (size_t)(SD.n_y > 0 ? SD.n_y : 0) * TEAM_INST) * sizeof(double);
}

}  // namespace clik
