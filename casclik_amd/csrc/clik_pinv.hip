// Kernel table of the PseudoInverseController path: AOT shape-specialised
// instantiations (this TU) + the dynamic-shape kernels (clik_pinv_dyn.hip).
#include "clik_pinv_kernels.hpp"

namespace clik {

// dynamic-shape launchers, instantiated in clik_pinv_dyn.hip
hipError_t dyn_solve_6(const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*, const double*, const double*, double*, double*, int32_t*, hipStream_t);
hipError_t dyn_solve_7(const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*, const double*, const double*, double*, double*, int32_t*, hipStream_t);
hipError_t dyn_solve_8(const DevSkill*, const WarmArgs&, const TickArgs&, long long, int, const double*, const double*, const double*, double*, double*, int32_t*, hipStream_t);
hipError_t dyn_rollout_8(const DevSkill*, const WarmArgs&, const double*, int, double, double, long long, int, double*, const double*, double*, int32_t*, hipStream_t);

// ---- AOT shapes ------------------------------------------------------------------
// One entry per skill structure that gets a guard-free kernel.  Adding a shape
// is one line here plus one line in kShapes[]; skills that match none of them
// run the DynShape kernel.  P/O/Y/Q are the CLIK_ROW_HAS_* feature flags.
namespace shapes {
constexpr int Q = CLIK_ROW_HAS_Q, P = CLIK_ROW_HAS_P, O = CLIK_ROW_HAS_O, Y = CLIK_ROW_HAS_Y;
constexpr int EQ = CLIK_CLS_EQ, SET = CLIK_CLS_SET;
//                                     n nt  cls            m           flags               const_j   aff fk qs ff md cl st
inline constexpr ShapeDesc kPos3N7  = {7, 1, {EQ},          {3},        {P | Y},            {0},       1, 1, 0, 1, 0, 0, 0};
inline constexpr ShapeDesc kPose6N7 = {7, 1, {EQ},          {6},        {P | O | Y},        {0},       1, 1, 2, 1, 0, 0, 0};
inline constexpr ShapeDesc kStackN7 = {7, 3, {SET, EQ, EQ}, {7, 6, 7},  {Q, P | O | Y, Q},  {1, 0, 1}, 1, 1, 2, 1, 1, 0, 0};
inline constexpr ShapeDesc kPos3N6  = {6, 1, {EQ},          {3},        {P | Y},            {0},       1, 1, 0, 1, 0, 0, 0};
inline constexpr ShapeDesc kPose6N6 = {6, 1, {EQ},          {6},        {P | O | Y},        {0},       1, 1, 2, 1, 0, 0, 0};
inline constexpr ShapeDesc kStackN6 = {6, 3, {SET, EQ, EQ}, {6, 6, 6},  {Q, P | O | Y, Q},  {1, 0, 1}, 1, 1, 2, 1, 1, 0, 0};
}  // namespace shapes

struct ShapeEntry {
    const ShapeDesc* sd;       // nullptr: dynamic kernel of width N
    int N;
    const char* name;
    solve_fn solve;
    rollout_fn rollout;
};

#define CLIK_STATIC_ENTRY(SD, NN) \
    {&shapes::SD, NN, #SD, &launch_solve<NN, StaticShape<shapes::SD>>, &launch_rollout<NN, StaticShape<shapes::SD>>}
#define CLIK_DYN_ENTRY(NN) {nullptr, NN, "dynamic", &dyn_solve_##NN, &dyn_rollout_8}

static const ShapeEntry kShapes[] = {
#ifdef CLIK_DEV_SINGLE   // developer builds: one instantiation, fast compile / ISA inspection
    CLIK_DEV_SINGLE
#else
    CLIK_STATIC_ENTRY(kPos3N7, 7),  CLIK_STATIC_ENTRY(kPose6N7, 7), CLIK_STATIC_ENTRY(kStackN7, 7),
    CLIK_STATIC_ENTRY(kPos3N6, 6),  CLIK_STATIC_ENTRY(kPose6N6, 6), CLIK_STATIC_ENTRY(kStackN6, 6),
    CLIK_DYN_ENTRY(6), CLIK_DYN_ENTRY(7), CLIK_DYN_ENTRY(8),
#endif
};
constexpr int kNumShapes = (int)(sizeof(kShapes) / sizeof(kShapes[0]));

static bool shape_equal(const ShapeDesc& a, const ShapeDesc& b)
{
    if (a.n != b.n || a.n_tasks != b.n_tasks || a.all_affine != b.all_affine || a.uses_fk != b.uses_fk ||
        a.quat_src != b.quat_src || a.feedforward != b.feedforward || a.multidim != b.multidim ||
        a.conv_last != b.conv_last || a.standard != b.standard)
        return false;
    for (int i = 0; i < a.n_tasks; ++i)
        if (a.cls[i] != b.cls[i] || a.m[i] != b.m[i] || a.flags[i] != b.flags[i] || a.const_j[i] != b.const_j[i])
            return false;
    return true;
}

// Index into kShapes for a skill: the matching static shape, else the dynamic
// kernel of the smallest sufficient width.  `allow_static` = 0 forces dynamic.
int pinv_pick_kernel(const DevSkill& S, int allow_static)
{
    if (allow_static && S.d.n_tasks <= SHAPE_MAX_TASKS)
        for (int k = 0; k < kNumShapes; ++k)
            if (kShapes[k].sd && shape_equal(*kShapes[k].sd, S.shape)) return k;
    int need = S.n;
    for (int ti = 0; ti < S.d.n_tasks; ++ti)
        if (S.d.tasks[ti].m > need) need = S.d.tasks[ti].m;
    for (int k = 0; k < kNumShapes; ++k)
        if (!kShapes[k].sd && kShapes[k].N >= need) return k;
    return -1;
}

const char* pinv_kernel_name(int k) { return (k >= 0 && k < kNumShapes) ? kShapes[k].name : "none"; }
int pinv_kernel_width(int k) { return (k >= 0 && k < kNumShapes) ? kShapes[k].N : 0; }

hipError_t pinv_launch_solve(int k, const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                             const double* q, const double* x, const double* y, double* dq,
                             double* dx, int32_t* mode, hipStream_t stream)
{
    if (k < 0 || k >= kNumShapes) return hipErrorInvalidValue;
    return kShapes[k].solve(dS, wa, tk, B, ny, q, x, y, dq, dx, mode, stream);
}

hipError_t pinv_launch_rollout(int k, const DevSkill* dS, const WarmArgs& wa, const double* d_tterms, int n_ticks,
                               double dt, double max_speed, long long B, int ny, double* q,
                               const double* y, double* dq, int32_t* mode, hipStream_t stream)
{
    if (k < 0 || k >= kNumShapes) return hipErrorInvalidValue;
    return kShapes[k].rollout(dS, wa, d_tterms, n_ticks, dt, max_speed, B, ny, q, y, dq, mode, stream);
}

int pinv_lds_slots_host(int N, int ny) { return pinv_lds_slots(N, ny); }

}  // namespace clik
