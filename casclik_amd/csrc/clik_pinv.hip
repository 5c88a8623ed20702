// Kernel table of the PseudoInverseController path: AOT shape-specialised
// instantiations (this TU, shapes from clik_shapes_gen.hpp) + the dynamic-shape
// kernels (clik_pinv_dyn.hip).
#define CLIK_LARGE_BATCH_VARIANT 1      // the ahead-of-time shapes also get the large-batch (occupancy 2) kernel
#include "clik_pinv_kernels.hpp"

namespace clik {

// dynamic-shape launchers, instantiated in clik_pinv_dyn.hip
hipError_t dyn_solve_6(const LaunchArgs&, const TickArgs&, long long, const double*, const double*, const double*, double*, double*, int32_t*, hipStream_t);
hipError_t dyn_solve_7(const LaunchArgs&, const TickArgs&, long long, const double*, const double*, const double*, double*, double*, int32_t*, hipStream_t);
hipError_t dyn_solve_8(const LaunchArgs&, const TickArgs&, long long, const double*, const double*, const double*, double*, double*, int32_t*, hipStream_t);
hipError_t dyn_rollout_8(const LaunchArgs&, const double*, int, double, double, long long, double*, const double*, double*, int32_t*, hipStream_t);

// ---- AOT shapes ------------------------------------------------------------------
// One ShapeDesc per (robot structure x skill structure) that gets a guard-free
// kernel.  The initialisers are GENERATED (tools/gen_shapes.py prints what
// clik_shape_describe() derives for a skill) - add a skill there to give it a
// specialised kernel; skills that match no entry run the dynamic kernel.
namespace shapes {
#include "clik_shapes_gen.hpp"
}  // namespace shapes

// ---- compile-time checks of the plans the AOT shapes get (regression guards) ----------------
#if defined(CLIK_GENERATED_SHAPES) && !defined(CLIK_DEV_SINGLE)
namespace plan_checks {
using shapes::kStackIiwa;
// config 3, mode 0 (set inactive): [pose (twice); centering] -> the centering task projects through the
// duplicated pose rows (6x6 push-through form), no Gram build
static_assert(Plan<kStackIiwa, 0u>::mode.t[0].skip && Plan<kStackIiwa, 0u>::mode.t[0].cone, "mode 0: set is a cone test");
static_assert(Plan<kStackIiwa, 0u>::mode.t[1].quirk && Plan<kStackIiwa, 0u>::mode.t[1].push_times == 2, "first equality twice");
static_assert(Plan<kStackIiwa, 0u>::mode.dup_task == 1 && Plan<kStackIiwa, 0u>::mode.dup_times == 2, "duplicated-row stack");
// mode 1 (set active): set rows stay explicit (7 <= 7), the pose projects through them with the
// host-side pinv of the unit rows, then the stack turns Gram (13 rows) for the centering task
static_assert(Plan<kStackIiwa, 1u>::mode.t[0].push_times == 1 && !Plan<kStackIiwa, 1u>::mode.t[0].gram_after, "set rows explicit");
static_assert(!Plan<kStackIiwa, 1u>::mode.t[1].quirk && Plan<kStackIiwa, 1u>::mode.t[1].wide_const_task == 0, "pose behind the set");
static_assert(Plan<kStackIiwa, 1u>::mode.t[2].gram_before && Plan<kStackIiwa, 1u>::mode.dup_task == -1, "Gram consumer");
static_assert(Plan<kStackIiwa, 1u>::mode.helper_ok && shape_split_ok<kStackIiwa>(), "role split applies");
static_assert(shape_unit(kStackIiwa, 0) && !shape_unit(kStackIiwa, 1) && shape_unit(kStackIiwa, 2), "joint-space tasks");
// mode order of pseudo_inverse.py:107-130 for two sets: 00, 10, 01, 11 with set 0 = bit 0
static_assert(shape_mode_act(shapes::kStackUr5, 0) == 0u && shape_mode_act(shapes::kStackUr5, 1) == 1u, "one set: 0, 1");
}  // namespace plan_checks
#endif

struct ShapeEntry {
    const ShapeDesc* sd;       // nullptr: dynamic kernel of width N
    int N;
    const char* name;
    solve_fn solve;
    rollout_fn rollout;
};

#define CLIK_STATIC_ENTRY(SD) \
    {&shapes::SD, shapes::SD.n, #SD, &launch_solve_static<shapes::SD>, &launch_rollout_static<shapes::SD>},
#define CLIK_DYN_ENTRY(NN) {nullptr, NN, "dynamic", &dyn_solve_##NN, &dyn_rollout_8},

static const ShapeEntry kShapes[] = {
#ifdef CLIK_DEV_SINGLE   // developer builds: one instantiation, fast compile / ISA inspection
    CLIK_DEV_SINGLE
#else
    CLIK_GENERATED_SHAPES(CLIK_STATIC_ENTRY)
    CLIK_DYN_ENTRY(6) CLIK_DYN_ENTRY(7) CLIK_DYN_ENTRY(8)
#endif
};
constexpr int kNumShapes = (int)(sizeof(kShapes) / sizeof(kShapes[0]));

// Index into kShapes for a skill: the matching static shape, else the dynamic
// kernel of the smallest sufficient width.  `allow_static` = 0 forces dynamic.
int pinv_pick_kernel(const DevSkill& S, int allow_static)
{
    if (allow_static && S.d.n_tasks <= SHAPE_MAX_TASKS && S.d.n_x == 0)
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
const char* pinv_static_variant(const ShapeDesc& sd, int mode_parallel, long long B) { return static_variant(sd, mode_parallel, B); }
bool shape_team_ok_rt(const ShapeDesc& sd) { return shape_team_ok(sd); }
bool shape_quad_front_ok_rt(const ShapeDesc& sd) { return shape_quad_front_ok(sd); }
long long pinv_team_max_batch() { return kTeamMaxBatch; }
bool shape_value_lane_ok_rt(const ShapeDesc& sd) { return shape_value_lane_ok(sd); }
long long pinv_value_lane_max_batch() { return kValueLaneMaxBatch; }
int pinv_kernel_width(int k) { return (k >= 0 && k < kNumShapes) ? kShapes[k].N : 0; }
int pinv_kernel_is_static(int k) { return (k >= 0 && k < kNumShapes && kShapes[k].sd) ? 1 : 0; }

hipError_t pinv_launch_solve(int k, const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,
                             const double* x, const double* y, double* dq, double* dx, int32_t* mode,
                             hipStream_t stream)
{
    if (k < 0 || k >= kNumShapes) return hipErrorInvalidValue;
    return kShapes[k].solve(a, tk, B, q, x, y, dq, dx, mode, stream);
}

hipError_t pinv_launch_rollout(int k, const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt,
                               double max_speed, long long B, double* q, const double* y, double* dq,
                               int32_t* mode, hipStream_t stream)
{
    if (k < 0 || k >= kNumShapes) return hipErrorInvalidValue;
    return kShapes[k].rollout(a, d_tterms, n_ticks, dt, max_speed, B, q, y, dq, mode, stream);
}

int pinv_lds_slots_host(int N, int ny) { return pinv_lds_slots(N, ny); }

// resident ticks (clik_pinv_team.hpp): waves per tick of the four-lanes-per-instance launch, and the reference producer
int team_waves_rt(long long B) { return (int)(((B + TEAM_INST - 1) / TEAM_INST) * TEAM_WAVES); }
hipError_t launch_ticket_feed(void* ticket, const unsigned* done, int n_ticks, int closed_loop, unsigned waves_per_tick,
                              unsigned long long timeout_ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(resident_feed_kernel<0>, dim3(1), dim3(waves_per_tick < 1024u ? ((waves_per_tick + 63u) & ~63u) : 1024u),
                       0, stream, (ResidentTicket*)ticket, done, n_ticks, closed_loop, waves_per_tick, timeout_ticks);
    return hipGetLastError();
}

}  // namespace clik
