// Ahead-of-time instantiations of the shape-specialised ReactiveQPController kernels (clik_qp_static.hpp) for the
// BASELINE skill structures of clik_shapes_gen.hpp, and the run-time copies of the shape predicates the API asks.
// Its own translation unit: it compiles beside clik_qp.hip (the dynamic-shape kernels) instead of behind it - together
// they were one 13-minute compile.
#include "clik_qp_static.hpp"

namespace clik {

namespace shapes {
#include "clik_shapes_gen.hpp"
}  // namespace shapes

// ---- shape-specialised QP kernels (clik_qp_static.hpp): AOT table -------------------------
struct QpStaticEntry {
    const char* name;
    const ShapeDesc* sd;
    qp_static_fn solve;
    qp_static_rollout_fn rollout;
};
#define CLIK_QP_STATIC_ENTRY(S) {"qp_static_" #S, &shapes::S, &launch_qp_static<shapes::S>, &launch_qp_rollout_static<shapes::S>},
static const QpStaticEntry kQpShapes[] = {
#ifdef CLIK_GENERATED_QP_SHAPES
    CLIK_GENERATED_QP_SHAPES(CLIK_QP_STATIC_ENTRY)
#endif
    {nullptr, nullptr, nullptr, nullptr}
};
constexpr int kNumQpShapes = (int)(sizeof(kQpShapes) / sizeof(kQpShapes[0])) - 1;

int qp_pick_static(const ShapeDesc& sd)
{
    for (int k = 0; k < kNumQpShapes; ++k)
        if (shape_equal(*kQpShapes[k].sd, sd)) return k;
    return -1;
}
const char* qp_static_name(int k) { return (k >= 0 && k < kNumQpShapes) ? kQpShapes[k].name : "none"; }
bool qp_box_family_rt(const ShapeDesc& sd) { return CLIK_QP_BOX_OK(sd); }
// rows the shape-specialised kernels hand to their active set: soft equalities folded, hard bounds on the same state merged
int qp_plan_rows_rt(const ShapeDesc& sd) { return make_qp_plan(sd).nr; }
// 64-double LDS slots a shape-specialised QP kernel keeps behind the skill image (QpLayout<SD>::SLOTS on the run-time
// copy of the shape): the primal families (bound-constrained, mixed) keep no dual Hessian there
int qp_layout_slots_rt(const ShapeDesc& sd)
{
    const QpPlanS p = make_qp_plan(sd);
    const bool primal = CLIK_QP_BOX_OK(sd) || CLIK_QP_MIXED_OK(sd);
    const int n = sd.n, ny = sd.n_y > 0 ? sd.n_y : 0, nra = p.nr > 0 ? p.nr : 1, nsa = p.ns > 0 ? p.ns : 1;
    const int nt = nra * (nra + 1) / 2;
    return n + ny + (primal ? 0 : nt) + 2 * nra + (primal ? 0 : nra) + (primal ? 0 : nra * n) + nsa;
}
hipError_t qp_launch_static(int k, const void* d_img, const TickArgs& tk, long long B, const double* q,
                            const double* x, const double* y, double* dq, double* dx, double* slack,
                            int32_t* status, int32_t* hot_set, int use_hot, hipStream_t stream,
                            const double* t_inst)
{
    if (k < 0 || k >= kNumQpShapes) return hipErrorInvalidValue;
    return kQpShapes[k].solve(d_img, tk, B, q, x, y, dq, dx, slack, status, hot_set, use_hot, stream, t_inst);
}

hipError_t qp_launch_rollout_static(int k, const void* d_img, const double* d_tterms, int n_ticks, double dt,
                                    double max_speed, long long B, double* q, const double* y, double* dq,
                                    double* slack, int32_t* status, double* x, double* dx, hipStream_t stream,
                                    int stages)
{
    if (k < 0 || k >= kNumQpShapes) return hipErrorInvalidValue;
    return kQpShapes[k].rollout(d_img, d_tterms, n_ticks, dt, max_speed, B, q, y, dq, slack, status, x, dx, stream,
                                stages);
}

}  // namespace clik
