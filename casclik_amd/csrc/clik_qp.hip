// Batched ReactiveQPController tick on gfx950, dynamic-shape kernels: the variant table and its look-up.  The kernels
// and their launch templates are clik_qp_dyn.hpp; their instantiations compile in clik_qp_dyn_[a-d].hip.
#include "clik_qp_dyn.hpp"

namespace clik {

// (instantiated elsewhere: nothing of the kernels is compiled in this translation unit)
CLIK_QP_VARIANTS_A(CLIK_QP_DECL_EXACT, CLIK_QP_DECL_GUARD, CLIK_QP_DECL_GLOBAL)
CLIK_QP_VARIANTS_B(CLIK_QP_DECL_EXACT, CLIK_QP_DECL_GUARD, CLIK_QP_DECL_GLOBAL)
CLIK_QP_VARIANTS_C(CLIK_QP_DECL_EXACT, CLIK_QP_DECL_GUARD, CLIK_QP_DECL_GLOBAL)
CLIK_QP_VARIANTS_D(CLIK_QP_DECL_EXACT, CLIK_QP_DECL_GUARD, CLIK_QP_DECL_GLOBAL)

#define CLIK_QP_EXACT(N, NC) {N, NC, 1, &qp_solve_launch<N, NC, true>, &qp_data_launch<N, NC>},
#define CLIK_QP_GUARD(N, NC) {N, NC, 0, &qp_solve_launch<N, NC, false>, &qp_data_launch<N, NC>},
#define CLIK_QP_GLOBAL(N, NC) {N, NC, 2, &qp_solve_launch_gws<N, NC>, &qp_data_launch_gws<N, NC>},
static const QpVariant kQpVariants[] = {
    CLIK_QP_VARIANTS_A(CLIK_QP_EXACT, CLIK_QP_GUARD, CLIK_QP_GLOBAL)
    CLIK_QP_VARIANTS_B(CLIK_QP_EXACT, CLIK_QP_GUARD, CLIK_QP_GLOBAL)
    CLIK_QP_VARIANTS_C(CLIK_QP_EXACT, CLIK_QP_GUARD, CLIK_QP_GLOBAL)
    CLIK_QP_VARIANTS_D(CLIK_QP_EXACT, CLIK_QP_GUARD, CLIK_QP_GLOBAL)
};
constexpr int kNumQpVariants = (int)(sizeof(kQpVariants) / sizeof(kQpVariants[0]));

// smallest variant that holds the skill; -1 if none (more than 16 rows)
int qp_pick_variant(int n, int nv, int nc)
{
    (void)nv;
    int best = -1;
    for (int k = 0; k < kNumQpVariants; ++k)
        if (kQpVariants[k].exact && kQpVariants[k].N == n && kQpVariants[k].NC == nc) return k;
    for (int k = 0; k < kNumQpVariants; ++k) {
        const QpVariant& v = kQpVariants[k];
        if (v.exact == 1 || v.N < n || v.NC < nc) continue;
        // task_eval works on N x N blocks: a constraint may have up to N rows
        // (the variants with their work area in LDS first, the global-memory ones when nothing else fits)
        const int cost = (v.exact == 2 ? 100000 : 0) + v.N * 100 + v.NC;
        const int best_cost = best < 0 ? 0 : (kQpVariants[best].exact == 2 ? 100000 : 0) + kQpVariants[best].N * 100 + kQpVariants[best].NC;
        if (best < 0 || cost < best_cost) best = k;
    }
    return best;
}

int qp_variant_width(int k) { return (k >= 0 && k < kNumQpVariants) ? kQpVariants[k].N : 0; }

size_t qp_variant_lds(int k, int ny)
{
    if (k < 0 || k >= kNumQpVariants) return 0;
    const QpVariant& v = kQpVariants[k];
    if (v.exact == 2) return 0;          // (work area in global memory)
    const int wk = v.NC * v.N > 6 * v.N ? v.NC * v.N : 6 * v.N;
    return (size_t)(v.N + ny + wk + v.NC * (v.NC + 1) / 2 + 3 * v.NC) * WAVE * sizeof(double);
}

hipError_t qp_launch_solve(int k, const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                           const double* q, const double* x, const double* y, double* dq, double* dx,
                           double* slack, int32_t* status, hipStream_t stream, GwsOwner* owner)
{
    if (k < 0 || k >= kNumQpVariants) return hipErrorInvalidValue;
    return kQpVariants[k].solve(dS, wa, tk, B, ny, q, x, y, dq, dx, slack, status, stream, owner);
}

hipError_t qp_launch_data(int k, const DevSkill* dS, const WarmArgs& wa, const TickArgs& tk, long long B, int ny,
                          const double* q, const double* x, const double* y, double* Hd, double* A, double* lb,
                          double* ub, hipStream_t stream, GwsOwner* owner)
{
    if (k < 0 || k >= kNumQpVariants) return hipErrorInvalidValue;
    return kQpVariants[k].data(dS, wa, tk, B, ny, q, x, y, Hd, A, lb, ub, stream, owner);
}
bool qp_variant_uses_workspace(int k) { return k >= 0 && k < kNumQpVariants && kQpVariants[k].exact == 2; }

}  // namespace clik
