// Dynamic-shape ReactiveQPController kernels, group C of the variants (clik_qp_dyn.hpp): explicit instantiations.
#include "clik_qp_dyn.hpp"

namespace clik {
CLIK_QP_VARIANTS_C(CLIK_QP_DEF_EXACT, CLIK_QP_DEF_GUARD, CLIK_QP_DEF_GLOBAL)
}  // namespace clik
