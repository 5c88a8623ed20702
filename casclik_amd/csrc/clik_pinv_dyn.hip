// Dynamic-shape instantiations of the PseudoInverseController kernels (any
// skill within the limits of include/clik.h; run-time guards).  Separate TU so
// it compiles in parallel with the shape-specialised kernels.
#include "clik_pinv_kernels.hpp"

namespace clik {
#define CLIK_DYN_SOLVE(NN)                                                                                 \
    hipError_t dyn_solve_##NN(const LaunchArgs& a, const TickArgs& tk, long long B, const double* q,       \
                              const double* x, const double* y, double* dq, double* dx, int32_t* mode,      \
                              hipStream_t stream)                                                           \
    {                                                                                                       \
        return launch_solve<NN, DynShape>(a, tk, B, q, x, y, dq, dx, mode, stream);                         \
    }
CLIK_DYN_SOLVE(6)
CLIK_DYN_SOLVE(7)
CLIK_DYN_SOLVE(8)

// the rollout uses the widest kernel for every dynamic skill (n <= 8)
hipError_t dyn_rollout_8(const LaunchArgs& a, const double* d_tterms, int n_ticks, double dt, double max_speed,
                         long long B, double* q, const double* y, double* dq, int32_t* mode, hipStream_t stream)
{
    return launch_rollout<8, DynShape>(a, d_tterms, n_ticks, dt, max_speed, B, q, y, dq, mode, stream);
}
}  // namespace clik
