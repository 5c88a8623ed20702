// placeholder until the QP kernel lands
#include "clik_device.hpp"
namespace clik {
int qp_pick_variant(int, int, int) { return -1; }
hipError_t qp_launch_solve(int, const DevSkill*, const TickArgs&, long long, const double*, const double*,
                           const double*, double*, double*, double*, int32_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t qp_launch_data(const DevSkill*, const TickArgs&, long long, const double*, const double*,
                          const double*, double*, double*, double*, double*, hipStream_t) { return hipErrorNotSupported; }
}
