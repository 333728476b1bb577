// placeholder (replaced below in the same round)
#include "common.hpp"
using namespace mlmc;
extern "C" {
int mlmc_maxent_solve(const mlmc_basis *, const double *, const double *, int32_t, double, double, const mlmc_maxent_opts *,
                      const double *, int32_t, double *, double *, mlmc_maxent_info *) { return fail("maxent: not built yet"); }
int mlmc_density_eval(const mlmc_basis *, const double *, const double *, int32_t, const double *, int64_t, double *, int) { return fail("maxent: not built yet"); }
int mlmc_density_integrate(const mlmc_basis *, const double *, const double *, int32_t, const double *, const double *, int64_t, int32_t, double *) { return fail("maxent: not built yet"); }
}
