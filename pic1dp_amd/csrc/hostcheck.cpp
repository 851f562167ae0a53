// hostcheck.cpp -- div_lx's and div_const's algorithm with the host's FMA (libm fma is exact) against the true
// quotient: create() vouches for a species' divisor constants with host_divc_check before it enables the fast form.
#include <cmath>

#include "check_values.hpp"
#include "kernels.hpp"

namespace pic1dp {

// div_const's algorithm with the host's FMA (libm fma is exact)
int64_t host_divc_check(double c, uint64_t seed, int64_t n) {
  const double rc = 1.0 / c;
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double a = divc_check_value(seed, i);
    const double q0 = a * rc;
    const double q1 = fma(fma(-c, q0, a), rc, q0);
    const double q = fma(fma(-c, q1, a), rc, q1);
    const double b = a / c;
    if (std::memcmp(&q, &b, 8) != 0) ++bad;
  }
  return bad;
}

// the same check with the host's FMA (libm fma is exact): div_lx's algorithm
int64_t host_div_check(double lx, int nx, uint64_t seed, int64_t n) {
  const double y = 1.0 / lx;
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double x = div_check_value(seed, i, lx, nx);
    const double ax = fabs(x);
    double a;
    if (ax > 0x1p-500 && ax < 0x1p+500) {
      const double q0 = x * y;
      const double q1 = fma(fma(-lx, q0, x), y, q0);
      a = fma(fma(-lx, q1, x), y, q1);
    } else {
      a = x / lx;
    }
    const double b = x / lx;
    if (std::memcmp(&a, &b, 8) != 0) ++bad;
  }
  return bad;
}

// the histogram geometry of output_ptcldist with its two constant divisors prepared (kernels.hpp DistGeom)
DistGeom make_dist_geom(double lx, double vmax, int nxo, int nvo) {
  DistGeom dg{};
  dg.lx = lx;
  dg.vmax = vmax;
  dg.nxo = nxo;
  dg.nvo = nvo;
  dg.rlx = 1.0 / lx;
  dg.dv = vmax * 2.0;  // src/pic1dp_output.F90:247
  dg.rdv = 1.0 / dg.dv;
  const double ad = std::fabs(dg.dv);
  dg.vfast = (ad > 0x1p-200 && ad < 0x1p+200 && host_divc_check(dg.dv, 0xD157ull, 50000) == 0) ? 1 : 0;
  return dg;
}

}  // namespace pic1dp
