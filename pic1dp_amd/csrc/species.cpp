// species.cpp -- one species of the input (src/pic1dp_input.F90:43-72) turned into the constants the marker
// kernels use: the divisor constants of -f0'/f0 and the push formed exactly as the reference's compile-time
// folding forms them (src/pic1dp_interaction.F90:274-337), their correctly rounded reciprocals, which of the
// bit-identical division short cuts apply, and the folded constants of the one-exp form of -f0'/f0.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace pic1dp {

namespace {

bool is_pow2(double c) {
  if (!(c > 0.0) || !std::isfinite(c)) return false;
  int e;
  return std::frexp(c, &e) == 0.5;
}

}  // namespace

SpeciesConst make_species_const(const SpeciesInput &in, int s) {
  SpeciesConst c{};
  const double T = in.temperature, T2 = in.temperature2;
  c.Z = in.charge;
  c.m = in.mass;
  c.den = in.density;
  c.beam = 1.0 - c.den;
  c.v0 = in.v0;
  c.T = T;
  c.tm = T / c.m;
  c.tm2 = T2 / c.m;
  c.two_tm = 2.0 * T / c.m;
  c.two_tm2 = 2.0 * T2 / c.m;
  c.stm = std::sqrt(c.tm);
  c.stm2 = std::sqrt(c.tm2);
  c.r_m = 1.0 / c.m;
  c.r_T = 1.0 / T;
  c.r_tm = 1.0 / c.tm;
  c.r_tm2 = 1.0 / c.tm2;
  c.r_two_tm = 1.0 / c.two_tm;
  c.r_two_tm2 = 1.0 / c.two_tm2;
  c.r_stm = 1.0 / c.stm;
  c.r_stm2 = 1.0 / c.stm2;
  c.pow2 = is_pow2(c.m) && is_pow2(T) && is_pow2(c.tm) && is_pow2(c.tm2) && is_pow2(c.two_tm) &&
           is_pow2(c.two_tm2) && is_pow2(c.stm) && is_pow2(c.stm2);
  c.unit = c.m == 1.0 && T == 1.0 && T2 == 1.0 && c.tm == 1.0 && c.tm2 == 1.0 && c.stm == 1.0 &&
           c.stm2 == 1.0 && c.two_tm == 2.0 && c.two_tm2 == 2.0;
  if (const char *e = tuning_env("PIC1DP_UNIT_SPECIALISATION")) c.unit = c.unit && std::atoi(e) != 0;
  // general divisors: a/c through div_const (device_math.hpp) if every one of the
  // eight is in a sane range and a randomised host comparison with the true
  // quotient finds no difference (the theorem behind it holds for every finite
  // c; this guards the implementation, not the mathematics)
  c.fastc = 0;
  if (!c.pow2) {
    const double divisors[8] = {c.m, T, c.tm, c.tm2, c.two_tm, c.two_tm2, c.stm, c.stm2};
    bool ok = true;
    for (double d : divisors) {
      const double ad = std::fabs(d);
      ok = ok && ad > 0x1p-200 && ad < 0x1p+200 && host_divc_check(d, 0x5EEDull + static_cast<uint64_t>(s), 50000) == 0;
    }
    c.fastc = ok ? 1 : 0;
  }
  if (const char *e = tuning_env("PIC1DP_FAST_DIVC")) c.fastc = c.fastc && std::atoi(e) != 0;

  // One-exp form of -f0'/f0 (device_math.hpp dlnf0_one_exp).  L(v) = (fq2 v + fq1) v + fq0 is the log of the
  // ratio of the second Maxwellian to the first, tmp2 = (fm1 v + fm0) + (fd1 v + fd0) tanh(L / 2).
  c.one_exp = 0;
  if (in.iptcldist == 3) {  // bump-on-tail, src/pic1dp_interaction.F90:294-321
    const double h1 = 0.5 / c.tm, h2 = 0.5 / c.tm2;               // 1/(2T/m), 1/(2T2/m)
    const double lnK = std::log(c.beam * c.stm) - std::log(c.den * c.stm2);  // -inf / +inf for a missing beam / bulk
    c.fq2 = h1 - h2;
    c.fq1 = 2.0 * h2 * c.v0;
    c.fq0 = lnK - h2 * c.v0 * c.v0;
    c.fm1 = 0.5 * (c.r_tm + c.r_tm2);   // (A + B)/2, A = v/(T/m), B = (v - v0)/(T2/m)
    c.fm0 = -0.5 * c.v0 * c.r_tm2;
    c.fd1 = 0.5 * (c.r_tm2 - c.r_tm);   // (B - A)/2
    c.fd0 = -0.5 * c.v0 * c.r_tm2;
    c.one_exp = std::isnan(c.fq0) || std::isnan(c.fq2) ? 0 : 1;    // den = beam = 0 and the like: the reference's 0/0
  } else if (in.iptcldist == 2) {  // two-stream2, :278-292: q = (vp ep + vm em)/(ep + em) m/T, ep/em = exp(-2 v0 v/(T/m))
    c.fq2 = 0.0;
    c.fq1 = -2.0 * c.v0 * c.r_tm;
    c.fq0 = 0.0;
    c.fm1 = c.r_tm;             // (vm + vp)/2 m/T = v/(T/m)
    c.fm0 = 0.0;
    c.fd1 = 0.0;
    c.fd0 = c.v0 * c.r_tm;      // (vp - vm)/2 m/T
    c.one_exp = 1;
  }
  if (!std::isfinite(c.fq1) || !std::isfinite(c.fm1) || !std::isfinite(c.fd0)) c.one_exp = 0;
  if (const char *e = std::getenv("PIC1DP_DLNF0"))
    if (std::strcmp(e, "ref") == 0 || std::strcmp(e, "0") == 0) c.one_exp = 0;
  return c;
}

}  // namespace pic1dp
