// device_math.hpp -- device functions shared by the gfx950 kernel translation units (kernels_*.hip) and by the
// measurement library (probe.hip): loads/stores, the exact divisions by run-time constants, cell location,
// periodic wrap, the table-driven exp, -f0'/f0, the push and the deposit of one marker, the LDS rho tile.
// Everything here is inline device code in an unnamed namespace: every translation unit gets its own copy, no
// device symbol crosses a translation unit (the library is linked without relocatable device code).
#pragma once
#include "kernels.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace pic1dp {
namespace {

constexpr int MODE_DF_NL = 0;   // deltaf=1, linear=0
constexpr int MODE_DF_LIN = 1;  // deltaf=1, linear=1
constexpr int MODE_FULLF = 2;   // deltaf=0, linear=0

__device__ __forceinline__ void lds_add(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Marker arrays are streamed: every element is touched once per kernel, so the
// loads and stores carry the non-temporal hint (global_load/store_dwordx4 ... nt),
// which on MI355X raises the streaming rate of these access shapes by 5-15 %
// (tools/probe_sweep.py).  PIC1DP_NT=0 at compile time restores plain accesses.
#ifndef PIC1DP_NT
#define PIC1DP_NT 1
#endif
// k_step_one runs two workgroups of 768 threads per CU = 6 waves per SIMD: its register allocation is
// held to 512 / 6 VGPRs (4 spilled registers; measured best, tools/ab_waves.sh).  k_step_half / k_step_full
// are left alone: held to the same budget k_step_full spills 15-19 registers and runs 1.12-1.39 ms instead
// of 0.96-0.98 ms (it then keeps 96 VGPRs and fewer waves).
// 0: every x / lx through the hardware division sequence (tuning / cross-check builds).  A compile-time choice:
// as a run-time flag the second code path cost the marker kernels registers (four more spilled in k_step_one)
#ifndef PIC1DP_FAST_DIV
#define PIC1DP_FAST_DIV 1
#endif
#ifndef PIC1DP_WAVES_PER_EU
#define PIC1DP_WAVES_PER_EU 6
#endif
#define PIC1DP_SIX_WAVES __attribute__((amdgpu_waves_per_eu(PIC1DP_WAVES_PER_EU)))
typedef double v2d __attribute__((ext_vector_type(2)));
// scheduling fence: independent instruction chains on either side are not interleaved (register pressure)
#define PAIR_FENCE() __builtin_amdgcn_sched_barrier(0)

template <bool NT>
__device__ __forceinline__ double2 ld2t(const double2 *p) {
  if constexpr (NT) {
    const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
    return make_double2(t.x, t.y);
  } else {
    return *p;
  }
}
template <bool NT>
__device__ __forceinline__ void st2t(double2 *p, double a, double b) {
  if constexpr (NT) {
    v2d t;
    t.x = a;
    t.y = b;
    __builtin_nontemporal_store(t, reinterpret_cast<v2d *>(p));
  } else {
    *p = make_double2(a, b);
  }
}
__device__ __forceinline__ double2 ld2(const double2 *p) { return ld2t<PIC1DP_NT != 0>(p); }
__device__ __forceinline__ void st2(double2 *p, double a, double b) { st2t<PIC1DP_NT != 0>(p, a, b); }

// a / c for a run-time constant c, correctly rounded, without the hardware
// division sequence: rc = RN(1/c) from the host, q0 = RN(a*rc) is within 2 ulp,
// one FMA correction (r0 = a - c*q0 exact, q1 = RN(q0 + r0*rc)) makes it
// faithful, and by Markstein's theorem (faithful q, correctly rounded
// reciprocal, exact residual) the second correction returns exactly RN(a/c), the
// reference's quotient.  5 full-rate FP64 ops instead of ~14 issue slots.  Zero /
// tiny / huge dividends (outside the theorem's no-underflow premise) and
// constants the host could not vouch for (fast = 0) take the hardware division.
__device__ __forceinline__ double div_const(double a, double c, double rc, int fast) {
  const double aa = fabs(a);
  if (fast && aa > 0x1p-500 && aa < 0x1p+500) {
    const double q0 = a * rc;
    const double r0 = fma(-c, q0, a);
    const double q1 = fma(r0, rc, q0);
    const double r1 = fma(-c, q1, a);
    return fma(r1, rc, q1);
  }
  return a / c;
}

// Division by a species constant c, bit-identical to a / c in all three forms:
//   POW2 = 2  a                 (unit species: m = T = T2 = 1, so T/m = sqrt(T/m) = 1
//                                and 2T/m = 2 -- the reference's default input)
//   POW2 = 1  a * (1/c)         (every divisor constant is a power of two)
//   POW2 = 0  general constants: through a divider object D, either
//       DivTrue   the hardware's IEEE division, or
//       DivFast   div_const's five operations WITHOUT its per-division range test: the
//                 divider only tracks the smallest and largest |dividend| it was given, and
//                 the caller checks ok() ONCE after the whole -f0'/f0 + push evaluation (ten
//                 divisions for bump-on-tail); a marker that fails (a zero, tiny or huge
//                 dividend: measure zero) is evaluated again with DivTrue.  One test per
//                 marker instead of ten, and ten independent FMA chains with no branches
//                 between them.
struct DivTrue {
  __device__ __forceinline__ double operator()(double a, double c, double) const { return a / c; }
};
struct DivFast {
  double lo = 1.0, hi = 1.0;
  __device__ __forceinline__ double operator()(double a, double c, double rc) {
    const double aa = fabs(a);
    lo = fmin(lo, aa);
    hi = fmax(hi, aa);
    const double q0 = a * rc;
    const double r0 = fma(-c, q0, a);
    const double q1 = fma(r0, rc, q0);
    const double r1 = fma(-c, q1, a);
    return fma(r1, rc, q1);
  }
  // Markstein's premises (no underflow in the residuals) hold for every dividend seen
  __device__ __forceinline__ bool ok() const { return lo > 0x1p-500 && hi < 0x1p+500; }
};

template <int POW2, class D>
__device__ __forceinline__ double divc(double a, double c, double rc, D &d) {
  if constexpr (POW2 == 2) {
    return a;
  } else if constexpr (POW2 == 1) {
    return a * rc;
  } else {
    return d(a, c, rc);
  }
}
// the same for the constants 2T/m, 2T2/m (= 2 for a unit species)
template <int POW2, class D>
__device__ __forceinline__ double divh(double a, double c, double rc, D &d) {
  if constexpr (POW2 == 2) {
    return a * 0.5;
  } else if constexpr (POW2 == 1) {
    return a * rc;
  } else {
    return d(a, c, rc);
  }
}

// x / lx, correctly rounded, without the hardware division sequence.
// y = RN(1/lx).  q0 = RN(x*y) is within 2 ulp of x/lx; one FMA correction
// (r0 = x - lx*q0, q1 = RN(q0 + r0*y)) makes it faithful; by Markstein's theorem
// (faithful q, |y - 1/b| < 2^-53/b, r = a - b*q exact, q' = RN(q + r*y)  =>
// q' = RN(a/b)) the second correction returns exactly RN(x/lx), the value the
// reference's division produces.  5 full-rate FP64 ops instead of ~14 slots.
// Tiny / huge / zero operands (outside the theorem's no-underflow premise) take
// the hardware division.  tests: test_exact_division_by_lx (GPU and host).
__device__ __forceinline__ double div_lx(double x, const GridConst &g) {
  const double ax = fabs(x);
  if (PIC1DP_FAST_DIV && ax > 0x1p-500 && ax < 0x1p+500) {
    const double y = g.rlx;
    const double q0 = x * y;
    const double r0 = fma(-g.lx, q0, x);
    const double q1 = fma(r0, y, q0);
    const double r1 = fma(-g.lx, q1, x);
    return fma(r1, y, q1);
  }
  return x / g.lx;
}

// cell index and left weight of position x (already inside [0, lx]):
// sx = x/lx*nx; ix = floor(sx); wl = 1 - (sx - ix)
// src/pic1dp_interaction.F90:106-108 and :250-252
// The division skips div_lx's range test (one compare chain and branch per locate, four locates per
// marker in k_step_one): positions reaching locate are wrapped into [0, lx] or come from memory in that
// range, and outside the theorem's range the outcome cannot change anyway -- for |x| < 2^-500 (0, -0,
// subnormals included) any quotient within a few ulp gives s < 2^-490, hence ix = 0 and wl = 1 exactly
// as the IEEE quotient does; NaN stays NaN and folds to cell 0 below like NaN / lx; |x| > 2^500 would
// index outside the grid in the reference and folds to cell 0 here either way.
__device__ __forceinline__ double div_lx_unchecked(double x, const GridConst &g) {
#if !PIC1DP_FAST_DIV
  return x / g.lx;
#endif
  const double y = g.rlx;
  const double q0 = x * y;
  const double r0 = fma(-g.lx, q0, x);
  const double q1 = fma(r0, y, q0);
  const double r1 = fma(-g.lx, q1, x);
  return fma(r1, y, q1);
}
__device__ __forceinline__ void locate(double x, const GridConst &g, int &ix, double &wl) {
  const double s = div_lx_unchecked(x, g) * g.dnx;
  const double fl = floor(s);
  ix = static_cast<int>(fl);
  wl = 1.0 - (s - fl);
  // memory safety only (x + lx may round to lx; NaN): fold to cell 0
  if (static_cast<unsigned>(ix) >= static_cast<unsigned>(g.nx)) ix = 0;
}

// periodic wrap: x = mod(x, lx); if (x < 0) x = x + lx
// src/pic1dp_interaction.F90:102-104.  fmod is exact; the common cases are
// resolved without the general routine, with identical results:
//   0 <= x < lx          -> x
//   lx <= x < 2 lx       -> x - lx       (exact, Sterbenz)
//   -lx < x < 0          -> fmod = x, then x + lx (one rounding, as the reference)
__device__ __forceinline__ double wrap(double x, double lx) {
  if (x >= 0.0 && x < lx) return x;
  if (x >= lx && x < 2.0 * lx) return x - lx;
  if (x < 0.0 && x > -lx) return x + lx;
  double r = fmod(x, lx);
  if (r < 0.0) r = r + lx;
  return r;
}

// exp(x) for the weight equation's arguments x = -(v -+ v0)^2 / (2T/m) <= 0 (src/pic1dp_interaction.F90:
// 278-321).  The library exp costs ~25 FP64 instructions, and the one-pass kernel evaluates four per
// marker at an FP64-issue-bound pace.  Table-driven instead: x = n ln2/64 + r, |r| <= ln2/128,
// e^x = 2^(n>>6) * T[n&63] * e^r with T[j] = 2^(j/64) as a correctly rounded hi + lo pair (LDS, 1 KiB
// per workgroup) and e^r - 1 by its Taylor polynomial of degree 5 (truncation 3.5e-17): 13 VALU
// instructions + one ds_read_b128, within 1 ulp of libm on [-745, 0] (test_device_exp_against_libm;
// 97 % of arguments bit-identical) -- the same distance the library exp keeps.  Arguments below -750
// (a run that has blown up) are clamped: the result underflows to 0 either way.
__device__ const double2 kExpTab[64] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56},
    {0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55},
    {0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57},
    {0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54},
    {0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59},
    {0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54},
    {0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54},
    {0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55},
    {0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55},
    {0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54},
    {0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55},
    {0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54},
    {0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55},
    {0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55},
    {0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54},
    {0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55},
    {0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54},
    {0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54},
    {0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56},
    {0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55},
    {0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58},
    {0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59},
    {0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56},
    {0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56},
    {0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54},
    {0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55},
    {0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54},
    {0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54},
    {0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54},
    {0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54},
    {0x1.6623882552225p+0, -0x1.bb60987591c34p-54},
    {0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54},
    {0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57},
    {0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55},
    {0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54},
    {0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55},
    {0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56},
    {0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54},
    {0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54},
    {0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54},
    {0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55},
    {0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57},
    {0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54},
    {0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56},
    {0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54},
    {0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54},
    {0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54},
    {0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54},
    {0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57},
    {0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56},
    {0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55},
    {0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55},
    {0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54},
    {0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56},
    {0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54},
    {0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55},
    {0x1.da9e603db3285p+0, 0x1.c2300696db532p-54},
    {0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54},
    {0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55},
    {0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54},
    {0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54},
    {0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54},
    {0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55}};

__device__ __forceinline__ double2 *exp_table() {
  __shared__ __attribute__((aligned(16))) double2 sExpT[64];
  return sExpT;
}
// every kernel that evaluates -f0'/f0 calls this before its first workgroup barrier
__device__ __forceinline__ void exp_table_init() {
  if (threadIdx.x < 64) exp_table()[threadIdx.x] = kExpTab[threadIdx.x];
}
// HI_ONLY (the one-exp form of -f0'/f0, whose result owes the CPU arithmetic nothing beyond rounding): the
// kernel that evaluates it is LDS-bound (three table reads per marker among eight atomics and four gathers), and
// half the bytes per read are worth 2 % of its time (profiles/r03/experiments/ab_variants.log)
template <bool HI_ONLY = false>
__device__ __forceinline__ double pexp(double x) {
  x = fmax(x, -750.0);
  const double fn = rint(x * 0x1.71547652b82fep+6);            // n = round(x * 64/ln2)
  const int n = static_cast<int>(fn);
  double r;
  if constexpr (HI_ONLY) {
    // x - n ln2/64 with ln2/64 as ONE double: the product's error, |n| 2^-60, enters e^x relatively and the
    // blend of the one-exp form with the weight 2 E/(1 + E)^2 <= 2 e^x: at most 2 |n| e^(-|n| ln2/64) 2^-60 < 6e-17
    r = fma(-fn, 0x1.62e42fefa39efp-7, x);
  } else {
    r = fma(-fn, 0x1.62e42fefa0000p-7, x);                    // x - n ln2/64, ln2/64 as hi (36 bits) + lo
    r = fma(-fn, 0x1.cf79abc9e3b3ap-46, r);
  }
  double q = 0x1.1111111111111p-7;                            // 1/120
  q = fma(q, r, 0x1.5555555555555p-5);                        // 1/24
  q = fma(q, r, 0x1.5555555555555p-3);                        // 1/6
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  const double p = q * r;                                     // e^r - 1
  if constexpr (HI_ONLY) {  // 8-byte read, the table's low part dropped: within ~2 ulp instead of 1
    const double th = exp_table()[n & 63].x;
    return ldexp(fma(th, p, th), n >> 6);
  } else {
    const double2 t = exp_table()[n & 63];
    return ldexp(t.x + fma(t.x, p, t.y), n >> 6);
  }
}

// DIST values of the marker kernels: iptcldist 0..3 with -f0'/f0 in the reference's operation order, and the
// one-exp forms of the two exp-bearing distributions (dlnf0_one_exp below)
constexpr int DIST_TS2_ONE_EXP = 4;   // iptcldist 2
constexpr int DIST_BUMP_ONE_EXP = 5;  // iptcldist 3

// 1 / d for d in [1, 2]: v_rcp_f64 and two Newton steps (no scaling, no fix-up: d is a normal number near 1)
// Newton steps on v_rcp_f64's seed (the divisor is 1 + E in [1, 2]): one step leaves the quotient within a
// few ulp (measured by test_dlnf0_forms_against_extended_precision against its derived bound); the second was
// worth 1.5 % of k_step_one (profiles/r03/experiments/ab_variants.log)
#ifndef PIC1DP_RCP_NR
#define PIC1DP_RCP_NR 1
#endif
#ifndef PIC1DP_ONE_EXP_TAB_HI
#define PIC1DP_ONE_EXP_TAB_HI true
#endif
__device__ __forceinline__ double rcp_1to2(double d) {
  double r = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int k = 0; k < PIC1DP_RCP_NR; ++k) r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// -f0'/f0 of the two-Maxwellian distributions with ONE exp and no division by a species constant.
// The reference (src/pic1dp_interaction.F90:278-321) evaluates both Maxwellians and divides their weighted sum
// by their sum: bump-on-tail tmp2 = (A c + B d) / (c + d) with c = den e1 / stm, d = beam e2 / stm2,
// A = v / (T/m), B = (v - v0) / (T2/m); two-stream2 q = (vp ep + vm em) / (ep + em) m/T.  Only the RATIO of the
// two Maxwellians enters: with rho = d / c = exp(L),
//     L(v) = ln(beam stm / (den stm2)) + v^2 / (2T/m) - (v - v0)^2 / (2T2/m)    (a quadratic in v; two-stream2:
//     L = -2 v0 v / (T/m)),
//     tmp2 = (A + rho B) / (1 + rho) = (A + B)/2 + (B - A)/2 tanh(L / 2),
// and tanh(|L|/2) = (1 - E) / (1 + E) with E = exp(-|L|) in (0, 1] -- an argument pexp is made for, a divisor
// in [1, 2], no overflow wherever v goes.  (A + B)/2 and (B - A)/2 are linear in v: every species constant is
// folded on the host (SpeciesConst f*), so a species with general T, T2, m pays what the unit species pays.
// The same function of v to rounding: the numerator's cancellation (A and B of opposite sign where f0 has its
// minimum) is the reference form's own, and the rounding of L enters E as the rounding of the two exp
// arguments enters the reference's ratio -- w keeps the distance to the CPU arithmetic it had (the tests derive the bound),
// x and v never see tmp2 and stay bit-exact.  PIC1DP_DLNF0=ref keeps the reference's operation order.
template <int DIST, int POW2>
__device__ __forceinline__ double dlnf0_one_exp(double v, const SpeciesConst &c) {
  double L, M, D;
  if constexpr (DIST == DIST_TS2_ONE_EXP) {
    L = v * c.fq1;
    M = POW2 == 2 ? v : v * c.fm1;
    D = c.fd0;
  } else if constexpr (POW2 == 2) {  // T = T2 = m = 1: L is linear in v, (A + B)/2 = v - v0/2, (B - A)/2 = -v0/2
    L = fma(c.fq1, v, c.fq0);
    M = v + c.fm0;
    D = c.fd0;
  } else {
    L = fma(fma(c.fq2, v, c.fq1), v, c.fq0);
    M = fma(v, c.fm1, c.fm0);
    D = fma(v, c.fd1, c.fd0);
  }
  const double E = pexp<PIC1DP_ONE_EXP_TAB_HI>(-fabs(L));
  const double t = (1.0 - E) * rcp_1to2(1.0 + E);
  return fma(D, copysign(t, L), M);
}

// -(d f0/dv)/f0 at v, src/pic1dp_interaction.F90:274-326
template <int DIST, int POW2, class D>
__device__ __forceinline__ double dlnf0(double v, const SpeciesConst &c, D &dv) {
  if constexpr (DIST == DIST_TS2_ONE_EXP || DIST == DIST_BUMP_ONE_EXP) {
    return dlnf0_one_exp<DIST, POW2>(v, c);
  } else if constexpr (DIST == 1) {  // two-stream1 :276
    return v - 2.0 / v;
  } else if constexpr (DIST == 2) {  // two-stream2 :278-292
    const double vp = v + c.v0, vm = v - c.v0;
    const double ep = pexp(-divh<POW2>(vp * vp, c.two_tm, c.r_two_tm, dv));
    PAIR_FENCE();
    const double em = pexp(-divh<POW2>(vm * vm, c.two_tm, c.r_two_tm, dv));
    const double q = (vp * ep + vm * em) / (ep + em);
    return divc<POW2>(q * c.m, c.T, c.r_T, dv);
  } else if constexpr (DIST == 3) {  // bump-on-tail :294-321
    const double vm = v - c.v0;
    const double e1 = pexp(-divh<POW2>(v * v, c.two_tm, c.r_two_tm, dv));
    PAIR_FENCE();
    const double e2 = pexp(-divh<POW2>(vm * vm, c.two_tm2, c.r_two_tm2, dv));
    const double a = divc<POW2>(divc<POW2>(c.den * v, c.tm, c.r_tm, dv) * e1, c.stm, c.r_stm, dv);
    const double b = divc<POW2>(divc<POW2>(c.beam * vm, c.tm2, c.r_tm2, dv) * e2, c.stm2, c.r_stm2, dv);
    const double cc = divc<POW2>(c.den * e1, c.stm, c.r_stm, dv);
    const double d = divc<POW2>(c.beam * e2, c.stm2, c.r_stm2, dv);
    return (a + b) / (cc + d);
  } else {  // (shifted) Maxwellian :323-325
    return divc<POW2>(v - c.v0, c.tm, c.r_tm, dv);
  }
}

struct One {
  double x, v, w;
};

// the weight and velocity updates of one marker given its field e,
// src/pic1dp_interaction.F90:261-338.  T2MODE: 0 evaluate tmp2 = -f0'/f0(v); 1 evaluate it and
// hand it out through t2io; 2 take it from t2io (the whole-step kernels can carry it from the
// first sub-step's kernel to the second's instead of evaluating it twice, see k_step_half)
template <int DIST, int MODE, int POW2, int T2MODE, class D>
__device__ __forceinline__ One push_core(double v, double w, double p, double xb, double vb, double wb, double e,
                                         double dt, const SpeciesConst &s, D &dv, double *t2io) {
  One o;
  o.x = xb + dt * v;                     // :261
  o.w = w;
  if constexpr (MODE != MODE_FULLF) {
    const double tmp1 = (MODE == MODE_DF_LIN) ? p * e : (p - w) * e;   // :268-272
    double tmp2;
    if constexpr (T2MODE == 2) {
      tmp2 = *t2io;
    } else {
      tmp2 = dlnf0<DIST, POW2>(v, s, dv);
      if constexpr (T2MODE == 1) *t2io = tmp2;
    }
    if constexpr (POW2 == 0 && (DIST == DIST_TS2_ONE_EXP || DIST == DIST_BUMP_ONE_EXP)) {
      // with the one-exp form tmp2 -- and w -- match the CPU arithmetic to rounding, not bit for bit: the
      // division by a general mass may be the product with its reciprocal (one operation instead of five)
      o.w = wb + (dt * tmp1 * tmp2 * s.Z) * s.r_m;
    } else {
      o.w = wb + divc<POW2>(dt * tmp1 * tmp2 * s.Z, s.m, s.r_m, dv);  // :329
    }
  }
  if constexpr (MODE == MODE_DF_LIN) {
    o.v = v;
  } else {
    o.v = vb + divc<POW2>(dt * e * s.Z, s.m, s.r_m, dv);  // :336
  }
  return o;
}

// gather + push of one marker, src/pic1dp_interaction.F90:246-338:
// derivatives at (x, v, w), base (xb, vb, wb), field tile sE, step dt
// where push_one takes the field at a grid point from: a staged tile, or -- for a mode-filter field of one kept
// mode whose tile is not staged (k_step_sums) -- the mode's tables A = 2 fre, B = 2 fim and its amplitudes:
// (fre re + fim im) * 2 as the solve writes it (src/pic1dp_field.F90:251-257) and A re + B im are the same
// bits (the factor 2 commutes with every rounding; no contraction in this build)
struct ModeField {
  const double *A, *B;
  double re, im;
};
__device__ __forceinline__ double field_at(const double *t, int i) { return t[i]; }
__device__ __forceinline__ double field_at(const ModeField &m, int i) {
  double e = m.A[i] * m.re;
  e = e + m.B[i] * m.im;
  return e;
}

template <int DIST, int MODE, int POW2, int T2MODE = 0, class FS = const double *>
__device__ __forceinline__ One push_one(double x, double v, double w, double p, double xb,
                                        double vb, double wb, const FS &sE, double dt,
                                        const GridConst &g, const SpeciesConst &s, double *t2io = nullptr) {
  int ix;
  double wl;
  locate(x, g, ix, wl);
  double e = field_at(sE, ix) * wl;                // :254
  e = e + field_at(sE, ix + 1) * (1.0 - wl);       // :257 (cell nx holds E[0])
  if constexpr (POW2 == 0) {
    if (s.fastc) {
      DivFast dv;
      const One o = push_core<DIST, MODE, POW2, T2MODE>(v, w, p, xb, vb, wb, e, dt, s, dv, t2io);
      if (dv.ok()) return o;
    }
  }
  DivTrue dv;
  return push_core<DIST, MODE, POW2, T2MODE>(v, w, p, xb, vb, wb, e, dt, s, dv, t2io);
}

// wrap + linear deposit of one marker into the LDS copy of rho,
// src/pic1dp_interaction.F90:102-113; returns the wrapped position
__device__ __forceinline__ double deposit_one(double x, double q, double *sR, const GridConst &g, int *ix_out = nullptr,
                                              double *wl_out = nullptr) {
  const double px = wrap(x, g.lx);
  int ix;
  double wl;
  locate(px, g, ix, wl);
  if (ix_out) {  // cell and left weight of the wrapped position, for a caller that gathers there next
    *ix_out = ix;
    *wl_out = wl;
  }
  // :110, :113.  The right-hand neighbour of the last cell is cell 0: of the next copy of the tile, or the
  // guard cell behind the last copy (flush_rho adds them all) -- no wrap-around of the index, one address
  lds_add(&sR[ix], wl * q);
  lds_add(&sR[ix + 1], (1.0 - wl) * q);
  return px;
}

// The workgroup's pairs: row k = pairs (blockIdx + k gridDim) blockDim ... of the grid-stride split.  `dealt` rows go to
// its threads as they stand; the 64-pair chunks of the remaining rows (StepArgs::dyn_tail sixteenths of them) are DRAWN:
// every wave takes the next one from a counter in the LDS (one ds_add_rtn_u32 per chunk, no device-scope traffic), so the
// waves that run ahead take more and the workgroup meets its final barrier together (round 5: the skew between a
// workgroup's waves was worth 1.4-4.7 % of k_step_one<PRIV>, which has this loop spelled out for its register budget).
// Used by k_step_half, k_step_full, the tiles' k_step_one (kernels_step.hip) and k_ptcldist (kernels_diag.hip).
struct PairRows {
  int64_t first, stride;
  int dealt, drawn_total;
};
__device__ __forceinline__ PairRows pair_rows(int64_t npair, int dyn_tail) {
  PairRows r;
  r.first = static_cast<int64_t>(blockIdx.x) * blockDim.x;
  r.stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  r.dealt = r.first < npair ? static_cast<int>((npair - r.first + r.stride - 1) / r.stride) : 0;
  r.drawn_total = 0;
  if (dyn_tail > 0) {
    const int drawn_rows = (r.dealt * dyn_tail) >> 4;
    r.dealt -= drawn_rows;
    r.drawn_total = drawn_rows * static_cast<int>(blockDim.x >> 6);
  }
  return r;
}
// the pair of this lane in the next drawn chunk, or false: the workgroup's pairs are exhausted (wave-uniform)
__device__ __forceinline__ bool draw_chunk(const PairRows &r, unsigned *ctr, int64_t &j) {
  int c = 0;
  if ((threadIdx.x & 63) == 0) c = static_cast<int>(__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  c = __builtin_amdgcn_readfirstlane(c);
  if (c >= r.drawn_total) return false;
  const int waves = static_cast<int>(blockDim.x >> 6);
  j = r.first + static_cast<int64_t>(r.dealt + c / waves) * r.stride + (c % waves) * 64 + (threadIdx.x & 63);
  return true;
}

// The workgroup's LDS tile of rho: nx cells + a guard cell (= cell 0) behind them.  (Replicated tiles -- lane l depositing
// into copy l % k -- were measured in rounds 1-2 and bought nothing at any grid: HISTORY.md.)
__device__ __forceinline__ void zero_rho(double *sR, const GridConst &g) {
  for (int i = threadIdx.x; i < g.nx + 1; i += blockDim.x) sR[i] = 0.0;  // + the guard cell
}
__device__ __forceinline__ void flush_rho(const double *sR, double *rho, const GridConst &g) {
  // one global atomic per cell per workgroup; start cell rotated by workgroup
  const int nx = g.nx;
  rho += static_cast<size_t>(blockIdx.x & (g.gcopies - 1)) * g.gstride;
  const int rot = static_cast<int>((static_cast<long long>(blockIdx.x) * nx) / gridDim.x);
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    int j = i + rot;
    if (j >= nx) j -= nx;
    double val = sR[j];
    if (j == 0) val += sR[nx];  // the guard cell behind the tile is cell 0
    if (val != 0.0) glb_add(&rho[j], val);  // rho: this workgroup's copy of the accumulator
  }
}

// sum reduced over the workgroup (tree order); valid on thread 0
__device__ __forceinline__ double block_sum(double v, double *scratch) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (blockDim.x >> 6); ++w) t += scratch[w];
  return t;
}

// six per-thread values summed over the workgroup and added to dst[0..5] (one barrier; scr: 96 doubles nobody else
// uses any more): wave reductions, [6][16] partials, then 16-lane groups of the first 96 threads
__device__ __forceinline__ void block_sum6_add(const double (&v)[6], double *scr, double *dst) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double t = v[k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) scr[k * 16 + wave] = t;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 96; t += blockDim.x) {
    double u = (t & 15) < nwaves ? scr[t] : 0.0;
    for (int off = 8; off > 0; off >>= 1) u += __shfl_down(u, off, 16);
    if ((t & 15) == 0) glb_add(dst + (t >> 4), u);
  }
}

}  // namespace
}  // namespace pic1dp
