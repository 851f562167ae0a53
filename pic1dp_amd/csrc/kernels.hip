// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the PIC1D
// time-step hot path.  Compiled with -ffp-contract=off: every product/sum that
// the reference rounds separately is rounded separately here, so positions,
// velocities and cell indices are bit-identical to the CPU arithmetic; only
// exp() (OCML vs libm, <= 1 ulp) and the order of the charge sums differ.
//
// Design (DESIGN.md has the numbers):
//  * All kernels are HBM-bound streaming passes over SoA FP64 particle arrays;
//    there is no dense contraction, so no MFMA.  Loads/stores are 16 B per lane
//    (double2), fully coalesced, grid-stride, >= 4 workgroups per CU.
//  * The field E (nx doubles) is staged once per workgroup into LDS with a
//    wrap-around guard cell, so the 2-point gather is two ds_read_b64.
//  * Deposition accumulates into a per-workgroup LDS copy of rho with
//    ds_add_f64, and is flushed with one global_atomic_add_f64 per cell per
//    workgroup, start cell staggered by workgroup to spread contention.
//  * Sub-step kernels (k_push, k_deposit; the drop-in call sites) use two
//    particle sets (ping-pong): sub-step 1 reads the step-start set and writes
//    the half-step set; sub-step 2 reads both and overwrites the step-start
//    set.  The reference's x_bak/v_bak/w_bak copies
//    (src/pic1dp_interaction.F90:178-189) are never materialised.
//  * Whole-step kernels (k_step_half, k_step_full; pic1dp_hip_step) go further:
//    the half-step state is recomputed bit-identically in the second kernel
//    instead of being stored and re-read, and the state is updated in place.
//  * x/lx, evaluated three times per marker and step, uses a correctly rounded
//    reciprocal-plus-two-FMA division (div_lx) instead of the hardware sequence;
//    species whose divisor constants are powers of two (or all 1) multiply instead
//    of dividing -- every shortcut is bit-identical to the true division.
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
// tuning builds of the whole-step kernels' marker loop: 0 one pair per lane per trip (default),
// 1 next trip's loads before this trip's arithmetic (k_step_full), 2 two pairs per trip
#ifndef PIC1DP_STEP_PIPE
#define PIC1DP_STEP_PIPE 0
#endif
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
__device__ __forceinline__ double pexp(double x) {
  x = fmax(x, -750.0);
  const double fn = rint(x * 0x1.71547652b82fep+6);            // n = round(x * 64/ln2)
  const int n = static_cast<int>(fn);
  double r = fma(-fn, 0x1.62e42fefa0000p-7, x);               // x - n ln2/64, ln2/64 as hi (36 bits) + lo
  r = fma(-fn, 0x1.cf79abc9e3b3ap-46, r);
  double q = 0x1.1111111111111p-7;                            // 1/120
  q = fma(q, r, 0x1.5555555555555p-5);                        // 1/24
  q = fma(q, r, 0x1.5555555555555p-3);                        // 1/6
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  const double p = q * r;                                     // e^r - 1
  const double2 t = exp_table()[n & 63];
  return ldexp(t.x + fma(t.x, p, t.y), n >> 6);
}

// -(d f0/dv)/f0 at v, src/pic1dp_interaction.F90:274-326
template <int DIST, int POW2, class D>
__device__ __forceinline__ double dlnf0(double v, const SpeciesConst &c, D &dv) {
  if constexpr (DIST == 1) {  // two-stream1 :276
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
    o.w = wb + divc<POW2>(dt * tmp1 * tmp2 * s.Z, s.m, s.r_m, dv);  // :329
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

// Measurement build only (-DPIC1DP_DEPOSIT_PREREDUCE=1, tools/ab_prereduce.sh): the wave-level
// pre-reduction the north_star names -- lanes of a wave that hit the same cell combine their
// values (shuffle butterfly) and ONE lane issues ONE ds_add_f64 per distinct cell.  gfx950 has no
// lane-matching instruction, so the groups are peeled off one by one: leader = first lane still
// to do, ballot of the lanes with its cell, wave sum of their values.  With unsorted markers a
// wave holds ~55 distinct cells of 192 (~62 of 1024): ~60 rounds of ~20 instructions against two
// atomics -- see DESIGN.md 3.3 for the measured table.
#ifndef PIC1DP_DEPOSIT_PREREDUCE
#define PIC1DP_DEPOSIT_PREREDUCE 0
#endif
#if PIC1DP_DEPOSIT_PREREDUCE
__device__ __forceinline__ void lds_add_matched(double *sR, int ix, double val) {
  const int lane = static_cast<int>(__lane_id());
  unsigned long long todo = __ballot(1);
  while (todo) {
    const int lead = __ffsll(static_cast<long long>(todo)) - 1;
    const int lix = __shfl(ix, lead, 64);
    const bool mine = ix == lix && ((todo >> lane) & 1ULL);
    const unsigned long long grp = __ballot(mine);
    double v = mine ? val : 0.0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == lead) lds_add(&sR[lix], v);
    todo &= ~grp;
  }
}
#endif

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
#if PIC1DP_DEPOSIT_PREREDUCE
  lds_add_matched(sR, ix, wl * q);
  ix = ix + 1;
  if (ix > g.nx - 1) ix = 0;
  lds_add_matched(sR, ix, (1.0 - wl) * q);
#else
  // :110, :113.  The right-hand neighbour of the last cell is cell 0: of the next copy of the tile, or the
  // guard cell behind the last copy (flush_rho adds them all) -- no wrap-around of the index, one address
  lds_add(&sR[ix], wl * q);
  lds_add(&sR[ix + 1], (1.0 - wl) * q);
#endif
  return px;
}

// The workgroup's LDS copy of rho may be replicated (g.rcopies = 1, 2, 4 or 8 copies,
// lane l deposits into copy l % rcopies): neighbouring lanes of a wave that hit the same
// cell then hit different addresses, which matters for small grids (at nx = 192 a wave's
// 64 lanes share 192 cells).  The copies are added up in the flush.
__device__ __forceinline__ double *my_rho_copy(double *sR, const GridConst &g) {
  return sR + (threadIdx.x & (g.rcopies - 1)) * g.nx;
}
__device__ __forceinline__ void zero_rho(double *sR, const GridConst &g) {
  for (int i = threadIdx.x; i < g.nx * g.rcopies + 1; i += blockDim.x) sR[i] = 0.0;  // + the guard cell
}
__device__ __forceinline__ void flush_rho(const double *sR, double *rho, const GridConst &g) {
  // one global atomic per cell per workgroup; start cell rotated by workgroup
  if (g.debug_noflush) return;
  const int nx = g.nx;
  rho += static_cast<size_t>(blockIdx.x & (g.gcopies - 1)) * g.gstride;
  const int rot = static_cast<int>((static_cast<long long>(blockIdx.x) * nx) / gridDim.x);
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    int j = i + rot;
    if (j >= nx) j -= nx;
    double val = sR[j];
    for (int c = 1; c < g.rcopies; ++c) val += sR[c * nx + j];
    if (j == 0) val += sR[g.rcopies * nx];  // the guard cell behind the last copy is cell 0
    if (val != 0.0) glb_add(&rho[j], val);  // rho: this workgroup's copy of the accumulator
  }
}

template <int DIST, int MODE, int POW2, bool IRK2, bool FUSED>
__global__ void __launch_bounds__(1024) k_push(const PushArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exp_table_init();
  double *sE = reinterpret_cast<double *>(smem);
  const int nx = a.g.nx;
  double *sR0 = sE + ((nx + 2) & ~1);
  for (int i = threadIdx.x; i < nx; i += blockDim.x) sE[i] = a.E[i];
  if constexpr (FUSED) zero_rho(sR0, a.g);
  if (threadIdx.x == 0) sE[nx] = a.E[0];
  __syncthreads();
  double *sR = my_rho_copy(sR0, a.g);

  constexpr bool HAS_W = (MODE != MODE_FULLF);
  constexpr bool PUSH_V = (MODE != MODE_DF_LIN);
  const int64_t npair = a.np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const double2 *sx2 = reinterpret_cast<const double2 *>(a.src.x);
  const double2 *sv2 = reinterpret_cast<const double2 *>(a.src.v);
  const double2 *sw2 = reinterpret_cast<const double2 *>(a.src.w);
  const double2 *bx2 = reinterpret_cast<const double2 *>(a.base.x);
  const double2 *bv2 = reinterpret_cast<const double2 *>(a.base.v);
  const double2 *bw2 = reinterpret_cast<const double2 *>(a.base.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
  double2 *dx2 = reinterpret_cast<double2 *>(a.dst.x);
  double2 *dv2 = reinterpret_cast<double2 *>(a.dst.v);
  double2 *dw2 = reinterpret_cast<double2 *>(a.dst.w);

  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair;
       j += stride) {
    const int64_t o = tidx2(j);
    const double2 X = ld2(sx2 + o), V = ld2(sv2 + o);
    double2 W = make_double2(0.0, 0.0), P = make_double2(0.0, 0.0);
    if constexpr (HAS_W) W = ld2(sw2 + o);
    if constexpr (MODE != MODE_FULLF || FUSED) P = ld2(p2 + o);
    double2 XB = X, VB = V, WB = W;
    if constexpr (IRK2) {
      XB = ld2(bx2 + o);
      if constexpr (PUSH_V) VB = ld2(bv2 + o);
      if constexpr (HAS_W) WB = ld2(bw2 + o);
    }
    One o0 = push_one<DIST, MODE, POW2>(X.x, V.x, W.x, P.x, XB.x, VB.x, WB.x, sE, a.dt, a.g, a.s);
    One o1 = push_one<DIST, MODE, POW2>(X.y, V.y, W.y, P.y, XB.y, VB.y, WB.y, sE, a.dt, a.g, a.s);
    if constexpr (FUSED) {
      o0.x = deposit_one(o0.x, HAS_W ? o0.w : P.x, sR, a.g);
      o1.x = deposit_one(o1.x, HAS_W ? o1.w : P.y, sR, a.g);
    }
    st2(dx2 + o, o0.x, o1.x);
    if constexpr (PUSH_V) st2(dv2 + o, o0.v, o1.v);
    if constexpr (HAS_W) st2(dw2 + o, o0.w, o1.w);
  }
  // odd tail marker
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double x = a.src.x[i], v = a.src.v[i];
    const double w = HAS_W ? a.src.w[i] : 0.0;
    const double p = a.p[i];
    double xb = x, vb = v, wb = w;
    if constexpr (IRK2) {
      xb = a.base.x[i];
      if constexpr (PUSH_V) vb = a.base.v[i];
      if constexpr (HAS_W) wb = a.base.w[i];
    }
    One o = push_one<DIST, MODE, POW2>(x, v, w, p, xb, vb, wb, sE, a.dt, a.g, a.s);
    if constexpr (FUSED) o.x = deposit_one(o.x, HAS_W ? o.w : p, sR, a.g);
    a.dst.x[i] = o.x;
    if constexpr (PUSH_V) a.dst.v[i] = o.v;
    if constexpr (HAS_W) a.dst.w[i] = o.w;
  }
  if constexpr (FUSED) {
    __syncthreads();
    flush_rho(sR0, a.rho, a.g);
  }
}

// ---------------------------------------------------------------------------
// diagnostics of output_all, per marker (used by k_ptcldist and by the DIAG variant of
// k_step_full): src/pic1dp_output.F90:126-151 (kinetic sums) and :239-315 (histograms)
// ---------------------------------------------------------------------------
struct DistBins {
  double *h;       // base of [markr_xv | total_xv | pertb_xv | markr_v | total_v | pertb_v]
  int nxv, nv;     // nx_opd*nv_opd, nv_opd
  __device__ __forceinline__ double *xv(int k) const { return h + static_cast<size_t>(k) * nxv; }
  __device__ __forceinline__ double *vv(int k) const { return h + static_cast<size_t>(3) * nxv + k * nv; }
};
struct DistSums {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;  // sum v^2, v^2 p, v^2 w of this thread
};

template <bool LDS>
__device__ __forceinline__ void bin_add(double *p, double v) {
  if constexpr (LDS) {
    lds_add(p, v);
  } else {
    glb_add(p, v);
  }
}

template <bool LDS, bool DELTAF>
__device__ __forceinline__ void ptcldist_one(double px, double pv, double pp, double pw, const DistGeom &dg,
                                             const DistBins &b, DistSums &sm) {
  const int nxo = dg.nxo, nvo = dg.nvo;
  const double v2 = pv * pv;
  sm.s0 += v2;
  sm.s1 += v2 * pp;
  if constexpr (DELTAF) sm.s2 += v2 * pw;
  if (fabs(pv) >= dg.vmax) return;                      // :241
  double sx = px / dg.lx * static_cast<double>(nxo);    // :243
  const double fx = floor(sx);
  int ix = static_cast<int>(fx);
  sx = 1.0 - (sx - fx);
  double sv = (pv + dg.vmax) / (dg.vmax * 2.0) * static_cast<double>(nvo - 1);  // :247
  const double fv = floor(sv);
  const int iv = static_cast<int>(fv);
  sv = 1.0 - (sv - fv);
  // memory safety only (the reference would write out of bounds)
  if (static_cast<unsigned>(ix) >= static_cast<unsigned>(nxo) || iv < 0 || iv + 1 >= nvo) return;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int a = iv * nxo + ix, c = (iv + 1) * nxo + ix;
    bin_add<LDS>(&b.xv(0)[a], sx * sv);
    bin_add<LDS>(&b.xv(1)[a], sx * sv * pp);
    if constexpr (DELTAF) bin_add<LDS>(&b.xv(2)[a], sx * sv * pw);
    bin_add<LDS>(&b.xv(0)[c], sx * (1.0 - sv));
    bin_add<LDS>(&b.xv(1)[c], sx * (1.0 - sv) * pp);
    if constexpr (DELTAF) bin_add<LDS>(&b.xv(2)[c], sx * (1.0 - sv) * pw);
    ix = ix + 1;                                        // :274-276
    if (ix > nxo - 1) ix = 0;
    sx = 1.0 - sx;
  }
  if constexpr (!LDS) {                                 // :300-314
    bin_add<LDS>(&b.vv(0)[iv], sv);
    bin_add<LDS>(&b.vv(1)[iv], sv * pp);
    if constexpr (DELTAF) bin_add<LDS>(&b.vv(2)[iv], sv * pw);
    bin_add<LDS>(&b.vv(0)[iv + 1], 1.0 - sv);
    bin_add<LDS>(&b.vv(1)[iv + 1], (1.0 - sv) * pp);
    if constexpr (DELTAF) bin_add<LDS>(&b.vv(2)[iv + 1], (1.0 - sv) * pw);
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

// per-workgroup partial kinetic sums, v histograms as row sums, flush of the LDS copy
template <bool LDS, bool DELTAF>
__device__ __forceinline__ void ptcldist_finish(const DistGeom &dg, const DistBins &b, const DistSums &sm, double *scr,
                                                double *out, double *partial) {
  const int nxo = dg.nxo, nvo = dg.nvo, ntot = 3 * nxo * nvo + 3 * nvo;
  if (partial) {
    const double t0 = block_sum(sm.s0, scr);
    const double t1 = block_sum(sm.s1, scr);
    const double t2 = block_sum(sm.s2, scr);
    if (threadIdx.x == 0) {
      partial[blockIdx.x * 3 + 0] = t0;
      partial[blockIdx.x * 3 + 1] = t1;
      partial[blockIdx.x * 3 + 2] = t2;
    }
  }
  if constexpr (LDS) {
    __syncthreads();
    // v histograms = row sums of the (x,v) histograms, one thread per (k, iv)
    for (int t = threadIdx.x; t < (DELTAF ? 3 : 2) * nvo; t += blockDim.x) {
      const int k = t / nvo, iv = t - k * nvo;
      const double *row = b.xv(k) + static_cast<size_t>(iv) * nxo;
      double acc = 0.0;
      for (int ix = 0; ix < nxo; ++ix) acc += row[ix];
      b.vv(k)[iv] = acc;
    }
    __syncthreads();
    const int rot = static_cast<int>((static_cast<long long>(blockIdx.x) * ntot) / gridDim.x);
    for (int i = threadIdx.x; i < ntot; i += blockDim.x) {
      int j = i + rot;
      if (j >= ntot) j -= ntot;
      const double val = b.h[j];
      if (val != 0.0) glb_add(&out[j], val);
    }
  }
}

// ---------------------------------------------------------------------------
// Whole-time-step kernels (pic1dp_hip_step): the half-step state is never
// written to memory.  RK2 (midpoint) needs, for the second sub-step, the state
// after the first one; instead of storing it (24 B) and loading it back (24 B)
// it is recomputed from the step-start state and the step-start field E0 with
// the very same instruction sequence, hence bit-identical:
//   k_step_half : x0,v0,w0,p (32 B in, 0 B out) -> half-step x', w' -> deposit
//   k_step_full : x0,v0,w0,p (32 B in)          -> recompute x',v',w' from E0,
//                 push from the base with the half-step field Eh, wrap, deposit,
//                 store x,v,w in place (24 B out)
// 88 B per marker per time step instead of 136 B (ping-pong) or the reference's
// 256 B data flow; arithmetic per marker roughly doubles (still under the
// FP64 rate at the HBM-bound pace).
// ---------------------------------------------------------------------------
struct StepArgsDev {
  double *x, *v, *w;
  const double *p;
  const double *E0, *Eh;
  double *rho;
  int64_t np;
  double dt_half, dt_full;
  GridConst g;
  SpeciesConst s;
  int nt;
  double *t2;  // [np + 2] -f0'/f0 at the step-start velocity, carried from k_step_half to k_step_full (or null)
  // DIAG variant of k_step_full: the diagnostics of output_all taken on the new state in the same pass
  DistGeom dg;
  double *dist_out;      // histograms [3*nxo*nvo + 3*nvo] of this species (accumulated with atomics), or null
  double *dist_partial;  // [gridDim][3] kinetic sums per workgroup
  // k_step_one: prediction of the next step's first-sub-step charge
  const double *tabA, *tabB;  // [pred_nm][nx]
  double *pred;               // [1 + 2*pred_nm][nx]; k_step_sums: [8]
  int pred_nm, t2_mode;
  const double *eh_re, *eh_im;  // k_step_sums: the kept mode of Eh
  double snx, pred_k;           // k_step_one's prediction: nx / lx, and dt/2 Z/m
};

// CARRY: a species whose divisor constants are general numbers spends most of either kernel
// in -f0'/f0 (two exp, eight constant divisions, one true division: FP64-issue-bound).  The
// second kernel evaluates it twice -- at the step-start velocity again, to recompute the
// half-step state, and at the half-step velocity.  With CARRY the first kernel stores its
// value (8 B per marker, contiguous array) and the second loads it: 16 B more traffic per
// marker and step for a third less arithmetic.  Same value, same bits.
// NT: non-temporal loads and stores.  They win once the marker state no longer
// fits the 256 MiB Infinity Cache (+15 % at 2e7 markers); below that, plain
// accesses keep the state cache-resident between the two kernels of a step
// (+5 % at the reference's default 6.4e6 markers).  Chosen per launch.
template <int DIST, int MODE, int POW2, bool NT, bool CARRY>
__global__ void __launch_bounds__(1024) k_step_half(const StepArgsDev a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exp_table_init();
  double *sE = reinterpret_cast<double *>(smem);
  const int nx = a.g.nx;
  double *sR0 = sE + ((nx + 2) & ~1);
  for (int i = threadIdx.x; i < nx; i += blockDim.x) sE[i] = a.E0[i];
  zero_rho(sR0, a.g);
  if (threadIdx.x == 0) sE[nx] = a.E0[0];
  __syncthreads();
  double *sR = my_rho_copy(sR0, a.g);
  constexpr bool HAS_W = (MODE != MODE_FULLF);
  const int64_t npair = a.np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const double2 *x2 = reinterpret_cast<const double2 *>(a.x);
  const double2 *v2 = reinterpret_cast<const double2 *>(a.v);
  const double2 *w2 = reinterpret_cast<const double2 *>(a.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
#if PIC1DP_STEP_PIPE == 2
  // tuning variant: two pairs per lane per trip, all eight loads in flight before the arithmetic
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += 2 * stride) {
    const bool hb = j + stride < npair;
    const int64_t oa = tidx2(j), ob = tidx2(hb ? j + stride : j);
    const double2 XA = ld2t<NT>(x2 + oa), VA = ld2t<NT>(v2 + oa), PA = ld2t<NT>(p2 + oa);
    const double2 XB = ld2t<NT>(x2 + ob), VB = ld2t<NT>(v2 + ob), PB = ld2t<NT>(p2 + ob);
    double2 WA = make_double2(0.0, 0.0), WB = WA;
    if constexpr (HAS_W) {
      WA = ld2t<NT>(w2 + oa);
      WB = ld2t<NT>(w2 + ob);
    }
    {
      const One h0 = push_one<DIST, MODE, POW2>(XA.x, VA.x, WA.x, PA.x, XA.x, VA.x, WA.x, sE, a.dt_half, a.g, a.s);
      const One h1 = push_one<DIST, MODE, POW2>(XA.y, VA.y, WA.y, PA.y, XA.y, VA.y, WA.y, sE, a.dt_half, a.g, a.s);
      deposit_one(h0.x, HAS_W ? h0.w : PA.x, sR, a.g);
      deposit_one(h1.x, HAS_W ? h1.w : PA.y, sR, a.g);
    }
    if (hb) {
      const One h0 = push_one<DIST, MODE, POW2>(XB.x, VB.x, WB.x, PB.x, XB.x, VB.x, WB.x, sE, a.dt_half, a.g, a.s);
      const One h1 = push_one<DIST, MODE, POW2>(XB.y, VB.y, WB.y, PB.y, XB.y, VB.y, WB.y, sE, a.dt_half, a.g, a.s);
      deposit_one(h0.x, HAS_W ? h0.w : PB.x, sR, a.g);
      deposit_one(h1.x, HAS_W ? h1.w : PB.y, sR, a.g);
    }
  }
#else
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += stride) {
    const int64_t o = tidx2(j);
    const double2 X = ld2t<NT>(x2 + o), V = ld2t<NT>(v2 + o), P = ld2t<NT>(p2 + o);
    double2 W = make_double2(0.0, 0.0);
    if constexpr (HAS_W) W = ld2t<NT>(w2 + o);
    double t0 = 0.0, t1 = 0.0;
    const One h0 = push_one<DIST, MODE, POW2, CARRY ? 1 : 0>(X.x, V.x, W.x, P.x, X.x, V.x, W.x, sE, a.dt_half, a.g, a.s, &t0);
    deposit_one(h0.x, HAS_W ? h0.w : P.x, sR, a.g);
    PAIR_FENCE();
    const One h1 = push_one<DIST, MODE, POW2, CARRY ? 1 : 0>(X.y, V.y, W.y, P.y, X.y, V.y, W.y, sE, a.dt_half, a.g, a.s, &t1);
    deposit_one(h1.x, HAS_W ? h1.w : P.y, sR, a.g);
    if constexpr (CARRY) st2t<NT>(reinterpret_cast<double2 *>(a.t2) + j, t0, t1);
  }
#endif
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double x = a.x[i], v = a.v[i], p = a.p[i];
    const double w = HAS_W ? a.w[i] : 0.0;
    double t0 = 0.0;
    const One h = push_one<DIST, MODE, POW2, CARRY ? 1 : 0>(x, v, w, p, x, v, w, sE, a.dt_half, a.g, a.s, &t0);
    if constexpr (CARRY) a.t2[a.np - 1] = t0;
    deposit_one(h.x, HAS_W ? h.w : p, sR, a.g);
  }
  __syncthreads();
  flush_rho(sR0, a.rho, a.g);
}

// one marker through the second half of the time step
template <int DIST, int MODE, int POW2, bool CARRY = false, class FH = const double *>
__device__ __forceinline__ One step_full_one(double x, double v, double w, double p, const double *sE0,
                                             const FH &sEh, double *sR, const StepArgsDev &a, double t2 = 0.0,
                                             int *ix_out = nullptr, double *wl_out = nullptr) {
  constexpr bool HAS_W = (MODE != MODE_FULLF);
  // sub-step 1 again (identical arithmetic), with the wrap the deposit applied
  One h = push_one<DIST, MODE, POW2, CARRY ? 2 : 0>(x, v, w, p, x, v, w, sE0, a.dt_half, a.g, a.s, &t2);
  h.x = wrap(h.x, a.g.lx);
  // sub-step 2: derivatives at the half-step state, base = step-start state
  One n = push_one<DIST, MODE, POW2>(h.x, h.v, h.w, p, x, v, w, sEh, a.dt_full, a.g, a.s);
  n.x = deposit_one(n.x, HAS_W ? n.w : p, sR, a.g, ix_out, wl_out);
  return n;
}

// DIAG: on a step after which the host will call output_all, the histograms of output_ptcldist and
// the kinetic sums of output_field (k_ptcldist's work: another 32 B per marker read) are taken here on
// the state just computed, into an LDS copy of the histograms next to the grid tiles (one workgroup
// of 1024 threads per CU then).
template <int DIST, int MODE, int POW2, bool NT, bool CARRY, bool DIAG>
__global__ void __launch_bounds__(1024) k_step_full(const StepArgsDev a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exp_table_init();
  const int nx = a.g.nx;
  const int ne = (nx + 2) & ~1;
  double *sE0 = reinterpret_cast<double *>(smem);
  double *sEh = sE0 + ne;
  double *sR0 = sEh + ne;
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    sE0[i] = a.E0[i];
    sEh[i] = a.Eh[i];
  }
  zero_rho(sR0, a.g);
  if (threadIdx.x == 0) {
    sE0[nx] = a.E0[0];
    sEh[nx] = a.Eh[0];
  }
  constexpr bool HAS_W = (MODE != MODE_FULLF);
  constexpr bool PUSH_V = (MODE != MODE_DF_LIN);
  // DIAG: histograms behind the rho copies (16-byte aligned), then the block_sum scratch
  double *sH = sR0 + ((nx * a.g.rcopies + 2) & ~1);
  const int ntot = DIAG ? 3 * a.dg.nxo * a.dg.nvo + 3 * a.dg.nvo : 0;
  const DistBins bins{sH, a.dg.nxo * a.dg.nvo, a.dg.nvo};
  DistSums sums;
  if constexpr (DIAG)
    for (int i = threadIdx.x; i < ntot; i += blockDim.x) sH[i] = 0.0;
  __syncthreads();
  double *sR = my_rho_copy(sR0, a.g);
  const int64_t npair = a.np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 *x2 = reinterpret_cast<double2 *>(a.x);
  double2 *v2 = reinterpret_cast<double2 *>(a.v);
  double2 *w2 = reinterpret_cast<double2 *>(a.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
#if PIC1DP_STEP_PIPE == 2
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += 2 * stride) {
    const bool hb = j + stride < npair;
    const int64_t oa = tidx2(j), ob = tidx2(hb ? j + stride : j);
    const double2 XA = ld2t<NT>(x2 + oa), VA = ld2t<NT>(v2 + oa), PA = ld2t<NT>(p2 + oa);
    const double2 XB = ld2t<NT>(x2 + ob), VB = ld2t<NT>(v2 + ob), PB = ld2t<NT>(p2 + ob);
    double2 WA = make_double2(0.0, 0.0), WB = WA;
    if constexpr (HAS_W) {
      WA = ld2t<NT>(w2 + oa);
      WB = ld2t<NT>(w2 + ob);
    }
    {
      const One n0 = step_full_one<DIST, MODE, POW2>(XA.x, VA.x, WA.x, PA.x, sE0, sEh, sR, a);
      const One n1 = step_full_one<DIST, MODE, POW2>(XA.y, VA.y, WA.y, PA.y, sE0, sEh, sR, a);
      st2t<NT>(x2 + oa, n0.x, n1.x);
      if constexpr (PUSH_V) st2t<NT>(v2 + oa, n0.v, n1.v);
      if constexpr (HAS_W) st2t<NT>(w2 + oa, n0.w, n1.w);
    }
    if (hb) {
      const One n0 = step_full_one<DIST, MODE, POW2>(XB.x, VB.x, WB.x, PB.x, sE0, sEh, sR, a);
      const One n1 = step_full_one<DIST, MODE, POW2>(XB.y, VB.y, WB.y, PB.y, sE0, sEh, sR, a);
      st2t<NT>(x2 + ob, n0.x, n1.x);
      if constexpr (PUSH_V) st2t<NT>(v2 + ob, n0.v, n1.v);
      if constexpr (HAS_W) st2t<NT>(w2 + ob, n0.w, n1.w);
    }
  }
#elif PIC1DP_STEP_PIPE == 1
  // tuning variant: the next trip's loads are issued before this trip's arithmetic
  {
    int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    double2 X = make_double2(0.0, 0.0), V = X, P = X, W = X;
    if (j < npair) {
      const int64_t o = tidx2(j);
      X = ld2t<NT>(x2 + o), V = ld2t<NT>(v2 + o), P = ld2t<NT>(p2 + o);
      if constexpr (HAS_W) W = ld2t<NT>(w2 + o);
    }
    for (; j < npair; j += stride) {
      const int64_t o = tidx2(j), jn = j + stride, on = tidx2(jn < npair ? jn : j);
      const double2 Xn = ld2t<NT>(x2 + on), Vn = ld2t<NT>(v2 + on), Pn = ld2t<NT>(p2 + on);
      double2 Wn = make_double2(0.0, 0.0);
      if constexpr (HAS_W) Wn = ld2t<NT>(w2 + on);
      const One n0 = step_full_one<DIST, MODE, POW2>(X.x, V.x, W.x, P.x, sE0, sEh, sR, a);
      const One n1 = step_full_one<DIST, MODE, POW2>(X.y, V.y, W.y, P.y, sE0, sEh, sR, a);
      st2t<NT>(x2 + o, n0.x, n1.x);
      if constexpr (PUSH_V) st2t<NT>(v2 + o, n0.v, n1.v);
      if constexpr (HAS_W) st2t<NT>(w2 + o, n0.w, n1.w);
      X = Xn, V = Vn, P = Pn, W = Wn;
    }
  }
#else
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += stride) {
    const int64_t o = tidx2(j);
    const double2 X = ld2t<NT>(x2 + o), V = ld2t<NT>(v2 + o), P = ld2t<NT>(p2 + o);
    double2 W = make_double2(0.0, 0.0);
    if constexpr (HAS_W) W = ld2t<NT>(w2 + o);
    double2 T = make_double2(0.0, 0.0);
    if constexpr (CARRY) T = ld2t<NT>(reinterpret_cast<const double2 *>(a.t2) + j);
    const One n0 = step_full_one<DIST, MODE, POW2, CARRY>(X.x, V.x, W.x, P.x, sE0, sEh, sR, a, T.x);
    PAIR_FENCE();
    const One n1 = step_full_one<DIST, MODE, POW2, CARRY>(X.y, V.y, W.y, P.y, sE0, sEh, sR, a, T.y);
    st2t<NT>(x2 + o, n0.x, n1.x);
    if constexpr (PUSH_V) st2t<NT>(v2 + o, n0.v, n1.v);
    if constexpr (HAS_W) st2t<NT>(w2 + o, n0.w, n1.w);
    if constexpr (DIAG) {
      ptcldist_one<true, HAS_W>(n0.x, n0.v, P.x, n0.w, a.dg, bins, sums);
      ptcldist_one<true, HAS_W>(n1.x, n1.v, P.y, n1.w, a.dg, bins, sums);
    }
  }
#endif
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double w = HAS_W ? a.w[i] : 0.0;
    const One n = step_full_one<DIST, MODE, POW2, CARRY>(a.x[i], a.v[i], w, a.p[i], sE0, sEh, sR, a,
                                                         CARRY ? a.t2[a.np - 1] : 0.0);
    a.x[i] = n.x;
    if constexpr (PUSH_V) a.v[i] = n.v;
    if constexpr (HAS_W) a.w[i] = n.w;
    if constexpr (DIAG) ptcldist_one<true, HAS_W>(n.x, n.v, a.p[i], n.w, a.dg, bins, sums);
  }
  __syncthreads();
  flush_rho(sR0, a.rho, a.g);
  if constexpr (DIAG) ptcldist_finish<true, HAS_W>(a.dg, bins, sums, sH + ntot, a.dist_out, a.dist_partial);
}

// ---------------------------------------------------------------------------
// k_step_one: ONE pass over the markers per time step.
// The first sub-step's kernel exists only to deposit the half-step charge, from which the
// half-step field Eh follows.  But the half push is LINEAR in the field it sees:
//     x' = x + dt/2 v                                     (no field at all, :261)
//     w' = w + dt/2 (p - w) (-f0'/f0)(v) Z/m * E(x)        (:268-329; linear: p instead of p - w)
// and the field is the kept modes' amplitudes times fixed tables (src/pic1dp_field.F90:251-257):
//     E(x) = sum_m  re_m A_m(x) + im_m B_m(x),   A_m = gather of 2 cos, B_m = gather of -2 sin.
// Hence the charge the NEXT step's first sub-step would deposit is
//     rho_h = R0 + sum_m re_m RA_m + im_m RB_m,
//     R0 = deposit of w at x',  RA_m = deposit of c A_m(x) at x',  RB_m likewise,
//     c = dt/2 (p - w)(-f0'/f0)(v) Z/m,
// and R0, RA_m, RB_m depend on the markers only -- this kernel, which has just computed the new
// (x, v, w), deposits them as well.  When the field of the new state is solved (re, im known),
// rho_h is one small combination (k_pred_combine), Eh one more solve, and the next step needs no
// first-sub-step pass: 56 B per marker and step instead of 88.  The second sub-step's push is
// untouched (same operations, same order as the reference given E0 and Eh); rho_h differs from a
// marker-by-marker deposit of w' by rounding only (same algebra, different grouping: ~1e-15 relative,
// the order of magnitude the atomics' order contributes anyway).  Full-f: rho_h = R0 (p at x').
// Falls back to k_step_half + k_step_full when nmode > PRED_MAX_MODES or the tiles outgrow the LDS.
// ---------------------------------------------------------------------------

// c = dt/2 * (p - w) * (-f0'/f0)(v) * Z / m  (linear: p), and -f0'/f0(v) itself for the carry
template <int DIST, int MODE, int POW2, class D>
__device__ __forceinline__ double pred_coef_core(double v, double w, double p, double dt, const SpeciesConst &s, D &dv,
                                                 double &t2) {
  const double tmp1 = (MODE == MODE_DF_LIN) ? p : (p - w);
  t2 = dlnf0<DIST, POW2>(v, s, dv);
  return divc<POW2>(dt * tmp1 * t2 * s.Z, s.m, s.r_m, dv);
}
template <int DIST, int MODE, int POW2>
__device__ __forceinline__ double pred_coef(double v, double w, double p, double dt, const SpeciesConst &s, double &t2) {
  if constexpr (POW2 == 0) {
    if (s.fastc) {
      DivFast dv;
      const double c = pred_coef_core<DIST, MODE, POW2>(v, w, p, dt, s, dv, t2);
      if (dv.ok()) return c;
    }
  }
  DivTrue dv;
  return pred_coef_core<DIST, MODE, POW2>(v, w, p, dt, s, dv, t2);
}

// the prediction deposits of one marker in its NEW state n (x wrapped); returns -f0'/f0(n.v)
// (ix, wl): cell and left weight of n.x, where the next step gathers its field (:250-257) -- the deposit
// of the new state has just computed them.
// The kernel runs at the package power limit with its FP64 pipes ~77 % busy at the clock that leaves
// (DESIGN.md 7), so instructions are what this part is written for:
// * the tables lie cell by cell, sAB[cell][A_0 B_0 (A_1 B_1)] with a guard cell, and the accumulators likewise,
//   sP[cell][R0 RA_0 RB_0 (...)] with TWO guard cells (folded into cells 0 and 1 at the flush): one address
//   per cell instead of one per tile and cell, no wrap-around of the right-hand cell, no clamp;
// * the cell of x' needs no exact division and no wrap of the position: the prediction equals a marker-by-
//   marker deposit to rounding anyway, and a deposit is continuous across a cell boundary (a position within
//   an ulp of one puts ~0 into the far cell either way) -- s = x' * (nx / lx), one multiplication, and the
//   CELL is wrapped (x' in (-lx, 2 lx) unless a marker crosses a box length in half a step: cells 0 ... nx,
//   right-hand neighbour up to nx + 1; cvt(NaN) = 0);
// * the constants of c are folded (pred_k = dt/2 Z/m).
template <int DIST, int MODE, int POW2>
__device__ __forceinline__ double pred_one(const One &n, double p, int ix, double wl, const double *sAB, double *sP,
                                           const StepArgsDev &a) {
  const int nm = a.pred_nm, np1 = 1 + 2 * nm;
  const double xh = fma(a.dt_half, n.v, n.x);     // the next step's half push of x (:261), to rounding
  const double sh = xh * a.snx;                   // its cell, wrapped as an integer (:102-108 to rounding)
  const double fh = floor(sh);
  int ih = static_cast<int>(fh);
  const double wr = sh - fh, wh = 1.0 - wr;
  ih = ih < 0 ? ih + a.g.nx : ih;
  ih = ih > a.g.nx ? ih - a.g.nx : ih;            // (cell nx is a guard cell)
  if (static_cast<unsigned>(ih) > static_cast<unsigned>(a.g.nx)) {  // more than a box length in half a step, NaN
    ih = ih % a.g.nx;
    if (ih < 0) ih += a.g.nx;
  }
  double *cl = sP + __mul24(ih, np1), *cr = cl + np1;
  double t2 = 0.0;
  if constexpr (MODE == MODE_FULLF) {
    lds_add(cl, wh * p);
    lds_add(cr, wr * p);
  } else {
    lds_add(cl, wh * n.w);
    lds_add(cr, wr * n.w);
    const double tmp1 = (MODE == MODE_DF_LIN) ? p : (p - n.w);
    if constexpr (POW2 == 0) {
      if (a.s.fastc) {
        DivFast dv;
        t2 = dlnf0<DIST, POW2>(n.v, a.s, dv);
        if (!dv.ok()) {
          DivTrue dt;
          t2 = dlnf0<DIST, POW2>(n.v, a.s, dt);
        }
      } else {
        DivTrue dt;
        t2 = dlnf0<DIST, POW2>(n.v, a.s, dt);
      }
    } else {
      DivTrue dt;
      t2 = dlnf0<DIST, POW2>(n.v, a.s, dt);
    }
    const double c = tmp1 * t2 * a.pred_k;
    const double *gl = sAB + __mul24(ix, 2 * nm), *gr = gl + 2 * nm;
    const double wlr = 1.0 - wl;
    for (int m = 0; m < nm; ++m) {
      const double2 tl = *reinterpret_cast<const double2 *>(gl + 2 * m), tr = *reinterpret_cast<const double2 *>(gr + 2 * m);
      const double A = fma(tr.x, wlr, tl.x * wl), B = fma(tr.y, wlr, tl.y * wl);  // (contraction is fine here)
      const double cA = c * A, cB = c * B;
      lds_add(cl + 1 + m, wh * cA);
      lds_add(cr + 1 + m, wr * cA);
      lds_add(cl + 1 + nm + m, wh * cB);
      lds_add(cr + 1 + nm + m, wr * cB);
    }
  }
  return t2;
}

// T2: 0 no carry of -f0'/f0; 1 this step evaluates it, the next step's value is stored; 2 this
// step's value is loaded (stored by the previous k_step_one), the next step's stored
template <int DIST, int MODE, int POW2, bool NT, int T2>
__global__ void __launch_bounds__(1024) PIC1DP_SIX_WAVES k_step_one(const StepArgsDev a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exp_table_init();
  const int nx = a.g.nx, nm = a.pred_nm, np1 = 1 + 2 * nm;
  const int ne = (nx + 2) & ~1;
  double *sE0 = reinterpret_cast<double *>(smem);
  double *sEh = sE0 + ne;
  double *sAB = sEh + ne;                                        // [nx + 1][2 nm]: A_0 B_0 (A_1 B_1) per cell
  double *sR0 = sAB + static_cast<size_t>(nx + 1) * 2 * nm;
  double *sP = sR0 + ((nx * a.g.rcopies + 2) & ~1);              // [nx + 2][1 + 2 nm]: R0 RA_m RB_m per cell
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    sE0[i] = a.E0[i];
    sEh[i] = a.Eh[i];
  }
  for (int i = threadIdx.x; i < nm * (nx + 1); i += blockDim.x) {
    const int m = i / (nx + 1), c = i - m * (nx + 1), cs = c < nx ? c : 0;  // cell nx: the guard, = cell 0
    sAB[c * 2 * nm + 2 * m] = a.tabA[m * nx + cs];
    sAB[c * 2 * nm + 2 * m + 1] = a.tabB[m * nx + cs];
  }
  zero_rho(sR0, a.g);
  for (int i = threadIdx.x; i < np1 * (nx + 2); i += blockDim.x) sP[i] = 0.0;
  if (threadIdx.x == 0) {
    sE0[nx] = a.E0[0];
    sEh[nx] = a.Eh[0];
  }
  __syncthreads();
  double *sR = my_rho_copy(sR0, a.g);
  constexpr bool HAS_W = (MODE != MODE_FULLF);
  constexpr bool PUSH_V = (MODE != MODE_DF_LIN);
  constexpr bool CARRY_IN = (T2 == 2) && HAS_W;
  constexpr bool CARRY_OUT = (T2 != 0) && HAS_W;
  const int64_t npair = a.np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 *x2 = reinterpret_cast<double2 *>(a.x);
  double2 *v2 = reinterpret_cast<double2 *>(a.v);
  double2 *w2 = reinterpret_cast<double2 *>(a.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
  double2 *t2 = reinterpret_cast<double2 *>(a.t2);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += stride) {
    const int64_t o = tidx2(j);
    const double2 X = ld2t<NT>(x2 + o), V = ld2t<NT>(v2 + o), P = ld2t<NT>(p2 + o);
    double2 W = make_double2(0.0, 0.0), T = make_double2(0.0, 0.0);
    if constexpr (HAS_W) W = ld2t<NT>(w2 + o);
    if constexpr (CARRY_IN) T = ld2t<NT>(t2 + j);
    int i0, i1;
    double l0, l1;
    // the two markers of a pair one after the other (PAIR_FENCE): interleaving their four exp chains
    // costs more registers than six waves per SIMD leave
    const One n0 = step_full_one<DIST, MODE, POW2, CARRY_IN>(X.x, V.x, W.x, P.x, sE0, sEh, sR, a, T.x, &i0, &l0);
    const double u0 = pred_one<DIST, MODE, POW2>(n0, P.x, i0, l0, sAB, sP, a);
    PAIR_FENCE();
    const One n1 = step_full_one<DIST, MODE, POW2, CARRY_IN>(X.y, V.y, W.y, P.y, sE0, sEh, sR, a, T.y, &i1, &l1);
    const double u1 = pred_one<DIST, MODE, POW2>(n1, P.y, i1, l1, sAB, sP, a);
    st2t<NT>(x2 + o, n0.x, n1.x);
    if constexpr (PUSH_V) st2t<NT>(v2 + o, n0.v, n1.v);
    if constexpr (HAS_W) st2t<NT>(w2 + o, n0.w, n1.w);
    if constexpr (CARRY_OUT) st2t<NT>(t2 + j, u0, u1);
  }
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double w = HAS_W ? a.w[i] : 0.0, p = a.p[i];
    int ic;
    double lc;
    const One n = step_full_one<DIST, MODE, POW2, CARRY_IN>(a.x[i], a.v[i], w, p, sE0, sEh, sR, a,
                                                            CARRY_IN ? a.t2[a.np - 1] : 0.0, &ic, &lc);
    a.x[i] = n.x;
    if constexpr (PUSH_V) a.v[i] = n.v;
    if constexpr (HAS_W) a.w[i] = n.w;
    const double u = pred_one<DIST, MODE, POW2>(n, p, ic, lc, sAB, sP, a);
    if constexpr (CARRY_OUT) a.t2[a.np - 1] = u;
  }
  __syncthreads();
  flush_rho(sR0, a.rho, a.g);
  // the guard cells nx, nx + 1 are cells 0, 1 (mod nx); then one global atomic per cell and slice
  if (threadIdx.x < 2 * np1) {
    const int g = threadIdx.x / np1, k = threadIdx.x - g * np1;
    const int to = (nx + g) % nx;
    if (g == 0 || to != 0 || nx > 1) lds_add(&sP[to * np1 + k], sP[(nx + g) * np1 + k]);
  }
  __syncthreads();
  {
    const int rot = static_cast<int>((static_cast<long long>(blockIdx.x) * nx) / gridDim.x);
    for (int i = threadIdx.x; i < np1 * nx; i += blockDim.x) {
      const int k = i / nx;
      int c = i - k * nx + rot;
      if (c >= nx) c -= nx;
      const double val = sP[c * np1 + k];
      if (val != 0.0) glb_add(&a.pred[static_cast<size_t>(k) * nx + c], val);
    }
  }
}

// ---------------------------------------------------------------------------
// k_step_sums: the one pass per step for grids whose prediction tiles outgrow the LDS (nx > ~2400: the
// Landau scaling run's nx = 4096), one kept mode.  Two observations replace five of k_step_one's tiles:
// * the solve that turns the predicted half-step charge into Eh only ever looks at its projections on the
//   kept mode's tables (src/pic1dp_field.F90:231-240), and the projection of a linear (CIC) deposit of q at x'
//   is q times the gather of the table at x':  sum_c fre[c] deposit[c] = q A(x') / 2.  Hence, with
//   c = dt/2 (p - w)(-f0'/f0)(v) Z/m at the marker's NEW state,
//       sum_c fre[c] rho_h[c] = 1/2 [ K0c + re K1c + im K2c ],   K0c = sum_i Z w_i A(x'_i),
//       K1c = sum_i Z c_i A(x_i) A(x'_i),  K2c = sum_i Z c_i B(x_i) A(x'_i),  and K0s, K1s, K2s with B(x'_i):
//   six scalars per rank instead of three tiles -- accumulated in registers, reduced per workgroup, one
//   global atomic each; when the new state's field is solved (re, im known) Eh follows from them
//   (k_field_solve_pair_sums);
// * Eh is its kept mode times the tables the kernel holds anyway: gathered from A, B and (re_h, im_h)
//   (ModeField: the same bits as a staged tile of Eh).
// LDS: E0, A, B, rho = 4 tiles (128 KiB at nx = 4096).  The six accumulators and the extra gathers cost
// registers (101-117 VGPRs for the exp-bearing distributions: four waves per SIMD), which is why k_step_one
// stays the kernel wherever its tiles fit (profiles/r02/experiments/pred_six_sums_*.log).
// The second sub-step's push is untouched; Eh differs from the solve of a marker-by-marker deposit by
// rounding only, as with k_step_one.  Full-f: q = p, K1 = K2 = 0.
// ---------------------------------------------------------------------------
struct PredSums {
  double k0c = 0.0, k1c = 0.0, k2c = 0.0, k0s = 0.0, k1s = 0.0, k2s = 0.0;
};

// the six sums' terms of one marker in its NEW state n (x wrapped); (ix, wl): cell and left weight of n.x,
// where the next step gathers its field (:250-257).  Returns -f0'/f0(n.v) for the carry.
template <int DIST, int MODE, int POW2>
__device__ __forceinline__ double pred_one_sums(const One &n, double p, int ix, double wl, const double *sA,
                                                const double *sB, PredSums &k, const StepArgsDev &a) {
  double t2 = 0.0;
  double cA = 0.0, cB = 0.0;
  if constexpr (MODE != MODE_FULLF) {
    // the long chain first (-f0'/f0: exp, division), with little else alive
    const double c = a.s.Z * pred_coef<DIST, MODE, POW2>(n.v, n.w, p, a.dt_half, a.s, t2);
    PAIR_FENCE();
    double A = sA[ix] * wl;                       // tables at x: the field the half push will see
    A = A + sA[ix + 1] * (1.0 - wl);
    double B = sB[ix] * wl;
    B = B + sB[ix + 1] * (1.0 - wl);
    cA = c * A, cB = c * B;
  }
  double xh = n.x + a.dt_half * n.v;              // the next step's half push of x (:261)
  xh = wrap(xh, a.g.lx);                          // and the wrap + cell of its deposit (:102-108)
  int ih;
  double wh;
  locate(xh, a.g, ih, wh);
  double Ah = sA[ih] * wh;                        // tables at x': the deposit's projection weights
  Ah = Ah + sA[ih + 1] * (1.0 - wh);
  double Bh = sB[ih] * wh;
  Bh = Bh + sB[ih + 1] * (1.0 - wh);
  const double q = a.s.Z * (MODE == MODE_FULLF ? p : n.w);
  k.k0c += q * Ah;
  k.k0s += q * Bh;
  if constexpr (MODE != MODE_FULLF) {
    k.k1c += cA * Ah;
    k.k2c += cB * Ah;
    k.k1s += cA * Bh;
    k.k2s += cB * Bh;
  }
  return t2;
}

template <int DIST, int MODE, int POW2, bool NT, int T2>
__global__ void __launch_bounds__(1024) k_step_sums(const StepArgsDev a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exp_table_init();
  const int nx = a.g.nx;
  const int ne = (nx + 2) & ~1;
  double *sE0 = reinterpret_cast<double *>(smem);
  double *sA = sE0 + ne;
  double *sB = sA + ne;
  double *sR0 = sB + ne;
  double *sScr = sR0 + ((nx * a.g.rcopies + 2) & ~1);  // [16] reduction scratch
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    sE0[i] = a.E0[i];
    sA[i] = a.tabA[i];
    sB[i] = a.tabB[i];
  }
  zero_rho(sR0, a.g);
  if (threadIdx.x == 0) {
    sE0[nx] = a.E0[0];
    sA[nx] = a.tabA[0];
    sB[nx] = a.tabB[0];
  }
  __syncthreads();
  const ModeField sEh{sA, sB, *a.eh_re, *a.eh_im};
  double *sR = my_rho_copy(sR0, a.g);
  constexpr bool HAS_W = (MODE != MODE_FULLF);
  constexpr bool PUSH_V = (MODE != MODE_DF_LIN);
  constexpr bool CARRY_IN = (T2 == 2) && HAS_W;
  constexpr bool CARRY_OUT = (T2 != 0) && HAS_W;
  PredSums ks;
  const int64_t npair = a.np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 *x2 = reinterpret_cast<double2 *>(a.x);
  double2 *v2 = reinterpret_cast<double2 *>(a.v);
  double2 *w2 = reinterpret_cast<double2 *>(a.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
  double2 *t2 = reinterpret_cast<double2 *>(a.t2);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += stride) {
    const int64_t o = tidx2(j);
    const double2 X = ld2t<NT>(x2 + o), V = ld2t<NT>(v2 + o), P = ld2t<NT>(p2 + o);
    double2 W = make_double2(0.0, 0.0), T = make_double2(0.0, 0.0);
    if constexpr (HAS_W) W = ld2t<NT>(w2 + o);
    if constexpr (CARRY_IN) T = ld2t<NT>(t2 + j);
    int i0, i1;
    double l0, l1;
    const One n0 = step_full_one<DIST, MODE, POW2, CARRY_IN>(X.x, V.x, W.x, P.x, sE0, sEh, sR, a, T.x, &i0, &l0);
    PAIR_FENCE();
    const double u0 = pred_one_sums<DIST, MODE, POW2>(n0, P.x, i0, l0, sA, sB, ks, a);
    PAIR_FENCE();
    const One n1 = step_full_one<DIST, MODE, POW2, CARRY_IN>(X.y, V.y, W.y, P.y, sE0, sEh, sR, a, T.y, &i1, &l1);
    PAIR_FENCE();
    const double u1 = pred_one_sums<DIST, MODE, POW2>(n1, P.y, i1, l1, sA, sB, ks, a);
    st2t<NT>(x2 + o, n0.x, n1.x);
    if constexpr (PUSH_V) st2t<NT>(v2 + o, n0.v, n1.v);
    if constexpr (HAS_W) st2t<NT>(w2 + o, n0.w, n1.w);
    if constexpr (CARRY_OUT) st2t<NT>(t2 + j, u0, u1);
  }
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double w = HAS_W ? a.w[i] : 0.0, p = a.p[i];
    int ic;
    double lc;
    const One n = step_full_one<DIST, MODE, POW2, CARRY_IN>(a.x[i], a.v[i], w, p, sE0, sEh, sR, a,
                                                            CARRY_IN ? a.t2[a.np - 1] : 0.0, &ic, &lc);
    a.x[i] = n.x;
    if constexpr (PUSH_V) a.v[i] = n.v;
    if constexpr (HAS_W) a.w[i] = n.w;
    const double u = pred_one_sums<DIST, MODE, POW2>(n, p, ic, lc, sA, sB, ks, a);
    if constexpr (CARRY_OUT) a.t2[a.np - 1] = u;
  }
  __syncthreads();
  flush_rho(sR0, a.rho, a.g);
  // the six sums: workgroup reduction, one global atomic each
  const double r0 = block_sum(ks.k0c, sScr), r1 = block_sum(ks.k1c, sScr), r2 = block_sum(ks.k2c, sScr);
  const double r3 = block_sum(ks.k0s, sScr), r4 = block_sum(ks.k1s, sScr), r5 = block_sum(ks.k2s, sScr);
  if (threadIdx.x == 0) {
    glb_add(a.pred + 0, r0);
    glb_add(a.pred + 1, r1);
    glb_add(a.pred + 2, r2);
    glb_add(a.pred + 3, r3);
    glb_add(a.pred + 4, r4);
    glb_add(a.pred + 5, r5);
  }
}

template <typename K>
hipError_t launch_step_kernel(K kern, const StepArgsDev &d, const LaunchCfg &lc, hipStream_t st) {
  if (lc.lds > 64 * 1024) {  // opt in to > 64 KiB of dynamic LDS (idempotent, cheap)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PARTICLE_LDS_CAP);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(lc.blocks), dim3(lc.threads), lc.lds, st, d);
  return hipGetLastError();
}

template <int DIST, int MODE, int POW2, bool CARRY = false>
hipError_t launch_step_dmp(const StepArgsDev &d, bool full, const LaunchCfg &lc, hipStream_t st) {
#if PIC1DP_STEP_PIPE == 0
  if (full && d.pred && d.pred_nm < 0) {  // one pass per step, prediction as six sums (large grids)
    const int t2m = d.t2 ? d.t2_mode : 0;
    if (d.nt) {
      if (t2m == 2) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 2>, d, lc, st);
      if (t2m == 1) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 1>, d, lc, st);
      return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 0>, d, lc, st);
    }
    if (t2m == 2) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 2>, d, lc, st);
    if (t2m == 1) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 1>, d, lc, st);
    return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 0>, d, lc, st);
  }
  if (full && d.pred) {  // one pass per step: also predicts the next step's first-sub-step charge
    const int t2m = d.t2 ? d.t2_mode : 0;
    if (d.nt) {
      if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 2>, d, lc, st);
      if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 1>, d, lc, st);
      return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 0>, d, lc, st);
    }
    if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 2>, d, lc, st);
    if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 1>, d, lc, st);
    return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 0>, d, lc, st);
  }
  if (full && d.dist_out)  // with the diagnostics of output_all
    return d.nt ? launch_step_kernel(k_step_full<DIST, MODE, POW2, true, CARRY, true>, d, lc, st)
                : launch_step_kernel(k_step_full<DIST, MODE, POW2, false, CARRY, true>, d, lc, st);
#endif
  if (d.nt)
    return full ? launch_step_kernel(k_step_full<DIST, MODE, POW2, true, CARRY, false>, d, lc, st)
                : launch_step_kernel(k_step_half<DIST, MODE, POW2, true, CARRY>, d, lc, st);
  return full ? launch_step_kernel(k_step_full<DIST, MODE, POW2, false, CARRY, false>, d, lc, st)
              : launch_step_kernel(k_step_half<DIST, MODE, POW2, false, CARRY>, d, lc, st);
}

template <int DIST>
hipError_t launch_step_d(const StepArgsDev &d, int deltaf, int linear, bool full, const LaunchCfg &lc,
                         hipStream_t st) {
  const bool pow2 = d.s.pow2 != 0;
  // full-f evaluates no f0 derivative (one instantiation serves all DIST), but
  // still divides by the mass in the v push
  if (!deltaf)
    return pow2 ? launch_step_dmp<0, MODE_FULLF, 1>(d, full, lc, st)
                : launch_step_dmp<0, MODE_FULLF, 0>(d, full, lc, st);
  // general divisor constants and an exp-bearing distribution: -f0'/f0 carried between the kernels
  const bool carry = !pow2 && d.t2 != nullptr && (DIST == 2 || DIST == 3) && PIC1DP_STEP_PIPE == 0;
  if (linear) {
    if constexpr (DIST == 2 || DIST == 3)
      if (carry) return launch_step_dmp<DIST, MODE_DF_LIN, 0, true>(d, full, lc, st);
    return pow2 ? launch_step_dmp<DIST, MODE_DF_LIN, 1>(d, full, lc, st)
                : launch_step_dmp<DIST, MODE_DF_LIN, 0>(d, full, lc, st);
  }
  if (d.s.unit) return launch_step_dmp<DIST, MODE_DF_NL, 2>(d, full, lc, st);
  if constexpr (DIST == 2 || DIST == 3)
    if (carry) return launch_step_dmp<DIST, MODE_DF_NL, 0, true>(d, full, lc, st);
  return pow2 ? launch_step_dmp<DIST, MODE_DF_NL, 1>(d, full, lc, st)
              : launch_step_dmp<DIST, MODE_DF_NL, 0>(d, full, lc, st);
}

// stand-alone wrap + deposit (interaction_collect_charge loop :96-114)
__global__ void __launch_bounds__(1024)
k_deposit(double *x, const double *q, double *rho, int64_t np, const GridConst g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sR0 = reinterpret_cast<double *>(smem);
  zero_rho(sR0, g);
  __syncthreads();
  double *sR = my_rho_copy(sR0, g);
  const int64_t npair = np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 *x2 = reinterpret_cast<double2 *>(x);
  const double2 *q2 = reinterpret_cast<const double2 *>(q);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair;
       j += stride) {
    const int64_t o = tidx2(j);
    double2 X = ld2(x2 + o);
    const double2 Q = ld2(q2 + o);
    X.x = deposit_one(X.x, Q.x, sR, g);
    X.y = deposit_one(X.y, Q.y, sR, g);
    st2(x2 + o, X.x, X.y);
  }
  if ((np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(np - 1);
    x[i] = deposit_one(x[i], q[i], sR, g);
  }
  __syncthreads();
  flush_rho(sR0, rho, g);
}

template <int DIST, int MODE, int POW2, bool IRK2, bool FUSED>
hipError_t launch_push_t(const PushArgs &a, const LaunchCfg &lc, hipStream_t st) {
  auto kern = k_push<DIST, MODE, POW2, IRK2, FUSED>;
  static bool big_lds_ok = false;  // opt in once to > 64 KiB of dynamic LDS (nx >= 4096)
  if (lc.lds > 64 * 1024 && !big_lds_ok) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PARTICLE_LDS_CAP);
    if (e != hipSuccess) return e;
    big_lds_ok = true;
  }
  hipLaunchKernelGGL(kern, dim3(lc.blocks), dim3(lc.threads), lc.lds, st, a);
  return hipGetLastError();
}

template <int DIST, int MODE, int POW2>
hipError_t launch_push_dm(const PushArgs &a, bool fused, const LaunchCfg &lc, hipStream_t st) {
  const bool irk2 = a.irk == 2;
  if (irk2) {
    return fused ? launch_push_t<DIST, MODE, POW2, true, true>(a, lc, st)
                 : launch_push_t<DIST, MODE, POW2, true, false>(a, lc, st);
  }
  return fused ? launch_push_t<DIST, MODE, POW2, false, true>(a, lc, st)
               : launch_push_t<DIST, MODE, POW2, false, false>(a, lc, st);
}

template <int DIST>
hipError_t launch_push_d(const PushArgs &a, bool fused, const LaunchCfg &lc, hipStream_t st) {
  const int mode = a.deltaf ? (a.linear ? MODE_DF_LIN : MODE_DF_NL) : MODE_FULLF;
  const bool pow2 = a.s.pow2 != 0;
  switch (mode) {
    case MODE_DF_NL:
      return pow2 ? launch_push_dm<DIST, MODE_DF_NL, true>(a, fused, lc, st)
                  : launch_push_dm<DIST, MODE_DF_NL, false>(a, fused, lc, st);
    case MODE_DF_LIN:
      return pow2 ? launch_push_dm<DIST, MODE_DF_LIN, true>(a, fused, lc, st)
                  : launch_push_dm<DIST, MODE_DF_LIN, false>(a, fused, lc, st);
    default:
      // full-f evaluates no f0 derivative (one instantiation serves all DIST)
      // but still divides by the mass in the v push
      return pow2 ? launch_push_dm<0, MODE_FULLF, true>(a, fused, lc, st)
                  : launch_push_dm<0, MODE_FULLF, false>(a, fused, lc, st);
  }
}

}  // namespace

hipError_t launch_push(const PushArgs &a, bool fused_deposit, const LaunchCfg &lc,
                       hipStream_t st) {
  switch (a.iptcldist) {
    case 1: return launch_push_d<1>(a, fused_deposit, lc, st);
    case 2: return launch_push_d<2>(a, fused_deposit, lc, st);
    case 3: return launch_push_d<3>(a, fused_deposit, lc, st);
    default: return launch_push_d<0>(a, fused_deposit, lc, st);
  }
}

hipError_t launch_step(const StepArgs &a, bool full, const LaunchCfg &lc, hipStream_t st) {
  StepArgsDev d{};
  d.x = a.x;
  d.v = a.v;
  d.w = a.w;
  d.p = a.p;
  d.E0 = a.E0;
  d.Eh = a.Eh;
  d.rho = a.rho;
  d.np = a.np;
  d.dt_half = a.dt_half;
  d.dt_full = a.dt_full;
  d.g = a.g;
  d.s = a.s;
  d.nt = a.stream_nt;
  d.t2 = a.t2;
  d.dg = a.dg;
  d.dist_out = a.dist_out;
  d.dist_partial = a.dist_partial;
  d.tabA = a.tabA;
  d.tabB = a.tabB;
  d.pred = a.pred;
  d.pred_nm = a.pred_kind == 2 ? -1 : a.pred_nm;  // -1: k_step_sums
  d.t2_mode = a.t2_mode;
  d.eh_re = a.eh_re;
  d.eh_im = a.eh_im;
  d.snx = a.g.dnx / a.g.lx;
  d.pred_k = a.dt_half * a.s.Z / a.s.m;
  switch (a.iptcldist) {
    case 1: return launch_step_d<1>(d, a.deltaf, a.linear, full, lc, st);
    case 2: return launch_step_d<2>(d, a.deltaf, a.linear, full, lc, st);
    case 3: return launch_step_d<3>(d, a.deltaf, a.linear, full, lc, st);
    default: return launch_step_d<0>(d, a.deltaf, a.linear, full, lc, st);
  }
}

hipError_t launch_deposit(double *x, const double *q, double *rho, int64_t np, const GridConst &g,
                          const LaunchCfg &lc, hipStream_t st) {
  static bool big_lds_ok = false;
  if (lc.lds > 64 * 1024 && !big_lds_ok) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_deposit),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    big_lds_ok = true;
  }
  hipLaunchKernelGGL(k_deposit, dim3(lc.blocks), dim3(lc.threads), lc.lds, st, x, q, rho, np, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// field kernels (nx <= a few thousand: one workgroup, latency-bound, tiny)
// ---------------------------------------------------------------------------
namespace {

constexpr int FIELD_THREADS = 256;
#ifndef PIC1DP_CHAIN_W
#define PIC1DP_CHAIN_W 16
#endif
constexpr int CHAIN_W = PIC1DP_CHAIN_W;  // prefetch depth of the serial mode sums

// charge2(:) = charge2(:) + charge1(:)*Z over species, from 0
// (src/pic1dp_interaction.F90:81,126-127); accumulators are re-zeroed
__device__ __forceinline__ double charge_local_one(const FieldArgs &f, int ix) {
  double c2 = 0.0;
  for (int s = 0; s < f.nspecies; ++s) {
    double *r = f.rho_sp + static_cast<size_t>(s) * f.nx + ix;
    double c1 = *r;
    *r = 0.0;
    for (int g = 1; g < f.rho_copies; ++g) {  // the copies the workgroups flushed into
      c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
      r[static_cast<size_t>(g) * f.rho_stride] = 0.0;
    }
    c2 = c2 + c1 * f.Z[s];
  }
  f.charge[ix] = c2;
  return c2;
}

__global__ void __launch_bounds__(FIELD_THREADS) k_charge_local(const FieldArgs f) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) charge_local_one(f, ix);
}

// chargeden = charge1*nx/lx (- Z*n0 per species for full-f)
// src/pic1dp_interaction.F90:138-148
__device__ __forceinline__ double chargeden_from(const FieldArgs &f, double charge1) {
  double cd = charge1 * f.dnx / f.lx;
  if (!f.deltaf)
    for (int s = 0; s < f.nspecies; ++s) cd = cd - f.Z[s] * f.n0[s];
  return cd;
}

template <bool WITH_LOCAL>
__global__ void __launch_bounds__(FIELD_THREADS) k_chargeden(const FieldArgs f) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x)
    f.chargeden[ix] = chargeden_from(f, WITH_LOCAL ? charge_local_one(f, ix) : f.charge[ix]);
}

// k_step_one's prediction turned into this rank's charge2 of the next step's first sub-step:
//   charge2_h = sum_s Z_s * (R0_s + sum_m re_m RA_sm + im_m RB_sm),   re / im = the kept modes of the
// field the markers were just advanced to.  The accumulators are consumed (re-zeroed).  The caller
// reduces f.charge over ranks and scales it (k_chargeden<false>) like any other charge2.
__global__ void __launch_bounds__(FIELD_THREADS) k_pred_combine(const FieldArgs f, double *pred, int nm_pred) {
  const int nx = f.nx, np1 = 1 + 2 * nm_pred;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    for (int s = 0; s < f.nspecies; ++s) {
      double *r = pred + static_cast<size_t>(s) * np1 * nx + ix;
      double c1 = r[0];
      r[0] = 0.0;
      for (int m = 0; m < nm_pred; ++m) {
        double *ra = r + static_cast<size_t>(1 + m) * nx, *rb = r + static_cast<size_t>(1 + nm_pred + m) * nx;
        c1 = c1 + f.mode_re[m] * *ra;
        c1 = c1 + f.mode_im[m] * *rb;
        *ra = 0.0;
        *rb = 0.0;
      }
      c2 = c2 + c1 * f.Z[s];
    }
    f.charge[ix] = c2;
  }
}

// RCCL path of a one-pass step: everything this rank contributes to the two charge sums of the step,
// packed for ONE all-reduce: pack[0] = charge2 of the new state, pack[1 + k] = sum_s Z_s * (R0, RA_m, RB_m)_s
// (the combination with the kept modes is linear, so the species sum and the sum over ranks commute with it).
__global__ void __launch_bounds__(FIELD_THREADS) k_charge_pack(const FieldArgs f, double *pred, int nm_pred, double *pack) {
  const int nx = f.nx, np1 = 1 + 2 * nm_pred;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    pack[ix] = charge_local_one(f, ix);
    for (int k = 0; k < np1; ++k) {
      double c2 = 0.0;
      for (int s = 0; s < f.nspecies; ++s) {
        double *r = pred + (static_cast<size_t>(s) * np1 + k) * nx + ix;
        c2 = c2 + *r * f.Z[s];
        *r = 0.0;
      }
      pack[static_cast<size_t>(1 + k) * nx + ix] = c2;
    }
  }
}

// ---- prediction as six sums (k_step_sums) ----
// The forward sums sum_c fre[c] cd_h[c], sum_c fim[c] cd_h[c] of the NEXT first sub-step's charge density
// from the six sums K (summed over species with Z, and over ranks) and the kept mode (re, im) of the field the
// markers were just advanced to (derivation at k_step_sums); PredTab: what the host knows of the tables
__device__ __forceinline__ void pred_forward_sums(const FieldArgs &f, const PredTab &pt, const double *K, double re, double im,
                                                  double &acc_c, double &acc_s) {
  double off = 0.0;
  if (!f.deltaf)
    for (int s = 0; s < f.nspecies; ++s) off = off + f.Z[s] * f.n0[s];  // chargeden -= Z n0, :142-148
  acc_c = 0.5 * (K[0] + re * K[1] + im * K[2]) * f.dnx / f.lx - off * pt.sum_fre;
  acc_s = 0.5 * (K[3] + re * K[4] + im * K[5]) * f.dnx / f.lx - off * pt.sum_fim;
}

// Call-site path: collect_charge after a noted push(1).  The host will call solve_field next, which works
// from field_chargeden -- so chargeden gets the kept mode's content of the half-step charge density,
//     cd[c] = alpha fre[c] + beta fim[c]   with   sum fre cd = acc_c,  sum fim cd = acc_s,
// from which the ordinary solve reproduces the predicted Eh (to rounding).  What the filter drops is absent
// from this chargeden; nothing in the reference driver reads chargeden between the sub-steps.
// K: the six sums (already summed over ranks); pred (or null) is re-zeroed.
__global__ void __launch_bounds__(FIELD_THREADS) k_pred_chargeden(const FieldArgs f, const PredTab pt, double *pred,
                                                                 const double *K) {
  __shared__ double sab[2];
  if (threadIdx.x == 0) {
    double ac, as;
    pred_forward_sums(f, pt, K, f.mode_re[0], f.mode_im[0], ac, as);
    const double det = pt.g11 * pt.g22 - pt.g12 * pt.g12;
    sab[0] = (ac * pt.g22 - as * pt.g12) / det;
    sab[1] = (as * pt.g11 - ac * pt.g12) / det;
    if (pred)
      for (int k = 0; k < 8; ++k) pred[k] = 0.0;
  }
  __syncthreads();
  const double alpha = sab[0], beta = sab[1];
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) f.chargeden[ix] = alpha * f.fre[ix] + beta * f.fim[ix];
}

// this rank's six sums into the head of f.charge (rest zero) for a reduction over ranks (call-site path)
__global__ void __launch_bounds__(FIELD_THREADS) k_pred_to_charge(const FieldArgs f, double *pred) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) f.charge[ix] = ix < 6 ? pred[ix] : 0.0;
  __syncthreads();
  if (threadIdx.x < 8) pred[threadIdx.x] = 0.0;
}

// k_charge_pack for the six sums: pack[0..nx) = charge2 of the new state, pack[nx..nx+8) = the sums (+ pad)
__global__ void __launch_bounds__(FIELD_THREADS) k_charge_pack_sums(const FieldArgs f, double *pred, double *pack) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) pack[ix] = charge_local_one(f, ix);
  if (threadIdx.x < 8) {
    pack[f.nx + threadIdx.x] = pred[threadIdx.x];
    pred[threadIdx.x] = 0.0;
  }
}

// field_solve_electric, src/pic1dp_field.F90:231-257, with the one-rank PETSc
// summation order: forward sums run over ascending ix in ONE thread per
// (mode, re/im) so the result is bit-identical to the sequential CPU loop.
// chargeden into sCD (and memory) from: the all-reduced charge (neither flag), the
// raw species deposits (WITH_LOCAL, one rank), or field_chargeden itself (FROM_CD)
template <bool WITH_LOCAL, bool FROM_CD>
__device__ __forceinline__ void solve_fill_chargeden(const FieldArgs &f, double *sCD) {
  const int nx = f.nx;
  // four grid points per thread per trip, all loads issued before the first use
  // (one memory round trip instead of four for nx = 1024)
  constexpr int U = 4;
  for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
    double c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      c[u] = 0.0;
      if (ix < nx) {
        if constexpr (FROM_CD) {
          c[u] = f.chargeden[ix];
        } else if constexpr (WITH_LOCAL) {  // src/pic1dp_interaction.F90:126-127
          for (int sp = 0; sp < f.nspecies; ++sp) {
            const double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            double c1 = *r;
            for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
            c[u] = c[u] + c1 * f.Z[sp];
          }
        } else {
          c[u] = f.charge[ix];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      if (ix < nx) {
        double cd = c[u];
        if constexpr (!FROM_CD) {
          if constexpr (WITH_LOCAL) {
            for (int sp = 0; sp < f.nspecies; ++sp)
              for (int g = 0; g < f.rho_copies; ++g)
                f.rho_sp[static_cast<size_t>(g) * f.rho_stride + static_cast<size_t>(sp) * nx + ix] = 0.0;
            f.charge[ix] = c[u];
          }
          cd = chargeden_from(f, c[u]);
          f.chargeden[ix] = cd;
        }
        sCD[ix] = cd;
      }
    }
  }
}

// the solve proper: chargeden in sCD -> mode_re/im, E (+ field energy)
// TREE: the forward sums as workgroup reductions instead of the reference's serial ascending-ix chains.
// Only for a field that has no reference order to keep: the half-step field predicted by k_step_one, whose
// charge already differs from a marker-by-marker deposit by rounding (0.6 us instead of 5.8 us at nx = 1024).
// sum of prod[0..nx) in ascending order, one lane, bit-identical to the sequential loop.
// One wave issues this whole chain, so every instruction counts (a wave64
// VALU or LDS instruction occupies its SIMD for 4 cycles whatever the exec
// mask): two register batches in ping-pong, no copies between them, and
// 16-byte LDS loads when the row is aligned.  16 dependent adds per batch
// cover the LDS round trip of the next one.
__device__ __forceinline__ double chain_sum_lds(const double *prod, int nx) {
  double acc = 0.0;
  int ix = 0;
  constexpr int W = CHAIN_W;
  if ((reinterpret_cast<uintptr_t>(prod) & 15) == 0) {
    double A[W], B[W];
    const int nb = nx / W;
    auto load = [](double (&r)[W], const double *q) {
#pragma unroll
      for (int k = 0; k < W; k += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(q + k);
        r[k] = t.x;
        r[k + 1] = t.y;
      }
    };
    if (nb > 0) load(A, prod);
    int b = 0;
    for (; b + 2 <= nb; b += 2) {
      load(B, prod + (b + 1) * W);
#pragma unroll
      for (int k = 0; k < W; ++k) acc = acc + A[k];
      if (b + 2 < nb) load(A, prod + (b + 2) * W);
#pragma unroll
      for (int k = 0; k < W; ++k) acc = acc + B[k];
    }
    if (b < nb) {
#pragma unroll
      for (int k = 0; k < W; ++k) acc = acc + A[k];
    }
    ix = nb * W;
  }
  for (; ix < nx; ++ix) acc = acc + prod[ix];
  return acc;
}

template <bool TREE = false>
__device__ __forceinline__ void solve_body(const FieldArgs &f, double *sCD, double *sMode, double *sScr,
                                           double *sTab) {
  const int nx = f.nx, nm = f.nmode;
  constexpr int U = 4;
  if constexpr (TREE) {
    for (int c = 0; c < 2 * nm; ++c) {  // chain c -> mode c>>1, (c&1 ? cos-table : -sin-table)
      const int m = c >> 1;
      const bool use_cos = c & 1;
      const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
      double part = 0.0;
      for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) part = part + tab[ix] * sCD[ix];
      const double acc = block_sum(part, sScr);
      if (threadIdx.x == 0) {
        if (use_cos) {
          const double im = acc * f.sc_im * f.grad_inv[m];
          sMode[nm + m] = im;
          f.mode_im[m] = im;
        } else {
          const double re = acc * f.sc_re * f.grad_inv[m];
          sMode[m] = re;
          f.mode_re[m] = re;
        }
      }
    }
    __syncthreads();
  } else {
  // forward partial DFT.  Every term table[ix]*chargeden[ix] is rounded on its
  // own in the reference too (no FMA), so the products are formed by all threads
  // at once (coalesced table reads) and only the additions run serially, in the
  // reference's ascending-ix order.
  if (f.tab_lds) {
    const int n = nm * nx;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double cd = sCD[i % nx];
      sTab[i] = f.fre[i] * cd;
      sTab[n + i] = f.fim[i] * cd;
    }
    __syncthreads();
  }
  // thread t -> mode t>>1, (t&1 ? cos-table : -sin-table)
  if (threadIdx.x < 2 * nm) {
    const int m = threadIdx.x >> 1;
    const bool use_cos = threadIdx.x & 1;
    double acc = 0.0;
    int ix = 0;
    if (f.tab_lds) {
      const double *prod = sTab + (use_cos ? 0 : nm * nx) + m * nx;
      acc = chain_sum_lds(prod, nx);
    } else {
      const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
      for (; ix + 8 <= nx; ix += 8) {
        double t[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          t[k] = tab[ix + k];
          r[k] = sCD[ix + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = acc + t[k] * r[k];
      }
      for (; ix < nx; ++ix) acc = acc + tab[ix] * sCD[ix];
    }
    // :234/:239 VecScale by -1/nx resp. 1/nx, then :243-247 times 1/k
    if (use_cos) {
      const double im = acc * f.sc_im * f.grad_inv[m];
      sMode[nm + m] = im;
      f.mode_im[m] = im;
    } else {
      const double re = acc * f.sc_re * f.grad_inv[m];
      sMode[m] = re;
      f.mode_re[m] = re;
    }
  }
  __syncthreads();
  }  // !TREE

  // inverse: E = 2*(Fre*mode_re + Fim*mode_im), ascending mode order :251-256
  double e2 = 0.0;
  if (nm == 1) {  // the usual case: both table reads of four grid points in flight together
    for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
      double tr[U], ti[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * FIELD_THREADS;
        tr[u] = ix < nx ? f.fre[ix] : 0.0;
        ti[u] = ix < nx ? f.fim[ix] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * FIELD_THREADS;
        if (ix < nx) {
          double s = 0.0;
          s = s + tr[u] * sMode[0];
          s = s + ti[u] * sMode[1];
          const double e = s * 2.0;
          f.E[ix] = e;
          e2 += e * e;
        }
      }
    }
  } else {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      double s = 0.0;
      for (int m = 0; m < nm; ++m) s = s + f.fre[static_cast<size_t>(m) * nx + ix] * sMode[m];
      for (int m = 0; m < nm; ++m) s = s + f.fim[static_cast<size_t>(m) * nx + ix] * sMode[nm + m];
      const double e = s * 2.0;
      f.E[ix] = e;
      e2 += e * e;
    }
  }
  if (f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

template <bool WITH_LOCAL, bool FROM_CD>
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve(const FieldArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);       // [nx]
  double *sMode = sCD + f.nx;                           // [2*nmode]: re then im
  double *sScr = sMode + 2 * f.nmode;                   // [16]
  double *sTab = sScr + 16;                             // [2][nmode][nx] when tab_lds
  solve_fill_chargeden<WITH_LOCAL, FROM_CD>(f, sCD);
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
}

// Call sites, one rank: collect_charge after a noted push(1) whose charge k_step_one has predicted, and the
// solve_field that follows, in one launch -- k_pred_combine (with the kept modes of the field as it is), the
// scaling, the solve.  The prediction accumulators are consumed.
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve_pred(const FieldArgs f, double *pred, int nm_pred) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  const int nx = f.nx, np1 = 1 + 2 * nm_pred;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    for (int s = 0; s < f.nspecies; ++s) {
      double *r = pred + static_cast<size_t>(s) * np1 * nx + ix;
      double c1 = r[0];
      r[0] = 0.0;
      for (int m = 0; m < nm_pred; ++m) {
        double *ra = r + static_cast<size_t>(1 + m) * nx, *rb = r + static_cast<size_t>(1 + nm_pred + m) * nx;
        c1 = c1 + f.mode_re[m] * *ra;
        c1 = c1 + f.mode_im[m] * *rb;
        *ra = 0.0;
        *rb = 0.0;
      }
      c2 = c2 + c1 * f.Z[s];
    }
    f.charge[ix] = c2;
    const double cd = chargeden_from(f, c2);
    f.chargeden[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();  // every thread has read mode_re / mode_im before solve_body overwrites them
  solve_body(f, sCD, sMode, sScr, sTab);
}

// ---------------------------------------------------------------------------
// One-hop charge exchange (replaces MPI_Allreduce, src/pic1dp_interaction.F90:130-135,
// for N processes = N GPUs of one node).  Every rank owns an exchange
// area (fine-grained device memory, mapped into every peer through hipIpc handles):
//     flags[2][XCHG_MAX_RANKS]   epoch of the last charge rank q delivered, per parity
//     slots[2][nranks][nx]       the charge2 vectors, one slot per source rank
// Exchange number e (1, 2, ...), parity e & 1:
//   1. charge2 = sum_s rho_s * Z_s (accumulators re-zeroed), stored into slot [rank]
//      of EVERY rank's area (system-scope stores: over xGMI for the peers),
//   2. every storing wave drains its stores (system-scope release fence), the
//      workgroup meets, then one lane per destination stores the flag e,
//   3. lane q of the first wave polls flag q of the OWN area until it reads e
//      (bounded by a wall-clock limit: on expiry the error word is set and the
//      kernel goes on, so the grid always drains),
//   4. charge1[ix] = slots[0][ix] + slots[1][ix] + ... in rank order: the same
//      additions in the same order on every GPU, so charge1 -- and with it E and the
//      marker trajectories -- are bit-identical on all ranks and from run to run,
//      which RCCL's choice of algorithm does not promise.
// Two parities suffice: a rank can start exchange e+2 (same parity as e) only after
// every peer has flagged e+1, which a peer does after it has finished reading e.
// ---------------------------------------------------------------------------
#define PIC1DP_SYS __HIP_MEMORY_SCOPE_SYSTEM

// n values per rank (n <= x.vstride), this rank's in sV -- every thread has filled the elements
// threadIdx.x + k * blockDim.x and only ever touches those -- summed over ranks in rank order, in place
__device__ __forceinline__ void exchange_vectors(const XchgArgs &x, double *sV, int n) {
  const int nr = x.nranks, par = static_cast<int>(x.epoch & 1);
  for (int k = 0; k < nr; ++k) {
    int q = x.rank + k;  // start with the own area, then the peers in ring order
    if (q >= nr) q -= nr;
    double *dst = x.slots[q] + (static_cast<size_t>(par) * nr + x.rank) * x.vstride;
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(dst + i, sV[i], __ATOMIC_RELAXED, PIC1DP_SYS);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this wave's stores have landed
  __syncthreads();
  if (threadIdx.x < nr) {
    const int q = threadIdx.x;
    __hip_atomic_store(x.flags[q] + par * XCHG_MAX_RANKS + x.rank, x.epoch, __ATOMIC_RELEASE, PIC1DP_SYS);
    const unsigned long long *fl = x.flags[x.rank] + par * XCHG_MAX_RANKS + q;
    const long long t0 = wall_clock64();
    // a run that already timed out once does not wait again: its remaining launches drain at once
    const long long limit = __hip_atomic_load(x.err, __ATOMIC_RELAXED, PIC1DP_SYS) ? 0 : x.timeout_ticks;
    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, PIC1DP_SYS) < x.epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > limit) {  // give up: report, never hang
        __hip_atomic_store(x.err, (x.epoch << 8) | static_cast<unsigned long long>(q + 1), __ATOMIC_RELAXED, PIC1DP_SYS);
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  const double *mine = x.slots[x.rank] + static_cast<size_t>(par) * nr * x.vstride;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double t[XCHG_MAX_RANKS];
#pragma unroll
    for (int q = 0; q < XCHG_MAX_RANKS; ++q)
      t[q] = q < nr ? __hip_atomic_load(mine + static_cast<size_t>(q) * x.vstride + i, __ATOMIC_RELAXED, PIC1DP_SYS) : 0.0;
    double sum = t[0];
#pragma unroll
    for (int q = 1; q < XCHG_MAX_RANKS; ++q)
      if (q < nr) sum = sum + t[q];
    sV[i] = sum;
  }
}

__device__ __forceinline__ void exchange_charge(const FieldArgs &f, const XchgArgs &x, double *sC) {
  const int nx = f.nx;
  // this rank's charge2: from the species accumulators, or already formed in f.charge (k_pred_combine)
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sC[ix] = x.local_in_charge ? f.charge[ix] : charge_local_one(f, ix);
  exchange_vectors(x, sC, nx);
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) f.charge[ix] = sC[ix];
}

// exchange only: charge1 into field charge (collect_charge call site, many-mode solve)
__global__ void __launch_bounds__(FIELD_THREADS) k_charge_exchange(const FieldArgs f, const XchgArgs x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exchange_charge(f, x, reinterpret_cast<double *>(smem));
}

// local charge -> exchange -> chargeden -> solve: one launch per sub-step
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve_xchg(const FieldArgs f, const XchgArgs x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  exchange_charge(f, x, sCD);
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) {  // own elements again
    const double cd = chargeden_from(f, sCD[ix]);
    f.chargeden[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
}

// One launch for both fields of the one-pass-per-step scheme (k_step_one): the field of the new
// state from its deposited charge (as k_field_solve / k_field_solve_xchg), then -- with the kept modes
// just found -- the predicted charge of the next first sub-step (as k_pred_combine), summed over
// ranks when XCHG, scaled, and solved into the half-step field of the NEXT step.
// SRC: 0 one rank (charges from the local accumulators), 1 one-hop exchange (two exchanges inside this
// launch), 2 packed (pa.pack holds the rank-summed charge2 and Z-weighted prediction slices, k_charge_pack +
// one all-reduce)
template <int SRC>
__global__ void __launch_bounds__(FIELD_THREADS)
k_field_solve_pair(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  const int nx = f.nx, nm = f.nmode, np1 = 1 + 2 * nm;
  // SRC 1: [charge2 | Z-weighted prediction slices] of this rank, then of all ranks, behind the solve's tiles
  double *sV = sTab + (f.tab_lds ? 2 * static_cast<size_t>(nm) * nx : 0);
  const double *pk = pa.pack;  // SRC 2: the same slices, all-reduced in memory
  if constexpr (SRC == 1) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      sV[ix] = charge_local_one(f, ix);
      for (int k = 0; k < np1; ++k) {
        double c2 = 0.0;
        for (int s = 0; s < f.nspecies; ++s) {
          double *r = pa.pred + (static_cast<size_t>(s) * np1 + k) * nx + ix;
          c2 = c2 + *r * f.Z[s];
          *r = 0.0;
        }
        sV[static_cast<size_t>(1 + k) * nx + ix] = c2;
      }
    }
    // element i of the packed vector belongs to thread i % blockDim; with nx a multiple of blockDim that is
    // the thread that wrote it -- otherwise meet first
    __syncthreads();
    exchange_vectors(x1, sV, (1 + np1) * nx);
    __syncthreads();
    pk = sV;
  }
  if constexpr (SRC != 0) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      const double c = pk[ix];
      f.charge[ix] = c;
      const double cd = chargeden_from(f, c);
      f.chargeden[ix] = cd;
      sCD[ix] = cd;
    }
  } else {
    solve_fill_chargeden<true, false>(f, sCD);
  }
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
  __syncthreads();  // E, mode_re/im (also in sMode) are final; sCD and sTab are free again
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    if constexpr (SRC != 0) {
      const double *r = pk + nx + ix;  // Z-weighted, summed over species and ranks
      c2 = r[0];
      for (int m = 0; m < nm; ++m) {
        c2 = c2 + sMode[m] * r[static_cast<size_t>(1 + m) * nx];
        c2 = c2 + sMode[nm + m] * r[static_cast<size_t>(1 + nm + m) * nx];
      }
    } else {
      for (int s = 0; s < f.nspecies; ++s) {
        double *r = pa.pred + static_cast<size_t>(s) * np1 * nx + ix;
        double c1 = r[0];
        r[0] = 0.0;
        for (int m = 0; m < nm; ++m) {
          double *ra = r + static_cast<size_t>(1 + m) * nx, *rb = r + static_cast<size_t>(1 + nm + m) * nx;
          c1 = c1 + sMode[m] * *ra;
          c1 = c1 + sMode[nm + m] * *rb;
          *ra = 0.0;
          *rb = 0.0;
        }
        c2 = c2 + c1 * f.Z[s];
      }
    }
    const double cd = chargeden_from(f, c2);
    pa.cd_h[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();
  FieldArgs g = f;
  g.E = pa.E_h;
  g.mode_re = pa.mode_h;
  g.mode_im = pa.mode_h + nm;
  g.history = nullptr;
  solve_body<true>(g, sCD, sMode, sScr, sTab);
}

// k_field_solve_pair for the usual case -- ONE kept mode, tables that fit the LDS -- with everything that does
// not wait for the serial sums moved in front of them.  The launch is latency-bound (one workgroup; at 1e7
// markers per GPU it is 7-10 % of the time step), and k_field_solve_pair spends it in a row of dependent
// round trips: charge, tables, [chain], prediction tiles, two workgroup reductions, inverse.  Here every
// thread issues all its loads at once (charge, the three prediction slices, both tables), forms the chain's
// products AND the prediction's forward sums before the chain runs -- the predicted charge density is
// cd_h = g0 + re ga + im gb with g0 = chargeden(R0), ga = RA nx/lx, gb = RB nx/lx, so its projections are
// S0 + re Sa + im Sb with six sums that need no mode: wave reductions, no barrier -- and after the chain one
// thread combines them; both inverse transforms then run in one loop.  The field of the new state: the same
// products in the same order as k_field_solve (bit for bit).  Eh: regrouped sums, as before (TREE).
template <int SRC>
__global__ void __launch_bounds__(FIELD_THREADS)
k_field_solve_pair1(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nx = f.nx;
  const int ne = (nx + 1) & ~1;
  double *sPc = reinterpret_cast<double *>(smem);  // [ne] fre * chargeden   (16-byte aligned rows for the chain)
  double *sPs = sPc + ne;                           // [ne] fim * chargeden
  double *sW = sPs + ne;                            // [FIELD_THREADS / 64][6] wave partials of the six sums
  double *sMode = sW + (FIELD_THREADS / 64) * 6;    // re, im, then the six sums of the workgroup
  double *sScr = sMode + 8;                         // [16]
  double *sV = sScr + 16;                           // SRC 1: [charge2 | R0 | RA | RB] of this rank, then of all
  const double *pk = pa.pack;                       // SRC 2: the same, all-reduced in memory
  const size_t np1 = 3;
  if constexpr (SRC == 1) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      sV[ix] = charge_local_one(f, ix);
      for (size_t k = 0; k < np1; ++k) {
        double c2 = 0.0;
        for (int sp = 0; sp < f.nspecies; ++sp) {
          double *r = pa.pred + (static_cast<size_t>(sp) * np1 + k) * nx + ix;
          c2 = c2 + *r * f.Z[sp];
          *r = 0.0;
        }
        sV[(1 + k) * nx + ix] = c2;
      }
    }
    __syncthreads();
    exchange_vectors(x1, sV, 4 * nx);
    __syncthreads();
    pk = sV;
  }
  double off = 0.0;
  if (!f.deltaf)
    for (int sp = 0; sp < f.nspecies; ++sp) off = off + f.Z[sp] * f.n0[sp];
  double s0c = 0.0, sac = 0.0, sbc = 0.0, s0s = 0.0, sas = 0.0, sbs = 0.0;
  constexpr int U = 4;
  const double ginv = f.grad_inv[0];  // off the critical path behind the chain
  double tr[U], ti[U];                // the tables of the last trip stay in registers for the inverse
  for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
    double c[U], r0[U], ra[U], rb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // all loads of the trip in flight together
      const int ix = base + u * FIELD_THREADS;
      c[u] = r0[u] = ra[u] = rb[u] = tr[u] = ti[u] = 0.0;
      if (ix < nx) {
        tr[u] = f.fre[ix];
        ti[u] = f.fim[ix];
        if constexpr (SRC != 0) {
          c[u] = pk[ix];
          r0[u] = pk[nx + ix];
          ra[u] = pk[2 * static_cast<size_t>(nx) + ix];
          rb[u] = pk[3 * static_cast<size_t>(nx) + ix];
        } else {
          for (int sp = 0; sp < f.nspecies; ++sp) {  // src/pic1dp_interaction.F90:126-127
            double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            double c1 = *r;
            for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
            c[u] = c[u] + c1 * f.Z[sp];
            const double *q = pa.pred + static_cast<size_t>(sp) * np1 * nx + ix;
            r0[u] = r0[u] + q[0] * f.Z[sp];
            ra[u] = ra[u] + q[nx] * f.Z[sp];
            rb[u] = rb[u] + q[2 * static_cast<size_t>(nx)] * f.Z[sp];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      if (ix < nx) {
        if constexpr (SRC == 0) {  // accumulators consumed: zero for the next kernels
          for (int sp = 0; sp < f.nspecies; ++sp) {
            double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            for (int g = 0; g < f.rho_copies; ++g) r[static_cast<size_t>(g) * f.rho_stride] = 0.0;
            double *q = pa.pred + static_cast<size_t>(sp) * np1 * nx + ix;
            q[0] = 0.0;
            q[nx] = 0.0;
            q[2 * static_cast<size_t>(nx)] = 0.0;
          }
        }
        f.charge[ix] = c[u];
        const double cd = chargeden_from(f, c[u]);  // :138-148
        f.chargeden[ix] = cd;
        sPc[ix] = tr[u] * cd;
        sPs[ix] = ti[u] * cd;
        const double g0 = r0[u] * f.dnx / f.lx - off, ga = ra[u] * f.dnx / f.lx, gb = rb[u] * f.dnx / f.lx;
        s0c += tr[u] * g0;
        sac += tr[u] * ga;
        sbc += tr[u] * gb;
        s0s += ti[u] * g0;
        sas += ti[u] * ga;
        sbs += ti[u] * gb;
      }
    }
  }
  {
    double v[6] = {s0c, sac, sbc, s0s, sas, sbs};
#pragma unroll
    for (int k = 0; k < 6; ++k)
      for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    if ((threadIdx.x & 63) == 0)
      for (int k = 0; k < 6; ++k) sW[(threadIdx.x >> 6) * 6 + k] = v[k];
  }
  __syncthreads();
  // the serial sums, ascending ix (:231-240): lane 0 the cos table -> im, lane 1 the -sin table -> re;
  // beside them the second wave adds up the wave partials of the six sums
  if (threadIdx.x < 2) {
    const bool use_cos = threadIdx.x == 0;
    const double acc = chain_sum_lds(use_cos ? sPc : sPs, nx);
    if (use_cos) {
      const double im = acc * f.sc_im * ginv;
      sMode[1] = im;
      f.mode_im[0] = im;
    } else {
      const double re = acc * f.sc_re * ginv;
      sMode[0] = re;
      f.mode_re[0] = re;
    }
  } else if (threadIdx.x >= 64 && threadIdx.x < 70) {
    const int k = threadIdx.x - 64;
    double t = 0.0;
    for (int w = 0; w < FIELD_THREADS / 64; ++w) t += sW[w * 6 + k];
    sMode[2 + k] = t;
  }
  __syncthreads();
  // the kept mode of the next step's half-step field (every thread for itself), both inverse transforms (:251-257)
  double e2 = 0.0;
  const double re = sMode[0], im = sMode[1];
  const double ac = sMode[2] + re * sMode[3] + im * sMode[4], as = sMode[5] + re * sMode[6] + im * sMode[7];
  const double im_h = ac * f.sc_im * ginv, re_h = as * f.sc_re * ginv;
  if (threadIdx.x == 0) {
    pa.mode_h[0] = re_h;
    pa.mode_h[1] = im_h;
  }
  const bool one_trip = nx <= U * FIELD_THREADS;
  for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
    if (!one_trip) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * FIELD_THREADS;
        tr[u] = ix < nx ? f.fre[ix] : 0.0;
        ti[u] = ix < nx ? f.fim[ix] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      if (ix < nx) {
        double a = 0.0;
        a = a + tr[u] * re;
        a = a + ti[u] * im;
        const double e = a * 2.0;
        f.E[ix] = e;
        e2 += e * e;
        double b = 0.0;
        b = b + tr[u] * re_h;
        b = b + ti[u] * im_h;
        pa.E_h[ix] = b * 2.0;
      }
    }
  }
  if (f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

// k_field_solve_pair for k_step_sums' prediction: the field of the new state from its deposited charge, then --
// with the kept mode just found -- the half-step field of the NEXT step from the six sums: no second forward
// transform, the sums ARE the projections (pred_forward_sums), only the inverse (:251-257).
// SRC: 0 one rank (charge from the local accumulators, sums from pa.pred), 1 one-hop exchange (charge2 and the
// six sums travel together, one exchange of nx + 8 doubles), 2 packed (pa.pack holds both, already all-reduced)
template <int SRC>
__global__ void __launch_bounds__(FIELD_THREADS)
k_field_solve_pair_sums(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  const int nx = f.nx;
  double *sV = sTab + (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * nx : 0);  // SRC 1: [charge2 | six sums | pad]
  __shared__ double sK[8];
  if constexpr (SRC == 1) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sV[ix] = charge_local_one(f, ix);
    if (threadIdx.x < 8) {
      sV[nx + threadIdx.x] = pa.pred[threadIdx.x];
      pa.pred[threadIdx.x] = 0.0;
    }
    __syncthreads();  // element i of the packed vector belongs to thread i % blockDim in the exchange
    exchange_vectors(x1, sV, nx + 8);
    __syncthreads();
    if (threadIdx.x < 8) sK[threadIdx.x] = sV[nx + threadIdx.x];
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      const double c = sV[ix];
      f.charge[ix] = c;
      const double cd = chargeden_from(f, c);
      f.chargeden[ix] = cd;
      sCD[ix] = cd;
    }
  } else if constexpr (SRC == 2) {
    if (threadIdx.x < 8) sK[threadIdx.x] = pa.pack[nx + threadIdx.x];
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      const double c = pa.pack[ix];
      f.charge[ix] = c;
      const double cd = chargeden_from(f, c);
      f.chargeden[ix] = cd;
      sCD[ix] = cd;
    }
  } else {
    if (threadIdx.x < 8) {
      sK[threadIdx.x] = pa.pred[threadIdx.x];
      pa.pred[threadIdx.x] = 0.0;
    }
    solve_fill_chargeden<true, false>(f, sCD);
  }
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
  __syncthreads();  // E, mode_re / mode_im (also in sMode: re, im) are final
  if (threadIdx.x == 0) {
    double ac, as;
    pred_forward_sums(f, pa.pt, sK, sMode[0], sMode[1], ac, as);
    const double im_h = ac * f.sc_im * f.grad_inv[0];   // :234, :243-247
    const double re_h = as * f.sc_re * f.grad_inv[0];   // :239
    sMode[0] = re_h;
    sMode[1] = im_h;
    pa.mode_h[0] = re_h;
    pa.mode_h[1] = im_h;
  }
  __syncthreads();
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {  // inverse, :251-257
    double sacc = 0.0;
    sacc = sacc + f.fre[ix] * sMode[0];
    sacc = sacc + f.fim[ix] * sMode[1];
    pa.E_h[ix] = sacc * 2.0;
  }
}

// k_field_solve_pair_sums trimmed like k_field_solve_pair1, and launched with as many threads as the grid has
// cells (up to 1024: at nx = 4096 every thread owns four cells and has all its loads in flight at once -- with 256
// threads each of the kernel's loops is four dependent round trips): charge -> products, [chain | the six sums
// fetched beside it], both inverse transforms in one loop.  41 -> 33 us at nx 4096, of which the chain is 23.
template <int SRC>
__global__ void __launch_bounds__(1024)
k_field_solve_pair_sums1(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nx = f.nx;
  const int ne = (nx + 1) & ~1;
  double *sPc = reinterpret_cast<double *>(smem);  // [ne] fre * chargeden
  double *sPs = sPc + ne;                           // [ne] fim * chargeden
  double *sMode = sPs + ne;                         // re, im, then the six sums
  double *sScr = sMode + 8;                         // [16]
  double *sV = sScr + 16;                           // SRC 1: [charge2 | six sums | pad]
  const double *pk = pa.pack;
  if constexpr (SRC == 1) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sV[ix] = charge_local_one(f, ix);
    if (threadIdx.x < 8) {
      sV[nx + threadIdx.x] = pa.pred[threadIdx.x];
      pa.pred[threadIdx.x] = 0.0;
    }
    __syncthreads();
    exchange_vectors(x1, sV, nx + 8);
    __syncthreads();
    pk = sV;
  }
  const double ginv = f.grad_inv[0];
  constexpr int U = 4;
  const int T = blockDim.x;
  double tr[U], ti[U];
  for (int base = threadIdx.x; base < nx; base += U * T) {
    double c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * T;
      c[u] = tr[u] = ti[u] = 0.0;
      if (ix < nx) {
        tr[u] = f.fre[ix];
        ti[u] = f.fim[ix];
        if constexpr (SRC != 0) {
          c[u] = pk[ix];
        } else {
          for (int sp = 0; sp < f.nspecies; ++sp) {  // src/pic1dp_interaction.F90:126-127
            const double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            double c1 = *r;
            for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
            c[u] = c[u] + c1 * f.Z[sp];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * T;
      if (ix < nx) {
        if constexpr (SRC == 0)
          for (int sp = 0; sp < f.nspecies; ++sp) {
            double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            for (int g = 0; g < f.rho_copies; ++g) r[static_cast<size_t>(g) * f.rho_stride] = 0.0;
          }
        f.charge[ix] = c[u];
        const double cd = chargeden_from(f, c[u]);  // :138-148
        f.chargeden[ix] = cd;
        sPc[ix] = tr[u] * cd;
        sPs[ix] = ti[u] * cd;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2) {  // the serial sums, ascending ix (:231-240)
    const bool use_cos = threadIdx.x == 0;
    const double acc = chain_sum_lds(use_cos ? sPc : sPs, nx);
    if (use_cos) {
      const double im = acc * f.sc_im * ginv;
      sMode[1] = im;
      f.mode_im[0] = im;
    } else {
      const double re = acc * f.sc_re * ginv;
      sMode[0] = re;
      f.mode_re[0] = re;
    }
  } else if (threadIdx.x >= 64 && threadIdx.x < 72) {  // beside them: the six sums (+ pad) of this step
    const int k = threadIdx.x - 64;
    if constexpr (SRC == 0) {
      if (k < 6) sMode[2 + k] = pa.pred[k];
      pa.pred[k] = 0.0;
    } else {
      if (k < 6) sMode[2 + k] = pk[nx + k];
    }
  }
  __syncthreads();
  const double re = sMode[0], im = sMode[1];
  double ac, as;
  pred_forward_sums(f, pa.pt, sMode + 2, re, im, ac, as);
  const double im_h = ac * f.sc_im * ginv, re_h = as * f.sc_re * ginv;  // :234, :239, :243-247
  if (threadIdx.x == 0) {
    pa.mode_h[0] = re_h;
    pa.mode_h[1] = im_h;
  }
  double e2 = 0.0;
  const bool one_trip = nx <= U * T;
  for (int base = threadIdx.x; base < nx; base += U * T) {
    if (!one_trip) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * T;
        tr[u] = ix < nx ? f.fre[ix] : 0.0;
        ti[u] = ix < nx ? f.fim[ix] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * T;
      if (ix < nx) {  // both inverse transforms, :251-257
        double a = 0.0;
        a = a + tr[u] * re;
        a = a + ti[u] * im;
        const double e = a * 2.0;
        f.E[ix] = e;
        e2 += e * e;
        double b = 0.0;
        b = b + tr[u] * re_h;
        b = b + ti[u] * im_h;
        pa.E_h[ix] = b * 2.0;
      }
    }
  }
  if (f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

// Many kept modes (2*nmode > FIELD_THREADS, up to the full spectrum nmode = nx/2,
// SURVEY N4): the same arithmetic in the same order, spread over workgroups.
// The reference's operators are then O(nx^2) dense matrices exactly as here
// (doc/formulation.tex:288-290); one thread still owns one serial sum.
constexpr int WIDE_THREADS = 64;

// forward sums: chain t -> mode t>>1, (t&1 ? cos-table : -sin-table), ascending ix
__global__ void __launch_bounds__(WIDE_THREADS) k_field_modes_wide(const FieldArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);  // [nx]
  const int nx = f.nx, nm = f.nmode;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sCD[ix] = f.chargeden[ix];
  __syncthreads();
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= 2 * nm) return;
  const int m = chain >> 1;
  const bool use_cos = chain & 1;
  const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
  double acc = 0.0;
  int ix = 0;
  for (; ix + 8 <= nx; ix += 8) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tab[ix + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = acc + t[k] * sCD[ix + k];
  }
  for (; ix < nx; ++ix) acc = acc + tab[ix] * sCD[ix];
  if (use_cos)
    f.mode_im[m] = acc * f.sc_im * f.grad_inv[m];
  else
    f.mode_re[m] = acc * f.sc_re * f.grad_inv[m];
}

// inverse: one grid point per thread, serial over ascending mode (:251-256)
__global__ void __launch_bounds__(WIDE_THREADS) k_field_inverse_wide(const FieldArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sMode = reinterpret_cast<double *>(smem);  // [2*nmode]: re then im
  const int nx = f.nx, nm = f.nmode;
  for (int m = threadIdx.x; m < nm; m += blockDim.x) {
    sMode[m] = f.mode_re[m];
    sMode[nm + m] = f.mode_im[m];
  }
  __syncthreads();
  const int ix = blockIdx.x * blockDim.x + threadIdx.x;
  if (ix >= nx) return;
  double s = 0.0;
  for (int m = 0; m < nm; ++m) s = s + f.fre[static_cast<size_t>(m) * nx + ix] * sMode[m];
  for (int m = 0; m < nm; ++m) s = s + f.fim[static_cast<size_t>(m) * nx + ix] * sMode[nm + m];
  f.E[ix] = s * 2.0;
}

__global__ void __launch_bounds__(FIELD_THREADS)
k_field_energy(const double *E, int nx, double lx, double dnx, double *out) {
  __shared__ double scr[16];
  double e2 = 0.0;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) e2 += E[ix] * E[ix];
  const double tot = block_sum(e2, scr);
  if (threadIdx.x == 0) {
    const double nrm = sqrt(tot);
    *out = nrm * nrm * lx / dnx;
  }
}

// ---------------------------------------------------------------------------
// Opt-in ALTERNATIVE field solve (NOT the reference's algorithm, which is the
// mode-filtered partial DFT above): second-order finite differences
// keeping every mode,
//     (phi[i-1] - 2 phi[i] + phi[i+1]) / h^2 = -(rho[i] - <rho>),
//     E[i] = -(phi[i+1] - phi[i-1]) / (2 h),      periodic, gauge phi[0] = 0.
// The nx-1 unknowns form a tridiagonal system, solved by parallel cyclic
// reduction held in LDS (ceil(log2(nx-1)) sweeps, every row eliminated against
// its neighbours at distance 1, 2, 4, ...).  One workgroup; nx <= 4096.
// ---------------------------------------------------------------------------
constexpr int FD_THREADS = 1024;
constexpr int FD_MAX_NX = 4096;
constexpr int FD_PER_THREAD = FD_MAX_NX / FD_THREADS;

__global__ void __launch_bounds__(FD_THREADS)
k_field_fd(const double *chargeden, double *E, double *history, int nx, double lx, double dnx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double scr[16];
  __shared__ double s_mean;
  double *A = reinterpret_cast<double *>(smem), *B = A + nx, *Cc = B + nx, *D = Cc + nx;
  const int n = nx - 1;
  const double h = lx / dnx;
  double part = 0.0;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) part += chargeden[ix];
  const double tot = block_sum(part, scr);
  if (threadIdx.x == 0) s_mean = tot / dnx;
  __syncthreads();
  const double mean = s_mean;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    A[j] = j == 0 ? 0.0 : -1.0;
    B[j] = 2.0;
    Cc[j] = j == n - 1 ? 0.0 : -1.0;
    D[j] = h * h * (chargeden[j + 1] - mean);
  }
  __syncthreads();
  for (int s = 1; s < n; s <<= 1) {
    double na[FD_PER_THREAD], nb[FD_PER_THREAD], nc[FD_PER_THREAD], nd[FD_PER_THREAD];
    int k = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x, ++k) {
      const int lo = j - s, hi = j + s;
      double a = A[j], b = B[j], c = Cc[j], d = D[j];
      double a2 = 0.0, c2 = 0.0;
      if (lo >= 0) {
        const double al = -a / B[lo];
        a2 = al * A[lo];
        b += al * Cc[lo];
        d += al * D[lo];
      }
      if (hi < n) {
        const double ga = -c / B[hi];
        c2 = ga * Cc[hi];
        b += ga * A[hi];
        d += ga * D[hi];
      }
      na[k] = a2;
      nb[k] = b;
      nc[k] = c2;
      nd[k] = d;
    }
    __syncthreads();
    k = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x, ++k) {
      A[j] = na[k];
      B[j] = nb[k];
      Cc[j] = nc[k];
      D[j] = nd[k];
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < n; j += blockDim.x) D[j] = D[j] / B[j];  // phi[j+1]
  __syncthreads();
  double e2 = 0.0;
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    const int ip = i + 1 == nx ? 0 : i + 1, im = i == 0 ? nx - 1 : i - 1;
    const double pp = ip == 0 ? 0.0 : D[ip - 1], pm = im == 0 ? 0.0 : D[im - 1];
    const double e = -(pp - pm) / (2.0 * h);
    E[i] = e;
    e2 += e * e;
  }
  if (history) {
    const double t2 = block_sum(e2, scr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(t2);
      *history = nrm * nrm * lx / dnx;
    }
  }
}

}  // namespace

hipError_t launch_field_fd(const double *chargeden, double *E, double *history, int nx, double lx,
                           double dnx, hipStream_t st) {
  if (nx < 3 || nx > FD_MAX_NX) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * 4 * static_cast<size_t>(nx);
  static bool big_lds_ok = false;
  if (lds > 64 * 1024 && !big_lds_ok) {
    // the kernel also holds 136 B of static LDS: leave room for it
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_fd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    big_lds_ok = true;
  }
  hipLaunchKernelGGL(k_field_fd, dim3(1), dim3(FD_THREADS), lds, st, chargeden, E, history, nx, lx, dnx);
  return hipGetLastError();
}

hipError_t launch_charge_local(const FieldArgs &f, hipStream_t st) {
  hipLaunchKernelGGL(k_charge_local, dim3(1), dim3(FIELD_THREADS), 0, st, f);
  return hipGetLastError();
}

hipError_t launch_chargeden(const FieldArgs &f, bool with_local, hipStream_t st) {
  if (with_local) {
    hipLaunchKernelGGL(k_chargeden<true>, dim3(1), dim3(FIELD_THREADS), 0, st, f);
  } else {
    hipLaunchKernelGGL(k_chargeden<false>, dim3(1), dim3(FIELD_THREADS), 0, st, f);
  }
  return hipGetLastError();
}

hipError_t launch_field_solve(const FieldArgs &f, bool with_local, bool from_chargeden,
                              hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) {  // many modes: chargeden, forward, inverse, energy
    if (!from_chargeden) {
      hipError_t e = launch_chargeden(f, with_local, st);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_field_modes_wide, dim3((2 * f.nmode + WIDE_THREADS - 1) / WIDE_THREADS),
                       dim3(WIDE_THREADS), sizeof(double) * f.nx, st, f);
    hipLaunchKernelGGL(k_field_inverse_wide, dim3((f.nx + WIDE_THREADS - 1) / WIDE_THREADS), dim3(WIDE_THREADS),
                       sizeof(double) * 2 * f.nmode, st, f);
    if (f.history) hipLaunchKernelGGL(k_field_energy, dim3(1), dim3(FIELD_THREADS), 0, st, f.E, f.nx, f.lx, f.dnx, f.history);
    return hipGetLastError();
  }
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  if (from_chargeden) {
    hipLaunchKernelGGL((k_field_solve<false, true>), dim3(1), dim3(FIELD_THREADS), lds, st, f);
  } else if (with_local) {
    hipLaunchKernelGGL((k_field_solve<true, false>), dim3(1), dim3(FIELD_THREADS), lds, st, f);
  } else {
    hipLaunchKernelGGL((k_field_solve<false, false>), dim3(1), dim3(FIELD_THREADS), lds, st, f);
  }
  return hipGetLastError();
}

hipError_t launch_pred_combine(const FieldArgs &f, double *pred, int nm_pred, hipStream_t st) {
  hipLaunchKernelGGL(k_pred_combine, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred, nm_pred);
  return hipGetLastError();
}

hipError_t launch_charge_exchange(const FieldArgs &f, const XchgArgs &x, hipStream_t st) {
  hipLaunchKernelGGL(k_charge_exchange, dim3(1), dim3(FIELD_THREADS), sizeof(double) * f.nx, st, f, x);
  return hipGetLastError();
}

hipError_t launch_field_solve_xchg(const FieldArgs &f, const XchgArgs &x, hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) {  // many modes: exchange, then the wide kernels from the summed charge
    hipError_t e = launch_charge_exchange(f, x, st);
    if (e != hipSuccess) return e;
    return launch_field_solve(f, false, false, st);
  }
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  hipLaunchKernelGGL(k_field_solve_xchg, dim3(1), dim3(FIELD_THREADS), lds, st, f, x);
  return hipGetLastError();
}

hipError_t launch_field_solve_pair(const FieldArgs &f, const PairArgs &pa, const XchgArgs *x1, hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) return hipErrorInvalidValue;
  size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                 (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  const XchgArgs none{};
  if (pa.kind != 2 && f.nmode == 1 && f.tab_lds && !pa.plain) {  // the lean kernel of the usual case
    const size_t ne = (static_cast<size_t>(f.nx) + 1) & ~static_cast<size_t>(1);
    size_t l1 = sizeof(double) * (2 * ne + (FIELD_THREADS / 64) * 6 + 8 + 16);
    if (x1) {
      l1 += sizeof(double) * 4 * static_cast<size_t>(f.nx);
      if (l1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair1<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair1<1>, dim3(1), dim3(FIELD_THREADS), l1, st, f, *x1, pa);
    } else if (pa.pack) {
      hipLaunchKernelGGL(k_field_solve_pair1<2>, dim3(1), dim3(FIELD_THREADS), l1, st, f, none, pa);
    } else {
      hipLaunchKernelGGL(k_field_solve_pair1<0>, dim3(1), dim3(FIELD_THREADS), l1, st, f, none, pa);
    }
    return hipGetLastError();
  }
  if (pa.kind == 2 && f.nmode == 1 && !pa.plain) {  // the lean kernel
    const size_t ne = (static_cast<size_t>(f.nx) + 1) & ~static_cast<size_t>(1);
    size_t l1 = sizeof(double) * (2 * ne + 8 + 16);
    const int threads = f.nx > 2048 ? 1024 : (f.nx > 1024 ? 512 : FIELD_THREADS);
    if (x1) {
      l1 += sizeof(double) * pack_doubles(f.nx, 1, 2);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums1<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_field_solve_pair_sums1<1>, dim3(1), dim3(threads), l1, st, f, *x1, pa);
    } else if (pa.pack) {
      if (l1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums1<2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair_sums1<2>, dim3(1), dim3(threads), l1, st, f, none, pa);
    } else {
      if (l1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums1<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair_sums1<0>, dim3(1), dim3(threads), l1, st, f, none, pa);
    }
    return hipGetLastError();
  }
  if (pa.kind == 2) {
    if (f.nmode != 1) return hipErrorInvalidValue;
    if (x1) {
      lds += sizeof(double) * pack_doubles(f.nx, 1, 2);
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair_sums<1>, dim3(1), dim3(FIELD_THREADS), lds, st, f, *x1, pa);
    } else if (pa.pack) {
      hipLaunchKernelGGL(k_field_solve_pair_sums<2>, dim3(1), dim3(FIELD_THREADS), lds, st, f, none, pa);
    } else {
      hipLaunchKernelGGL(k_field_solve_pair_sums<0>, dim3(1), dim3(FIELD_THREADS), lds, st, f, none, pa);
    }
    return hipGetLastError();
  }
  if (x1) {
    lds += sizeof(double) * (2 + 2 * static_cast<size_t>(f.nmode)) * f.nx;  // the packed vector
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_field_solve_pair<1>, dim3(1), dim3(FIELD_THREADS), lds, st, f, *x1, pa);
  } else if (pa.pack) {
    hipLaunchKernelGGL(k_field_solve_pair<2>, dim3(1), dim3(FIELD_THREADS), lds, st, f, none, pa);
  } else {
    hipLaunchKernelGGL(k_field_solve_pair<0>, dim3(1), dim3(FIELD_THREADS), lds, st, f, none, pa);
  }
  return hipGetLastError();
}

hipError_t launch_field_solve_pred(const FieldArgs &f, double *pred, int nm_pred, hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  hipLaunchKernelGGL(k_field_solve_pred, dim3(1), dim3(FIELD_THREADS), lds, st, f, pred, nm_pred);
  return hipGetLastError();
}

hipError_t launch_charge_pack(const FieldArgs &f, double *pred, int nm_pred, int kind, double *pack, hipStream_t st) {
  if (kind == 2)
    hipLaunchKernelGGL(k_charge_pack_sums, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred, pack);
  else
    hipLaunchKernelGGL(k_charge_pack, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred, nm_pred, pack);
  return hipGetLastError();
}

hipError_t launch_pred_chargeden(const FieldArgs &f, const PredTab &pt, double *pred, const double *K, hipStream_t st) {
  if (f.nmode != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pred_chargeden, dim3(1), dim3(FIELD_THREADS), 0, st, f, pt, pred, K);
  return hipGetLastError();
}

hipError_t launch_pred_to_charge(const FieldArgs &f, double *pred, hipStream_t st) {
  hipLaunchKernelGGL(k_pred_to_charge, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred);
  return hipGetLastError();
}

hipError_t launch_field_energy(const double *E, int nx, double lx, double dnx, double *out,
                               hipStream_t st) {
  hipLaunchKernelGGL(k_field_energy, dim3(1), dim3(FIELD_THREADS), 0, st, E, nx, lx, dnx, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// diagnostics
// ---------------------------------------------------------------------------
namespace {

// sum v^2, v^2 p, v^2 w (src/pic1dp_output.F90:126-151): per-workgroup partials,
// the host adds them in workgroup order
__global__ void __launch_bounds__(256)
k_energy_sums(const double *v, const double *p, const double *w, int64_t i0, int64_t n, double *partial) {
  __shared__ double scr[16];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride) {
    const int64_t i = tidx(i0 + k);
    const double v2 = v[i] * v[i];
    s0 += v2;
    s1 += v2 * p[i];
    if (w) s2 += v2 * w[i];
  }
  const double t0 = block_sum(s0, scr);
  const double t1 = block_sum(s1, scr);
  const double t2 = block_sum(s2, scr);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = t0;
    partial[blockIdx.x * 3 + 1] = t1;
    partial[blockIdx.x * 3 + 2] = t2;
  }
}

__global__ void __launch_bounds__(256)
k_cell_indices(const double *x, int64_t np, const GridConst g, int32_t *ixo,
               unsigned long long *count) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np; i += stride) {
    int ix;
    double wl;
    locate(x[tidx(i)], g, ix, wl);
    if (ixo) ixo[i] = ix;
    if (count) atomicAdd(&count[ix], 1ULL);
  }
}

}  // namespace

namespace {

// One pass over a species for everything output_all needs from the markers:
// * the (x,v) and v histograms of output_ptcldist, src/pic1dp_output.F90:239-315:
//   4-point bilinear weights on an nx_opd x nv_opd grid, markers with
//   |v| >= v_max skipped (:241);
// * the kinetic sums of output_field, sum v^2, v^2 p, v^2 w over ALL markers
//   (:126-151), as per-workgroup partials the host adds in workgroup order.
// LDS = true keeps a private copy of the histograms per workgroup
// (3*(nxo*nvo)+3*nvo doubles) and flushes it with global atomics.  There the v
// histograms are not accumulated marker by marker (64 hot bins: the LDS atomics
// of a wave collide) but formed once per workgroup as the row sums of its (x,v)
// histograms -- the same numbers in exact arithmetic, (sx + (1-sx))*sv = sv, and
// within rounding (<= 1e-15 relative per term) of the separate accumulation.
// LDS = false (grids too large for 160 KiB) adds everything straight to memory.
// The per-marker part and the finish are shared with the DIAG variant of k_step_full.
template <bool LDS, bool DELTAF>
__global__ void __launch_bounds__(1024)
k_ptcldist(const double *x, const double *v, const double *p, const double *w, int64_t np, const DistGeom dg,
           double *out, double *partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntot = 3 * dg.nxo * dg.nvo + 3 * dg.nvo;
  DistBins b{LDS ? reinterpret_cast<double *>(smem) : out, dg.nxo * dg.nvo, dg.nvo};
  double *scr = reinterpret_cast<double *>(smem) + (LDS ? ntot : 0);  // [16]
  if constexpr (LDS) {
    for (int i = threadIdx.x; i < ntot; i += blockDim.x) b.h[i] = 0.0;
    __syncthreads();
  }
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  DistSums sm;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < np; k += stride) {
    const int64_t i = tidx(k);
    ptcldist_one<LDS, DELTAF>(x[i], v[i], p[i], DELTAF ? w[i] : 0.0, dg, b, sm);
  }
  ptcldist_finish<LDS, DELTAF>(dg, b, sm, scr, out, partial);
}

}  // namespace

int ptcldist_blocks(int64_t np, int nxo, int nvo, int num_cu) {
  const size_t bytes = sizeof(double) * (3 * static_cast<size_t>(nxo) * nvo + 3 * static_cast<size_t>(nvo));
  const bool lds = bytes <= 150 * 1024;
  int64_t blocks = lds ? num_cu : static_cast<int64_t>(num_cu) * 2;
  const int64_t need = (np + 1023) / 1024;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  return static_cast<int>(blocks);
}

hipError_t launch_ptcldist(const double *x, const double *v, const double *p, const double *w,
                           int64_t np, double lx, double vmax, int nxo, int nvo, bool deltaf,
                           double *out, double *partial, int num_cu, hipStream_t st) {
  const size_t hist = sizeof(double) * (3 * static_cast<size_t>(nxo) * nvo + 3 * static_cast<size_t>(nvo));
  const bool lds = hist <= 150 * 1024;
  const size_t bytes = (lds ? hist : 0) + 16 * sizeof(double);  // + block_sum scratch
  const int threads = 1024;
  const int blocks = ptcldist_blocks(np, nxo, nvo, num_cu);
  const DistGeom dg{lx, vmax, nxo, nvo};
  auto go = [&](auto kern) -> hipError_t {
    if (lds && bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(threads), bytes, st, x, v, p, w, np, dg, out,
                       partial);
    return hipGetLastError();
  };
  if (lds) return deltaf ? go(k_ptcldist<true, true>) : go(k_ptcldist<true, false>);
  return deltaf ? go(k_ptcldist<false, true>) : go(k_ptcldist<false, false>);
}

namespace {

// test positions for div_lx: uniform over three periods, cell boundaries and
// their neighbours (where a wrong last bit would change the cell index), raw
// bit patterns over a wide exponent range, small and large magnitudes
__host__ __device__ inline double div_check_value(uint64_t seed, int64_t i, double lx, int nx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * static_cast<uint64_t>(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  const double u = static_cast<double>(z >> 11) * 0x1p-53;  // [0,1)
  const int kind = static_cast<int>(z & 7);
  if (kind <= 2) return lx * (u * 3.0 - 1.0);
  if (kind == 3 || kind == 4) {
    const int k = static_cast<int>((z >> 3) % static_cast<uint64_t>(nx + 1));
    double x = lx * static_cast<double>(k) / static_cast<double>(nx);
    const int steps = static_cast<int>((z >> 40) & 7) - 3;  // -3..4 ulps around the boundary
    union { double d; int64_t b; } c;
    c.d = x;
    if (x != 0.0) c.b += steps;
    return c.d;
  }
  if (kind == 5) {
    union { double d; uint64_t b; } c;
    const uint64_t e = 1023 - 400 + (z >> 12) % 800;
    c.b = (z & 0x800FFFFFFFFFFFFFULL) | (e << 52);
    return c.d;
  }
  if (kind == 6) return lx * u * 0x1p-30;
  return lx * (u - 0.5) * 1e6;
}

__global__ void k_div_check(GridConst g, uint64_t seed, int64_t n, unsigned long long *bad) {
  GridConst gf = g;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = div_check_value(seed, i, g.lx, g.nx);
    const double a = div_lx(x, gf), b = x / g.lx;
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(bad, 1ULL);
  }
}

}  // namespace

namespace {

// bandwidth probe with the access pattern of the particle kernels: NR input
// streams and NW output streams of doubles, 16 B per lane, grid-stride
struct ProbeArgs {
  const double2 *in[8];
  double2 *out[4];
  int64_t npair;
};

template <int NR, int NW, int VARIANT>
__global__ void __launch_bounds__(1024) k_stream_probe(const ProbeArgs a) {
  // VARIANT 0: plain loads/stores; 1: non-temporal; 2: plain, two pairs per lane per trip
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 acc = make_double2(0.0, 0.0);
  constexpr int U = VARIANT == 2 ? 2 : 1;
  for (int64_t j0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j0 < a.npair; j0 += U * stride) {
    double2 s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = make_double2(0.0, 0.0);
      const int64_t j = j0 + u * stride;
      if (j < a.npair) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          double2 t;
          if constexpr (VARIANT == 1) {
            t.x = __builtin_nontemporal_load(&a.in[k][j].x);
            t.y = __builtin_nontemporal_load(&a.in[k][j].y);
          } else {
            t = a.in[k][j];
          }
          s[u].x += t.x;
          s[u].y += t.y;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = j0 + u * stride;
      if (j < a.npair) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          if constexpr (VARIANT == 1) {
            __builtin_nontemporal_store(s[u].x + k, &a.out[k][j].x);
            __builtin_nontemporal_store(s[u].y - k, &a.out[k][j].y);
          } else {
            a.out[k][j] = make_double2(s[u].x + k, s[u].y - k);
          }
        }
      }
      if constexpr (NW == 0) {
        acc.x += s[u].x;
        acc.y += s[u].y;
      }
    }
  }
  if constexpr (NW == 0) {
    if (acc.x == 1.2345e300 && acc.y == -1.2345e300) a.out[0][0] = acc;  // keeps the loads alive
  }
}

template <int NR, int NW>
hipError_t launch_probe_v(const ProbeArgs &a, int variant, int blocks, int threads, hipStream_t st) {
  switch (variant) {
    case 1: hipLaunchKernelGGL((k_stream_probe<NR, NW, 1>), dim3(blocks), dim3(threads), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_stream_probe<NR, NW, 2>), dim3(blocks), dim3(threads), 0, st, a); break;
    default: hipLaunchKernelGGL((k_stream_probe<NR, NW, 0>), dim3(blocks), dim3(threads), 0, st, a); break;
  }
  return hipGetLastError();
}

template <int NR>
hipError_t launch_probe_nr(const ProbeArgs &a, int nw, int variant, int blocks, int threads, hipStream_t st) {
  switch (nw) {
    case 0: return launch_probe_v<NR, 0>(a, variant, blocks, threads, st);
    case 1: return launch_probe_v<NR, 1>(a, variant, blocks, threads, st);
    case 3: return launch_probe_v<NR, 3>(a, variant, blocks, threads, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

namespace {

// Layout probe (tuning only): the traffic of k_step_full -- four arrays read, three
// of them written back in place, 16 B per lane, non-temporal -- over ONE slab, with
// the four arrays either apart by `step` double2 (SoA, TILED = false) or interleaved
// in tiles of 2^lt2 pairs: [x tile | v tile | w tile | p tile] (TILED = true).
template <bool TILED, bool WRITE, bool WG_PER_TILE>
__global__ void __launch_bounds__(1024) k_layout_probe(double2 *base, int64_t step, int lt2, int64_t npair) {
  const int64_t mask = (static_cast<int64_t>(1) << lt2) - 1;
  double2 acc = make_double2(0.0, 0.0);
  auto body = [&](int64_t j) {
    const int64_t o = TILED ? (((j >> lt2) << (lt2 + 2)) + (j & mask)) : j;
    const int64_t d = TILED ? (static_cast<int64_t>(1) << lt2) : step;
    const double2 a = ld2t<true>(base + o), b = ld2t<true>(base + o + d), c = ld2t<true>(base + o + 2 * d),
                  e = ld2t<true>(base + o + 3 * d);
    const double sx = a.x + b.x + c.x + e.x, sy = a.y + b.y + c.y + e.y;
    if constexpr (WRITE) {
      st2t<true>(base + o, sx * 0.25, sy * 0.25);
      st2t<true>(base + o + d, sx * 0.125, sy * 0.125);
      st2t<true>(base + o + 2 * d, sx * 0.0625, sy * 0.0625);
    } else {
      acc.x += sx;
      acc.y += sy;
    }
  };
  if constexpr (WG_PER_TILE) {  // a workgroup walks whole tiles: [x|v|w|p] of one tile, then its next tile
    const int64_t tp = static_cast<int64_t>(1) << lt2, ntile = (npair + tp - 1) >> lt2;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x)
      for (int64_t l = threadIdx.x; l < tp; l += blockDim.x)
        if ((t << lt2) + l < npair) body((t << lt2) + l);
  } else {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += stride) body(j);
  }
  if constexpr (!WRITE) {
    if (acc.x == 1.2345e300 && acc.y == -1.2345e300) base[0] = acc;  // keeps the loads alive
  }
}

}  // namespace

// variant: bit 0 tiled, bit 1 read-only (k_step_half's shape), bit 2 one workgroup per tile
hipError_t launch_layout_probe(double *base, int64_t step_doubles, int log2_tile, int64_t n, int variant, int blocks,
                               int threads, hipStream_t st) {
  double2 *b2 = reinterpret_cast<double2 *>(base);
  const int64_t s2 = step_doubles >> 1, np = n >> 1;
  const int lt2 = log2_tile - 1;
#define PIC1DP_LP(T, W, G) hipLaunchKernelGGL((k_layout_probe<T, W, G>), dim3(blocks), dim3(threads), 0, st, b2, s2, lt2, np)
  switch (variant & 7) {
    case 0: PIC1DP_LP(false, true, false); break;
    case 1: PIC1DP_LP(true, true, false); break;
    case 2: PIC1DP_LP(false, false, false); break;
    case 3: PIC1DP_LP(true, false, false); break;
    case 5: PIC1DP_LP(true, true, true); break;
    case 7: PIC1DP_LP(true, false, true); break;
    default: return hipErrorInvalidValue;
  }
#undef PIC1DP_LP
  return hipGetLastError();
}

hipError_t launch_stream_probe(double *const *in, int nr, double *const *out, int nw, int64_t n,
                               int blocks, int threads, int variant, hipStream_t st) {
  ProbeArgs a{};
  for (int k = 0; k < nr && k < 8; ++k) a.in[k] = reinterpret_cast<const double2 *>(in[k]);
  for (int k = 0; k < 4; ++k) a.out[k] = reinterpret_cast<double2 *>(out[k < nw ? k : 0]);
  a.npair = n >> 1;
  switch (nr) {
    case 1: return launch_probe_nr<1>(a, nw, variant, blocks, threads, st);
    case 4: return launch_probe_nr<4>(a, nw, variant, blocks, threads, st);
    case 7: return launch_probe_nr<7>(a, nw, variant, blocks, threads, st);
    default: return hipErrorInvalidValue;
  }
}

namespace {

// dividends for the div_const check: random sign, exponent in [-300, 300],
// random significand -- every 16th one from the edges (0...0k, 1...1k) where
// rounding decisions are closest
__host__ __device__ inline double divc_check_value(uint64_t seed, int64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(i + 1);  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  uint64_t mant = z & 0xFFFFFFFFFFFFFull;
  if ((i & 15) == 0) mant = (z & 0x3FF) | ((z >> 10) & 1 ? 0xFFFFFFFFFFC00ull : 0ull);
  const uint64_t expo = 1023 - 300 + (z >> 52) % 601;
  const uint64_t bits = (z & 0x8000000000000000ull) | (expo << 52) | mant;
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double(static_cast<long long>(bits));
#else
  double d;
  std::memcpy(&d, &bits, 8);
  return d;
#endif
}

__global__ void k_divc_check(double c, double rc, uint64_t seed, int64_t n, unsigned long long *bad) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double a = divc_check_value(seed, i);
    const double q = div_const(a, c, rc, 1), b = a / c;
    if (__double_as_longlong(q) != __double_as_longlong(b)) atomicAdd(bad, 1ULL);
  }
}

}  // namespace

namespace {
// the push's transcendental on its own (tests bound it against libm)
__global__ void __launch_bounds__(256) k_exp_array(const double *x, double *y, int64_t n) {
  exp_table_init();
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = pexp(x[i]);
}
}  // namespace

hipError_t launch_exp_array(const double *x, double *y, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_exp_array, dim3(1024), dim3(256), 0, st, x, y, n);
  return hipGetLastError();
}

hipError_t launch_divc_check(double c, uint64_t seed, int64_t n, unsigned long long *bad, hipStream_t st) {
  hipLaunchKernelGGL(k_divc_check, dim3(2048), dim3(256), 0, st, c, 1.0 / c, seed, n, bad);
  return hipGetLastError();
}

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

hipError_t launch_div_check(const GridConst &g, uint64_t seed, int64_t n, unsigned long long *bad,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_div_check, dim3(2048), dim3(256), 0, st, g, seed, n, bad);
  return hipGetLastError();
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

hipError_t launch_energy_sums(const double *v, const double *p, const double *w, int64_t i0, int64_t n,
                              double *partial, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(k_energy_sums, dim3(blocks), dim3(256), 0, st, v, p, w, i0, n, partial);
  return hipGetLastError();
}

namespace {

// host arrays are contiguous, marker arrays tiled: the two meet in these kernels
__global__ void __launch_bounds__(256) k_tile_scatter(double *arr, int64_t i0, const double *src, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    arr[tidx(i0 + k)] = src[k];
}
__global__ void __launch_bounds__(256) k_tile_gather(const double *arr, int64_t i0, double *dst, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    dst[k] = arr[tidx(i0 + k)];
}
__global__ void __launch_bounds__(256) k_tile_copy(double *dst, const double *src, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    dst[tidx(k)] = src[tidx(k)];
}

int copy_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return static_cast<int>(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

hipError_t launch_tile_scatter(double *arr, int64_t i0, const double *src, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_scatter, dim3(copy_blocks(n)), dim3(256), 0, st, arr, i0, src, n);
  return hipGetLastError();
}
hipError_t launch_tile_gather(const double *arr, int64_t i0, double *dst, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_gather, dim3(copy_blocks(n)), dim3(256), 0, st, arr, i0, dst, n);
  return hipGetLastError();
}
hipError_t launch_tile_copy(double *dst, const double *src, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_copy, dim3(copy_blocks(n)), dim3(256), 0, st, dst, src, n);
  return hipGetLastError();
}

hipError_t launch_cell_indices(const double *x, int64_t np, const GridConst &g, int32_t *ix,
                               unsigned long long *count, hipStream_t st) {
  int blocks = static_cast<int>((np + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_cell_indices, dim3(blocks), dim3(256), 0, st, x, np, g, ix, count);
  return hipGetLastError();
}

}  // namespace pic1dp
