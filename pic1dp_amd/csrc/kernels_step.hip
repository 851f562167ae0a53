// kernels_step.hip -- the whole-time-step marker kernels (pic1dp_hip_step and the lazy call sites): k_step_half,
// k_step_full, k_step_one, k_step_sums.  gfx950 (CDNA4, wave64), compiled with -ffp-contract=off: every product /
// sum the reference rounds separately is rounded separately here, so positions, velocities and cell indices are
// bit-identical to the CPU arithmetic; only exp() and the order of the charge sums differ.
// All kernels are HBM-bound streaming passes over the tiled FP64 marker slabs (16 B per lane, coalesced,
// grid-stride); no dense contraction, so no MFMA.  Grid tiles (E0, Eh, rho, the prediction's tables and
// accumulators) live in LDS; deposits are ds_add_f64, flushed with one global atomic per cell and workgroup.
// DESIGN.md section 2 has the numbers.
//
// One translation unit per distribution: the build compiles this file once for every value of
// -DPIC1DP_STEP_DIST (0 Maxwellian, 1 two-stream1, 2 two-stream2, 3 bump-on-tail, 4 / 5 the one-exp forms of 2 / 3),
// in parallel; step_dispatch.cpp picks the instance.
#include "device_diag.hpp"
#include "device_field.hpp"
#include "device_math.hpp"
#include "device_xchg.hpp"
#include "step_args.hpp"

#include <cstdio>
#include <vector>

#ifndef PIC1DP_STEP_DIST
#error "compile with -DPIC1DP_STEP_DIST=0..5 (pic1dp_amd/build.py does)"
#endif

namespace pic1dp {

namespace {

// Tuning build -DPIC1DP_TUNE_STAMPS (tools/stamp_probe.sh; never the product): thread 0 of every workgroup of
// k_step_one / k_step_sums records the 100 MHz wall clock at the phase boundaries -- 0 entry, 1 tiles staged, 2 its
// own loop done, 3 the workgroup's loop done, 4 rho flushed, 5 end -- and 6 its hardware id; the launch numbered
// PIC1DP_STAMP_AT is written to PIC1DP_STAMP_FILE by the launch after it.
#ifdef PIC1DP_TUNE_STAMPS
#define STAMP(a, k)                                                                          \
  do {                                                                                       \
    if (threadIdx.x == 0) (a).stamps[static_cast<size_t>(blockIdx.x) * 8 + (k)] = wall_clock64(); \
  } while (0)
#define STAMP_HWID(a)                                                                                             \
  do {                                                                                                            \
    if (threadIdx.x == 0)                                                                                         \
      (a).stamps[static_cast<size_t>(blockIdx.x) * 8 + 6] =                                                       \
          (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11))) << 32) | \
          __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                                            \
  } while (0)
#else
#define STAMP(a, k) ((void)0)
#define STAMP_HWID(a) ((void)0)
#endif
// -DPIC1DP_TUNE_STAMPS -DPIC1DP_TUNE_STAMPS_SOLVE: stamps 2, 3, 4 are taken INSIDE the fused prologue's solve instead
// (2 charge / products staged, 3 the serial forward sums done, 4 the prediction's sums combined; 1 stays "tiles staged")
#if defined(PIC1DP_TUNE_STAMPS) && defined(PIC1DP_TUNE_STAMPS_SOLVE)
#define STAMP_SOLVE(p, k)                                                                   \
  do {                                                                                      \
    if (threadIdx.x == 0 && (p)) (p)[static_cast<size_t>(blockIdx.x) * 8 + (k)] = wall_clock64(); \
  } while (0)
#define STAMP_LOOP(a, k) ((void)0)
#else
#define STAMP_SOLVE(p, k) ((void)0)
#define STAMP_LOOP(a, k) STAMP(a, k)
#endif

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
  unsigned *sDraw = reinterpret_cast<unsigned *>(sR0 + ((nx + 2) & ~1));  // the drawn chunks' counter
  for (int i = threadIdx.x; i < nx; i += blockDim.x) sE[i] = a.E0[i];
  zero_rho(sR0, a.g);
  if (threadIdx.x == 0) {
    sE[nx] = a.E0[0];
    *sDraw = 0u;
  }
  __syncthreads();
  double *sR = sR0;
  constexpr bool HAS_W = (MODE != MODE_FULLF);
  const int64_t npair = a.np >> 1;
  const double2 *x2 = reinterpret_cast<const double2 *>(a.x);
  const double2 *v2 = reinterpret_cast<const double2 *>(a.v);
  const double2 *w2 = reinterpret_cast<const double2 *>(a.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
  const PairRows rows = pair_rows(npair, a.dyn_tail);
  int64_t j = rows.first + threadIdx.x;
  for (int k = 0;; ++k, j += rows.stride) {
    if (k >= rows.dealt && !draw_chunk(rows, sDraw, j)) break;
    if (j >= npair) continue;
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
// FX (DIAG): the LDS copy of the histograms as 64-bit fixed-point sums (device_diag.hpp DistScale), as in k_ptcldist
template <int DIST, int MODE, int POW2, bool NT, bool CARRY, bool DIAG, bool FX = false>
__global__ void __launch_bounds__(1024) k_step_full(const StepArgsDev a) {
  static_assert(!FX || DIAG, "fixed-point sums are the diagnostics'");
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
  // behind the rho copies (16-byte aligned): the drawn chunks' counter; DIAG: the histograms, then the block_sum scratch
  unsigned *sDraw = reinterpret_cast<unsigned *>(sR0 + ((nx + 2) & ~1));
  double *sH = sR0 + ((nx + 2) & ~1) + 2;
  const int ntot = DIAG ? 3 * a.dg.nxo * a.dg.nvo + 3 * a.dg.nvo : 0;
  const DistBins bins{sH, a.dg.nxo * a.dg.nvo, a.dg.nvo};
  DistSums sums;
  if constexpr (DIAG)
    for (int i = threadIdx.x; i < ntot; i += blockDim.x) sH[i] = 0.0;
  if (threadIdx.x == 0) *sDraw = 0u;
  __syncthreads();
  double *sR = sR0;
  const int64_t npair = a.np >> 1;
  double2 *x2 = reinterpret_cast<double2 *>(a.x);
  double2 *v2 = reinterpret_cast<double2 *>(a.v);
  double2 *w2 = reinterpret_cast<double2 *>(a.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
  const PairRows rows = pair_rows(npair, a.dyn_tail);
  int64_t j = rows.first + threadIdx.x;
  for (int k = 0;; ++k, j += rows.stride) {
    if (k >= rows.dealt && !draw_chunk(rows, sDraw, j)) break;
    if (j >= npair) continue;
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
      ptcldist_one<true, HAS_W, FX>(n0.x, n0.v, P.x, n0.w, a.dg, bins, sums, &a.dscale);
      ptcldist_one<true, HAS_W, FX>(n1.x, n1.v, P.y, n1.w, a.dg, bins, sums, &a.dscale);
    }
  }
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double w = HAS_W ? a.w[i] : 0.0;
    const One n = step_full_one<DIST, MODE, POW2, CARRY>(a.x[i], a.v[i], w, a.p[i], sE0, sEh, sR, a,
                                                         CARRY ? a.t2[a.np - 1] : 0.0);
    a.x[i] = n.x;
    if constexpr (PUSH_V) a.v[i] = n.v;
    if constexpr (HAS_W) a.w[i] = n.w;
    if constexpr (DIAG) ptcldist_one<true, HAS_W, FX>(n.x, n.v, a.p[i], n.w, a.dg, bins, sums, &a.dscale);
  }
  __syncthreads();
  flush_rho(sR0, a.rho, a.g);
  if constexpr (DIAG) ptcldist_finish<true, HAS_W, FX, 6>(a.dg, bins, sums, sH + ntot, a.dist_out, a.dist_partial, &a.dscale);
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
// (DESIGN.md 6), so instructions are what this part is written for:
// * the tables lie cell by cell, sAB[cell][A_0 B_0 (A_1 B_1)] with a guard cell, and the accumulators likewise,
//   sP[cell][R0 RA_0 RB_0 (...)] with TWO guard cells (folded into cells 0 and 1 at the flush): one address
//   per cell instead of one per tile and cell, no wrap-around of the right-hand cell, no clamp;
// * the cell of x' needs no exact division and no wrap of the position: the prediction equals a marker-by-
//   marker deposit to rounding anyway, and a deposit is continuous across a cell boundary (a position within
//   an ulp of one puts ~0 into the far cell either way) -- s = x' * (nx / lx), one multiplication, and the
//   CELL is wrapped (x' in (-lx, 2 lx) unless a marker crosses a box length in half a step: cells 0 ... nx,
//   right-hand neighbour up to nx + 1; cvt(NaN) = 0);
// * the constants of c are folded (pred_k = dt/2 Z/m).
// * the number of kept modes is a template parameter (1 or 2): the tile and accumulator addresses of a cell
//   are immediate offsets of one base address instead of a run-time loop's address arithmetic (a dozen integer
//   instructions per marker).
// The prediction tiles as 64-bit FIXED-POINT sums (round 6; the diagnostics' trick, device_diag.hpp):
// ten of the tiles' twelve LDS atomics per marker (two kept modes) hit random cells, where ds_add_u64 runs at 4.5 ns per
// wave-instruction against ds_add_f64's 8.7 (tools/lds_atomic_rate.hip) -- and the prediction only has to be right to
// rounding (the rho tile, whose sums are the reference's charge, stays in doubles).  Two kept modes at 1e8 markers:
// kernel 1.235 -> 1.08 ms (profiles/r06/experiments/ab_fx_tiles*.log).
//   * fxb[0], fxb[1] (device, per species): bounds on |q| (w, or p in full-f) and on |c| = dt/2 |p - w| |f0'/f0| |Z/m| -- seeded
//     by the host from the markers it loads and raised by the kernels (atomic max, monotone) to the largest value met AMONG
//     THE MARKERS THAT WENT THE FIXED-POINT WAY: the population's growth moves them, an outlier never does (one marker of
//     two-stream1 at v = 1e-9 would otherwise coarsen everybody's quantum by nine orders of magnitude, for good).  A
//     workgroup that finds NOT ONE of its markers within a bound -- the whole population has jumped: a field set by the host
//     -- raises that bound 256-fold per launch until the markers fit again;
//   * a term within 16x its bound (to be precise: below the cap the power-of-two scale leaves) is rounded ONCE to a power-of-two quantum chosen such that a workgroup's sums stay below
//     2^61 (about 2^-41 of the bound at 2e5 markers per workgroup: 1e-13 of a cell's sum), the sums themselves are exact and
//     independent of the atomics' order;
//   * a marker beyond 16x (two-stream1's v - 2/v near v = 0; weights scaled behind the library's back; NaN) adds its terms
//     straight into the global accumulators in doubles -- exact, slow, rare.
struct FxTiles {
  double s0, s1;      // quanta^-1 of the R0 slice and of the RA / RB slices (powers of two, wave-uniform: scalar registers; +inf with
                      // an unknown bound: every term then fails the test below and goes through the doubles)
  float mx0, mx1;     // largest |q|, |c| this thread has met among the markers within the bound (single precision: a register
                      // each); -1: markers met, none within the bound; -2: no marker met
};
// a value every lane holds alike, moved to scalar registers (the marker loop is at its VGPR budget)
__device__ __forceinline__ double wave_uniform(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(b)));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32)));
  return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}
__device__ __forceinline__ double fx_scale(double lim, double markers) {  // 2^k with lim * markers * 2^k <= 2^61
  const double t = lim * markers;
  return t > 0.0 && t < 0x1p900 ? ldexp(1.0, 60 - ilogb(t)) : __builtin_inf();   // (unknown bound: nothing passes the cap)
}
__device__ __forceinline__ void pred_add_fx(double *p, double v, double s) {  // RN(v s) added as a two's-complement integer
  const double t = fma(v, s, 6755399441055744.0);
  __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p),
                         static_cast<unsigned long long>(__double_as_longlong(t)) - 0x4338000000000000ull, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the workgroup's view of a bound: every wave folds its threads' mx into an LDS word as an integer CODE that orders like the
// values -- 0: no marker met, 1: markers met and none within the bound, 2 + the bits of the float (non-negative floats order as
// their bits) -- and behind the workgroup's barrier one thread raises the bound (positive doubles order as integers).
// (The code is exact: the weights of a delta-f run are ~1e-11 of order one at 1e8 markers, an encoding that shifted the
// value, mx + 3, rounded all of them to "0" and the bound never followed a growing mode.)
__device__ __forceinline__ void fx_note(unsigned *slot, float seen_f) {
  for (int off = 32; off > 0; off >>= 1) seen_f = fmaxf(seen_f, __shfl_down(seen_f, off, 64));
  if ((threadIdx.x & 63) == 0) {
    const unsigned code = seen_f >= 0.0f ? __float_as_uint(seen_f) + 2u : (seen_f > -1.5f ? 1u : 0u);
    __hip_atomic_fetch_max(slot, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__device__ __forceinline__ void fx_raise(double *bound, unsigned noted) {
  if (noted == 0u) return;
  const double cur = __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double want = 0.0;
  if (noted >= 2u) want = static_cast<double>(__uint_as_float(noted - 2u)) * (1.0 + 0x1p-18);   // (rounded twice on the way: not below what was met)
  else if (cur > 0.0) want = 256.0 * cur;                                                        // markers, and none of them within the bound
  if (want > cur && want < 0x1p120)
    __hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(bound), static_cast<unsigned long long>(__double_as_longlong(want)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DIST, int MODE, int POW2, int NM>
__device__ __forceinline__ double pred_one(const One &n, double p, int ix, double wl, const double *sAB, double *sP,
                                           const StepArgsDev &a, FxTiles &fx) {
  constexpr int nm = NM, np1 = 1 + 2 * NM;
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
  // the overflow path (rare): slice k of this marker straight into the global accumulators' cells of cl, cr (the guard
  // cells nx, nx + 1 are cells 0, 1)
  auto slow = [&](int k, double vl, double vr) {
    const int nx = a.g.nx;
    const int gl_c = ih >= nx ? ih - nx : ih;
    int gr_c = ih + 1;
    gr_c = gr_c >= nx ? gr_c - nx : gr_c;
    gr_c = gr_c >= nx ? gr_c - nx : gr_c;   // (nx = 1: cell nx + 1 is cell 0 as well)
    if (vl != 0.0) glb_add(a.pred + static_cast<size_t>(k) * nx + gl_c, vl);   // (zeros: a species without perturbation has no bound)
    if (vr != 0.0) glb_add(a.pred + static_cast<size_t>(k) * nx + gr_c, vr);
    if (vl != 0.0 || vr != 0.0)   // counted (pic1dp_hip_kernel_stats 13): a run whose count keeps growing has lost its bounds
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.fxb + 2), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // a term t goes the fixed-point way if |t| s <= fx_cap = 2^61 / (markers a workgroup takes): its sums then stay in 63 bits
  if constexpr (MODE == MODE_FULLF) {
    const double aq = fabs(p);
    const bool fixed = aq * fx.s0 <= a.fx_cap;
    fx.mx0 = fmaxf(fx.mx0, (fixed || fx.s0 > 0x1p1000) ? static_cast<float>(aq) : -1.0f);   // (no bound yet: whatever is met)
    if (fixed) {
      pred_add_fx(cl, wh * p, fx.s0);
      pred_add_fx(cr, wr * p, fx.s0);
    } else {
      slow(0, wh * p, wr * p);
    }
  } else {
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
    const double aq = fabs(n.w), ac = fabs(c);
    const bool ok0 = aq * fx.s0 <= a.fx_cap, ok1 = 2.0 * ac * fx.s1 <= a.fx_cap;   // (|A|, |B| <= 2; false for NaN)
    fx.mx0 = fmaxf(fx.mx0, (ok0 || fx.s0 > 0x1p1000) ? static_cast<float>(aq) : -1.0f);   // (no bound yet: whatever is met)
    fx.mx1 = fmaxf(fx.mx1, (ok1 || fx.s1 > 0x1p1000) ? static_cast<float>(ac) : -1.0f);
    const bool fixed = ok0 && ok1;
    if (fixed) {
      pred_add_fx(cl, wh * n.w, fx.s0);
      pred_add_fx(cr, wr * n.w, fx.s0);
    } else {
      slow(0, wh * n.w, wr * n.w);
    }
    const double *gl = sAB + __mul24(ix, 2 * nm), *gr = gl + 2 * nm;
    const double wlr = 1.0 - wl;
#pragma unroll
    for (int m = 0; m < nm; ++m) {
      const double2 tl = *reinterpret_cast<const double2 *>(gl + 2 * m), tr = *reinterpret_cast<const double2 *>(gr + 2 * m);
      const double A = fma(tr.x, wlr, tl.x * wl), B = fma(tr.y, wlr, tl.y * wl);  // (contraction is fine here)
      const double cA = c * A, cB = c * B;
      if (fixed) {
        pred_add_fx(cl + 1 + m, wh * cA, fx.s1);
        pred_add_fx(cr + 1 + m, wr * cA, fx.s1);
        pred_add_fx(cl + 1 + nm + m, wh * cB, fx.s1);
        pred_add_fx(cr + 1 + nm + m, wr * cB, fx.s1);
      } else {
        slow(1 + m, wh * cA, wr * cA);
        slow(1 + nm + m, wh * cB, wr * cB);
      }
    }
  }
  return t2;
}

// ---------------------------------------------------------------------------
// FUSED: the prologue of a one-pass launch solves the field of the PREVIOUS step itself (kernels.hpp FusedSolve).
// A time step then is ONE launch: the separate field launch (8-12 us of dependent round trips on one CU while 255
// idle) and the dependency gap in front of it (~3 us) go; what remains is the solve's own latency -- charge ->
// products -> the serial forward sums (the reference's order: 1 us at nx 192, 5.8 us at nx 1024 in the one-rank
// order) -> inverse --, run by EVERY workgroup for itself, side by side, with the very device functions of the
// field kernels (device_field.hpp: the same products in the same order, hence the same bits as
// k_field_solve_pair_sums1 gives for the same accumulators).  No workgroup waits for another, nothing is exchanged:
// the accumulators are complete when the launch starts, and every workgroup reads all of them (nx doubles, L2 hits).
// sE0: the tile E0 goes to -- and, until the sums have run, the products of the cos table; sX: products of the -sin
// table, then (TILE_EH) the tile of Eh; sSc: [88] scratch.  Workgroup 0 writes to memory what the field kernel
// would have: charge, chargeden, E, the kept mode, Eh and its kept mode, the field energy; and zeroes the
// accumulators the previous launch read.  Ends without a barrier: the caller's follows.
// ---------------------------------------------------------------------------
constexpr int FUSED_CHAIN_W = 8;  // register batch of the serial sums inside a kernel held to 80 VGPRs
// The marker loop's tables A = 2 fre, B = 2 fim (exact doublings: capi.cpp builds tabA / tabB that way) are staged by the
// solve itself from the fre / fim it loads for its products, and its inverse transforms read them back from the LDS
// (halved: exact again) -- one trip to memory in the prologue instead of three (tabA / tabB, fre / fim, fre / fim again).
struct TabCells {  // k_step_one: [cell][A B]
  double *ab;
  __device__ __forceinline__ void put(int c, double a, double b) const { ab[2 * c] = a, ab[2 * c + 1] = b; }
  __device__ __forceinline__ void get(int c, double &a, double &b) const { a = ab[2 * c], b = ab[2 * c + 1]; }
};
struct TabTiles {  // k_step_sums: a tile each
  double *sa, *sb;
  __device__ __forceinline__ void put(int c, double a, double b) const { sa[c] = a, sb[c] = b; }
  __device__ __forceinline__ void get(int c, double &a, double &b) const { a = sa[c], b = sb[c]; }
};
template <bool TILE_EH, class TAB>
__device__ __forceinline__ void fused_solve(const FusedSolve &fs, const TAB &tab, double *sE0, double *sX, double *sSc,
                                            double &re_h, double &im_h, unsigned long long *stamps = nullptr) {
  (void)stamps;
  const FieldArgs &f = fs.f;
  const int nx = f.nx;
  double *sMode = sSc, *sScr = sSc + 8, *sPart = sSc + 24;
  const bool lead = blockIdx.x == 0;
  double *sPc = sE0, *sPs = sX;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    for (int sp = 0; sp < f.nspecies; ++sp) {  // src/pic1dp_interaction.F90:126-127
      const double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
      double c1 = *r;
      for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
      c2 = c2 + c1 * f.Z[sp];
    }
    const double cd = chargeden_from(f, c2);  // :138-148
    if (lead) {
      f.charge[ix] = c2;
      f.chargeden[ix] = cd;
    }
    const double tr = f.fre[ix], ti = f.fim[ix];
    tab.put(ix, 2.0 * tr, 2.0 * ti);
    if (ix == 0) tab.put(nx, 2.0 * tr, 2.0 * ti);  // the guard cell behind the last one
    sPc[ix] = tr * cd;
    sPs[ix] = ti * cd;
  }
  if (lead) {  // the accumulators the previous launch read: nobody looks at them any more
    for (int64_t i = threadIdx.x; i < fs.zero_rho_n; i += blockDim.x) fs.zero_rho[i] = 0.0;
    for (int i = threadIdx.x; i < 8 * PRED_SUM_COPIES; i += blockDim.x) fs.zero_pred[i] = 0.0;
  }
  __syncthreads();
  STAMP_SOLVE(stamps, 2);
  const double ginv = f.grad_inv[0];
  {  // the forward sums (:231-240); beside them the last wave adds up the copies of the six sums, in copy order
    const double acc = lean_forward_sums<FUSED_CHAIN_W>(f, sPc, sPs, sPart, [&]() {
      const int k = static_cast<int>(threadIdx.x) - (static_cast<int>(blockDim.x) - 64);
      if (k >= 0 && k < 6) {
        double a = 0.0;
        static_assert(PRED_SUM_COPIES % 8 == 0, "eight copies in flight at a time");
        for (int c0 = 0; c0 < PRED_SUM_COPIES; c0 += 8) {
          double t[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) t[c] = fs.pred_in[(c0 + c) * 8 + k];
#pragma unroll
          for (int c = 0; c < 8; ++c) a = (c0 + c == 0) ? t[0] : a + t[c];
        }
        sMode[2 + k] = a;
      }
    });
    if (threadIdx.x == 0) {
      const double im = acc * f.sc_im * ginv;
      sMode[1] = im;
      if (lead) f.mode_im[0] = im;
    } else if (threadIdx.x == 1) {
      const double re = acc * f.sc_re * ginv;
      sMode[0] = re;
      if (lead) f.mode_re[0] = re;
    }
  }
  __syncthreads();
  STAMP_SOLVE(stamps, 3);
  const double re = sMode[0], im = sMode[1];
  double ac, as;
  pred_forward_sums(f, fs.pt, sMode + 2, re, im, ac, as);
  STAMP_SOLVE(stamps, 4);
  im_h = ac * f.sc_im * ginv;  // :234, :239, :243-247
  re_h = as * f.sc_re * ginv;
  if (lead && threadIdx.x == 0) {
    fs.mode_h[0] = re_h;
    fs.mode_h[1] = im_h;
  }
  double e2 = 0.0;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {  // both inverse transforms, :251-257
    double tA, tB;
    tab.get(ix, tA, tB);
    const double tr = 0.5 * tA, ti = 0.5 * tB;  // fre, fim: bit for bit
    double a = 0.0;
    a = a + tr * re;
    a = a + ti * im;
    const double e = a * 2.0;
    double b = 0.0;
    b = b + tr * re_h;
    b = b + ti * im_h;
    const double eh = b * 2.0;
    sE0[ix] = e;
    if constexpr (TILE_EH) sX[ix] = eh;
    if (ix == 0) {  // the guard cell behind the last one
      sE0[nx] = e;
      if constexpr (TILE_EH) sX[nx] = eh;
    }
    e2 += e * e;
    if (lead) {
      f.E[ix] = e;
      fs.E_h[ix] = eh;
    }
  }
  if (lead && f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

// PRIV: the prediction of ONE kept mode as six sums instead of three tiles (the algebra of k_step_sums below: the
// solve only ever projects the predicted charge on the kept mode's tables, and the projection of a CIC deposit of q
// at x' is q times the gather of the table at x'), accumulated in THREAD-PRIVATE LDS slots.  k_step_one is LDS-bound
// (profiles/r03/sq_counters_c3_onepass.json: the six prediction atomics at random cells cost ~12 cycles each in
// bank conflicts); an atomic into slot [k][thread] is conflict-free by construction (consecutive lanes, consecutive
// 8-byte words: 4 cycles), the second table gather is two reads, and -- unlike k_step_sums' register
// accumulators (126 VGPRs) -- the slots cost no registers: six waves per SIMD as before.
// sS: this thread's slot of sum 0; sum k lies k * PRIV_THREADS further (the kernel is launched with exactly that many
// threads).  K = [k0c k1c k2c k0s k1s k2s] as k_step_sums.
constexpr int PRIV_THREADS = STEP_PRIVATE_THREADS;
template <int DIST, int MODE, int POW2>
__device__ __forceinline__ double pred_one_private(const One &n, double p, int ix, double wl, const double *sAB, double *sS,
                                                   const StepArgsDev &a) {
  constexpr int stride = PRIV_THREADS;            // compile-time: the six slot addresses are immediate offsets of one
  const int nx = a.g.nx;
  // the long chain first (-f0'/f0: exp, reciprocal), with little else alive; then the two table gathers
  double t2 = 0.0, cA = 0.0, cB = 0.0;
  if constexpr (MODE != MODE_FULLF) {
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
    const double c = tmp1 * t2 * (a.pred_k * a.s.Z);
    const double *gl = sAB + 2 * ix;              // the tables at x: the field the half push will see
    const double2 tl = *reinterpret_cast<const double2 *>(gl), tr = *reinterpret_cast<const double2 *>(gl + 2);
    const double wlr = 1.0 - wl;
    cA = c * fma(tr.x, wlr, tl.x * wl);
    cB = c * fma(tr.y, wlr, tl.y * wl);
  }
  const double xh = fma(a.dt_half, n.v, n.x);     // the next step's half push of x (:261), to rounding
  const double sh = xh * a.snx;                   // its cell, wrapped as an integer (:102-108 to rounding)
  const double fh = floor(sh);
  int ih = static_cast<int>(fh);
  const double wr = sh - fh, wh = 1.0 - wr;
  ih = ih < 0 ? ih + nx : ih;
  ih = ih >= nx ? ih - nx : ih;
  if (static_cast<unsigned>(ih) >= static_cast<unsigned>(nx)) {  // more than a box length in half a step, NaN
    ih = ih % nx;
    if (ih < 0) ih += nx;
  }
  const double *hl = sAB + 2 * ih;                // [A B] of cell ih, then of cell ih + 1 (cell nx: the guard, = cell 0)
  const double2 ul = *reinterpret_cast<const double2 *>(hl), ur = *reinterpret_cast<const double2 *>(hl + 2);
  const double Ah = fma(ur.x, wr, ul.x * wh), Bh = fma(ur.y, wr, ul.y * wh);  // projection weights of the deposit at x'
  const double q = a.s.Z * (MODE == MODE_FULLF ? p : n.w);
  lds_add(sS, q * Ah);
  lds_add(sS + 3 * stride, q * Bh);
  if constexpr (MODE != MODE_FULLF) {
    lds_add(sS + stride, cA * Ah);
    lds_add(sS + 2 * stride, cB * Ah);
    lds_add(sS + 4 * stride, cA * Bh);
    lds_add(sS + 5 * stride, cB * Bh);
  }
  return t2;
}


// the private sums of one marker: six for one kept mode (two kept modes would take twenty: tried, HISTORY.md round 5)
template <int DIST, int MODE, int POW2, int NM>
__device__ __forceinline__ double priv_sums(const One &n, double p, int ix, double wl, const double *sAB, double *sS,
                                            const StepArgsDev &a) {
  return pred_one_private<DIST, MODE, POW2>(n, p, ix, wl, sAB, sS, a);
}

// T2: 0 no carry of -f0'/f0; 1 this step evaluates it, the next step's value is stored; 2 this
// step's value is loaded (stored by the previous k_step_one), the next step's stored
// NM: kept modes of the prediction tiles (1 .. PRED_MAX_MODES); PRIV (NM = 1): six sums in thread-private slots instead
template <int DIST, int MODE, int POW2, bool NT, int T2, int NM, bool PRIV = false, bool FUSED = false>
__global__ void __launch_bounds__(1024) PIC1DP_SIX_WAVES k_step_one(const StepArgsDev a) {
  static_assert(!FUSED || PRIV, "the fused solve serves the six-sum prediction");
  constexpr int NS = 6, PT = PRIV_THREADS;
  static_assert(!PRIV || NM == 1, "six sums: one kept mode");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  STAMP(a, 0);
  STAMP_HWID(a);
  exp_table_init();
  const int nx = a.g.nx;
  constexpr int nm = NM, np1 = 1 + 2 * NM;
  const int ne = (nx + 2) & ~1;
  double *sE0 = reinterpret_cast<double *>(smem);
  double *sEh = sE0 + ne;
  double *sAB = sEh + ne;                                        // [nx + 1][2 nm]: A_0 B_0 (A_1 B_1) per cell
  double *sR0 = sAB + static_cast<size_t>(nx + 1) * 2 * nm;
  double *sP = sR0 + ((nx + 2) & ~1);              // [nx + 2][1 + 2 nm]: R0 RA_m RB_m per cell
                                                                 // PRIV: [6][blockDim] private sums, then [16] scratch
  if constexpr (!FUSED) {
    for (int i = threadIdx.x; i < nx; i += blockDim.x) {
      sE0[i] = a.E0[i];
      sEh[i] = a.Eh[i];
    }
  }
  if constexpr (!FUSED) {
    for (int i = threadIdx.x; i < nm * (nx + 1); i += blockDim.x) {
      const int m = i / (nx + 1), c = i - m * (nx + 1), cs = c < nx ? c : 0;  // cell nx: the guard, = cell 0
      sAB[c * 2 * nm + 2 * m] = a.tabA[m * nx + cs];
      sAB[c * 2 * nm + 2 * m + 1] = a.tabB[m * nx + cs];
    }
  }
  zero_rho(sR0, a.g);
  if constexpr (FUSED) {  // E0 and Eh of this step from the previous launch's deposits and six sums (scratch: the
                          // head of the slots, zeroed behind it)
    double re_h, im_h;
#ifdef PIC1DP_TUNE_STAMPS
    fused_solve<true>(a.fused, TabCells{sAB}, sE0, sEh, sP, re_h, im_h, a.stamps);
#else
    fused_solve<true>(a.fused, TabCells{sAB}, sE0, sEh, sP, re_h, im_h);  // (stages sAB as well)
#endif
    __syncthreads();
  } else if (threadIdx.x == 0) {
    sE0[nx] = a.E0[0];
    sEh[nx] = a.Eh[0];
  }
  unsigned *sDraw = reinterpret_cast<unsigned *>(PRIV ? sP + NS * PT : sP + np1 * (nx + 2));  // the chunk counter of the drawn tail
  FxTiles fx{};
  fx.mx0 = fx.mx1 = -2.0f;
  if constexpr (PRIV) {
    for (int k = 0; k < NS; ++k) sP[k * PT + threadIdx.x] = 0.0;
  } else {
    // The tiles' fixed-point scales from the bounds as they stand when this WORKGROUP starts: read by ONE thread and handed
    // to the others through the LDS.  (Every thread reading them for itself was a race: the workgroups that finish first
    // raise the bounds while later ones start, two waves of one workgroup could see different values and add into the same
    // tile in different units -- found by the suite on a tuning build, one case in 596, energies off by 3e-4.)
    if (threadIdx.x == 0) {
      sP[0] = __hip_atomic_load(a.fxb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sP[1] = __hip_atomic_load(a.fxb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    fx.s0 = wave_uniform(fx_scale(16.0 * sP[0], a.fx_markers));
    fx.s1 = wave_uniform(fx_scale(32.0 * sP[1], a.fx_markers));   // |A|, |B| <= 2 (tables 2 cos, -2 sin)
    __syncthreads();
    for (int i = threadIdx.x; i < np1 * (nx + 2); i += blockDim.x) sP[i] = 0.0;
  }
  if (threadIdx.x == 0) sDraw[0] = sDraw[1] = sDraw[2] = 0u;   // ([1], [2]: the workgroup's view of the tiles' two bounds, fx_note)
  __syncthreads();
  STAMP(a, 1);
  double *sR = sR0;
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
  // The workgroup's pairs: row k = pairs (blockIdx + k gridDim) blockDim ..., dealt to its threads (static grid stride).
  // (PIC1DP_DYN_TAIL; pair_rows / draw_chunk above, spelled out here for this kernel's register budget):
  // the last dyn_tail / 16 of the rows are not dealt but DRAWN -- every wave
  // takes the next 64-pair chunk of them from a counter in the LDS (one ds_add_rtn_u32 per chunk, no device-scope traffic),
  // so that the waves that run ahead take more and the workgroup meets its final barrier together.
  const int64_t first = static_cast<int64_t>(blockIdx.x) * blockDim.x;
  int dealt = first < npair ? static_cast<int>((npair - first + stride - 1) / stride) : 0;  // rows of this workgroup
  int drawn_total = 0;
  if (a.dyn_tail > 0) {
    const int drawn_rows = (dealt * a.dyn_tail) >> 4;
    dealt -= drawn_rows;
    drawn_total = drawn_rows * static_cast<int>(blockDim.x >> 6);
  }
  int64_t j = first + threadIdx.x;
  for (int k = 0;; ++k, j += stride) {
    if (k >= dealt) {
      int c = 0;
      if ((threadIdx.x & 63) == 0) c = static_cast<int>(__hip_atomic_fetch_add(sDraw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      c = __builtin_amdgcn_readfirstlane(c);
      if (c >= drawn_total) break;
      const int waves = static_cast<int>(blockDim.x >> 6);
      j = first + static_cast<int64_t>(dealt + c / waves) * stride + (c % waves) * 64 + (threadIdx.x & 63);
    }
    if (j >= npair) continue;
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
    double u0, u1;
    if constexpr (PRIV)
      u0 = priv_sums<DIST, MODE, POW2, NM>(n0, P.x, i0, l0, sAB, sP + threadIdx.x, a);
    else
      u0 = pred_one<DIST, MODE, POW2, NM>(n0, P.x, i0, l0, sAB, sP, a, fx);
    PAIR_FENCE();
    const One n1 = step_full_one<DIST, MODE, POW2, CARRY_IN>(X.y, V.y, W.y, P.y, sE0, sEh, sR, a, T.y, &i1, &l1);
    if constexpr (PRIV)
      u1 = priv_sums<DIST, MODE, POW2, NM>(n1, P.y, i1, l1, sAB, sP + threadIdx.x, a);
    else
      u1 = pred_one<DIST, MODE, POW2, NM>(n1, P.y, i1, l1, sAB, sP, a, fx);
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
    double u;
    if constexpr (PRIV)
      u = priv_sums<DIST, MODE, POW2, NM>(n, p, ic, lc, sAB, sP + threadIdx.x, a);
    else
      u = pred_one<DIST, MODE, POW2, NM>(n, p, ic, lc, sAB, sP, a, fx);
    if constexpr (CARRY_OUT) a.t2[a.np - 1] = u;
  }
  STAMP_LOOP(a, 2);
  if constexpr (!PRIV) {
    fx_note(sDraw + 1, fx.mx0);
    fx_note(sDraw + 2, fx.mx1);
  }
  __syncthreads();
  STAMP_LOOP(a, 3);
  flush_rho(sR0, a.rho, a.g);
  STAMP_LOOP(a, 4);
  if constexpr (PRIV) {  // the six sums: wave k adds up the slots of sum k (12 reads per lane, a wave reduction), one
                         // global atomic each -- no barrier beyond the one above
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 6) {
      const double *s = sP + wave * PRIV_THREADS + lane;
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < PRIV_THREADS / 64; ++j) t += s[64 * j];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (lane == 0) glb_add(a.pred + (blockIdx.x % PRED_SUM_COPIES) * 8 + wave, t);
    }
    STAMP(a, 5);
    if constexpr (!FUSED)  // several ranks: the last workgroup to finish packs / posts this rank's charge (StepTail)
      if (a.tail.mode) step_tail(a.tail, reinterpret_cast<double *>(smem));
    return;
  }
  // the guard cells nx, nx + 1 are cells 0, 1 (mod nx); then one global atomic per cell and slice
  if (threadIdx.x < 2 * np1) {
    const int g = threadIdx.x / np1, k = threadIdx.x - g * np1;
    const int to = (nx + g) % nx;
    if (g == 0 || to != 0 || nx > 1)   // (64-bit integers: the fixed-point sums)
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&sP[to * np1 + k]),
                             static_cast<unsigned long long>(__double_as_longlong(sP[(nx + g) * np1 + k])), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  {
    const double q0 = 1.0 / fx.s0, q1 = 1.0 / fx.s1;   // (powers of two: exact; 1 / inf = 0 with nothing in the tiles)
    const int rot = static_cast<int>((static_cast<long long>(blockIdx.x) * nx) / gridDim.x);
    for (int i = threadIdx.x; i < np1 * nx; i += blockDim.x) {
      const int k = i / nx;
      int c = i - k * nx + rot;
      if (c >= nx) c -= nx;
      const long long fixed = __double_as_longlong(sP[c * np1 + k]);
      if (fixed != 0) glb_add(&a.pred[static_cast<size_t>(k) * nx + c], static_cast<double>(fixed) * (k == 0 ? q0 : q1));
    }
  }
  if (threadIdx.x == 0) {   // the bounds follow the population (rarely an atomic: they only ever grow)
    fx_raise(a.fxb, sDraw[1]);
    fx_raise(a.fxb + 1, sDraw[2]);
  }
  STAMP(a, 5);
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

// tuning builds: -DPIC1DP_SUMS_WAVES=n holds k_step_sums to the register budget of n waves per SIMD
#ifdef PIC1DP_SUMS_WAVES
#define PIC1DP_SUMS_ATTR __attribute__((amdgpu_waves_per_eu(PIC1DP_SUMS_WAVES)))
#else
#define PIC1DP_SUMS_ATTR
#endif
template <int DIST, int MODE, int POW2, bool NT, int T2, bool FUSED = false>
__global__ void __launch_bounds__(1024) PIC1DP_SUMS_ATTR k_step_sums(const StepArgsDev a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  STAMP(a, 0);
  STAMP_HWID(a);
  exp_table_init();
  const int nx = a.g.nx;
  const int ne = (nx + 2) & ~1;
  double *sE0 = reinterpret_cast<double *>(smem);
  double *sA = sE0 + ne;
  double *sB = sA + ne;
  double *sR0 = sB + ne;
  double *sScr = sR0 + ((nx + 2) & ~1);  // [6][16] reduction scratch
  unsigned *sDraw = reinterpret_cast<unsigned *>(sScr);  // the chunk counter of the drawn tail: the head of the scratch,
                                                         // which the reductions need only behind the loop's barrier
  if constexpr (!FUSED) {
    for (int i = threadIdx.x; i < nx; i += blockDim.x) {
      sE0[i] = a.E0[i];
      sA[i] = a.tabA[i];
      sB[i] = a.tabB[i];
    }
    if (threadIdx.x == 0) {
      sE0[nx] = a.E0[0];
      sA[nx] = a.tabA[0];
      sB[nx] = a.tabB[0];
    }
  }
  double re_h, im_h;
  if constexpr (FUSED) {  // E0 and the kept mode of Eh from the previous launch's deposits and six sums; the rho tile
                          // holds the second row of products meanwhile
    fused_solve<false>(a.fused, TabTiles{sA, sB}, sE0, sR0, sScr, re_h, im_h);  // (stages sA, sB as well; its sums have left
                                                                               // the rho tile behind its last barrier)
  } else {
    re_h = *a.eh_re;
    im_h = *a.eh_im;
  }
  zero_rho(sR0, a.g);
  if (threadIdx.x == 0) *sDraw = 0u;
  __syncthreads();
  STAMP(a, 1);
  const ModeField sEh{sA, sB, re_h, im_h};
  double *sR = sR0;
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
  // the workgroup's rows of pairs: the first ones dealt to its threads, the last dyn_tail / 16 drawn by its waves from the
  // LDS counter (k_step_one has the same loop and the reason)
  const int64_t first = static_cast<int64_t>(blockIdx.x) * blockDim.x;
  int dealt = first < npair ? static_cast<int>((npair - first + stride - 1) / stride) : 0;
  int drawn_total = 0;
  if (a.dyn_tail > 0) {
    const int drawn_rows = (dealt * a.dyn_tail) >> 4;
    dealt -= drawn_rows;
    drawn_total = drawn_rows * static_cast<int>(blockDim.x >> 6);
  }
  int64_t j = first + threadIdx.x;
  for (int k = 0;; ++k, j += stride) {
    if (k >= dealt) {
      int c = 0;
      if ((threadIdx.x & 63) == 0) c = static_cast<int>(__hip_atomic_fetch_add(sDraw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      c = __builtin_amdgcn_readfirstlane(c);
      if (c >= drawn_total) break;
      const int waves = static_cast<int>(blockDim.x >> 6);
      j = first + static_cast<int64_t>(dealt + c / waves) * stride + (c % waves) * 64 + (threadIdx.x & 63);
    }
    if (j >= npair) continue;
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
  STAMP_LOOP(a, 2);
  __syncthreads();
  STAMP_LOOP(a, 3);
  flush_rho(sR0, a.rho, a.g);
  STAMP_LOOP(a, 4);
  // the six sums: workgroup reduction, one global atomic each into one of the copies (kernels.hpp PRED_SUM_COPIES)
  const double mine[6] = {ks.k0c, ks.k1c, ks.k2c, ks.k0s, ks.k1s, ks.k2s};
  block_sum6_add(mine, sScr, a.pred + (blockIdx.x % PRED_SUM_COPIES) * 8);
  STAMP(a, 5);
  if constexpr (!FUSED)  // several ranks: the last workgroup to finish packs / posts this rank's charge (StepTail)
    if (a.tail.mode) step_tail(a.tail, reinterpret_cast<double *>(smem));
}

#ifdef PIC1DP_TUNE_STAMPS
constexpr size_t kStampGrid = 16384;
struct StampState {
  unsigned long long *dev = nullptr;
  long launches = 0, at = -1;
  int blocks = 0, threads = 0;
};
StampState &stamp_state() {
  static StampState s;
  return s;
}
// before launch number n + 1: the stamps of launch n = PIC1DP_STAMP_AT go to PIC1DP_STAMP_FILE
hipError_t stamp_hook(StepArgsDev &d, const LaunchCfg &lc, hipStream_t st) {
  StampState &s = stamp_state();
  if (!s.dev) {
    hipError_t e = hipMalloc(&s.dev, sizeof(unsigned long long) * 8 * kStampGrid);
    if (e != hipSuccess) return e;
    if (const char *a = tuning_env("PIC1DP_STAMP_AT")) s.at = std::atol(a);
  }
  if (s.launches == s.at + 1 && s.at >= 0) {
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    std::vector<unsigned long long> h(static_cast<size_t>(8) * s.blocks);
    e = hipMemcpy(h.data(), s.dev, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    if (const char *fn = tuning_env("PIC1DP_STAMP_FILE")) {
      if (FILE *f = std::fopen(fn, "w")) {
        std::fprintf(f, "# blocks %d threads %d\n", s.blocks, s.threads);
        for (int b = 0; b < s.blocks; ++b) {
          for (int k = 0; k < 7; ++k) std::fprintf(f, "%llu ", h[static_cast<size_t>(b) * 8 + k]);
          std::fprintf(f, "\n");
        }
        std::fclose(f);
      }
    }
  }
  s.launches++;
  s.blocks = lc.blocks;
  s.threads = lc.threads;
  d.stamps = s.dev;
  return static_cast<size_t>(lc.blocks) <= kStampGrid ? hipSuccess : hipErrorInvalidValue;
}
#endif

template <typename K>
hipError_t launch_step_kernel(K kern, const StepArgsDev &d0, const LaunchCfg &lc, hipStream_t st) {
#ifdef PIC1DP_TUNE_STAMPS
  StepArgsDev d = d0;
  if (hipError_t e = stamp_hook(d, lc, st); e != hipSuccess) return e;
#else
  const StepArgsDev &d = d0;
#endif
  if (lc.lds > 64 * 1024) {  // opt in to > 64 KiB of dynamic LDS (idempotent, cheap)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PARTICLE_LDS_CAP);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(lc.blocks), dim3(lc.threads), lc.lds, st, d);
  return hipGetLastError();
}

// -f0'/f0 carried to the next step through memory (T2 = 1, 2): what the reference-order form of the exp-bearing
// distributions does by default (DIST 2, 3).  For every other distribution the carry is a measured-and-dropped variant
// (DIST 4, 5: the one-exp form evaluates again, 56 B instead of 72; DIST 0, 1 have nothing worth carrying): built for a
// tuning build's PIC1DP_CARRY only -- two thirds of the one-pass instantiations of those units (round 6).
#ifdef PIC1DP_TUNING
template <int DIST>
constexpr bool kCarryBuilt = true;
template <int DIST>
constexpr bool kNoCarryBuilt = true;
#else
template <int DIST>
constexpr bool kCarryBuilt = DIST == 2 || DIST == 3;
// ... and the reference-order units (DIST 2, 3) ALWAYS carry (capi_step.cpp: carry_one), so their carry-less one-pass
// instantiations are a tuning build's too (PIC1DP_CARRY=0): a third of those two units, the largest of the library
template <int DIST>
constexpr bool kNoCarryBuilt = !(DIST == 2 || DIST == 3);
#endif

static_assert(PRED_MAX_MODES == 2 || PRED_MAX_MODES == 3, "k_step_one is instantiated for one and two (three) kept modes");
static_assert(PRIV_THREADS == STEP_PRIVATE_THREADS, "slot stride of k_step_one<PRIV> = its workgroup size");
template <int DIST, int MODE, int POW2, int NM>
hipError_t launch_step_one(const StepArgsDev &d, int t2m, const LaunchCfg &lc, hipStream_t st) {
  if (d.nt) {
    if constexpr (kCarryBuilt<DIST>) {
      if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 2, NM>, d, lc, st);
      if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 1, NM>, d, lc, st);
    }
    if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
    if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 0, NM>, d, lc, st);
    return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
  }
  if constexpr (kCarryBuilt<DIST>) {
    if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 2, NM>, d, lc, st);
    if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 1, NM>, d, lc, st);
  }
  if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
  if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 0, NM>, d, lc, st);
  return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
}

template <int DIST, int MODE, int POW2, bool CARRY = false>
hipError_t launch_step_dmp(const StepArgsDev &d, bool full, const LaunchCfg &lc, hipStream_t st) {
  if (full && d.pred && d.pred_nm == -2) {  // one pass per step, six sums in thread-private LDS slots
    const int t2m = d.t2 ? d.t2_mode : 0;
    if (d.fused.on) {  // ... and the previous step's field solved in the prologue
      if (d.nt) {
        if constexpr (kCarryBuilt<DIST>) {
          if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 2, 1, true, true>, d, lc, st);
          if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 1, 1, true, true>, d, lc, st);
        }
        if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
        if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 0, 1, true, true>, d, lc, st);
        return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
      }
      if constexpr (kCarryBuilt<DIST>) {
        if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 2, 1, true, true>, d, lc, st);
        if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 1, 1, true, true>, d, lc, st);
      }
      if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
      if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 0, 1, true, true>, d, lc, st);
      return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
    }
    if (d.nt) {
      if constexpr (kCarryBuilt<DIST>) {
        if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 2, 1, true>, d, lc, st);
        if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 1, 1, true>, d, lc, st);
      }
      if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
      if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_one<DIST, MODE, POW2, true, 0, 1, true>, d, lc, st);
      return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
    }
    if constexpr (kCarryBuilt<DIST>) {
      if (t2m == 2) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 2, 1, true>, d, lc, st);
      if (t2m == 1) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 1, 1, true>, d, lc, st);
    }
    if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
    if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_one<DIST, MODE, POW2, false, 0, 1, true>, d, lc, st);
    return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
  }
  if (full && d.pred && d.pred_nm < 0) {  // one pass per step, prediction as six sums (large grids)
    const int t2m = d.t2 ? d.t2_mode : 0;
    if (d.fused.on) {
      if (d.nt) {
        if constexpr (kCarryBuilt<DIST>) {
          if (t2m == 2) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 2, true>, d, lc, st);
          if (t2m == 1) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 1, true>, d, lc, st);
        }
        if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
        if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 0, true>, d, lc, st);
        return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
      }
      if constexpr (kCarryBuilt<DIST>) {
        if (t2m == 2) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 2, true>, d, lc, st);
        if (t2m == 1) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 1, true>, d, lc, st);
      }
      if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
      if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 0, true>, d, lc, st);
      return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
    }
    if (d.nt) {
      if constexpr (kCarryBuilt<DIST>) {
        if (t2m == 2) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 2>, d, lc, st);
        if (t2m == 1) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 1>, d, lc, st);
      }
      if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
      if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, true, 0>, d, lc, st);
      return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
    }
    if constexpr (kCarryBuilt<DIST>) {
      if (t2m == 2) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 2>, d, lc, st);
      if (t2m == 1) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 1>, d, lc, st);
    }
    if (t2m != 0) return hipErrorInvalidValue;  // (a carry this build has no kernel for)
    if constexpr (kNoCarryBuilt<DIST>) return launch_step_kernel(k_step_sums<DIST, MODE, POW2, false, 0>, d, lc, st);
    return hipErrorInvalidValue;  // (no carry-less kernel in this unit: its distribution always carries)
  }
  if (full && d.pred) {  // one pass per step: also predicts the next step's first-sub-step charge
    const int t2m = d.t2 ? d.t2_mode : 0;
    if (d.pred_nm == 1) return launch_step_one<DIST, MODE, POW2, 1>(d, t2m, lc, st);
    if (d.pred_nm == 2) return launch_step_one<DIST, MODE, POW2, 2>(d, t2m, lc, st);
    if constexpr (PRED_MAX_MODES >= 3)
      if (d.pred_nm == 3) return launch_step_one<DIST, MODE, POW2, 3>(d, t2m, lc, st);
    return hipErrorInvalidValue;  // PRED_MAX_MODES
  }
  if (full && d.dist_out) {  // with the diagnostics of output_all (their histograms as fixed-point sums where the host knows bounds)
    if (d.diag_fx)
      return d.nt ? launch_step_kernel(k_step_full<DIST, MODE, POW2, true, CARRY, true, true>, d, lc, st)
                  : launch_step_kernel(k_step_full<DIST, MODE, POW2, false, CARRY, true, true>, d, lc, st);
    return d.nt ? launch_step_kernel(k_step_full<DIST, MODE, POW2, true, CARRY, true>, d, lc, st)
                : launch_step_kernel(k_step_full<DIST, MODE, POW2, false, CARRY, true>, d, lc, st);
  }
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
  if (!deltaf) {
    if constexpr (DIST == 0)
      return pow2 ? launch_step_dmp<0, MODE_FULLF, 1>(d, full, lc, st)
                  : launch_step_dmp<0, MODE_FULLF, 0>(d, full, lc, st);
    return hipErrorInvalidValue;  // step_dispatch.cpp sends every full-f species to the DIST 0 unit
  }
  // general divisor constants and an exp-bearing distribution: -f0'/f0 carried between the kernels
  const bool carry = !pow2 && d.t2 != nullptr && (DIST == 2 || DIST == 3);
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

}  // namespace

template <>
hipError_t launch_step_dist<PIC1DP_STEP_DIST>(const StepArgsDev &d, int deltaf, int linear, bool full, const LaunchCfg &lc,
                                              hipStream_t st) {
  return launch_step_d<PIC1DP_STEP_DIST>(d, deltaf, linear, full, lc, st);
}

}  // namespace pic1dp
