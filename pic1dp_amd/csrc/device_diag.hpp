// device_diag.hpp -- the per-marker part and the finish of output_all's diagnostics, shared by k_ptcldist
// (kernels_diag.hip) and the DIAG variant of k_step_full (kernels_step.hip)
#pragma once
#include "device_math.hpp"

namespace pic1dp {
namespace {

// ---------------------------------------------------------------------------
// diagnostics of output_all, per marker (used by k_ptcldist and by the DIAG variant of
// k_step_full): src/pic1dp_output.F90:126-151 (kinetic sums) and :239-315 (histograms)
// ---------------------------------------------------------------------------
// The output is [markr_xv | total_xv | pertb_xv | markr_v | total_v | pertb_v], planes of nx_opd * nv_opd bins
// (index iv * nx_opd + ix) and rows of nv_opd.  A workgroup's LDS copy (LDS = true) holds the three (x, v) planes
// INTERLEAVED bin by bin, h[bin][markr total pertb]: the three atomics of a corner are neighbours -- one address per
// corner, the planes at immediate offsets (round 5; twelve separate addresses per marker before) --; the flush maps
// the copy back onto the output's planes.  LDS = false (histograms too large for a CU's LDS) adds straight into the
// output.
struct DistBins {
  double *h;       // LDS: [nxv][3] then [3][nv]; else the output itself
  int nxv, nv;     // nx_opd*nv_opd, nv_opd
  template <bool LDS>
  __device__ __forceinline__ double *bin(int cell) const { return LDS ? h + 3 * cell : h + cell; }
  template <bool LDS>
  __device__ __forceinline__ int plane() const { return LDS ? 1 : nxv; }   // doubles from a bin's plane k to k + 1
  __device__ __forceinline__ double *vv(int k) const { return h + static_cast<size_t>(3) * nxv + k * nv; }
};
struct DistSums {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;  // sum v^2, v^2 p, v^2 w of this thread
  double maxp = 0.0, maxw = 0.0;        // max |p|, |w| of its markers (FX: the bounds the NEXT pass scales with)
  int over = 0;                         // FX: a |p| or |w| beyond the bound the pass was scaled for
};
// FX (round 5): the LDS copy of the (x, v) histograms holds 64-bit FIXED-POINT sums instead of doubles.  What bounds a
// diagnostics pass is its twelve LDS atomics at random bins per marker, and ds_add_u64 runs 1.9x the rate of ds_add_f64
// there (4.5 against 8.7 ns per wave-instruction and CU, tools/lds_atomic_rate.hip).  Plane k is scaled by the power of
// two sc[k] chosen on the host from a bound on |q_k| (1, max |p|, max |w| of the previous pass, with a margin) such that
// a workgroup's sums stay below 2^62; a term is rounded ONCE to a multiple of 1 / sc[k] (<= 2^-44 of the bound: 6e-14
// relative, against the ~1e-16 sqrt(n) of a double sum in some order), the sums themselves are exact -- and independent of
// the atomics' order.  A marker beyond the bound sets `over`; the host then repeats the pass with double sums.
// (struct DistScale: kernels.hpp)
// RN(x * s) as a two's-complement 64-bit integer, |x * s| < 2^51: one FMA onto 1.5 * 2^52 and the magic's high word off
__device__ __forceinline__ unsigned long long to_fixed(double x, double s) {
  const double t = fma(x, s, 6755399441055744.0);
  return static_cast<unsigned long long>(__double_as_longlong(t)) - 0x4338000000000000ull;
}
__device__ __forceinline__ void lds_add_u64(double *p, unsigned long long v) {
  __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool LDS>
__device__ __forceinline__ void bin_add(double *p, double v) {
  if constexpr (LDS) {
    lds_add(p, v);
  } else {
    glb_add(p, v);
  }
}

// a / c for the constant c with rc = RN(1 / c): RN(a / c) by Markstein's construction (device_math.hpp div_const,
// without its range test: the dividends here are positions in [0, lx] and v + v_max in (0, 2 v_max) -- normal numbers
// or exact zeros, for which the five operations return the IEEE quotient)
__device__ __forceinline__ double diag_div(double a, double c, double rc) {
  const double q0 = a * rc;
  const double r0 = fma(-c, q0, a);
  const double q1 = fma(r0, rc, q0);
  const double r1 = fma(-c, q1, a);
  return fma(r1, rc, q1);
}

template <bool LDS, bool DELTAF, bool FX = false>
__device__ __forceinline__ void ptcldist_one(double px, double pv, double pp, double pw, const DistGeom &dg,
                                             const DistBins &b, DistSums &sm, const DistScale *scale = nullptr) {
  static_assert(!FX || LDS, "fixed-point sums live in the workgroup's LDS copy");
  const int nxo = dg.nxo, nvo = dg.nvo;
  const double v2 = pv * pv;
  sm.s0 += v2;
  sm.s1 += v2 * pp;
  if constexpr (DELTAF) sm.s2 += v2 * pw;
  sm.maxp = fmax(sm.maxp, fabs(pp));
  if constexpr (DELTAF) sm.maxw = fmax(sm.maxw, fabs(pw));
  if (fabs(pv) >= dg.vmax) return;                      // :241
#if PIC1DP_FAST_DIV
  double sx = diag_div(px, dg.lx, dg.rlx) * static_cast<double>(nxo);    // :243
#else
  double sx = px / dg.lx * static_cast<double>(nxo);
#endif
  const double fx = floor(sx);
  const int ix = static_cast<int>(fx);
  sx = 1.0 - (sx - fx);
  const double av = pv + dg.vmax;
  double sv = (dg.vfast ? diag_div(av, dg.dv, dg.rdv) : av / dg.dv) * static_cast<double>(nvo - 1);  // :247
  const double fv = floor(sv);
  const int iv = static_cast<int>(fv);
  sv = 1.0 - (sv - fv);
  // memory safety only (the reference would write out of bounds)
  if (static_cast<unsigned>(ix) >= static_cast<unsigned>(nxo) || iv < 0 || iv + 1 >= nvo) return;
  const int ixr = ix + 1 > nxo - 1 ? 0 : ix + 1;        // :274-276
  const double sxr = 1.0 - sx, svu = 1.0 - sv;
  const int pl = b.template plane<LDS>();
  double *c00 = b.template bin<LDS>(iv * nxo + ix), *c10 = b.template bin<LDS>(iv * nxo + ixr);
  double *c01 = b.template bin<LDS>((iv + 1) * nxo + ix), *c11 = b.template bin<LDS>((iv + 1) * nxo + ixr);
  // the reference's products, in its order: (sx * sv), (sx * sv) * p, (sx * sv) * w, ...
  const double w00 = sx * sv, w01 = sx * svu, w10 = sxr * sv, w11 = sxr * svu;
  if constexpr (FX) {
    if (fabs(pp) > scale->bound[1] || (DELTAF && fabs(pw) > scale->bound[2])) {  // (NaN compares false: it would poison a double sum too)
      sm.over = 1;
      return;
    }
    const double s0 = scale->sc[0], s1 = scale->sc[1], s2 = scale->sc[2];
    lds_add_u64(c00, to_fixed(w00, s0));
    lds_add_u64(c00 + pl, to_fixed(w00 * pp, s1));
    if constexpr (DELTAF) lds_add_u64(c00 + 2 * pl, to_fixed(w00 * pw, s2));
    lds_add_u64(c01, to_fixed(w01, s0));
    lds_add_u64(c01 + pl, to_fixed(w01 * pp, s1));
    if constexpr (DELTAF) lds_add_u64(c01 + 2 * pl, to_fixed(w01 * pw, s2));
    lds_add_u64(c10, to_fixed(w10, s0));
    lds_add_u64(c10 + pl, to_fixed(w10 * pp, s1));
    if constexpr (DELTAF) lds_add_u64(c10 + 2 * pl, to_fixed(w10 * pw, s2));
    lds_add_u64(c11, to_fixed(w11, s0));
    lds_add_u64(c11 + pl, to_fixed(w11 * pp, s1));
    if constexpr (DELTAF) lds_add_u64(c11 + 2 * pl, to_fixed(w11 * pw, s2));
    return;
  }
  bin_add<LDS>(c00, w00);
  bin_add<LDS>(c00 + pl, w00 * pp);
  if constexpr (DELTAF) bin_add<LDS>(c00 + 2 * pl, w00 * pw);
  bin_add<LDS>(c01, w01);
  bin_add<LDS>(c01 + pl, w01 * pp);
  if constexpr (DELTAF) bin_add<LDS>(c01 + 2 * pl, w01 * pw);
  bin_add<LDS>(c10, w10);
  bin_add<LDS>(c10 + pl, w10 * pp);
  if constexpr (DELTAF) bin_add<LDS>(c10 + 2 * pl, w10 * pw);
  bin_add<LDS>(c11, w11);
  bin_add<LDS>(c11 + pl, w11 * pp);
  if constexpr (DELTAF) bin_add<LDS>(c11 + 2 * pl, w11 * pw);
  if constexpr (!LDS) {                                 // :300-314
    bin_add<LDS>(&b.vv(0)[iv], sv);
    bin_add<LDS>(&b.vv(1)[iv], sv * pp);
    if constexpr (DELTAF) bin_add<LDS>(&b.vv(2)[iv], sv * pw);
    bin_add<LDS>(&b.vv(0)[iv + 1], svu);
    bin_add<LDS>(&b.vv(1)[iv + 1], svu * pp);
    if constexpr (DELTAF) bin_add<LDS>(&b.vv(2)[iv + 1], svu * pw);
  }
}


// per-workgroup partial kinetic sums, v histograms as row sums, flush of the LDS copy
// maximum over the workgroup (valid on thread 0); the values are >= 0
__device__ __forceinline__ double block_max(double v, double *scratch) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) t = fmax(t, scratch[w]);
  return t;
}

// PSTRIDE: doubles per workgroup in `partial` -- 3 (the kinetic sums; k_step_full<DIAG>) or 6 (+ max |p|, max |w|, the
// overflow flag; k_ptcldist)
template <bool LDS, bool DELTAF, bool FX = false, int PSTRIDE = 3>
__device__ __forceinline__ void ptcldist_finish(const DistGeom &dg, const DistBins &b, const DistSums &sm, double *scr,
                                                double *out, double *partial, const DistScale *fx = nullptr) {
  const int nxo = dg.nxo, nvo = dg.nvo, nxv = nxo * nvo, ntot = 3 * nxv + 3 * nvo;
  if (partial) {
    const double t0 = block_sum(sm.s0, scr);
    const double t1 = block_sum(sm.s1, scr);
    const double t2 = block_sum(sm.s2, scr);
    if (threadIdx.x == 0) {
      partial[blockIdx.x * PSTRIDE + 0] = t0;
      partial[blockIdx.x * PSTRIDE + 1] = t1;
      partial[blockIdx.x * PSTRIDE + 2] = t2;
    }
    if constexpr (PSTRIDE >= 6) {
      const double mp = block_max(sm.maxp, scr);
      const double mw = block_max(sm.maxw, scr);
      const double ov = block_max(sm.over ? 1.0 : 0.0, scr);
      if (threadIdx.x == 0) {
        partial[blockIdx.x * PSTRIDE + 3] = mp;
        partial[blockIdx.x * PSTRIDE + 4] = mw;
        partial[blockIdx.x * PSTRIDE + 5] = ov;
      }
    }
  }
  if constexpr (LDS) {
    __syncthreads();
    // v histograms = row sums of the (x,v) histograms, one thread per (k, iv) (FX: exact integer sums, then scaled)
    for (int t = threadIdx.x; t < (DELTAF ? 3 : 2) * nvo; t += blockDim.x) {
      const int k = t / nvo, iv = t - k * nvo;
      const double *row = b.h + static_cast<size_t>(3) * iv * nxo + k;
      if constexpr (FX) {
        long long acc = 0;
        for (int ix = 0; ix < nxo; ++ix) acc += __double_as_longlong(row[3 * ix]);
        b.vv(k)[iv] = static_cast<double>(acc) * fx->inv[k];
      } else {
        double acc = 0.0;
        for (int ix = 0; ix < nxo; ++ix) acc += row[3 * ix];
        b.vv(k)[iv] = acc;
      }
    }
    __syncthreads();
    const int rot = static_cast<int>((static_cast<long long>(blockIdx.x) * ntot) / gridDim.x);
    for (int i = threadIdx.x; i < ntot; i += blockDim.x) {
      int j = i + rot;                                  // j: index into the output
      if (j >= ntot) j -= ntot;
      double val;
      if (j < 3 * nxv) {
        const int k = j / nxv, cell = j - k * nxv;
        val = b.h[3 * cell + k];
        if constexpr (FX) val = static_cast<double>(__double_as_longlong(val)) * fx->inv[k];
      } else {
        val = b.h[j];
      }
      if (val != 0.0) glb_add(&out[j], val);
    }
  }
}

}  // namespace
}  // namespace pic1dp
