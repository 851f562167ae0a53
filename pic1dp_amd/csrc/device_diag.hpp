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
};

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

template <bool LDS, bool DELTAF>
__device__ __forceinline__ void ptcldist_one(double px, double pv, double pp, double pw, const DistGeom &dg,
                                             const DistBins &b, DistSums &sm) {
  const int nxo = dg.nxo, nvo = dg.nvo;
  const double v2 = pv * pv;
  sm.s0 += v2;
  sm.s1 += v2 * pp;
  if constexpr (DELTAF) sm.s2 += v2 * pw;
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
template <bool LDS, bool DELTAF>
__device__ __forceinline__ void ptcldist_finish(const DistGeom &dg, const DistBins &b, const DistSums &sm, double *scr,
                                                double *out, double *partial) {
  const int nxo = dg.nxo, nvo = dg.nvo, nxv = nxo * nvo, ntot = 3 * nxv + 3 * nvo;
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
      const double *row = b.h + static_cast<size_t>(3) * iv * nxo + k;
      double acc = 0.0;
      for (int ix = 0; ix < nxo; ++ix) acc += row[3 * ix];
      b.vv(k)[iv] = acc;
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
      } else {
        val = b.h[j];
      }
      if (val != 0.0) glb_add(&out[j], val);
    }
  }
}

}  // namespace
}  // namespace pic1dp
