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

}  // namespace
}  // namespace pic1dp
