// step_args.hpp -- what the whole-step marker kernels receive, and the per-distribution launch entries
// (kernels_step.hip is compiled once per distribution; step_dispatch.cpp picks the instance)
#pragma once
#include "kernels.hpp"

namespace pic1dp {

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
  FusedSolve fused;             // kernels.hpp: the prologue solves the previous step's field
  StepTail tail;                // kernels.hpp: the last workgroup packs / posts this rank's charge
  DistScale dscale;             // k_step_full<DIAG, FX>
  int diag_fx;
  int dyn_tail;                 // sixteenths of a workgroup's chunks drawn from an LDS counter
  double *fxb;                  // k_step_one's tiles: [2] bounds on |q|, |c| of this species (kernels_step.hip FxTiles)
  double fx_markers, fx_cap;    // ... the most markers one workgroup of this launch takes, and 2^61 over it
#ifdef PIC1DP_TUNE_STAMPS  // tuning build (tools/stamp_probe.sh): [gridDim][8] wall-clock stamps of the phases of a workgroup
  unsigned long long *stamps;
#endif
};

// DIST: 0 Maxwellian, 1 two-stream1, 2 two-stream2, 3 bump-on-tail (-f0'/f0 in the reference's operation order),
// 4 / 5: two-stream2 / bump-on-tail with the one-exp form (device_math.hpp)
template <int DIST>
hipError_t launch_step_dist(const StepArgsDev &d, int deltaf, int linear, bool full, const LaunchCfg &lc, hipStream_t st);
template <> hipError_t launch_step_dist<0>(const StepArgsDev &, int, int, bool, const LaunchCfg &, hipStream_t);
template <> hipError_t launch_step_dist<1>(const StepArgsDev &, int, int, bool, const LaunchCfg &, hipStream_t);
template <> hipError_t launch_step_dist<2>(const StepArgsDev &, int, int, bool, const LaunchCfg &, hipStream_t);
template <> hipError_t launch_step_dist<3>(const StepArgsDev &, int, int, bool, const LaunchCfg &, hipStream_t);
template <> hipError_t launch_step_dist<4>(const StepArgsDev &, int, int, bool, const LaunchCfg &, hipStream_t);
template <> hipError_t launch_step_dist<5>(const StepArgsDev &, int, int, bool, const LaunchCfg &, hipStream_t);

}  // namespace pic1dp
