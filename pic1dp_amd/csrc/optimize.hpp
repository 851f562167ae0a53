// optimize.hpp -- marker optimisation (merge / remove / split) of one reference
// rank block, host side.
//
// The reference's particle_merge / particle_remove / particle_split
// (src/pic1dp_particle.F90:411-715) are sequential by construction: markers are
// visited in storage order, a removed marker is overwritten by the last one and
// visited again, merge bins remember storage indices, remove and split consume
// the rank's random stream in visiting order.  They run a handful of times per
// simulation (input_nmerge + input_nremove + input_nsplit events, all zero by
// default).  opt_* are the routines on host arrays (the whole block: PIC1DP_OPT_HOST=1, and the statement the
// planners below are checked against); the engine's default moves keys only (plan_*).  Not part of the timed path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/pic1dp_hip.h"
#include "multirand.hpp"

namespace pic1dp {

// this block's contribution to |delta f|(v), added into hist[nv]
// (particle_compute_dist_pertb_abs_v, :356-403)
void opt_histogram(const pic1dp_input &in, int64_t np, const double *v, const double *w, double *hist);

// arrays hold the block's allocated slots; np is updated
void opt_merge(const pic1dp_input &in, double threshold, const double *hist, int64_t &np, double *x,
               double *v, double *p, double *w);
void opt_remove(const pic1dp_input &in, double threshold, const double *hist, Multirand &rng, int64_t &np,
                double *x, double *v, double *p, double *w);
void opt_split(const pic1dp_input &in, double threshold, const double *hist, Multirand &rng,
               int64_t nalloc, int64_t &np, double *x, double *v, double *p, double *w);

// ---- the sequential part alone, for the GPU driver (capi_optimize.cpp, kernels_opt.hip) ----
// What makes the three routines sequential -- visiting order, swap-with-last, the waiting merge partner, the random
// stream -- depends on one small key per marker, not on the markers.  The plan_* functions walk the keys exactly as
// opt_merge / opt_remove / opt_split walk the markers and record what the device has to do to the 32 B of each.
// Indices are block-local; `id` names a marker by the slot it had when the event began.
struct OptMoves {                    // marker id[t] (beyond the new count) -> the t-th hole below it, holes ascending: the
  std::vector<uint32_t> id;          // positions whose own marker is gone (the device finds them: kernels_opt.hip opt_holes)
  // The walk's last act may be to drop the marker it is looking at in the LAST valid slot (nothing moves in, the count
  // drops onto it): that slot, np_new, then keeps what was looked at -- possibly a marker moved in from the tail a
  // moment before.  ghost: that marker's id (-1: the walk did not end that way).  Outside the valid range, but the
  // reference's VecSum over the whole local vector sees it (src/pic1dp_output.F90:126-150).
  int64_t ghost = -1;
};
struct MergePlan {
  std::vector<uint32_t> dst, idk;      // marker idk merges into the marker that ends at position dst
  OptMoves moves;
  int64_t np_new = 0;
};
// keys[np]: (x cell * nv + v cell) * 2 + (w > 0), or 0xFFFFFFFF for a marker left alone; nslots = nx * nv * 2
void plan_merge(const uint32_t *keys, int64_t np, size_t nslots, MergePlan &plan);
struct RemovePlan {
  OptMoves moves;
  std::vector<uint32_t> gone_bits;   // bit i: the marker that began the event in slot i < np_new was removed
  int64_t np_new = 0;
};
// typeremove 1: skip[np] (1: |delta f| >= limit, left alone), df null; typeremove 2: df[np] = |delta f| / peak, skip null
void plan_remove(const pic1dp_input &in, const uint8_t *skip, const double *df, Multirand &rng, int64_t np, RemovePlan &plan);
struct SplitPlan {
  std::vector<uint32_t> ks;            // parents that split, in visiting order
  std::vector<double> dv;              // [ks.size()][split_ngroup] velocity offsets of their pairs (scaled)
  int64_t np_new = 0;
};
// flag[np]: 1 = resonant marker (|delta f| > limit)
void plan_split(const pic1dp_input &in, const uint8_t *flag, Multirand &rng, int64_t nalloc, int64_t np, SplitPlan &plan);

}  // namespace pic1dp
