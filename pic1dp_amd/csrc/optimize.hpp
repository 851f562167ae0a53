// optimize.hpp -- marker optimisation (merge / remove / split) of one reference
// rank block, host side.
//
// The reference's particle_merge / particle_remove / particle_split
// (src/pic1dp_particle.F90:411-715) are sequential by construction: markers are
// visited in storage order, a removed marker is overwritten by the last one and
// visited again, merge bins remember storage indices, remove and split consume
// the rank's random stream in visiting order.  They run a handful of times per
// simulation (input_nmerge + input_nremove + input_nsplit events, all zero by
// default), so they stay on the host: the engine downloads the block, applies
// the routine below and uploads the result.  Not part of the timed path.
#pragma once
#include <cstdint>

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

}  // namespace pic1dp
