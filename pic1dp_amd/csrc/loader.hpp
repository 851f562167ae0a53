// loader.hpp -- native initial-condition generator (host side).
// Builds the marker arrays of one reference rank block exactly as
// particle_load does (src/pic1dp_particle.F90:145-269).
#pragma once
#include <cstdint>

#include "../../include/pic1dp_hip.h"
#include "multirand.hpp"

namespace pic1dp {

// PETSC_DECIDE ownership: n/size + (rank < n%size)
// (PetscSplitOwnership, used by src/pic1dp_particle.F90:89-94,129)
int64_t block_alloc(int64_t nglobal, int rank, int size);
// particle_np of a block, src/pic1dp_particle.F90:240-248
int64_t block_np(const pic1dp_input &in, int isp, int mype, int npe);

// Fill x, v, p, w (n = allocated slots of the block) for species isp, drawing
// from g in the reference's order: all v, then all x (:180, :222).  The
// element-wise weight formulas run on `nthreads` host threads; the draws are
// sequential.  Results do not depend on nthreads.
void load_block_species(const pic1dp_input &in, int isp, Multirand &g, int64_t n, double *x,
                        double *v, double *p, double *w, int nthreads);

}  // namespace pic1dp
