/* pic1dp_probe.h -- C ABI of libpic1dp_probe.so: MEASUREMENT and test support, not part of the drop-in boundary
 * (that is include/pic1dp_hip.h).  Streaming-rate probes with the marker kernels' access shapes, and array
 * evaluations of the device functions the marker kernels call, built from the same device headers as the product
 * library (pic1dp_amd/csrc/device_math.hpp).  Loaded by bench.py, tools/ and tests/ only (pic1dp_amd/probe.py).
 * All functions return 0 on success; pic1dp_probe_last_error() describes the last failure of the calling thread. */
#ifndef PIC1DP_PROBE_H
#define PIC1DP_PROBE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *pic1dp_probe_last_error(void);

/* nread (1, 4, 7) arrays of n doubles read and nwrite (0, 1, 3) written, 16 B per lane, grid-stride, reps launches
 * after one warm-up; variant 0 plain, 1 non-temporal (the marker kernels' accesses), 2 plain with two pairs per
 * lane.  blocks / threads 0: the sub-step kernels' launch shape (four workgroups of 512 per CU). */
int pic1dp_probe_stream(int32_t device, int32_t nread, int32_t nwrite, int64_t n, int32_t reps, int32_t blocks,
                        int32_t threads, int32_t variant, double *gbytes_per_s);

/* the traffic of the second sub-step's kernel (4 arrays read, 3 written back in place, non-temporal) over a fresh
 * slab of n markers: ms[0] four arrays apart (SoA, a 2 MiB multiple + stagger_bytes), ms[1] interleaved in tiles of
 * 2^log2_tile markers (the product's layout, src: pic1dp_amd/csrc/kernels.hpp), ms[2], ms[3] the same read-only,
 * ms[4], ms[5] tiled with one workgroup per tile.  keep != 0 leaves the slab allocated (the next call lands in
 * other physical memory) until pic1dp_probe_release.  blocks / threads 0: two workgroups of 768 per CU. */
int pic1dp_probe_layout(int32_t device, int64_t n, int32_t log2_tile, int64_t stagger_bytes, int32_t reps, int32_t keep,
                        int32_t blocks, int32_t threads, double ms[6]);
int pic1dp_probe_release(void);

/* x / lx by reciprocal + two FMA corrections (div_lx) against the IEEE division on n generated positions (cell
 * boundaries +- a few ulp, wide exponent range): *mismatches counts results that differ in any bit.  On the device,
 * and the same algorithm with the host's fma. */
int pic1dp_probe_div_lx(int32_t device, double lx, int32_t nx, int64_t n, uint64_t seed, int64_t *mismatches);
int pic1dp_probe_host_div_lx(double lx, int32_t nx, int64_t n, uint64_t seed, int64_t *mismatches);
/* a / divisor for a species constant (div_const), likewise */
int pic1dp_probe_div_const(int32_t device, double divisor, int64_t n, uint64_t seed, int64_t *mismatches);
int pic1dp_probe_host_div_const(double divisor, int64_t n, uint64_t seed, int64_t *mismatches);

/* The sequential walks of the GPU marker optimisation (pic1dp_amd/csrc/optimize.hpp plan_merge / plan_remove /
 * plan_split: one key per marker) against the routines on whole markers they restate (opt_merge / opt_remove /
 * opt_split, src/pic1dp_particle.F90:411-746), on the HOST: np generated markers in nalloc slots, the event (kind 0
 * merge, 1 remove, 2 split) applied both ways, *mismatches = slots that differ in v, p, w (and x inside the new count);
 * -1: the marker counts differ, -2: the random stream was consumed differently; *np_after (may be NULL) = the marker
 * count the routine leaves.  No GPU needed. */
int pic1dp_probe_host_optimize(int32_t kind, int32_t typeremove, int32_t nx, int32_t nv, int32_t split_ngroup, double threshold,
                               uint64_t seed, int64_t np, int64_t nalloc, int64_t *mismatches, int64_t *np_after);

/* y[i] = exp(x[i]) as the marker kernels evaluate it (pexp; host arrays) */
int pic1dp_probe_exp(int32_t device, const double *x, double *y, int64_t n);

/* one species of the input (src/pic1dp_input.F90:43-72) */
typedef struct pic1dp_probe_species {
  int32_t iptcldist;
  double charge, mass, temperature, temperature2, density, v0;
} pic1dp_probe_species;
/* which division short cuts and which form of -f0'/f0 the library would take for it, and the folded constants of
 * the one-exp form f = {fq2, fq1, fq0, fm1, fm0, fd1, fd0}: L(v) = (fq2 v + fq1) v + fq0,
 * -f0'/f0 = (fm1 v + fm0) + (fd1 v + fd0) tanh(L / 2) */
int pic1dp_probe_species_const(const pic1dp_probe_species *sp, int32_t *pow2, int32_t *unit, int32_t *fastc, int32_t *one_exp,
                               double f[7]);
/* y[i] = -f0'/f0 at v[i] as the marker kernels evaluate it (src/pic1dp_interaction.F90:274-326): form 0 in the
 * reference's operation order, form 1 the one-exp form (iptcldist 2, 3) */
int pic1dp_probe_dlnf0(int32_t device, const pic1dp_probe_species *sp, int32_t form, const double *v, double *y, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
