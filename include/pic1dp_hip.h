/*
 * pic1dp_hip.h -- C ABI of the MI355X-native PIC1D time-step engine.
 *
 * This is the drop-in boundary for the hot path of wenjundeng/pic1dp
 * (PIC1D-PETSc): particle push + field gather, charge deposition (+ all-reduce)
 * and the mode-filtered spectral field solve.  The reference has no FFI: the
 * boundary there is three argument-less Fortran module procedures working on
 * module-global PETSc Vecs.  Each entry point below
 * names the reference procedure / call site it replaces (paths relative to the
 * reference tree).  The Fortran side binds these through ISO_C_BINDING
 * (pic1dp_amd/fortran/pic1dp_hip_mod.F90).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every function returns int: 0 = success, non-zero = error (the reference's
 *     PetscErrorCode / CHKERRQ convention, src/pic1dp_global.F90:59).
 *     pic1dp_hip_last_error() gives the message of the calling thread's last
 *     failure.
 *   - All state lives in an opaque context created per process/GPU.  Compute
 *     calls enqueue work on the context's HIP stream and return; calls that
 *     hand data to the host (get_*, download, energy, timers) synchronise.
 *   - Called from one host thread per context (the reference is single-threaded
 *     per MPI rank, src/multirand.F90:43-44).
 *   - FP64 everywhere (PetscReal/PetscScalar = real(8)); counts are int64.
 *   - There is NO CPU fallback: without a usable HIP device create() fails.
 */
#ifndef PIC1DP_HIP_H
#define PIC1DP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIC1DP_ABI_VERSION 5
#define PIC1DP_MAX_SPECIES 8
#define PIC1DP_MAX_MODES 4096 /* up to the full spectrum nx/2 of the largest grid */
#define PIC1DP_MAX_INIT_MODES 16
#define PIC1DP_COMM_ID_BYTES 128
#define PIC1DP_XCHG_HANDLE_BYTES 64
#define PIC1DP_MAX_OPT 32

/* error codes */
enum {
  PIC1DP_OK = 0,
  PIC1DP_ERR_ARG = 1,      /* bad argument / unsupported parameter value */
  PIC1DP_ERR_HIP = 2,      /* HIP runtime failure (message has the HIP error) */
  PIC1DP_ERR_NODEVICE = 3, /* no usable gfx950 device: no CPU fallback exists */
  PIC1DP_ERR_STATE = 4,    /* call out of sequence (e.g. step before load) */
  PIC1DP_ERR_COMM = 5,     /* RCCL failure / library not loadable */
  PIC1DP_ERR_RNG = 6,      /* multirand self-test failed or would hang */
  PIC1DP_ERR_NOMEM = 7
};

/* Run-time mirror of the compile-time parameters of src/pic1dp_input.F90:32-256.
 * Field names are the reference names without the "input_" prefix. */
typedef struct pic1dp_input {
  int32_t abi_version;          /* must be PIC1DP_ABI_VERSION */
  int32_t ntime_max;            /* :32 */
  int32_t linear;               /* :43  0 nonlinear, 1 linear */
  int32_t iptcldist;            /* :54  0 Maxwellian 1 two-stream1 2 two-stream2 3 bump-on-tail */
  int32_t nspecies;             /* :57 */
  int32_t nmode;                /* :75 */
  int32_t init_nmode;           /* :87 */
  int32_t deltaf;               /* :106 0 full-f, 1 delta-f */
  int32_t imarker;              /* :122 1 physical, 2 uniform in v */
  int32_t nx;                   /* :128 */
  int32_t nv;                   /* :131 (header of pic1dp.out only) */
  int32_t iptclshape;           /* :138 only 4 (on-the-fly linear shape) is built */
  int32_t nx_opd;               /* :253 */
  int32_t nv_opd;               /* :256 */
  int32_t multirand_al_int;     /* :217 */
  int32_t multirand_seed_type;  /* :223 */
  int32_t multirand_warmup;     /* :226 */
  int32_t multirand_selftest;   /* :233 */
  int64_t nparticle_max;        /* :113 global allocation per species */
  int64_t species_nparticle_init[PIC1DP_MAX_SPECIES]; /* :116 */
  double time_max;              /* :35 */
  double lx;                    /* :46 */
  double dt;                    /* :109 */
  double v_max;                 /* :125 */
  double output_interval;       /* :250 */
  double species_charge[PIC1DP_MAX_SPECIES];       /* :67 */
  double species_mass[PIC1DP_MAX_SPECIES];         /* :68 */
  double species_temperature[PIC1DP_MAX_SPECIES];  /* :69 */
  double species_temperature2[PIC1DP_MAX_SPECIES]; /* :70 */
  double species_density[PIC1DP_MAX_SPECIES];      /* :71 */
  double species_v0[PIC1DP_MAX_SPECIES];           /* :72 */
  int32_t modes[PIC1DP_MAX_MODES];                 /* :79 */
  int32_t init_mode[PIC1DP_MAX_INIT_MODES];        /* :90 */
  double init_mode_cos[PIC1DP_MAX_INIT_MODES];     /* :97 */
  double init_mode_sin[PIC1DP_MAX_INIT_MODES];     /* :98 */
  /* marker optimisation, :141-206 (all counts 0 = disabled, the default) */
  int32_t nmerge;               /* :146 */
  int32_t nremove;              /* :162 */
  int32_t nsplit;               /* :188 */
  int32_t typeremove;           /* :172 1 threshold + fraction, 2 by the |delta f| profile */
  int32_t split_ngroup;         /* :203 */
  int32_t reserved0;
  double remove_frac;           /* :184 */
  double split_dv_sig_frac;     /* :206 */
  double tmerge[PIC1DP_MAX_OPT];      /* :149 ascending */
  double thshmerge[PIC1DP_MAX_OPT];   /* :155 */
  double tremove[PIC1DP_MAX_OPT];     /* :165 */
  double thshremove[PIC1DP_MAX_OPT];  /* :177 */
  double tsplit[PIC1DP_MAX_OPT];      /* :191 */
  double thshsplit[PIC1DP_MAX_OPT];   /* :197 */
} pic1dp_input;

/* Placement of this process in the particle decomposition.
 * The reference splits every particle Vec into `npe` contiguous PETSC_DECIDE
 * blocks (src/pic1dp_particle.F90:89-94,129) and seeds one RNG stream per MPI
 * rank (:159-160).  Here `npe` is the number of REFERENCE ranks being
 * reproduced; this process (rank of nranks, one per GPU) owns the contiguous
 * group of npe/nranks reference blocks starting at rank*npe/nranks, so a
 * 1-GPU run can reproduce an 8-rank reference run ("virtual ranks").
 * npe must be a multiple of nranks; npe = nranks gives the plain mapping. */
typedef struct pic1dp_layout {
  int32_t rank;    /* global_mype of this process, 0-based */
  int32_t nranks;  /* processes = GPUs in the job */
  int32_t npe;     /* reference ranks reproduced (global_npe); 0 -> nranks */
  int32_t device;  /* HIP device ordinal; -1 -> rank % device count */
} pic1dp_layout;

typedef struct pic1dp_ctx pic1dp_ctx; /* opaque */

/* wall-clock timer ids: the reference's ids, src/pic1dp_global.F90:38-50 */
enum {
  PIC1DP_IWT_TOTAL = 1,
  PIC1DP_IWT_INIT = 2,
  PIC1DP_IWT_PARTICLE_LOAD = 3,
  PIC1DP_IWT_PUSH_PARTICLE = 4,
  PIC1DP_IWT_PARTICLE_SHAPE = 5,
  PIC1DP_IWT_COLLECT_CHARGE = 6,
  PIC1DP_IWT_FIELD_ELECTRIC = 7,
  PIC1DP_IWT_PARTICLE_OPTIMIZE = 8,
  PIC1DP_IWT_OUTPUT = 9,
  PIC1DP_IWT_FINAL = 10,
  PIC1DP_IWT_MPIALLREDU = 21,
  PIC1DP_IWT_SCATTER = 22
};

/* ---- library ---------------------------------------------------------- */
int pic1dp_hip_abi_version(void);
const char *pic1dp_hip_last_error(void);
/* 1 when the library is a -DPIC1DP_TUNING build (it then also reads the measurement knobs of
 * tools/README.md from the environment), 0 for the product build, which compiles none of them */
int pic1dp_hip_tuning_build(void);
/* number of visible HIP devices (0 when none; never fails) */
int pic1dp_hip_device_count(void);
/* fill `in` with the values of the reference's input file
 * (src/pic1dp_input.F90), except multirand_seed_type = 1 (reproducible) */
int pic1dp_hip_input_defaults(pic1dp_input *in);

/* sizeof(pic1dp_input) as the library was compiled: lets a binding in another
 * language verify its struct layout before the first call */
int pic1dp_hip_input_size(void);
/* the two validity checks of input_init (src/pic1dp_input.F90:287-308) plus the
 * range checks of this build, without touching a device */
int pic1dp_hip_input_validate(const pic1dp_input *in, const pic1dp_layout *layout);

/* ---- host-only helpers (no device needed) --------------------------------
 * allocated slots / valid markers of reference block `mype` of `npe`
 * (src/pic1dp_particle.F90:129, :240-248) */
int pic1dp_hip_block_sizes(const pic1dp_input *in, int32_t ispecies, int32_t mype,
                           int32_t npe, int64_t *nalloc, int64_t *np);
/* particle_load of ONE reference block into HOST arrays: x, v, p, w each hold
 * nspecies * nalloc doubles, species-major (species s at offset s*nalloc).
 * This is the generator pic1dp_hip_particle_load uses before the upload. */
int pic1dp_hip_host_particle_load(const pic1dp_input *in, int32_t mype, int32_t npe,
                                  double *x, double *v, double *p, double *w,
                                  int64_t nalloc);

/* n raw 64-bit draws (multirand_int64, as two's-complement int64) of the
 * loader's generator after multirand_init(al_int, seed_type, mype, warmup,
 * selftest) -- src/multirand.F90:132-383; lets a host or a test check the
 * stream against the reference's */
int pic1dp_hip_host_multirand_int64(int32_t al_int, int32_t seed_type, int32_t mype,
                                    int32_t warmup, int32_t selftest, int64_t *out,
                                    int64_t n);

/* ---- life cycle -------------------------------------------------------
 * create  <-> input_init + particle_init + field_init
 *             (src/pic1dp.F90:57-59; src/pic1dp_particle.F90:66-139;
 *              src/pic1dp_field.F90:55-212)
 * destroy <-> particle_final + field_final (src/pic1dp.F90:113-114) */
int pic1dp_hip_create(const pic1dp_input *in, const pic1dp_layout *layout,
                      pic1dp_ctx **out);
int pic1dp_hip_destroy(pic1dp_ctx *ctx);

/* local sizes of this process: allocated slots (sum of its reference blocks'
 * PETSC_DECIDE sizes) and valid markers particle_np (src/pic1dp_particle.F90:54,
 * :240-248) of one species */
int pic1dp_hip_local_sizes(pic1dp_ctx *ctx, int32_t ispecies, int64_t *nalloc,
                           int64_t *np);

/* ---- initial condition -------------------------------------------------
 * particle_load (src/pic1dp_particle.F90:145-269) with multirand
 * (src/multirand.F90): native host loader, one RNG stream per owned reference
 * block (seeded with that block's mype), then H2D.  Uses the multirand_*
 * fields of the input. */
int pic1dp_hip_particle_load(pic1dp_ctx *ctx);
/* Members of an ENSEMBLE of otherwise identical runs: reference block b of the next
 * particle_load draws from the stream multirand_init(..., mype = b + offset, ...)
 * instead of mype = b (src/pic1dp_particle.F90:159-160).  offset 0 (default) is the
 * reference's constant-seed run (seed_type 1) the parity tests are about; the
 * reference itself gets its ensembles from seed_type 2 / 3 (clock, /dev/urandom:
 * src/multirand.F90:291-306) and compares their growth rates (tools/runinfo.py:
 * 94-122,136-231) -- this is the reproducible form of that.  Block sizes, weights
 * and everything else are those of the npe-rank load. */
int pic1dp_hip_set_seed_offset(pic1dp_ctx *ctx, int32_t offset);
/* alternative for a host that ran the reference's own particle_load:
 * hand over HOST arrays of one species (n = allocated slots, np = valid) --
 * the VecGetArrayF90 view of particle_x/v/p/w (src/pic1dp_particle.F90:34-36) */
int pic1dp_hip_particles_upload(pic1dp_ctx *ctx, int32_t ispecies,
                                const double *x, const double *v,
                                const double *p, const double *w, int64_t n,
                                int64_t np);
/* copy the current particle_x/v/p/w of one species back to HOST arrays of
 * n = allocated slots (any pointer may be NULL) */
int pic1dp_hip_particles_download(pic1dp_ctx *ctx, int32_t ispecies, double *x,
                                  double *v, double *p, double *w, int64_t n);
/* RK backups particle_x_bak/v_bak/w_bak as the reference would hold them
 * (valid between push(1) and the end of push(2)) */
int pic1dp_hip_particles_download_bak(pic1dp_ctx *ctx, int32_t ispecies,
                                      double *xb, double *vb, double *wb,
                                      int64_t n);

/* ---- the hot path ------------------------------------------------------ */
/* interaction_collect_charge, iptclshape 4 (src/pic1dp_interaction.F90:79-151,
 * call sites src/pic1dp.F90:71,88): periodic wrap of x (stored back), linear
 * deposit, species charge scaling, all-reduce over ranks (RCCL), normalisation
 * into field_chargeden.  The last step (on one rank: the species sum too) runs in
 * the launch of the solve_field that follows, or as soon as anything else asks for
 * field_chargeden or touches the accumulators (one launch less per sub-step: at the
 * reference's default size a launch is 5 % of a step); PIC1DP_LAZY_CALLS=0: at once */
int pic1dp_hip_collect_charge(pic1dp_ctx *ctx);
/* field_solve_electric (src/pic1dp_field.F90:218-270, call sites
 * src/pic1dp.F90:72,89): mode-filtered partial DFT solve of field_chargeden
 * into field_electric, field_mode_re, field_mode_im */
int pic1dp_hip_solve_field(pic1dp_ctx *ctx);
/* interaction_push_particle (src/pic1dp_interaction.F90:161-370, call site
 * src/pic1dp.F90:80); irk = global_irk = 1 or 2.
 * In the reference's sequence push(1), collect_charge, solve_field, push(2),
 * collect_charge, solve_field the push is only noted and the collect_charge
 * that follows runs one whole-step kernel for both; from the second step on the
 * collect_charge after push(1) touches no marker at all -- the previous step's
 * kernel has predicted its charge -- so a step is ONE pass over
 * the markers (56 instead of 184 bytes per marker and step), and on one rank the
 * solve_field after the second collect_charge solves both fields of the step in one
 * launch: two launches per time step.  Whatever looks at the markers in between first gets the
 * ordinary kernels run, so every observable state is the eager one, bit for
 * bit.  PIC1DP_LAZY_CALLS=0 in the environment: one kernel per call, at once. */
int pic1dp_hip_push(pic1dp_ctx *ctx, int32_t irk);
/* particle_optimize (src/pic1dp_particle.F90:724-783; call site src/pic1dp.F90:82,
 * right after the push of sub-step irk): when global_time + dt has reached the
 * next entry of tmerge / tremove / tsplit and irk == 2, runs particle_merge /
 * particle_remove / particle_split (:411-715) on every owned reference block and
 * sets *flag_optimized = 1.  The markers stay on the device: the |delta f|(v)
 * histogram is folded there in the reference's order of additions, one small key per
 * marker (1-8 B) goes to the host, which walks the keys exactly as these sequential
 * routines walk the markers (visiting order, swap-with-last, the block's random
 * stream; the blocks side by side on host threads), and the decisions are applied on
 * the device (PIC1DP_OPT_HOST=1: the pass on host copies of the
 * markers, kept as the cross-check).  Off the timed path and disabled by default.  pic1dp_hip_step calls
 * this itself.  Needs markers loaded by pic1dp_hip_particle_load (the blocks'
 * generators continue from the load), delta-f only like the reference. */
int pic1dp_hip_particle_optimize(pic1dp_ctx *ctx, int32_t irk, int32_t *flag_optimized);
/* one Runge-Kutta sub-step = push(irk); [particle_optimize;] collect_charge;
 * solve_field (src/pic1dp.F90:80-89) with push+wrap+deposit fused into one kernel */
int pic1dp_hip_substep(pic1dp_ctx *ctx, int32_t irk);
/* nsteps time steps: {substep(1); substep(2); itime += 1; time += dt}
 * (src/pic1dp.F90:79-93).  The field energy int E^2 dx after every step is
 * appended to a device-side history (see pic1dp_hip_energy_history). */
int pic1dp_hip_step(pic1dp_ctx *ctx, int32_t nsteps);
/* how pic1dp_hip_step advances a time step (marker pushes are bit-identical given
 * identical fields; only the memory traffic differs):
 *   0 (default) whole-step kernels: the half-step state is recomputed in the
 *     second sub-step instead of being stored and re-read, state updated in place;
 *     from the second step on ONE pass over the markers per step (the second
 *     sub-step's kernel also deposits the next first sub-step's charge as
 *     coefficients of the kept field modes: 56-72 B per marker per step; the
 *     half-step charge then equals a marker-by-marker deposit up to rounding;
 *     PIC1DP_PREDICT=0 in the environment keeps the two passes, 88 B).  The
 *     prediction is held as LDS tiles (two kept modes; nx up to ~1700) or, with
 *     one kept mode (nx up to ~5000), as six
 *     sums over the markers -- then, through the call sites, the collect_charge after push(1)
 *     leaves in field_chargeden the kept mode's content of the half-step charge
 *     density only (all that solve_field looks at; nothing in the reference driver
 *     reads it there).  A host that DOES read it there gets the reference's vector:
 *     see pic1dp_hip_get_field.
 *     Falls back to two passes otherwise, and to mode 1 when nx is too large for
 *     three grid tiles in LDS
 *   1 two fused sub-steps through the RK ping-pong sets (136 B per marker) */
int pic1dp_hip_set_step_mode(pic1dp_ctx *ctx, int32_t mode);
/* how step mode 0 predicts the next first sub-step's charge for this input: 0 not at
 * all (two passes per step), 1 prediction tiles (k_step_one), 2 six sums (k_step_sums) */
int pic1dp_hip_predict_kind(pic1dp_ctx *ctx, int32_t *kind);
/* Diagnostics of output_all taken inside the time step that precedes it.  A step after which the
 * driver will call output_all (the cadence test of src/pic1dp.F90:98-107 evaluated one step ahead
 * from the library's time, see set_time; the last step of a pic1dp_hip_step call, or the
 * collect_charge that follows push(2)) can take the histograms of output_ptcldist and the kinetic
 * sums of output_field inside its marker kernel, on the state it has just computed
 * (k_step_full<DIAG>): pic1dp_hip_output_scalars / pic1dp_hip_ptcldist then cost no pass over the
 * markers.  Results equal the separate pass up to the summation order of the atomics.
 * Cost at 1e8 markers: inside a two-pass step +0.17 ms an output (the histograms as 64-bit
 * fixed-point sums in the LDS; +0.48 as double sums, +0.6 as a pass of its own).
 *   on = 0 (default): never -- a host that never asks for output would pay ~20 % on that launch;
 *   on = 1: where it pays.  Not on a predicted one-pass step (pic1dp_hip_predict_kind != 0):
 *           k_step_full<DIAG> cannot predict the next step's half-step charge, so the step after
 *           the output would run a first-sub-step pass again; there the step stays k_step_one and
 *           the diagnostics take their own pass (32 B per marker) when output_all asks for them;
 *   on = 2: always. */
int pic1dp_hip_set_output_fusion(pic1dp_ctx *ctx, int32_t on);
/* which field solve pic1dp_hip_solve_field / substep / step perform:
 *   0 (default) the reference's field_solve_electric: mode-filtered partial DFT
 *     (src/pic1dp_field.F90:218-270) -- the only solver with reference parity
 *   1 opt-in alternative, NOT in the reference: second-order finite differences
 *     keeping all modes, -phi'' = rho - <rho> as a tridiagonal system solved by
 *     parallel cyclic reduction on the GPU, E = -dphi/dx by central differences
 *     (3 <= nx <= 4096).  field_mode_re/im still hold the kept modes of the
 *     mode-filter solve. */
int pic1dp_hip_set_field_solver(pic1dp_ctx *ctx, int32_t kind);
/* field_electric as it was between the two sub-steps of the last time step
 * taken by pic1dp_hip_step ([nx]) */
int pic1dp_hip_get_field_half(pic1dp_ctx *ctx, double *electric_half);
/* wait for everything enqueued on the context's stream */
int pic1dp_hip_sync(pic1dp_ctx *ctx);

/* global_itime / global_time (src/pic1dp_global.F90:62-63) */
int pic1dp_hip_get_time(pic1dp_ctx *ctx, int32_t *itime, double *time);
int pic1dp_hip_set_time(pic1dp_ctx *ctx, int32_t itime, double time);
/* check_termination (src/pic1dp.F90:133-148): *flag = 1 to terminate */
int pic1dp_hip_check_termination(pic1dp_ctx *ctx, int32_t *flag);
/* output cadence test of src/pic1dp.F90:98-106 evaluated at the current time */
int pic1dp_hip_output_due(pic1dp_ctx *ctx, int32_t itermination, int32_t *flag);
/* how many iterations of the driver loop (src/pic1dp.F90:78-109) lie between the current counters and the next
 * output_all -- the cadence test of :98-106 and check_termination (:133-148) evaluated ahead, step by step, with
 * the very additions the loop makes (time = time + dt).  A host that hands the whole irk loop to
 * pic1dp_hip_step can ask for exactly that many steps in one call and loses no record: *nsteps >= 1 while the run
 * has not terminated, 0 once it has. */
int pic1dp_hip_steps_to_output(pic1dp_ctx *ctx, int32_t *nsteps);

/* ---- field access (what output_field reads, src/pic1dp_output.F90:173-186) */
/* any pointer may be NULL; E, chargeden: [nx]; mode_re, mode_im: [nmode].
 * HALF-STEP CHARGE DENSITY -- the one place where the state a host can look at differs
 * from what the reference would hold: between the collect_charge after push(1) and the
 * next collect_charge, when that half-step charge was predicted as six sums
 * (pic1dp_hip_predict_kind = 2: one kept mode), field_chargeden
 * holds its kept mode's content only.  Asking for chargeden here rebuilds the whole
 * vector on a one-rank context: the half-step state is pushed into memory after all and
 * deposited (that step then runs as two ordinary sub-steps -- a noted push(2) is run at
 * once as well; results unchanged to rounding).  On several ranks the rebuild would need
 * the charge sum -- a collective an inspection on one rank must not start -- so there
 * chargeden keeps the kept mode's content between the sub-steps; a host that needs the
 * full half-step vector on several ranks calls pic1dp_hip_set_step_mode(ctx, 1) or sets
 * PIC1DP_PREDICT=0.  Whether the vector handed out is the reference's or the kept mode's
 * content is never left to guessing: pic1dp_hip_chargeden_state says which.  Pass
 * chargeden = NULL to leave it alone. */
int pic1dp_hip_get_field(pic1dp_ctx *ctx, double *electric, double *chargeden,
                         double *mode_re, double *mode_im);
/* *kept_mode_only = 1: field_chargeden (as pic1dp_hip_get_field hands it out now) holds
 * only the kept mode's content of the half-step charge density -- the rebuild described
 * above was not possible (several ranks; or the markers have been moved on since by
 * calls outside the push / collect_charge / solve_field sequence); 0: it is the vector
 * the reference holds (src/pic1dp_interaction.F90:138-150). */
int pic1dp_hip_chargeden_state(pic1dp_ctx *ctx, int32_t *kept_mode_only);
/* overwrite field_electric (testing the push against a prescribed field) */
int pic1dp_hip_set_electric(pic1dp_ctx *ctx, const double *electric);
/* overwrite field_chargeden (testing the solve; field_test of
 * src/pic1dp_field.F90:276-309) */
int pic1dp_hip_set_chargeden(pic1dp_ctx *ctx, const double *chargeden);
/* int E^2 dx = ||E||_2^2 * lx / nx (src/pic1dp_output.F90:120-124) */
int pic1dp_hip_field_energy(pic1dp_ctx *ctx, double *energy);
/* energies recorded by step(): copies min(count, max) values, oldest first */
int pic1dp_hip_energy_history(pic1dp_ctx *ctx, double *energy, int64_t max,
                              int64_t *count);
int pic1dp_hip_energy_history_reset(pic1dp_ctx *ctx);
/* per-species sums of output_field (src/pic1dp_output.F90:126-172), local to
 * this process: out[0]=sum v^2, out[1]=sum v^2 p, out[2]=sum v^2 w */
int pic1dp_hip_energy_sums(pic1dp_ctx *ctx, int32_t ispecies, double out[3]);
/* cell index ix of every valid marker of one species and the per-cell marker
 * counts, as the deposit computes them (src/pic1dp_interaction.F90:106-107);
 * ix: [np] int32, count: [nx] int64, either may be NULL */
int pic1dp_hip_cell_indices(pic1dp_ctx *ctx, int32_t ispecies, int32_t *ix,
                            int64_t *count);

/* ---- diagnostics of output_all (called every output_interval, not timed) --
 * realbuf of output_field (src/pic1dp_output.F90:117-175): out[0] = time,
 * out[1] = int E^2 dx, then per species s: out[2+3s] = sum v^2,
 * out[3+3s] = total kinetic sum, out[4+3s] = perturbed kinetic sum (the linear /
 * full-f adjustments of :152-170 applied).  n must be 2 + 3*nspecies.  Sums are
 * all-reduced over ranks when a communicator exists.
 * energy_sums, output_scalars and ptcldist share one pass over a species'
 * markers (histograms and kinetic sums together); its results are kept until a
 * call changes the markers, so the sequence of output_all costs one pass. */
int pic1dp_hip_output_scalars(pic1dp_ctx *ctx, double *out, int32_t n);
/* (x,v) and v distributions of output_ptcldist (src/pic1dp_output.F90:196-477)
 * of one species, computed on the GPU: markr/total/pertb_xv hold
 * nx_opd*nv_opd doubles (index iv*nx_opd+ix), markr/total/pertb_v hold nv_opd.
 * finish = 0: this rank's raw sums (:239-315).
 * finish = 1: what the reference writes: summed over ranks (when a communicator
 *             exists), linear total += pertb (:328-331), scaled by the grid
 *             sizes (:361-369) and, for full-f, pertb = total - f0 (:370-453). */
int pic1dp_hip_ptcldist(pic1dp_ctx *ctx, int32_t ispecies, int32_t finish,
                        double *markr_xv, double *total_xv, double *pertb_xv,
                        double *markr_v, double *total_v, double *pertb_v);

/* output_all in ONE call (src/pic1dp_output.F90:100-189, 196-477; call sites src/pic1dp.F90:74,107): what
 * pic1dp_hip_output_scalars, pic1dp_hip_get_field and pic1dp_hip_ptcldist(finish = 1) of every species hand out,
 * with the diagnostics passes and every transfer enqueued together and ONE wait (the separate calls wait
 * three times and more per record).  scalars: [2 + 3 nspecies]; electric, chargeden: [nx]; mode_re, mode_im:
 * [nmode]; dist: [nspecies][3 nx_opd nv_opd + 3 nv_opd] = markr_xv | total_xv | pertb_xv | markr_v | total_v |
 * pertb_v of each species in turn.  Any pointer but scalars may be NULL.  On several ranks the call composes the
 * separate ones (RCCL communicator: they reduce; no communicator: an error, as pic1dp_hip_ptcldist). */
int pic1dp_hip_output_all(pic1dp_ctx *ctx, double *scalars, int32_t nscalars, double *electric, double *chargeden,
                          double *mode_re, double *mode_im, double *dist);
/* ---- split-phase diagnostics for a host that owns the reductions (MPI) ----
 * output_all sums its diagnostics over the ranks (VecSum / MPI_Reduce, src/pic1dp_output.F90:126-151,333-356).
 * With an RCCL communicator output_scalars and ptcldist(finish = 1) do that themselves; a host with its own
 * MPI takes the local sums -- pic1dp_hip_energy_sums per species, pic1dp_hip_ptcldist(finish = 0) --, reduces
 * them to its rank 0 and hands the sums back:
 *   output_scalars_from: sums[3*nspecies] = sum v^2, sum v^2 p, sum v^2 w per species, summed over ranks ->
 *     out[2 + 3*nspecies] as pic1dp_hip_output_scalars writes it (time, int E^2 dx, three energies per
 *     species, :152-170)
 *   ptcldist_finish: the six raw histograms summed over ranks -> what the reference writes (:328-331,
 *     :361-453), in place */
int pic1dp_hip_output_scalars_from(pic1dp_ctx *ctx, const double *sums, double *out, int32_t n);
int pic1dp_hip_ptcldist_finish(pic1dp_ctx *ctx, int32_t ispecies, double *markr_xv, double *total_xv,
                               double *pertb_xv, double *markr_v, double *total_v, double *pertb_v);

/* ---- split-phase deposit for a host that owns the reduction (MPI) ------
 * charge_local: everything of collect_charge up to the all-reduce
 *   (src/pic1dp_interaction.F90:81-128) -> this rank's charge2[nx] on the host
 * charge_reduced: hand back the globally summed array; finishes :138-150
 * The array is to be summed element by element over the ranks and handed back whole:
 * after a noted push(1) whose charge the previous step has predicted as six sums
 * (pic1dp_hip_predict_kind = 2) it carries those sums in its first six elements and
 * zeros behind, not a charge vector. */
int pic1dp_hip_charge_local(pic1dp_ctx *ctx, double *charge2);
int pic1dp_hip_charge_reduced(pic1dp_ctx *ctx, const double *charge1);

/* ---- multi-GPU: RCCL communicator (replaces MPI_Allreduce at
 * src/pic1dp_interaction.F90:132) ------------------------------------------
 * rank 0 obtains an id, the host distributes the 128 bytes to every rank
 * (MPI_Bcast / torch.distributed), every rank calls comm_init.  A context with
 * nranks = 1 needs no communicator; if comm_init is called on it anyway a
 * 1-rank communicator is created and the all-reduce path is taken (used to
 * exercise that path on a single GPU). */
int pic1dp_hip_comm_unique_id(unsigned char id[PIC1DP_COMM_ID_BYTES]);
int pic1dp_hip_comm_init(pic1dp_ctx *ctx,
                         const unsigned char id[PIC1DP_COMM_ID_BYTES]);

/* 0 when librccl can be loaded in this process (no device touched): lets every
 * rank of a job agree on RCCL before any of them enters ncclCommInitRank */
int pic1dp_hip_comm_available(void);

/* ---- multi-GPU alternative: one-hop charge exchange (also replaces MPI_Allreduce
 * at src/pic1dp_interaction.F90:132) ------------------------------
 * Every rank keeps one slot per source rank; after its deposit a rank stores its
 * charge vector into its slot on every GPU of the node (peer-mapped memory, one
 * xGMI hop), then every GPU adds the slots in rank order inside the field
 * solve's launch: one launch per sub-step instead of three, and the summed charge
 * -- hence E and the trajectories -- are bit-identical on all ranks and from run to
 * run.  Set-up: every rank calls xchg_create and hands its 64-byte handle to all
 * ranks (MPI_Allgather / torch.distributed.all_gather); every rank then calls
 * xchg_connect with the nranks handles in rank order (handles[rank] is its own),
 * and set_allreduce(2) on all ranks at the same point of the run.  Ranks may be
 * separate processes (the areas are mapped through hipIpc) or contexts of ONE
 * process, each driven by its own host thread (a handle made in this process is
 * recognised and its area addressed directly, with peer access enabled when it
 * lies on another device): the reference's `mpiexec -n N` either way.  A rank that
 * waits longer than PIC1DP_XCHG_TIMEOUT_MS (default 20 000) for a peer gives up:
 * the kernels always finish and the next synchronising call returns
 * PIC1DP_ERR_COMM. */
int pic1dp_hip_xchg_create(pic1dp_ctx *ctx, unsigned char handle[PIC1DP_XCHG_HANDLE_BYTES]);
int pic1dp_hip_xchg_connect(pic1dp_ctx *ctx, const unsigned char *handles);
/* which reduction collect_charge / substep / step use from now on:
 * 0 auto (RCCL when a communicator exists), 1 RCCL, 2 the one-hop exchange */
int pic1dp_hip_set_allreduce(pic1dp_ctx *ctx, int32_t kind);
/* memory kind of the exchange area (1 fine-grained, 2 uncached, 3 plain device
 * memory), exchanges performed so far; returns PIC1DP_ERR_COMM after a time-out */
int pic1dp_hip_xchg_info(pic1dp_ctx *ctx, int32_t *memkind, int64_t *exchanges);
/* device time spent INSIDE the exchanges (the stores into the peers' slots, the wait for
 * their flags, the rank-order sum) of the launches enqueued while the timers were on
 * (pic1dp_hip_timers_enable), from a 100 MHz wall clock read in the kernel: with the
 * exchange being the prologue of the field solve's launch, this is what splits the
 * reference's "mpiallredu" share (src/pic1dp_global.F90:38-50, timer 21) from "field
 * electric" (7) in that launch.  reset != 0: start over.  Synchronises the stream. */
int pic1dp_hip_xchg_time(pic1dp_ctx *ctx, double *ms, int64_t *exchanges_timed, int32_t reset);

/* ---- timers: accumulated milliseconds under the reference's timer ids
 * (src/pic1dp_global.F90:38-50), measured with HIP events on the stream.
 * timers_enable(on): 0 off; 1 every launch is bracketed by an event pair (exact; each
 * pair costs the stream ~3 us of dependency -- 10-28 % of a 70 us time step at the
 * reference's default size); n >= 2: launches under a timer id are bracketed in blocks of
 * 64 consecutive ones, every n-th block (inside a block as with on = 1, so that a run's
 * output steps and plain steps enter in their own proportions); the others are counted, and
 * pic1dp_hip_timer_ms reports the timed ones scaled by launches seen / launches timed
 * (what the Fortran host asks for: n = 17: a sixteenth of the cost, the time loop of the
 * default run 0.91 s against 1.10-1.17 with on = 1 and 0.90-0.93 with the timers off). ---- */
int pic1dp_hip_timers_enable(pic1dp_ctx *ctx, int32_t on);
int pic1dp_hip_timer_ms(pic1dp_ctx *ctx, int32_t iwt, double *ms);
int pic1dp_hip_timers_reset(pic1dp_ctx *ctx);

/* ---- tuning knobs (performance only; results do not depend on them apart
 * from floating-point summation order of the charge) ----------------------- */
/* threads per workgroup (multiple of 64, <= 1024; 0 = auto) and workgroups per
 * CU (0 = auto) of the particle kernels */
int pic1dp_hip_set_launch(pic1dp_ctx *ctx, int32_t threads, int32_t blocks_per_cu);
/* opaque HIP stream handle (hipStream_t) the context enqueues on */
int pic1dp_hip_get_stream(pic1dp_ctx *ctx, void **stream);
/* name and launch counters of the particle kernels, for bench.py's roofline:
 * accumulated device milliseconds (HIP events on the context's stream) and
 * launch count of the fused push+deposit kernel (which=0), the separate push
 * kernel (1), the separate deposit kernel (2), and the whole-step kernels:
 * first sub-step k_step_half (3), second sub-step k_step_full (4); which = 5: number
 * of separate diagnostics passes (k_ptcldist) launched so far, *ms = 0; which = 6: the
 * one-pass-per-step kernel k_step_one (second sub-step + prediction of the next first
 * sub-step's charge); which = 7: number of k_step_one / k_step_sums launches so far whose
 * prologue solved the field of the previous step (one launch per time step inside
 * pic1dp_hip_step: no field_solve_electric launch in between), *ms = 0; which = 8:
 * *launches = bytes marker optimisation events (pic1dp_hip_particle_optimize) have moved
 * between host and device so far, *ms = 0; which = 9: *launches = how the serial forward
 * sums of the field solve (one-rank order, up to eight kept modes) run: 0 chains of
 * additions in single lanes, 1 through the FP64 matrix unit (only where pic1dp_hip_create
 * found it to reproduce the sequential sums bit for bit; PIC1DP_CHAIN_MFMA=0 keeps the chains),
 * *ms = that self-test's verdict (1 identical, 0 differs, -1 it could not run: the chains then);
 * which = 10: *launches = marker launches so far whose last workgroup packed this rank's charge
 * for the sum over ranks or posted it into the peers' exchange slots itself (several ranks: no
 * separate packing launch in front of the all-reduce; PIC1DP_TAIL=0 keeps that launch), *ms = 0;
 * which = 11: *launches = pic1dp_hip_solve_field calls of a first sub-step that launched nothing
 * because the solve_field before them had solved both fields in one launch (one rank, the three
 * call sites: two launches per time step; PIC1DP_CALL_PAIR=0: three), *ms = 0;
 * which = 12: *launches = diagnostics passes (k_ptcldist) whose (x, v) histograms were summed as
 * 64-bit fixed-point numbers in the LDS (1.9x the atomic rate of double sums; scaled by the
 * species' max |p|, max |w| of the pass before; PIC1DP_DIAG_FX=0: always doubles), *ms = how many
 * of them met a marker beyond those bounds and were repeated with double sums;
 * which = 13: *launches = terms of the one-pass kernel's prediction tiles (two and three kept modes: 64-bit
 * fixed-point LDS sums scaled by per-species bounds that follow the markers) that lay beyond 16x their bound and
 * were added in doubles straight into the global accumulators -- rare by design, a count that grows with every
 * step says the bounds have lost the population; *ms = the first species' bound on |q| as it stands (waits for the stream) */
int pic1dp_hip_kernel_stats(pic1dp_ctx *ctx, int32_t which, double *ms,
                            int64_t *launches);
int pic1dp_hip_kernel_stats_enable(pic1dp_ctx *ctx, int32_t on);
/* what the marker kernel launched last under `which` (numbering of kernel_stats; not 5)
 * moves per marker and launch, in bytes: read_bytes (x, v, p (+ w); the RK base as well
 * for the second sub-step's k_push) and written_bytes (x (+ v) (+ w)) are compulsory for
 * its data flow; carry_bytes is the traffic the kernel CHOOSES to spend on handing
 * -f0'/f0 to the next launch instead of evaluating it again (k_step_one: 8 read + 8
 * written; 0 when it recomputes).  name: the kernel (k_step_one / k_step_sums / ...) and
 * the form of -f0'/f0 it was instantiated with.  bench.py prices its roofline on these. */
int pic1dp_hip_kernel_bytes(pic1dp_ctx *ctx, int32_t which, double *read_bytes,
                            double *written_bytes, double *carry_bytes, char *name,
                            int32_t name_len);

/* Debugging aid: checks the relations between the flags of the library's internal state machine (lazy call sites,
 * prediction, call-site pair, accumulator sets: DESIGN.md 0) that have to hold between any two calls, whatever the
 * calls were.  deep != 0 also synchronises and looks at device memory (accumulator sets nobody owes anything to are
 * zero).  PIC1DP_ERR_STATE + pic1dp_hip_last_error() name the relation that does not hold.  Replaces nothing in the
 * reference; the randomised call-sequence tests call it after every call. */
int pic1dp_hip_check_state(pic1dp_ctx *ctx, int32_t deep);

#ifdef __cplusplus
}
#endif
#endif /* PIC1DP_HIP_H */
