/*
 * pic1dp_oracle.h -- CPU restatement of the PIC1D-PETSc time-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pic1dp_amd/ (the product) may
 * include, link, import or execute anything in oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the timed CPU baseline -- never as the thing shipped.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - multirand (RNG) part: PINNED against the reference module itself
 *     (src/multirand.F90 compiled from /root/reference with flang by
 *     oracle/Makefile into oracle/_ref/) and against the reference's own KAT
 *     vectors (src/multirand.F90:396-425).
 *   - load / push / deposit / field-solve part: PARITY UNPINNED by reference
 *     outputs.  The reference cannot be built here (every hot-path file
 *     includes PETSc's finclude headers, PETSc is absent) and holds no golden
 *     vectors for these routines.  It is a restatement of
 *     the cited lines, cross-checked by the analytic field-solve identity and
 *     by linear growth rates against the Vlasov dispersion roots.
 *
 * Arithmetic contract: plain IEEE double, no FMA contraction, no fast-math,
 * libm exp/sin/cos/fmod/floor/sqrt -- the reference is built "-O3" for generic
 * x86-64 (reference Makefile:26).  Compile with -ffp-contract=off.
 */
#ifndef PIC1DP_ORACLE_H
#define PIC1DP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SPECIES 8
#define ORC_MAX_MODES 4096
#define ORC_MAX_INIT_MODES 16
#define ORC_MULTIRAND_NSEED 20635
#define ORC_MAX_OPT 32

/* mirrors the compile-time parameters of src/pic1dp_input.F90:32-256 */
typedef struct orc_input {
  int32_t ntime_max;
  int32_t linear;
  int32_t iptcldist;
  int32_t nspecies;
  int32_t nmode;
  int32_t init_nmode;
  int32_t deltaf;
  int32_t imarker;
  int32_t nx;
  int32_t nv;
  int32_t iptclshape;
  int32_t nx_opd;
  int32_t nv_opd;
  int32_t multirand_al_int;
  int32_t multirand_seed_type;
  int32_t multirand_warmup;
  int32_t multirand_selftest;
  int32_t pad0;
  int64_t nparticle_max;
  int64_t species_nparticle_init[ORC_MAX_SPECIES];
  double time_max;
  double lx;
  double dt;
  double v_max;
  double output_interval;
  double species_charge[ORC_MAX_SPECIES];
  double species_mass[ORC_MAX_SPECIES];
  double species_temperature[ORC_MAX_SPECIES];
  double species_temperature2[ORC_MAX_SPECIES];
  double species_density[ORC_MAX_SPECIES];
  double species_v0[ORC_MAX_SPECIES];
  int32_t modes[ORC_MAX_MODES];
  int32_t init_mode[ORC_MAX_INIT_MODES];
  double init_mode_cos[ORC_MAX_INIT_MODES];
  double init_mode_sin[ORC_MAX_INIT_MODES];
  /* marker optimisation, src/pic1dp_input.F90:141-206 */
  int32_t nmerge, nremove, nsplit, typeremove, split_ngroup, pad1;
  double remove_frac, split_dv_sig_frac;
  double tmerge[ORC_MAX_OPT], thshmerge[ORC_MAX_OPT];
  double tremove[ORC_MAX_OPT], thshremove[ORC_MAX_OPT];
  double tsplit[ORC_MAX_OPT], thshsplit[ORC_MAX_OPT];
} orc_input;

/* ---- multirand (src/multirand.F90) ---- */
typedef struct orc_multirand {
  uint64_t seeds[ORC_MULTIRAND_NSEED];
  int32_t iseed;
  int32_t al_int;
  int32_t gaussian64buf_filled;
  int32_t pad;
  double gaussian64buf;
} orc_multirand;

orc_multirand *orc_multirand_new(void);
void orc_multirand_free(orc_multirand *g);
/* returns 0 ok; 1 = selftest KAT mismatch; 2 = the reference would hang
 * (al_int=3, seed_type 1|2, selftest off: src/multirand.F90:346-348) */
int orc_multirand_init(orc_multirand *g, int al_int, int seed_type, int mype,
                       int warmup, int selftest);
int orc_multirand_selftest(orc_multirand *g, int al_int);
void orc_multirand_default_seeds(orc_multirand *g, int al_int);
int64_t orc_multirand_int64(orc_multirand *g);
double orc_multirand_real64(orc_multirand *g);
void orc_multirand_int_array64(orc_multirand *g, int64_t *a, int64_t n);
void orc_multirand_real_array64(orc_multirand *g, double *a, int64_t n);
void orc_multirand_gaussian_array64(orc_multirand *g, double *a, int64_t n);

/* ---- ownership (PETSC_DECIDE split, src/pic1dp_particle.F90:89-94,129) ---- */
int64_t orc_local_size(int64_t n, int rank, int size);
int64_t orc_particle_np(const orc_input *in, int isp, int mype, int npe);

/* ---- particle_load (src/pic1dp_particle.F90:145-269), one species, one rank.
 * Arrays have orc_local_size(nparticle_max, mype, npe) entries.  The caller
 * initialises g once per rank (multirand_init) and calls species in order. */
void orc_particle_load_species(const orc_input *in, int isp, orc_multirand *g,
                               int64_t nlocal, double *x, double *v, double *p,
                               double *w);

/* ---- interaction_collect_charge, mode 4 (src/pic1dp_interaction.F90:79-151) */
/* per-species local loop :96-114 (wraps x in place, accumulates charge1) */
void orc_deposit_species(const orc_input *in, int64_t np, double *x,
                         const double *q, double *charge1);
/* same, also returning the cell index of every marker and per-cell counts */
void orc_deposit_species_idx(const orc_input *in, int64_t np, double *x,
                             const double *q, double *charge1, int32_t *ix_out,
                             int64_t *count);
/* y[i] = exp(x[i]) with libm (the weight equation's transcendental) */
void orc_exp_array(const double *x, double *y, int64_t n);
/* :138-148: chargeden = charge1*nx/lx (- Z n0 for full-f) */
void orc_chargeden_from_charge(const orc_input *in, const double *charge1,
                               double *chargeden);

/* ---- interaction_push_particle loop (src/pic1dp_interaction.F90:238-339) --
 * caller performs the irk==1 backup copy (:178-189) via orc_push_backup */
void orc_push_backup(int64_t nalloc, const double *x, const double *v,
                     const double *w, double *xb, double *vb, double *wb,
                     int deltaf);
void orc_push_species(const orc_input *in, int isp, int irk, const double *E,
                      int64_t np, double *x, double *v, const double *p,
                      double *w, const double *xb, const double *vb,
                      const double *wb);

/* ---- field (src/pic1dp_field.F90:158-210 operators, :218-257 solve) ---- */
typedef struct orc_field {
  int32_t nx, nmode;
  double *fourier_re; /* [nx][nmode]  cos */
  double *fourier_im; /* [nx][nmode] -sin */
  double *grad_inv;   /* [nmode] */
} orc_field;
orc_field *orc_field_new(const orc_input *in);
void orc_field_free(orc_field *f);
void orc_field_solve(const orc_input *in, const orc_field *f,
                     const double *chargeden, double *E, double *mode_re,
                     double *mode_im);
/* the same with the forward sums in the order of an npe-rank run (MPI-AIJ:
 * every rank's row block summed from zero, the blocks added owner first, then in
 * rank order); npe = 1 is orc_field_solve */
void orc_field_solve_ranks(const orc_input *in, const orc_field *f, int npe,
                           const double *chargeden, double *E, double *mode_re,
                           double *mode_im);
/* the engine's opt-in finite-difference solver (not in the reference) */
void orc_field_solve_fd(const orc_input *in, const double *rho, double *E);
/* int E^2 dx as output_field does (src/pic1dp_output.F90:120-124) */
double orc_field_energy(const orc_input *in, const double *E);

/* ---- diagnostics of output_field / output_ptcldist ---- */
void orc_energy_sums(int64_t nalloc, const double *v, const double *p,
                     const double *w, int deltaf, double out[3]);
void orc_ptcldist(const orc_input *in, int64_t np, const double *x,
                  const double *v, const double *p, const double *w,
                  double *markr_xv, double *total_xv, double *pertb_xv,
                  double *markr_v, double *total_v, double *pertb_v);

/* ---- marker optimisation (src/pic1dp_particle.F90:356-813), one rank ---- */
/* :356-403 local |delta f|(v) histogram, added into hist[nv] */
void orc_dist_pertb_abs_v(const orc_input *in, int64_t np, const double *v,
                          const double *w, double *hist);
/* :411-519, :531-602, :610-715; hist = the rank-summed histogram; np in/out */
void orc_particle_merge(const orc_input *in, double thsh, const double *hist,
                        int64_t *np, double *x, double *v, double *p, double *w);
void orc_particle_remove(const orc_input *in, double thsh, const double *hist,
                         orc_multirand *g, int64_t *np, double *x, double *v,
                         double *p, double *w);
void orc_particle_split(const orc_input *in, double thsh, const double *hist,
                        orc_multirand *g, int64_t nalloc, int64_t *np, double *x,
                        double *v, double *p, double *w);

/* root-rank post-processing of output_ptcldist (src/pic1dp_output.F90:328-454) */
void orc_ptcldist_finish(const orc_input *in, int isp, double *markr_xv,
                         double *total_xv, double *pertb_xv, double *markr_v,
                         double *total_v, double *pertb_v);

/* ---- whole simulation with npe virtual reference ranks (driver of
 * src/pic1dp.F90:64-109).  nthreads>1 runs the rank blocks on OpenMP threads
 * (private charge per rank, summed in rank order = the npe-rank reference). */
typedef struct orc_sim orc_sim;
orc_sim *orc_sim_new(const orc_input *in, int npe);
void orc_sim_free(orc_sim *s);
int orc_sim_load(orc_sim *s);               /* particle_load on every rank */
void orc_sim_set_threads(orc_sim *s, int nthreads);
void orc_sim_collect_charge(orc_sim *s);
void orc_sim_solve_field(orc_sim *s);
void orc_sim_push(orc_sim *s, int irk);
void orc_sim_step(orc_sim *s, int nsteps);  /* nsteps x (irk=1,2) */
int32_t orc_sim_itime(const orc_sim *s);
double orc_sim_time(const orc_sim *s);
double orc_sim_field_energy(const orc_sim *s);
void orc_sim_get_field(const orc_sim *s, double *E, double *rho,
                       double *mode_re, double *mode_im);
void orc_sim_set_field(orc_sim *s, const double *E);
int64_t orc_sim_rank_np(const orc_sim *s, int rank, int isp);
int64_t orc_sim_rank_nalloc(const orc_sim *s, int rank);
/* pointers into rank-owned arrays: which = 0..6 -> x v p w xb vb wb */
double *orc_sim_array(orc_sim *s, int rank, int isp, int which);
void orc_sim_energy_sums(const orc_sim *s, int isp, double out[3]);
/* particle_optimize (:724-783) on every rank; returns 1 when something ran.
 * orc_sim_step calls it after each push like the driver (src/pic1dp.F90:82). */
int orc_sim_optimize(orc_sim *s, int irk);
/* the rank's generator as particle_load left it (continues in remove / split) */
orc_multirand *orc_sim_rank_rng(orc_sim *s, int rank);
void orc_sim_set_rank_np(orc_sim *s, int rank, int isp, int64_t np);
void orc_sim_ptcldist(const orc_sim *s, int isp, int finish, double *markr_xv,
                      double *total_xv, double *pertb_xv, double *markr_v,
                      double *total_v, double *pertb_v);
void orc_sim_output_scalars(const orc_sim *s, double *out);
/* termination / output cadence of src/pic1dp.F90:98-108,133-148 */
int orc_check_termination(const orc_input *in, int32_t itime, double time);
int orc_output_due(const orc_input *in, double time, int itermination);

#ifdef __cplusplus
}
#endif
#endif
