// kernels.hpp -- launch interface of the gfx950 kernels (kernels_step.hip, kernels_push.hip, kernels_field.hip, kernels_diag.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pic1dp_hip.h"

#include <cstdint>
#include <cstdlib>

// Environment variables.  The product library reads the fifteen of INTEGRATION.md section 6 with std::getenv: switches
// between paths that exist for a functional reason (eager / lazy call sites, two passes / one, tiles / sums, the solve in a
// launch of its own, ...), which the differential tests hold against each other, and operational settings.  Everything a
// MEASUREMENT wants to vary beyond them (launch shapes, thresholds, schedules; tools/ab_*.sh) goes through tuning_env(),
// which only a -DPIC1DP_TUNING build looks at -- `PIC1DP_EXTRA_FLAGS=-DPIC1DP_TUNING PIC1DP_LIB_OUT=.../v_tuning.so python
// pic1dp_amd/build.py --force`, loaded with PIC1DP_LIB; the default build compiles none of it.
#ifdef PIC1DP_TUNING
inline const char *tuning_env(const char *name) { return std::getenv(name); }
#else
inline const char *tuning_env(const char *) { return nullptr; }
#endif
#if defined(PIC1DP_TUNE_STAMPS) && !defined(PIC1DP_TUNING)
#error "-DPIC1DP_TUNE_STAMPS instruments the marker kernels: it belongs to a tuning build (-DPIC1DP_TUNING), never to the product"
#endif

namespace pic1dp {

// Constants of one species' push, pre-formed on the host exactly as the
// reference's compile-time folding would form them (src/pic1dp_input.F90
// parameters; expressions at src/pic1dp_interaction.F90:274-337).
struct SpeciesConst {
  double Z, m;          // charge, mass
  double den, beam;     // density, 1 - density
  double v0, T;
  double tm, tm2;       // T/m, T2/m
  double two_tm, two_tm2;  // 2*T/m, 2*T2/m
  double stm, stm2;     // sqrt(T/m), sqrt(T2/m)
  // reciprocals RN(1/c): exact products when every divisor is a power of two, else div_const
  double r_m, r_T, r_tm, r_tm2, r_two_tm, r_two_tm2, r_stm, r_stm2;
  int pow2;             // 1: all divisors above are powers of two
  int unit;             // 1: m = T = T2 = 1 (T/m = sqrt(T/m) = 1, 2T/m = 2): divisions vanish
  int fastc;            // 1: a/c by reciprocal + two FMA corrections for all eight divisors (host-verified)
  // one-exp form of -f0'/f0 (device_math.hpp dlnf0_one_exp; iptcldist 2 and 3): L(v) = (fq2 v + fq1) v + fq0 is
  // the log of the ratio of the two Maxwellians, tmp2 = (fm1 v + fm0) + (fd1 v + fd0) tanh(L / 2)
  int one_exp;          // 1: the marker kernels use it (default; PIC1DP_DLNF0=ref: the reference's operation order)
  double fq2, fq1, fq0, fm1, fm0, fd1, fd0;
};
// one species of the input file (src/pic1dp_input.F90:43-72) -> the constants above (host; species.cpp).
// PIC1DP_UNIT_SPECIALISATION, PIC1DP_FAST_DIVC, PIC1DP_DLNF0 (tuning / cross-check) are read here.
struct SpeciesInput {
  int iptcldist;
  double charge, mass, temperature, temperature2, density, v0;
};
SpeciesConst make_species_const(const SpeciesInput &in, int species_index);

struct GridConst {
  double lx, dnx, dt_full;
  double rlx;    // RN(1/lx), for the exact division by the constant lx
  int nx;
  int gcopies, gstride;  // copies of the species accumulators in memory (power of two) and doubles between them:
                         // workgroup b flushes into copy b % gcopies (fewer atomics per address), the field
                         // kernels add the copies up
};

// Marker storage.  The four arrays of a species (x, v, w, p) are NOT four separate
// allocations: they are interleaved in tiles of TILE markers,
//     [ x tile | v tile | w tile | p tile ] [ x tile | v tile | ... ] ...
// so that marker i of array k lives at slab[(i / TILE) * 4 * TILE + k * TILE + i % TILE].
// Why: the second sub-step's kernel runs seven streams at once (x, v, w, p read; x, v,
// w written back).  As four independent arrays their relative position in physical
// memory -- which the allocator, not the program, decides -- moved that kernel between
// 1.00 and 1.16 ms at 1e8 markers (DRAM bank/channel interference: tools/placement_probe.py
// maps kernel time against the array-to-array distance), and a pure 4-read / 3-write
// stream reached 5.0 TB/s.  Interleaved at 32 KiB the distances are fixed by the layout:
// the same stream runs at 6.3 TB/s wherever the slab lands (tools/layout_probe.py;
// 8 KiB: 5.3, 128 KiB: 5.3).  PIC1DP_TILE_LOG2=0 builds the untiled variant (arrays a
// 2 MiB multiple apart in one slab) for A/B measurements.
#ifndef PIC1DP_TILE_LOG2
#define PIC1DP_TILE_LOG2 12
#endif
constexpr int TILE_LOG2 = PIC1DP_TILE_LOG2;
constexpr int64_t TILE = static_cast<int64_t>(1) << TILE_LOG2;
static_assert(TILE_LOG2 == 0 || TILE_LOG2 >= 7, "a wave's 64 pairs must not straddle a tile");
// offset of marker i inside an array of a slab (arrays start k * TILE apart)
__host__ __device__ constexpr int64_t tidx(int64_t i) {
  return TILE_LOG2 ? (((i >> TILE_LOG2) << (TILE_LOG2 + 2)) + (i & (TILE - 1))) : i;
}
// the same for marker PAIRS, in double2 units
__host__ __device__ constexpr int64_t tidx2(int64_t j) {
  return TILE_LOG2 ? (((j >> (TILE_LOG2 - 1)) << (TILE_LOG2 + 1)) + (j & (TILE / 2 - 1))) : j;
}
// doubles of a slab holding n markers, and where array k (0 x, 1 v, 2 w, 3 p) starts
inline int64_t slab_array_stride(int64_t n) {
  if (TILE_LOG2) return TILE;
  // untiled A/B build only (-DPIC1DP_TILE_LOG2=0): arrays a 2 MiB multiple apart
  const int64_t unit = (2 << 20) / 8;
  return (n + unit - 1) / unit * unit;
}
inline int64_t slab_doubles(int64_t n) {
  return TILE_LOG2 ? (n + TILE - 1) / TILE * 4 * TILE : 4 * slab_array_stride(n);
}

// One particle set = the three pushed arrays of a species (array bases inside a slab).
struct PSet {
  double *x, *v, *w;
};

struct PushArgs {
  PSet src;       // state the derivatives are evaluated at (particle_x/v/w)
  PSet base;      // RK base (particle_*_bak); == src for irk 1
  PSet dst;       // where the pushed state goes
  const double *p;
  const double *E;  // [nx] field_electric
  double *rho;      // [nx] this species' charge accumulator (fused deposit)
  int64_t np;
  double dt;        // 0.5*dt (irk 1) or dt (irk 2)
  GridConst g;
  SpeciesConst s;
  int iptcldist, deltaf, linear, irk;
};

// geometry of the output_ptcldist histograms (src/pic1dp_output.F90:203-205,243-247)
struct DistGeom {
  double lx, vmax;
  int nxo, nvo;   // nx_opd, nv_opd
  // the two divisions of :243, :247 are divisions by run-time constants: rlx = RN(1/lx) (GridConst::rlx), dv = vmax * 2.0
  // and rdv = RN(1/dv); vfast = 1: the host vouches for div_const's five operations on dv (hostcheck.cpp), else the
  // hardware's division.  Bit for bit the IEEE quotients either way (device_math.hpp div_lx).
  double rlx, dv, rdv;
  int vfast;
};
DistGeom make_dist_geom(double lx, double vmax, int nxo, int nvo);
// scales of the histograms' 64-bit fixed-point sums in a workgroup's LDS copy (device_diag.hpp): plane k (markers, p, w)
// is summed as RN(term * sc[k]); a |p| / |w| beyond bound[k] makes the pass say so instead (the host repeats it in doubles)
struct DistScale {
  double sc[3];     // 2^e per plane
  double inv[3];    // 2^-e
  double bound[3];  // |q_k| the scale allows
};
// the scales for a pass of `blocks` workgroups over np markers whose |p|, |w| stay below bound_p, bound_w (<= 0: unknown);
// false: no fixed-point pass possible (sum in doubles)
bool make_dist_scale(int64_t np, int blocks, bool deltaf, double bound_p, double bound_w, DistScale *fx, int threads = 1024);

// dynamic LDS a particle kernel may ask for: 160 KiB per CU minus the 1 KiB static exp table
constexpr size_t PARTICLE_LDS_CAP = 159 * 1024;

struct LaunchCfg {
  int threads;   // per workgroup
  int blocks;    // grid size
  size_t lds;    // dynamic LDS bytes
};

struct FieldArgs {
  double *rho_sp;        // [rho_copies][nspecies][nx] raw per-species deposits (zeroed after use)
  int rho_copies, rho_stride;  // copies the particle kernels flush into, doubles between them (GridConst::gcopies)
  double *charge;        // [nx] charge2 / charge1 (local sum, all-reduced in place)
  double *chargeden;     // [nx]
  double *E;             // [nx]
  double *mode_re, *mode_im;   // [nmode]
  const double *fre, *fim;     // [nmode][nx] cos / -sin tables
  const double *grad_inv;      // [nmode]
  double *history;       // energy slot to write, or nullptr
  int nx, nmode, nspecies, deltaf;
  int npe;               // reference ranks reproduced: > 1 takes the forward sums in the npe-rank order (MPI-AIJ row
                         // blocks, device_field.hpp rank_block) and the inverse's per row with the row's own rank's mode
                         // entries first (inverse_row); 1: the one-rank ascending order
  int tab_lds;           // 1: stage the tables in LDS (they fit)
  int chain_mfma;        // 1: the serial forward sums of the one-rank order (up to eight kept modes) through the FP64
                         // matrix unit (device_field.hpp chain_rows_mfma) -- set by create() only if the device gives the
                         // sequential sums bit for bit that way (launch_chain_selftest); 0: a chain of additions in one lane
  double lx, dnx, sc_re, sc_im;
  double Z[8], n0[8];
};

// what the host knows of the one kept mode's tables (pred_kind 2): their sums (full-f offset) and their Gram
// matrix (kept-mode reconstruction of chargeden for the call-site path)
struct PredTab {
  double sum_fre, sum_fim, g11, g22, g12;
};

// one-hop charge exchange between the GPUs of a node (kernels_field.hip exchange_charge)
constexpr int XCHG_MAX_RANKS = 16;
struct XchgArgs {
  double *slots[XCHG_MAX_RANKS];               // every rank's slot area [2][nranks][nx] as mapped here
  unsigned long long *flags[XCHG_MAX_RANKS];   // every rank's flag area [2][XCHG_MAX_RANKS]
  unsigned long long *err;                     // host-visible error word (0 = fine)
  unsigned long long epoch;                    // number of this exchange, from 1
  long long timeout_ticks;                     // wall_clock64 ticks (100 MHz) a rank waits for its peers
  int rank, nranks;
  int local_in_charge;                         // 1: this rank's charge2 is already in FieldArgs::charge
  int vstride;                                 // doubles per (parity, rank) slot: XCHG_MAX_VEC * nx
  unsigned long long *ticks;                   // null, or [2] device words: wall-clock ticks (100 MHz) spent inside
                                               // exchanges so far (stores, the wait for the peers, the sum) and their number
};
// The TAIL of a one-pass marker launch on several ranks (device_xchg.hpp step_tail; k_step_one<PRIV>, k_step_sums, the
// last species launched): the last workgroup to finish forms this rank's packed vector [charge2 | six sums | pad] from the
// accumulators (re-zeroed) -- what k_charge_pack_sums did in a launch of its own -- and either leaves it in `pack` for
// the all-reduce that follows on the stream (mode 1) or stores it into every rank's exchange slots (mode 2), so that
// the field launch only waits for the flags, adds in rank order and solves.
struct StepTail {
  int mode;                    // 0 none, 1 pack for one all-reduce, 2 post into the peers' exchange slots
  unsigned int *ticket;        // device word, zero between launches: workgroups that have finished
  double *rho_sp;              // [rho_copies strides][nspecies][nx] the accumulators of ALL species (read, re-zeroed)
  int rho_copies, rho_stride, nspecies, nx;
  double Z[8];
  double *sums;                // [PRED_SUM_COPIES][8] the six sums' copies (read, re-zeroed)
  double *pack;                // mode 1: [nx + 8]
  XchgArgs x;                  // mode 2
};

// A whole-step launch whose PROLOGUE solves the field of the previous step itself (kernels_step.hip FUSED; one rank, one
// kept mode, prediction as six sums): every workgroup forms charge2 / chargeden from the accumulators the previous
// launch deposited into, runs the forward sums in the reference's order (the same device functions as the field
// kernels: device_field.hpp), and writes E0 and Eh of this step straight into its LDS tiles -- one launch per time
// step instead of two; workgroup 0 also writes what the field kernel would have left in memory.  The accumulators
// rotate through three buffers: read (deposited by the previous launch), deposit (this launch), zeroed by workgroup 0
// (read by the previous launch: nobody touches it any more).
struct FusedSolve {
  int on;                 // 0: the launch stages E0 / Eh from memory
  FieldArgs f;            // rho_sp: the accumulators to READ (left as they are); charge, chargeden, E, mode_re / mode_im,
                          // history: outputs of workgroup 0
  PredTab pt;
  const double *pred_in;  // [PRED_SUM_COPIES][8] the previous launch's six sums (read, left as they are)
  double *E_h, *mode_h;   // workgroup 0: this step's half-step field and its kept mode (re, im)
  double *zero_rho;       // workgroup 0 zeroes zero_rho[0 .. zero_rho_n) and zero_pred[0 .. 8 PRED_SUM_COPIES)
  int64_t zero_rho_n;
  double *zero_pred;
};

// whole-time-step path: state updated in place, half-step state recomputed
struct StepArgs {
  double *x, *v, *w;   // particle_x/v/w, updated in place by the full kernel
  const double *p;
  const double *E0;    // field at the start of the step
  const double *Eh;    // field after the first sub-step (full kernel only)
  double *rho;         // this species' charge accumulator
  int64_t np;
  double dt_half, dt_full;
  GridConst g;
  SpeciesConst s;
  int iptcldist, deltaf, linear;
  int stream_nt;       // 1: non-temporal loads/stores (state larger than the Infinity Cache)
  double *t2;          // [np + 2] carry of -f0'/f0 from the first kernel to the second, or null (kernels_step.hip CARRY)
  // full kernel only: take the diagnostics of output_all in the same pass (kernels_step.hip DIAG); dist_out null = no
  DistGeom dg;
  double *dist_out, *dist_partial;
  // full kernel only: also predict the charge of the NEXT step's first sub-step (kernels_step.hip k_step_one);
  // pred null = no.  tabA/tabB: [pred_nm][nx] mode tables with E = sum_m re_m*A_m + im_m*B_m;
  // pred: [1 + 2*pred_nm][nx] accumulators of this species (R0, RA_m, RB_m); t2_mode: 0 no carry of
  // -f0'/f0 through t2, 1 write it for the next step, 2 read this step's and write the next step's
  const double *tabA, *tabB;
  double *pred;
  int pred_nm, t2_mode;
  // pred_kind 2 (kernels_step.hip k_step_sums, for grids whose prediction tiles outgrow the LDS; one kept mode): pred is
  // [PRED_SUM_COPIES][8] -- six global sums K0c K1c K2c K0s K1s K2s shared by all species (Z folded in), in copies --, and Eh is not staged
  // but formed from the tables and its kept mode *eh_re, *eh_im (Eh = re A + im B, bit for bit what the solve wrote)
  int pred_kind;  // 1 tiles (k_step_one), 2 sums (k_step_sums)
  int pred_private;  // pred_kind 2 on a grid whose E0, Eh and table tiles fit the LDS: k_step_one<PRIV>, the six sums in
                     // thread-private LDS slots (Eh staged from memory like k_step_one's)
  const double *eh_re, *eh_im;
  FusedSolve fused;  // pred_kind 2 only: the prologue solves the previous step's field (E0, Eh, eh_re / eh_im unused)
  StepTail tail;     // pred_kind 2 only, several ranks: the last workgroup packs / posts this rank's charge (mode 0: no)
  DistScale dscale;  // k_step_full<DIAG>: the histograms as fixed-point sums (diag_fx != 0)
  int diag_fx;
  int dyn_tail;      // every whole-step kernel: sixteenths of a workgroup's chunks that its waves draw from an LDS counter (0: all dealt)
  double *fxb;       // pred_kind 1: [2] device bounds on |q|, |c| of this species, for the tiles' fixed-point sums
};
#ifndef PIC1DP_PRED_MAX_MODES
#define PIC1DP_PRED_MAX_MODES 3
#endif
constexpr int PRED_MAX_MODES = PIC1DP_PRED_MAX_MODES;  // kept modes k_step_one's prediction tiles are instantiated for: 1 ... 3.  (Round 4
                                    // built three and four with double sums and lost to the two passes; with the fixed-point
                                    // sums of round 6 three win -- 1.20 against 1.40 ms at 1e8 markers --, four were not rebuilt)
// pred_kind 2: the six sums (padded to 8) are kept in this many copies -- workgroup b of the marker kernel adds into
// copy b % PRED_SUM_COPIES, the field kernels add the copies up: six addresses shared by all workgroups serialise
constexpr int PRED_SUM_COPIES = 16;
// dynamic LDS of k_step_sums: E0, A, B tiles (with guard cell), rho copies, reduction scratch
inline size_t step_sums_lds_bytes(int nx) {
  const size_t ne = static_cast<size_t>((nx + 2) & ~1);
  return sizeof(double) * (3 * ne + ((static_cast<size_t>(nx) + 2) & ~static_cast<size_t>(1)) + 96);  // [6][16] scratch
}
// dynamic LDS of k_step_one: E0, Eh tiles (with guard cell), the tables cell by cell (nx + 1 cells of 2 nm), rho
// copies, the prediction accumulators cell by cell (nx + 2 cells of 1 + 2 nm)
inline size_t step_one_lds_bytes(int nx, int nm) {
  const size_t ne = static_cast<size_t>((nx + 2) & ~1);
  return sizeof(double) * (2 * ne + (static_cast<size_t>(nx) + 1) * 2 * nm +
                           ((static_cast<size_t>(nx) + 2) & ~static_cast<size_t>(1)) +
                           (static_cast<size_t>(nx) + 2) * (1 + 2 * nm) + 2);  // (+ the drawn chunks' counter)
}
// dynamic LDS of k_step_one<PRIV>: E0, Eh tiles, the one mode's tables cell by cell (nx + 1 cells of 2), rho copies,
// six private sums per thread, reduction scratch
#ifndef PIC1DP_PRIV_THREADS
#define PIC1DP_PRIV_THREADS 768  // (tuning builds: 1024 = one workgroup of sixteen waves per CU, tools/ab_variant_libs.sh)
#endif
constexpr int STEP_PRIVATE_THREADS = PIC1DP_PRIV_THREADS;  // k_step_one<PRIV> is launched with exactly this many threads per workgroup
inline size_t step_one_private_lds_bytes(int nx, int threads = STEP_PRIVATE_THREADS) {
  const size_t ne = static_cast<size_t>((nx + 2) & ~1);
  return sizeof(double) * (2 * ne + (static_cast<size_t>(nx) + 1) * 2 +
                           ((static_cast<size_t>(nx) + 2) & ~static_cast<size_t>(1)) +
                           6 * static_cast<size_t>(threads) + 16);
}
// dynamic LDS of the DIAG variant beyond the grid tiles: histograms + reduction scratch
inline size_t step_diag_lds_bytes(int nx, int nxo, int nvo) {
  (void)nx;  // the rho region of step_lds_bytes ends 16-byte aligned
  return sizeof(double) * (3 * static_cast<size_t>(nxo) * nvo + 3 * static_cast<size_t>(nvo) + 16);
}
// full = false: first sub-step (deposit of the half-step state, nothing stored)
// full = true : second sub-step (recompute half-step state, push, deposit, store)
hipError_t launch_step(const StepArgs &a, bool full, const LaunchCfg &lc, hipStream_t st);

// push (+gather) with or without the fused wrap+deposit
hipError_t launch_push(const PushArgs &a, bool fused_deposit, const LaunchCfg &lc, hipStream_t st);
// wrap + deposit of q (= w or p) at x, x stored back
hipError_t launch_deposit(double *x, const double *q, double *rho, int64_t np, const GridConst &g,
                          const LaunchCfg &lc, hipStream_t st);


// vectors of nx doubles one exchange can carry: charge2 + the 1 + 2 * PRED_MAX_MODES prediction slices
constexpr int XCHG_MAX_VEC = 2 + 2 * PRED_MAX_MODES;
// both fields of a one-pass step in one launch: the new state's field from its deposited charge, then the
// NEXT step's half-step field from k_step_one's prediction (x1: the ONE exchange of a multi-rank step -- charge2
// and the Z-weighted prediction slices travel together --, or null)
struct PairArgs {
  double *pred;     // kind 1: [nspecies][1 + 2 nmode][nx]; kind 2: [PRED_SUM_COPIES][8] the six sums; consumed (re-zeroed)
  double *E_h;      // [nx] half-step field of the next step
  double *mode_h;   // [2 nmode] its kept modes (re..., im...)
  double *cd_h;     // [nx] its charge density (kind 1: scratch; kind 2: the kept mode's content of it, for the call sites' adoption)
  double *pack;     // null, or what k_charge_pack made, already summed over ranks
  int kind;         // 1 tiles, 2 sums
  PredTab pt;       // kind 2
  int posted;       // kind 2 with x1: the marker launch's tail has stored this rank's vector and flag already (StepTail mode 2)
};
// doubles k_charge_pack writes / one exchange of a one-pass step carries per rank:
// kind 1: charge2 + the 1 + 2 nmode Z-weighted prediction slices; kind 2: charge2 + the six sums (padded to 8)
inline size_t pack_doubles(int nx, int nmode, int kind) {
  return kind == 2 ? static_cast<size_t>(nx) + 8 : static_cast<size_t>(2 + 2 * nmode) * nx;
}
// this rank's charge2 and prediction packed for one all-reduce (accumulators re-zeroed)
hipError_t launch_charge_pack(const FieldArgs &f, double *pred, int nm_pred, int kind, double *pack, hipStream_t st);
hipError_t launch_field_solve_pair(const FieldArgs &f, const PairArgs &pa, const XchgArgs *x1, hipStream_t st);
// kind 2, call-site path: the six sums K (summed over ranks) + the field's kept mode -> field_chargeden with the
// kept-mode content of the next first sub-step's charge density; pred (or null) is re-zeroed
hipError_t launch_pred_chargeden(const FieldArgs &f, const PredTab &pt, double *pred, const double *K, hipStream_t st);
// kind 2, call-site path, one rank: launch_pred_chargeden (from this rank's copies of the sums) and the mode-filter
// solve in one launch
hipError_t launch_field_solve_pred_sums(const FieldArgs &f, const PredTab &pt, double *pred, hipStream_t st);
// kind 2, call-site path, several ranks: this rank's six sums into the head of f.charge (rest zero), pred re-zeroed
hipError_t launch_pred_to_charge(const FieldArgs &f, double *pred, hipStream_t st);
// k_step_one's prediction accumulators [nspecies][1 + 2 nm][nx] + the field's kept modes -> this rank's
// charge2 of the next first sub-step in f.charge; the accumulators are re-zeroed
hipError_t launch_pred_combine(const FieldArgs &f, double *pred, int nm_pred, hipStream_t st);
// launch_pred_combine, the scaling into f.chargeden and the mode-filter solve in one launch (call sites, one rank)
hipError_t launch_field_solve_pred(const FieldArgs &f, double *pred, int nm_pred, hipStream_t st);
// local charge + exchange: the summed charge1 into f.charge
hipError_t launch_charge_exchange(const FieldArgs &f, const XchgArgs &x, hipStream_t st);
// local charge + exchange + chargeden + field solve in one launch
hipError_t launch_field_solve_xchg(const FieldArgs &f, const XchgArgs &x, hipStream_t st);

// charge2 = sum_s rho_sp[s]*Z_s ; rho_sp = 0      (src/pic1dp_interaction.F90:81-128)
hipError_t launch_charge_local(const FieldArgs &f, hipStream_t st);
// chargeden from charge (:138-148) only; with_local folds launch_charge_local in
hipError_t launch_chargeden(const FieldArgs &f, bool with_local, hipStream_t st);
// chargeden from charge (:138-148) then field_solve_electric; with_local runs
// launch_charge_local's work first in the same kernel (single-rank path)
hipError_t launch_field_solve(const FieldArgs &f, bool with_local, bool from_chargeden,
                              hipStream_t st);
// opt-in alternative solve (all modes): finite-difference Poisson equation as a
// tridiagonal system, parallel cyclic reduction in LDS; chargeden -> E (+ energy)
hipError_t launch_field_fd(const double *chargeden, double *E, double *history, int nx, double lx,
                           double dnx, hipStream_t st);
// The serial sums of nrows (<= 16) rows of n generated doubles two ways: out[0 .. nrows) chain_sum_lds, out[16 .. 16 + nrows)
// chain_rows_mfma -- create() compares them with the host's sequential sums
hipError_t launch_chain_selftest(const double *v, int nrows, int n, double *out, hipStream_t st);
// int E^2 dx into *out (device)
// output_all's record gathered on the device: up to PACK_MAX_SEGMENTS ranges copied into one contiguous buffer, which then
// crosses PCIe in ONE transfer (a small device-to-host copy costs the stream ~10 us whatever its size; a record had 5 + 2 per species)
constexpr int PACK_MAX_SEGMENTS = 5 + 2 * 8;
struct PackArgs {
  const double *src[PACK_MAX_SEGMENTS];
  unsigned dst[PACK_MAX_SEGMENTS];  // offset in the record, in doubles
  unsigned n[PACK_MAX_SEGMENTS];
  int count;
};
hipError_t launch_pack_record(const PackArgs &a, double *out, hipStream_t st);
hipError_t launch_field_energy(const double *E, int nx, double lx, double dnx, double *out,
                               hipStream_t st);

// per-block partial sums of v^2, v^2 p, v^2 w over markers [i0, i0 + n) -> partial[blocks][3]
hipError_t launch_energy_sums(const double *v, const double *p, const double *w, int64_t i0, int64_t n,
                              double *partial, int blocks, hipStream_t st);
// contiguous buffer <-> markers [i0, i0 + n) of one array of a slab
hipError_t launch_tile_scatter(double *arr, int64_t i0, const double *src, int64_t n, hipStream_t st);
hipError_t launch_tile_gather(const double *arr, int64_t i0, double *dst, int64_t n, hipStream_t st);
// markers [0, n) of one array copied between two slabs of the same geometry
hipError_t launch_tile_copy(double *dst, const double *src, int64_t n, hipStream_t st);
// raw (x,v) and v histograms of output_ptcldist into out =
// [markr_xv | total_xv | pertb_xv | markr_v | total_v | pertb_v] (accumulated)
// In the same pass: partial[blocks][6] = per-workgroup sums of v^2, v^2 p, v^2 w over all np markers, max |p|, max |w|,
// and whether a marker exceeded the bounds of a fixed-point pass; blocks = ptcldist_blocks(...).
// bound_p, bound_w > 0: the (x, v) histograms as 64-bit fixed-point sums in the LDS (device_diag.hpp DistScale)
hipError_t launch_ptcldist(const double *x, const double *v, const double *p, const double *w,
                           int64_t np, const DistGeom &dg, bool deltaf, double bound_p, double bound_w,
                           double *out, double *partial, int num_cu, int dyn_tail, hipStream_t st, bool *fixed_point);
int ptcldist_blocks(int64_t np, int nxo, int nvo, int num_cu);
// ---- marker optimisation events (kernels_opt.hip; host side of the sequential part: optimize.hpp plan_*) ----
// one reference rank block of a species inside the species' packed (tiled) arrays: block-local marker i lies at global
// marker voff + i while i < nvalid0 (the block's valid markers when the event began), else at toff + (i - nvalid0) (its
// tail slots); nvalid0 is the LAYOUT's split point, fixed during an event -- the block's current count travels apart
struct OptBlock {
  double *x, *v, *p, *w;
  int64_t voff, nvalid0, toff;
};
struct OptGrid {
  double lx, v_max;
  int nx, nv;
};
constexpr uint32_t OPT_KEY_IMPORTANT = 0xFFFFFFFFu;
// this block's |delta f|(v) into hist[nv] (device), every bin summed in storage order; synchronises the stream
// scratch: device memory of at least opt_hist_scratch_bytes(np, nv) (the sort's items and temporaries)
size_t opt_hist_scratch_bytes(int64_t np, int nv);
hipError_t opt_hist_block(const OptBlock &b, const OptGrid &g, int64_t np, double *hist, void *scratch, hipStream_t st);
hipError_t opt_merge_keys(const OptBlock &b, const OptGrid &g, const double *hist, double limit, int64_t np, uint32_t *keys,
                          hipStream_t st);
hipError_t opt_remove_vals(const OptBlock &b, const OptGrid &g, const double *hist, double peak, double limit, int by_threshold,
                           int64_t np, uint8_t *skip, double *df, hipStream_t st);
hipError_t opt_split_flags(const OptBlock &b, const OptGrid &g, const double *hist, double limit, int64_t np, uint8_t *flag,
                           hipStream_t st);
// holes[nholes]: the positions below np_new whose marker is gone, ascending (gone: the ids listed, or -- gone_bits
// non-null -- one bit per position); fails when their number is not nholes; scratch: opt_holes_scratch_bytes(np_new)
size_t opt_holes_scratch_bytes(int64_t np_new);
hipError_t opt_holes(const uint32_t *gone_ids, int64_t ngone, const uint32_t *gone_bits, int64_t np_new, int64_t nholes,
                     uint32_t *holes, void *scratch, hipStream_t st);
// scratch: [4 npairs] doubles.  Indices are block-local, 32 bits (a block holds < 2^32 slots)
hipError_t opt_merge_apply(const OptBlock &b, const OptGrid &g, const double *hist, double limit, const uint32_t *dst,
                           const uint32_t *idk, int64_t npairs, const uint32_t *move_pos, const uint32_t *move_id, int64_t nmoves,
                           int64_t ghost, int64_t np_new, double *scratch, hipStream_t st);
hipError_t opt_remove_apply(const OptBlock &b, const OptGrid &g, const double *hist, double peak, double limit, int by_threshold,
                            double keep_scale, const uint32_t *move_pos, const uint32_t *move_id, int64_t nmoves, int64_t ghost,
                            int64_t np_new, hipStream_t st);
hipError_t opt_split_apply(const OptBlock &b, int64_t parents, const uint32_t *ks, const double *dv, int64_t nsplit, int ng, int deltaf,
                           hipStream_t st);
hipError_t opt_copy_segment(const OptBlock &b, int64_t i0, int64_t n, double *dx, double *dv, double *dp, double *dw, int64_t doff,
                            hipStream_t st);

// div_lx's / div_const's algorithm (reciprocal + two FMA corrections) with the host's fma against the true
// quotient on n generated operands: the number of results that differ in any bit (hostcheck.cpp)
int64_t host_div_check(double lx, int nx, uint64_t seed, int64_t n);
// the planners of the GPU marker optimisation against the host routines they restate (optcheck.cpp; test support):
// slots that differ, -1 / -2 when the marker counts / the random streams' positions differ
int64_t host_optimize_check(const pic1dp_input &in, int kind, double threshold, uint64_t seed, int64_t np0, int64_t nalloc,
                            int64_t *np_after);
int64_t host_divc_check(double c, uint64_t seed, int64_t n);
// cell index per marker and per-cell counts from (wrapped) x
hipError_t launch_cell_indices(const double *x, int64_t np, const GridConst &g, int32_t *ix,
                               unsigned long long *count, hipStream_t st);

}  // namespace pic1dp
