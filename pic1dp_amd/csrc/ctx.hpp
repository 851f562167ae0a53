// ctx.hpp -- what the translation units behind the C ABI share: the context (one process = one GPU = one HIP stream),
// the error convention, the event spans, and the internal entry points one unit offers the others.
//   capi.cpp           context, input, transfers, field access, timers and knobs
//   capi_step.cpp      the hot path: the three call sites and their lazy state machine, pic1dp_hip_step, the prediction
//   capi_comm.cpp      RCCL communicator, the one-hop exchange's set-up, the charge sum over ranks
//   capi_diag.cpp      diagnostics of output_all
//   capi_optimize.cpp  marker optimisation events (merge / remove / split)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/pic1dp_hip.h"
#include "kernels.hpp"
#include "loader.hpp"
#include "multirand.hpp"
#include "optimize.hpp"
#include "rccl_dyn.hpp"

using namespace pic1dp;

namespace pic1dp_host {

// the message pic1dp_hip_last_error() hands out (thread-local, capi.cpp); returns code
int fail(int code, const char *fmt, ...);


#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(PIC1DP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

#define CHECK_CTX(c) \
  if (!(c)) return fail(PIC1DP_ERR_ARG, "null context")

constexpr double kPi = 3.14159265358979323846264;        // PETSC_PI
constexpr double kSqrtEps = 1.490116119384766e-08;       // PETSC_SQRT_MACHINE_EPSILON
constexpr int kTagFused = 100, kTagPush = 101, kTagDeposit = 102, kTagStepHalf = 103, kTagStepFull = 104, kTagStepOne = 106,
              kNumTags = 128;
constexpr int64_t kHistCap = 1 << 20;
constexpr bool kCarryOneExpDefault = false;  // k_step_one with the one-exp form of -f0'/f0: carry it (72 B) or evaluate it again (56 B)
constexpr int kEnergyBlocks = 1024;
constexpr size_t kCuLds = 160 * 1024, kStaticLds = 1024;  // LDS of a CU; static LDS of a marker kernel (the exp table)
// states of the lazy call sites (pic1dp_ctx::lz)
// ---------------------------------------------------------------------------
// The call sites' state (capi_step.cpp "lazy call sites").  Two variables, each an enum, and a table of
// the pairs that can occur -- until round 6 this was lz, half_pair, half_solved and cd_lazy, 4 x 2 x 2 x 6 combinations of
// which 20 are legal, policed from outside by pic1dp_hip_check_state; now an illegal one cannot be stored (set_call_state).
//
// Seq: where the host stands in the reference's sequence push(1), collect_charge, solve_field, push(2), collect_charge,
// solve_field of a time step (src/pic1dp.F90:79-90), and -- the "Pair" states, one rank, six sums -- whether the half
// step is being served from the PREVIOUS step's pair solve: the half-step field then lies in d_Ehn / d_mode_h,
// field_electric still holds the step-start field, d_E0 is not filled.
enum class Seq : uint8_t {
  Clean,            // memory is what eager calls would have left; nothing noted
  Push1,            // push(1) noted
  Half,             // ... and its collect_charge served (k_step_half ran, or the prediction was taken); the half-step markers are not in memory
  HalfPair,         // Half, served from the pair solve: nothing was launched; the host has not yet called solve_field for the half step
  HalfPairSolved,   // ... and now it has: what it may SEE from here on is the half-step field (every reader settles first)
  Push2,            // push(2) noted after Half
  Push2Pair,        // push(2) noted after HalfPair (the host skipped the half step's solve_field: it pushes with the old field)
  Push2PairSolved,  // push(2) noted after HalfPairSolved: the next collect_charge runs k_step_one with E0 = field_electric, Eh = d_Ehn
  N
};
// Owed: what collect_charge has left to the launch of the solve_field that follows (one launch less per sub-step)
enum class Owed : uint8_t {
  Nothing,         // field_chargeden is current
  Scale,           // d_charge holds the summed charge1: the scaling into chargeden
  SumScale,        // (one rank) the species accumulators hold the deposits: species sum + scaling -- with a usable six-sum prediction: the pair solve
  PredTiles,       // (one rank, tiles) the prediction accumulators hold the half-step charge as coefficients: combine + sum + scale
  PredSums,        // (one rank, six sums) the kept mode's content of chargeden follows from the six sums
  AdoptHalfField,  // nothing to launch: field_electric / its kept mode / chargeden <- what the pair solve left in d_Ehn / d_mode_h / d_cd_h
  N
};
enum { LZ_CLEAN = 0, LZ_PUSH1, LZ_HALF, LZ_PUSH2 };   // the four positions of the sequence (Seq without the pair's colours)
constexpr int lz_of(Seq s) {
  return s == Seq::Clean ? LZ_CLEAN : s == Seq::Push1 ? LZ_PUSH1 : (s == Seq::Half || s == Seq::HalfPair || s == Seq::HalfPairSolved) ? LZ_HALF : LZ_PUSH2;
}
constexpr bool pair_of(Seq s) { return s == Seq::HalfPair || s == Seq::HalfPairSolved || s == Seq::Push2Pair || s == Seq::Push2PairSolved; }
constexpr bool solved_of(Seq s) { return s == Seq::HalfPairSolved || s == Seq::Push2PairSolved; }
// the same position with the pair settled (settle_half_pair: memory as the eager calls leave it)
constexpr Seq unpaired(Seq s) { return lz_of(s) == LZ_HALF ? Seq::Half : lz_of(s) == LZ_PUSH2 ? Seq::Push2 : s; }
// push(2) noted at a half step
constexpr Seq push2_noted(Seq s) { return s == Seq::HalfPair ? Seq::Push2Pair : s == Seq::HalfPairSolved ? Seq::Push2PairSolved : Seq::Push2; }
// Which (Seq, Owed) pairs occur.  Scale / SumScale: right behind a collect_charge (every other call that looks at the
// charge settles it first, materialize_cd).  PredTiles / PredSums: behind the collect_charge of a noted push(1) -- and
// still owed after a look at the MARKERS has put the half-step state into memory (particles_download materialises the
// markers, not the charge: Clean with a prediction owed; the fuzz campaign of round 6 found that one, 5 seeds in 4 500).
// AdoptHalfField is set with HalfPair and outlives it when an inspection settles the pair before solve_field came (the
// field is then adopted by copying); it ends with the solve_field of the half step, hence never with a Solved state.
constexpr bool kCallStateLegal[static_cast<int>(Seq::N)][static_cast<int>(Owed::N)] = {
    //                   Nothing Scale  SumScale PredTiles PredSums Adopt
    /* Clean           */ {true, true,  true,    true,     true,    true},
    /* Push1           */ {true, false, false,   false,    false,   true},
    /* Half            */ {true, true,  true,    true,     true,    true},
    /* HalfPair        */ {false, false, false,  false,    false,   true},
    /* HalfPairSolved  */ {true, false, false,   false,    false,   false},
    /* Push2           */ {true, false, false,   false,    false,   true},
    /* Push2Pair       */ {false, false, false,  false,    false,   true},
    /* Push2PairSolved */ {true, false, false,   false,    false,   false},
};

struct Species {
  int64_t nalloc = 0, np = 0;
  PSet set[2] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  double *p = nullptr;
  double *t2 = nullptr;   // carry of -f0'/f0 between the whole-step kernels
  uint64_t t2_version = 0;  // state_version whose step-start velocities the values in t2 belong to (0: none)
  double *slab[2] = {nullptr, nullptr};  // tiled storage of set 0 (+ p) and of the RK ping-pong set (kernels.hpp)
  double *rho = nullptr;  // slice of rho_sp
  double *fxb = nullptr;  // device [2]: bounds on |q| and |c| of this species' markers for the prediction tiles' fixed-point
                          // sums (kernels_step.hip FxTiles): seeded from the markers the host loads, raised by the kernels
  SpeciesConst sc{};
};

struct EvPair {
  hipEvent_t a, b;
  int tag;
  bool head;  // one of the first 64 spans of its timer id since the last reset (sampled timers: taken as they are, not scaled)
};

}  // namespace pic1dp_host
using namespace pic1dp_host;

struct pic1dp_ctx {
  pic1dp_input in{};
  pic1dp_layout lay{};
  int device = 0, num_cu = 256;
  hipStream_t st = nullptr;
  int cur = 0;  // which particle set is particle_x/v/w right now
  std::vector<Species> sp;
  int blk0 = 0, nblk = 1;  // owned reference blocks [blk0, blk0+nblk)
  std::vector<int64_t> blk_alloc;                 // [nblk] allocated slots of each owned block
  std::vector<std::vector<int64_t>> blk_np;       // [nspecies][nblk] valid markers
  std::vector<Multirand> blk_rng;                 // [nblk] generators as particle_load left them
  bool rng_ready = false;
  int imerge = 0, iremove = 0, isplit = 0;        // particle_imerge / _iremove / _isplit
  bool loaded = false;
  bool charge_pending = false;  // charge_local ran, waiting for charge_reduced
  // field
  double *d_rho_sp = nullptr, *d_charge = nullptr, *d_chargeden = nullptr, *d_E = nullptr;
  // The species accumulators and the six sums of the prediction exist three times (kernels.hpp FusedSolve): d_rho_sp /
  // Species::rho / fa.rho_sp / d_pred always name the set the marker kernels deposit into NOW (acc_idx); all sets are
  // zero whenever no fused launch sequence is under way
  double *d_rho_all = nullptr, *d_pred_all = nullptr;
  size_t rho_set_doubles = 0, pred_set_doubles = 0;
  int acc_idx = 0;
  int fuse_solve = 1;           // PIC1DP_FUSE_SOLVE=0: the field solve always in a launch of its own; 2: fused whatever the grid
  bool fused_pending = false;   // step(): the last marker launch left the solve of its step to the next launch's prologue
  int fused_dirty = -1;         // accumulator set the last fused launch read (still holding that step's deposits), or -1
  FusedSolve fuse_args{};       // what the next marker launch's prologue has to solve (on = 1), consumed by step_particles
  double *d_mode_re = nullptr, *d_mode_im = nullptr, *d_fre = nullptr, *d_fim = nullptr;
  double *d_ginv = nullptr, *d_hist = nullptr, *d_scratch = nullptr, *d_dist = nullptr;
  // one pass per step (kernels_step.hip k_step_one): mode tables with E = sum re_m A_m + im_m B_m, the
  // prediction accumulators [nspecies][1 + 2 nm][nx], the combined half-step charge density
  double *d_tabA = nullptr, *d_tabB = nullptr, *d_pred = nullptr, *d_cd_h = nullptr, *d_mode_h = nullptr;
  int osub_req = 0;                // PIC1DP_OSUB: grid size of the marker kernels in units of the resident one (0: auto)
  int dyn_tail_full = 16;          // ... of k_step_full (PIC1DP_DYN_TAIL sets both, PIC1DP_DYN_TAIL_FULL this one)
  int dyn_tail = 8;                // PIC1DP_DYN_TAIL: sixteenths of a workgroup's 64-pair chunks its waves DRAW from an LDS counter (every whole-step kernel)
  int pred_kind = 0;               // 0 no one-pass step here, 1 prediction tiles (k_step_one), 2 six sums (k_step_sums)
  int pred_private = 0;            // pred_kind 2 and E0, Eh, the table tiles and the private sums of two workgroups fit a
                                   // CU's LDS: the sums are taken by k_step_one<PRIV> (thread-private LDS slots)
  PredTab pred_tab{};              // kind 2: sums / Gram matrix of the kept mode's tables (host, libm)
  int eh_modes = 0;                // kind 2: where the kept mode of the Eh about to be used lies: 0 nowhere, 1 fa.mode_*, 2 d_mode_h
  bool charge_pending_pred = false;  // kind 2: charge_local handed out the six sums, not a charge vector
  double *d_Ehn = nullptr;         // half-step field predicted for the NEXT step (d_Eh stays the last step's)
  double *d_pack = nullptr;        // [2 + 2 nmode][nx] one all-reduce per one-pass step (RCCL path)
  // several ranks, six-sum prediction: the marker launch's last workgroup packs / posts this rank's charge (kernels.hpp
  // StepTail) instead of a launch of its own in front of the sum over ranks.  PIC1DP_TAIL=0: the separate launch
  unsigned int *d_ticket = nullptr;  // the tail's arrival counter (zero between launches)
  int tail_on = 1;
  int tail_done = 0;                 // what the last marker launch's tail did: 0 nothing, 1 packed into d_pack, 2 posted (tail_x)
  XchgArgs tail_x{};                 // tail_done == 2: the exchange the field launch has to finish
  int64_t tail_launches = 0;         // marker launches that carried a tail so far (kernel_stats 10)
  int predict = 1;                 // PIC1DP_PREDICT=0: always two passes per step
  uint64_t pred_version = 0;       // state_version the accumulators in d_pred belong to (0: none)
  uint64_t eh_version = 0;         // state_version d_Eh has been predicted for (step() path)
  uint64_t field_version = 1, eh_field_version = 0, modes_field_version = 0;  // who wrote d_E last
  double *d_stage = nullptr;  // contiguous staging buffer between host arrays and the tiled marker arrays
  double *d_Eh = nullptr;  // field after the first sub-step of the last whole-step call
  // The reference's three call sites at whole-step cost (see "lazy call sites"
  // below): a push is only noted; the collect_charge that follows runs the
  // whole-step kernel instead of push + deposit.
  int lazy_calls = 1;            // PIC1DP_LAZY_CALLS=0: every call launches its own kernel at once
  // Call sites, one rank: the solve_field that follows the collect_charge of push(2) solves BOTH fields in one launch
  // (the pair kernels of pic1dp_hip_step) -- the new state's into field_electric, the next step's half-step field from the
  // prediction into d_Ehn / d_mode_h.  The next push(1) / collect_charge / solve_field then launch nothing: the
  // "Pair" states of Seq say that the half-step field lies in d_Ehn (field_electric still holds the step-start field, d_E0 is
  // not filled), the "Solved" ones that the host has called solve_field for it -- what it may look at from then on is the
  // half-step field, so every inspection first settles (capi_step.cpp settle_half_pair: copies, memory as eager calls leave it).
  // PIC1DP_CALL_PAIR=0: the three launches per step of rounds 2-4.
  int call_pair = 1;
  int64_t call_pair_skips = 0;   // solve_field calls of a half step served without a launch (kernel_stats 11)
  // collect_charge leaves its last step to the solve_field that follows (one launch less per sub-step):
  // 0 field_chargeden is current; 1 d_charge holds the summed charge1, its scaling is pending; 2 (one rank) the
  // species accumulators hold the deposits, species sum and scaling pending; 3 (one rank, mode-filter solve, few
  // modes) k_step_one's prediction accumulators hold the charge, combination with the kept modes, species sum and
  // scaling pending; 4 (one rank, mode-filter solve) the six sums of the prediction are pending: the kept mode's
  // content of chargeden follows from them.  materialize_cd() before anything else looks at charge, chargeden or the
  // accumulators.
  Owed owed = Owed::Nothing;
  // field_chargeden holds only the kept mode's content of the half-step charge density (collect_charge after a
  // noted push(1) served from the six sums, pred_kind 2): all solve_field looks at, but not what the reference
  // holds there.  get_field rebuilds the full vector on one rank (rebuild_half_step_chargeden); cleared by
  // everything that writes field_chargeden.
  bool cd_kept_mode_only = false;
  Seq seq = Seq::Clean;          // written by set_call_state only (capi_step.cpp)
  double *d_E0 = nullptr;        // field the noted push(1) saw
  double *d_rho_dummy = nullptr; // accumulator of a wrap-only deposit
  int carry = -1;          // whole-step kernels carry -f0'/f0 between them: -1 where measured to pay, 0 never
                           // (PIC1DP_CARRY=0), 1 wherever -f0'/f0 bears an exp, 2 also two-stream2 between k_step_half / _full
  int step_mode = 0;       // 0 auto (recompute path when the LDS allows), 1 two fused sub-steps
  int field_solver = 0;    // 0 the reference's mode-filter DFT solve, 1 finite-difference tridiagonal (opt-in)
  // marker state (bytes) above which k_step_half / k_step_full stream non-temporally
  // The two kernels leave the caches to each other, so the pairs were compared inside
  // one process on the same arrays (tools/ab_nt.py, nx = 1024, half + full in ms):
  //   markers   plain/plain   nt/nt    half nt, full plain   half plain, full nt
  //   6.4e6       0.102*      0.114         0.105                 0.106
  //   1e7         0.159       0.169         0.158*                0.158*
  //   2e7         0.372       0.332         0.326                 0.319*
  //   3e7         0.539       0.497         0.492                 0.483*
  //   5e7         0.886       0.829*        0.835                 0.828*
  //   1e8         1.745       1.657*        1.676                 1.678
  // => both plain below 288 MiB of marker state, the full kernel non-temporal above
  //    it, the half kernel only above 2 GiB
  double nt_threshold_half = 2048.0 * 1048576.0, nt_threshold_full = 288.0 * 1048576.0;
  int64_t hist_count = 0;
  // marker diagnostics of output_all: one fused pass per species (histograms +
  // kinetic sums), kept until the markers change
  uint64_t state_version = 1;              // bumped by everything that writes marker arrays
  std::vector<uint64_t> diag_version;      // [nspecies] version the cached results belong to
  std::vector<double> diag_sums;           // [nspecies][3]
  double *d_diag_part = nullptr;           // [nspecies][3 * diag_max_blocks] per-workgroup partial sums of the pass
  std::vector<char> diag_pending;          // [nspecies] a pass ran, its partial sums are still on the device
  std::vector<int> diag_blocks;            // [nspecies] workgroups of that pass
  std::vector<int> diag_stride;            // [nspecies] doubles per workgroup in its partial sums: 3 (k_step_full<DIAG>) or 6 (k_ptcldist)
  // k_ptcldist with 64-bit fixed-point histogram sums (device_diag.hpp DistScale): max |p| and max |w| of the species as
  // the last pass saw them (0: unknown -- the next pass sums in doubles and finds out); PIC1DP_DIAG_FX=0: always doubles
  std::vector<double> diag_max_p, diag_max_w;
  std::vector<char> diag_fixed;            // [nspecies] the pending pass summed its histograms in fixed point (an overflow: once more in doubles)
  std::vector<uint64_t> diag_max_p_version;   // state_version-independent stamp: p changes with load / upload / events only
  int diag_fx = 1;
  double diag_fx_margin_w = 16.0;   // bound on |w| = this x the last pass's max |w| (PIC1DP_DIAG_FX_MARGIN: tests)
  int64_t diag_fx_passes = 0, diag_fx_repeats = 0;   // fixed-point passes so far; passes repeated in doubles after an overflow
  int fuse_output = 0;                     // take the diagnostics inside k_step_full on steps output_all follows
  double *d_rec = nullptr;                 // output_all's record gathered on the device (kernels.hpp PackArgs), grow-only
  size_t d_rec_doubles = 0;
  double *h_pin = nullptr;                 // pinned host staging of the small device-to-host transfers (output_all's calls:
  size_t h_pin_doubles = 0;                // pageable copies cost a synchronous call each, ~20 us; grow-only, capi_diag.cpp pinned)
  DistGeom dist_geom_v{};                  // output_ptcldist's histogram geometry (capi_diag.cpp dist_geom)
  bool dist_geom_ready = false;
  int64_t opt_pcie_bytes = 0;              // bytes marker optimisation events have moved between host and device
  std::vector<void *> opt_workers;         // streams and staging of the events' block workers (capi_optimize.cpp OptWorker), kept between events
  int64_t diag_passes = 0;                 // separate k_ptcldist passes launched so far
  int64_t fused_solves = 0;                // marker launches whose prologue solved the previous step's field
  int chain_selftest = 0;                  // create()'s verdict on the serial sums through the matrix unit: 1 identical, 0 differs, -1 could not run
  int32_t itime = 0;
  double time = 0.0;
  int seed_offset = 0;                     // ensemble member: block b draws from stream mype = b + seed_offset (pic1dp_hip_set_seed_offset)
  GridConst grid{};
  FieldArgs fa{};
  // comm
  ncclComm_t comm = nullptr;
  // one-hop charge exchange (kernels_field.hip exchange_charge): the own area, the peers'
  // areas as mapped through hipIpc, and the running exchange number
  struct Xchg {
    void *local = nullptr;                       // flags + slots of this rank
    void *peer[XCHG_MAX_RANKS] = {nullptr};      // every rank's area as mapped here (own: local)
    bool opened[XCHG_MAX_RANKS] = {false};       // peer[q] came from hipIpcOpenMemHandle
    unsigned long long *err = nullptr;           // pinned host word the kernel reports a time-out in
    unsigned long long epoch = 0;
    long long timeout_ticks = 0;
    unsigned long long *ticks = nullptr;         // device [2]: ticks inside exchanges, exchanges timed (while timers are on)
    bool connected = false;
    int memkind = 0;                             // 1 fine-grained, 2 uncached, 3 plain hipMalloc
  } xc;
  int allreduce_kind = 0;  // 0 auto (RCCL when a communicator exists), 1 RCCL, 2 one-hop exchange
  // launch
  int threads_req = 0, bpc_req = 0;
  // timing
  bool timers_on = false, stats_on = false;
  std::vector<EvPair> evpool;
  size_t ev_used = 0;
  double acc_ms[kNumTags] = {0};
  int64_t acc_n[kNumTags] = {0};
  // timers (tags < 100) may be SAMPLED: only every timer_every-th block of 64 spans of a tag is bracketed by events (an event pair costs
  // the stream ~3 us of dependency: 10-28 % of a 70 us step at the reference's default size), the others only counted;
  // pic1dp_hip_timer_ms scales the sampled time by spans seen / spans timed
  int timer_every = 1;
  int64_t span_seen[kNumTags] = {0};
  // the first block of a timer id (first launches: code loading, cold caches, the run's first step) is always timed and
  // enters as it is; only the later blocks stand for the ones between them
  double head_ms[kNumTags] = {0};
  int64_t head_n[kNumTags] = {0};
  // what the marker kernel launched last under a tag moves per marker (pic1dp_hip_kernel_bytes)
  struct KernelBytes {
    double rd = 0.0, wr = 0.0, carry = 0.0;
    char name[64] = {0};
  } kbytes[kNumTags];
};

namespace pic1dp_host {

inline int ev_resolve(pic1dp_ctx *c) {
  if (c->ev_used == 0) return 0;
  HIP_TRY(hipStreamSynchronize(c->st));
  for (size_t i = 0; i < c->ev_used; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->evpool[i].a, c->evpool[i].b));
    if (c->evpool[i].head) {
      c->head_ms[c->evpool[i].tag] += ms;
      c->head_n[c->evpool[i].tag] += 1;
    } else {
      c->acc_ms[c->evpool[i].tag] += ms;
      c->acc_n[c->evpool[i].tag] += 1;
    }
  }
  c->ev_used = 0;
  return 0;
}

// bracket helper: records a start event on construction (if enabled) and the
// stop event in end(); pairs are resolved to milliseconds lazily (ev_resolve)
struct Span {
  pic1dp_ctx *c;
  long idx = -1;
  int rc = 0;
  Span(pic1dp_ctx *c_, int tag, bool on) : c(c_) {
    if (!on) return;
    // sampled timers: blocks of 64 consecutive spans of a timer id are timed, every timer_every-th block -- inside a block
    // everything is bracketed as with exact timers (a run's output steps, optimisation steps and plain steps enter in
    // their own proportions), the other blocks are only counted
    const int64_t seq = tag < 100 ? c->span_seen[tag]++ : 0;
    if (tag < 100 && ((seq >> 6) % c->timer_every) != 0) return;
    if (c->ev_used == c->evpool.size() && c->evpool.size() >= (1u << 16)) {
      // a long run that reads its timers only at the end: fold what has been recorded into
      // the accumulators (one stream synchronisation per 65 536 spans) and reuse the pool
      if ((rc = ev_resolve(c)) != 0) return;
    }
    if (c->ev_used == c->evpool.size()) {
      EvPair p{};
      // timing only (the pairs are read after a stream synchronisation): no system-scope fence at the event -- with it
      // every bracketed launch pays a write-back of the caches its markers live in (a 70 us step at the reference's default
      // size: +10-28 %), and the time it reports contains that write-back
      // (A/B at the reference's default size: profiles/r05/experiments, tools/ab_event_fence.sh of that round)
      auto make = [](hipEvent_t *e) {
        if (hipEventCreateWithFlags(e, hipEventDisableSystemFence) == hipSuccess) return true;
        (void)hipGetLastError();
        return hipEventCreate(e) == hipSuccess;
      };
      if (!make(&p.a) || !make(&p.b)) {
        rc = fail(PIC1DP_ERR_HIP, "hipEventCreate failed");
        return;
      }
      c->evpool.push_back(p);
    }
    idx = static_cast<long>(c->ev_used++);
    c->evpool[idx].tag = tag;
    c->evpool[idx].head = tag < 100 && c->timer_every > 1 && seq < 64;
    if (hipEventRecord(c->evpool[idx].a, c->st) != hipSuccess) rc = fail(PIC1DP_ERR_HIP, "hipEventRecord failed");
  }
  int end() {
    if (idx >= 0 && hipEventRecord(c->evpool[idx].b, c->st) != hipSuccess)
      return fail(PIC1DP_ERR_HIP, "hipEventRecord failed");
    return rc;
  }
};

// ---- capi.cpp ----
int ensure_second_set(pic1dp_ctx *c);     // the second slab (RK ping-pong set; re-packing target of the optimiser)
int put_range(pic1dp_ctx *c, double *arr, int64_t off, const double *host, int64_t cnt);
int get_range(pic1dp_ctx *c, const double *arr, int64_t off, double *host, int64_t cnt);
// ---- capi_step.cpp (the hot path and the state machine of the lazy call sites) ----
int require_loaded(pic1dp_ctx *c);        // markers loaded, no charge_local pending; memory as eager calls would have left it
int materialize(pic1dp_ctx *c);           // a noted push becomes memory
int materialize_cd(pic1dp_ctx *c);        // what collect_charge left to the next solve_field becomes field_chargeden
int rebuild_half_step_chargeden(pic1dp_ctx *c);  // the whole vector where only the kept mode's content was formed
void field_written(pic1dp_ctx *c, bool by_solve);  // d_E changed: versions, what a noted push may still assume
size_t step_lds_bytes(int nx, bool full);  // dynamic LDS of a whole-step kernel
// ---- capi_comm.cpp ----
int allreduce_charge(pic1dp_ctx *c);
int allreduce_doubles(pic1dp_ctx *c, double *d, size_t n);
int reduce_charge(pic1dp_ctx *c);
bool xchg_active(const pic1dp_ctx *c);
XchgArgs next_xchg_args(pic1dp_ctx *c);
int xchg_check(pic1dp_ctx *c);
void comm_release(pic1dp_ctx *c);         // communicator and exchange mappings, for destroy
int set_call_state(pic1dp_ctx *c, Seq seq, Owed owed);  // the one writer of the call sites' state: refuses pairs that cannot occur
int set_seq(pic1dp_ctx *c, Seq seq);
int set_owed(pic1dp_ctx *c, Owed owed);
int settle_half_pair(pic1dp_ctx *c);      // call sites: the half-step field the pair solve left aside becomes field_electric (ctx.hpp half_pair)
int settle_field_view(pic1dp_ctx *c);     // ... for readers of the field only
int adopt_half_field(pic1dp_ctx *c);
void optimize_release(pic1dp_ctx *c);     // the optimisation events' workers (streams, pinned and device staging), for destroy
// ---- capi_optimize.cpp ----
void optimize_due_at(const pic1dp_ctx *c, double time0, bool due[3]);  // which events a step starting at time0 fires
void optimize_due(const pic1dp_ctx *c, bool due[3]);
bool optimize_due_any(const pic1dp_ctx *c);
// ---- capi_diag.cpp ----
int diag_buffers(pic1dp_ctx *c);
int diag_max_blocks(const pic1dp_ctx *c);
const DistGeom &dist_geom(pic1dp_ctx *c);
int pinned(pic1dp_ctx *c, size_t ndoubles, double **out);   // the context's pinned staging with room for ndoubles
size_t dist_len(const pic1dp_input &in);

}  // namespace pic1dp_host
