// capi_step.cpp -- the hot path behind the C ABI of include/pic1dp_hip.h: the three call sites and their lazy state
// machine (push / collect_charge / solve_field), the whole-step kernels' sequencing (pic1dp_hip_step), the
// prediction of the half-step charge, the solve inside the marker launch, and the split-phase deposit.
// Context, transfers, field access, timers: capi.cpp; the other units: ctx.hpp.
#include "ctx.hpp"

namespace {

LaunchCfg particle_launch(const pic1dp_ctx *c, int64_t np, bool with_E, bool with_rho) {
  const int nx = c->in.nx;
  LaunchCfg lc{};
  lc.lds = sizeof(double) * ((with_E ? static_cast<size_t>((nx + 2) & ~1) : 0) +
                             (with_rho ? static_cast<size_t>(nx) + 1 : 0));  // + guard cell
  int by_lds = lc.lds ? static_cast<int>(kCuLds / (lc.lds + kStaticLds)) : 8;
  if (by_lds < 1) by_lds = 1;
  int threads = c->threads_req > 0 ? c->threads_req : 512;
  if (c->threads_req <= 0 && by_lds * threads < 2048) threads = 1024;
  int bpc = 2048 / threads;
  if (bpc > by_lds) bpc = by_lds;
  if (c->bpc_req > 0) bpc = c->bpc_req < by_lds ? c->bpc_req : by_lds;
  if (bpc < 1) bpc = 1;
  int64_t blocks = static_cast<int64_t>(c->num_cu) * bpc;
  const int64_t need = ((np >> 1) + threads - 1) / threads;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  lc.threads = threads;
  lc.blocks = static_cast<int>(blocks);
  return lc;
}

PushArgs make_push_args(pic1dp_ctx *c, int isp, int irk, const double *E) {
  Species &S = c->sp[isp];
  PushArgs a{};
  a.src = S.set[c->cur];
  a.base = irk == 2 ? S.set[1 - c->cur] : S.set[c->cur];
  a.dst = S.set[1 - c->cur];
  a.p = S.p;
  a.E = E ? E : c->d_E;
  a.rho = S.rho;
  a.np = S.np;
  a.dt = irk == 1 ? 0.5 * c->in.dt : c->in.dt;  // src/pic1dp_interaction.F90:179,192
  a.g = c->grid;
  a.s = S.sc;
  a.iptcldist = c->in.iptcldist;
  a.deltaf = c->in.deltaf;
  a.linear = c->in.linear;
  a.irk = irk;
  return a;
}

int enqueue_push(pic1dp_ctx *c, int irk, bool fused, const double *E = nullptr) {
  if (int rc = ensure_second_set(c)) return rc;
  c->state_version++;
  for (int s = 0; s < c->in.nspecies; ++s) {
    PushArgs a = make_push_args(c, s, irk, E);
    if (a.np <= 0) continue;
    LaunchCfg lc = particle_launch(c, a.np, true, fused);
    {  // irk 1 reads x, v, p (+ w); irk 2 also the RK base x (+ v) (+ w); both write x (+ v) (+ w)
      pic1dp_ctx::KernelBytes &kb = c->kbytes[fused ? kTagFused : kTagPush];
      const double pushed = 8.0 * (1 + (c->in.linear ? 0 : 1) + (c->in.deltaf ? 1 : 0));
      kb.rd = 8.0 * (3 + (c->in.deltaf ? 1 : 0)) + (irk == 2 ? pushed : 0.0);
      kb.wr = pushed;
      kb.carry = 0.0;
      std::snprintf(kb.name, sizeof kb.name, "%s", fused ? "k_push<FUSED>" : "k_push");
    }
    Span tm(c, PIC1DP_IWT_PUSH_PARTICLE, c->timers_on);
    Span ks(c, fused ? kTagFused : kTagPush, c->stats_on);
    HIP_TRY(launch_push(a, fused, lc, c->st));
    if (int rc = ks.end()) return rc;
    if (int rc = tm.end()) return rc;
  }
  c->cur = 1 - c->cur;
  return 0;
}

int enqueue_deposit(pic1dp_ctx *c) {
  c->state_version++;  // the wrap is stored back into x
  for (int s = 0; s < c->in.nspecies; ++s) {
    Species &S = c->sp[s];
    if (S.np <= 0) continue;
    double *x = S.set[c->cur].x;
    const double *q = c->in.deltaf ? S.set[c->cur].w : S.p;  // :84-91
    LaunchCfg lc = particle_launch(c, S.np, false, true);
    {
      pic1dp_ctx::KernelBytes &kb = c->kbytes[kTagDeposit];
      kb.rd = 16.0, kb.wr = 8.0, kb.carry = 0.0;  // x, q read; wrapped x written
      std::snprintf(kb.name, sizeof kb.name, "k_deposit");
    }
    Span ks(c, kTagDeposit, c->stats_on);
    HIP_TRY(launch_deposit(x, q, S.rho, S.np, c->grid, lc, c->st));
    if (int rc = ks.end()) return rc;
  }
  return 0;
}

}  // namespace


static bool step_recompute_ok(const pic1dp_ctx *c);
static int step_particles(pic1dp_ctx *c, bool full, const double *E0, const double *Eh, bool diag = false,
                          bool pred = false, bool tail_ok = false);
static bool predict_capable(const pic1dp_ctx *c);
static LaunchCfg step_launch(const pic1dp_ctx *c, int64_t np, bool full);
static LaunchCfg pred_launch(const pic1dp_ctx *c, int64_t np, bool priv, int64_t *resident);
static bool output_follows(const pic1dp_ctx *c);
static bool output_follows_at(const pic1dp_ctx *c, int32_t itime0, double time0);
static bool diag_in_step(const pic1dp_ctx *c);
static int finish_pending_solve(pic1dp_ctx *c);
static int solve_phase(pic1dp_ctx *c, double *Eout, bool record, bool pred);
static bool pred_usable(const pic1dp_ctx *c);
static int pred_to_chargeden(pic1dp_ctx *c, const FieldArgs &f, bool defer);

// ---------------------------------------------------------------------------
// hot path
// (the entry points are declared extern "C" in include/pic1dp_hip.h; their definitions here inherit that linkage)
// ---------------------------------------------------------------------------
// field_chargeden (and d_charge, and zeroed accumulators) as collect_charge would have left them at once
int pic1dp_host::materialize_cd(pic1dp_ctx *c) {
  const Owed pending = c->owed;
  if (pending == Owed::AdoptHalfField) return 0;  // (a half-step FIELD waiting to be adopted by solve_field: no charge to settle)
  if (pending == Owed::Nothing) return 0;
  if (int rc = set_owed(c, Owed::Nothing)) return rc;
  if (pending == Owed::PredSums) {  // the six sums of a predicted push(1): the kept mode's content of chargeden, directly
    HIP_TRY(launch_pred_chargeden(c->fa, c->pred_tab, c->d_pred, nullptr, c->st));
    return 0;
  }
  if (pending == Owed::PredTiles) HIP_TRY(launch_pred_combine(c->fa, c->d_pred, c->in.nmode, c->st));
  HIP_TRY(launch_chargeden(c->fa, pending == Owed::SumScale, c->st));
  return 0;
}

// THE place the call sites' state is written: a pair that cannot occur (ctx.hpp kCallStateLegal) is an internal error
// here, at the call that would have produced it, instead of a wrong field some calls later
int pic1dp_host::set_call_state(pic1dp_ctx *c, Seq seq, Owed owed) {
  if (seq >= Seq::N || owed >= Owed::N || !kCallStateLegal[static_cast<int>(seq)][static_cast<int>(owed)])
    return fail(PIC1DP_ERR_STATE, "internal: call-site state (%d, %d) cannot occur (from (%d, %d))", static_cast<int>(seq),
                static_cast<int>(owed), static_cast<int>(c->seq), static_cast<int>(c->owed));
  if (seq != Seq::Clean && !c->lazy_calls) return fail(PIC1DP_ERR_STATE, "internal: a push noted by eager call sites");
  c->seq = seq;
  c->owed = owed;
  return 0;
}
int pic1dp_host::set_seq(pic1dp_ctx *c, Seq seq) { return set_call_state(c, seq, c->owed); }
int pic1dp_host::set_owed(pic1dp_ctx *c, Owed owed) { return set_call_state(c, c->seq, owed); }

// checks only: for the call sites that take part in the lazy scheme themselves
static int require_loaded_keep_lazy(pic1dp_ctx *c) {
  if (!c->loaded) return fail(PIC1DP_ERR_STATE, "no particles: call particle_load or particles_upload first");
  if (c->charge_pending) return fail(PIC1DP_ERR_STATE, "charge_local is waiting for charge_reduced");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(PIC1DP_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  return materialize_cd(c);
}

// every other entry point that reads or writes markers or charge accumulators:
// memory first becomes what the eager calls would have left
int pic1dp_host::require_loaded(pic1dp_ctx *c) {
  if (int rc = require_loaded_keep_lazy(c)) return rc;
  return materialize(c);
}

// ---------------------------------------------------------------------------
// Lazy call sites.  The reference driver calls push(irk), collect_charge,
// solve_field per sub-step (src/pic1dp.F90:80-89).  Run one kernel per call, that
// sequence streams 136 B + 2 x 24 B per marker and step; the whole-step kernels
// need 88 B.  So push() only notes the request and the collect_charge() that
// follows runs k_step_half (after push(1): deposit of the half-step state,
// nothing stored) or k_step_full (after push(2): the whole RK2 step in place).
//   LZ_CLEAN --push(1)--> LZ_PUSH1 --collect--> LZ_HALF --push(2)--> LZ_PUSH2 --collect--> LZ_CLEAN
// Memory then lags behind what the reference would hold (the half-step state is
// never written).  Every other entry point that looks at or changes markers,
// accumulators or the field a noted push depends on first calls materialize(),
// which runs the ordinary kernels (k_push with the right field, wrap) so that
// memory is bit for bit what the eager calls would have left; results of the lazy
// path itself are the whole-step path's (tests: test_lazy_call_sites_*).
// ---------------------------------------------------------------------------
static bool lazy_ok(const pic1dp_ctx *c) {
  return c->lazy_calls && step_recompute_ok(c) && !optimize_due_any(c);
}

// wrap of x stored back, as the deposit of collect_charge does, charge discarded
static int enqueue_wrap_only(pic1dp_ctx *c) {
  for (int s = 0; s < c->in.nspecies; ++s) {
    Species &S = c->sp[s];
    if (S.np <= 0) continue;
    const double *q = c->in.deltaf ? S.set[c->cur].w : S.p;
    LaunchCfg lc = particle_launch(c, S.np, false, true);
    HIP_TRY(launch_deposit(S.set[c->cur].x, q, c->d_rho_dummy, S.np, c->grid, lc, c->st));
  }
  return 0;
}

// Call sites (ctx.hpp Seq, the Pair states): the half-step field the pair solve left in d_Ehn / d_mode_h, and the step-start
// field still in field_electric, put where the eager calls would have them -- d_E0 <- E, and, once the host has called
// solve_field for the half step, E <- the half-step field with its kept modes.  Everything that looks at the field, or
// leaves the sequence push(1), collect_charge, solve_field, push(2), collect_charge, comes through here first.
int pic1dp_host::settle_half_pair(pic1dp_ctx *c) {
  if (!pair_of(c->seq)) return 0;
  const size_t nx = c->in.nx;
  HIP_TRY(hipMemcpyAsync(c->d_E0, c->d_E, sizeof(double) * nx, hipMemcpyDeviceToDevice, c->st));
  const bool solved = solved_of(c->seq);
  if (int rc = set_seq(c, unpaired(c->seq))) return rc;
  // solve_field has been called for the half step: field_electric has to BE the half-step field now (field_version was
  // bumped when it was called); not yet called: it stays owed (AdoptHalfField) and copies when it comes
  if (solved) return adopt_half_field(c);
  return 0;
}
// for readers of the field only: nothing to do until the host has called solve_field for the half step -- until then
// field_electric is the step-start field, which is what d_E holds
int pic1dp_host::settle_field_view(pic1dp_ctx *c) { return solved_of(c->seq) ? settle_half_pair(c) : 0; }
// field_electric and its kept modes <- the half-step field the pair solve left in d_Ehn / d_mode_h, and field_chargeden
// <- the kept mode's content of the half-step charge density it left in d_cd_h (what collect_charge leaves there when it
// is served from the six sums, Owed::PredSums): a solve_field that solves from field_chargeden again reproduces the half-step
// field, as it does after eager calls
int pic1dp_host::adopt_half_field(pic1dp_ctx *c) {
  const size_t nx = c->in.nx, nm = c->in.nmode;
  HIP_TRY(hipMemcpyAsync(c->d_chargeden, c->d_cd_h, sizeof(double) * nx, hipMemcpyDeviceToDevice, c->st));
  HIP_TRY(hipMemcpyAsync(c->d_E, c->d_Ehn, sizeof(double) * nx, hipMemcpyDeviceToDevice, c->st));
  HIP_TRY(hipMemcpyAsync(c->fa.mode_re, c->d_mode_h, sizeof(double) * nm, hipMemcpyDeviceToDevice, c->st));
  HIP_TRY(hipMemcpyAsync(c->fa.mode_im, c->d_mode_h + nm, sizeof(double) * nm, hipMemcpyDeviceToDevice, c->st));
  return 0;
}

int pic1dp_host::materialize(pic1dp_ctx *c) {
  if (int rc = settle_half_pair(c)) return rc;
  const int lz = lz_of(c->seq);
  if (lz == LZ_CLEAN) return 0;
  if (int rc = set_seq(c, Seq::Clean)) return rc;
  if (lz == LZ_PUSH1) return enqueue_push(c, 1, false);           // the field has not changed since
  // LZ_HALF / LZ_PUSH2: push(1) saw d_E0; its deposit wrapped x
  if (int rc = enqueue_push(c, 1, false, c->d_E0)) return rc;
  if (int rc = enqueue_wrap_only(c)) return rc;
  if (lz == LZ_PUSH2) return enqueue_push(c, 2, false);
  return 0;
}

// the deposit of collect_charge / charge_local into the species accumulators:
// the whole-step kernel of a noted push, or the plain wrap + deposit
static int deposit_or_step(pic1dp_ctx *c) {
  if (c->seq == Seq::Push1) {
    HIP_TRY(hipMemcpyAsync(c->d_E0, c->d_E, sizeof(double) * c->in.nx, hipMemcpyDeviceToDevice, c->st));
    if (int rc = step_particles(c, false, c->d_E, c->d_Eh)) return rc;
    return set_seq(c, Seq::Half);
  }
  if (lz_of(c->seq) == LZ_PUSH2) {
    if (c->seq == Seq::Push2Pair)  // (push(2) without the solve_field of the half step: it sees the old field)
      if (int rc = settle_half_pair(c)) return rc;
    c->state_version++;
    const bool diag = diag_in_step(c) && output_follows(c);
    if (c->seq == Seq::Push2PairSolved) {  // the half-step field came out of the previous step's pair solve: E0 is still in d_E
      std::swap(c->d_Eh, c->d_Ehn);
      c->eh_modes = 2;  // its kept mode: d_mode_h
      if (int rc = step_particles(c, true, c->d_E, c->d_Eh, diag, !diag)) return rc;
      return set_seq(c, Seq::Clean);
    }
    // Eh = d_E: the kept modes describe it when the mode-filter solve wrote it last
    c->eh_modes = (c->field_solver == 0 && c->modes_field_version == c->field_version) ? 1 : 0;
    if (int rc = step_particles(c, true, c->d_E0, c->d_E, diag, !diag)) return rc;
    return set_seq(c, Seq::Clean);
  }
  if (int rc = materialize(c)) return rc;
  return enqueue_deposit(c);
}

int pic1dp_hip_collect_charge(pic1dp_ctx *c) {
  CHECK_CTX(c);
  if (int rc = require_loaded_keep_lazy(c)) return rc;
  // a whole-step kernel run for a noted push is booked under "push particle"
  // (step_particles); "collect charge" then covers the reduction and scaling only
  // after a noted push(1) whose charge the previous step's kernel has predicted (k_step_one): no
  // pass over the markers at all -- combine, reduce, scale
  if (pair_of(c->seq) && c->seq != Seq::Push2PairSolved)  // out of sequence: memory as the eager calls leave it
    if (int rc = settle_half_pair(c)) return rc;
  // after a noted push(1) whose half-step FIELD the previous solve_field has already solved (the pair, below): nothing
  // to launch at all
  // (the six sums of one kept mode: with the tiles collect_charge owes the host the whole half-step charge density)
  if (c->seq == Seq::Push1 && c->call_pair && c->lazy_calls && c->lay.nranks == 1 && c->comm == nullptr && predict_capable(c) &&
      c->pred_kind == 2 && c->in.nmode == 1 && c->eh_version == c->state_version && c->eh_field_version == c->field_version) {
    c->cd_kept_mode_only = true;  // (field_chargeden is not the half step's: asking for it rebuilds, get_field)
    return set_call_state(c, Seq::HalfPair, Owed::AdoptHalfField);  // what the solve_field that follows has to do: adopt the half-step field
  }
  if (c->seq == Seq::Push1 && pred_usable(c)) {
    HIP_TRY(hipMemcpyAsync(c->d_E0, c->d_E, sizeof(double) * c->in.nx, hipMemcpyDeviceToDevice, c->st));
    if (int rc = set_call_state(c, Seq::Half, Owed::Nothing)) return rc;   // (a stale AdoptHalfField ends here)
    Span tm(c, PIC1DP_IWT_COLLECT_CHARGE, c->timers_on);
    if (int rc = pred_to_chargeden(c, c->fa, c->lazy_calls != 0)) return rc;
    return tm.end();
  }
  c->cd_kept_mode_only = false;  // a deposit follows: the whole vector again
  if (c->owed == Owed::AdoptHalfField)   // (left over by an inspection that settled a pair before its solve_field: void now)
    if (int rc = set_owed(c, Owed::Nothing)) return rc;
  const bool noted = lz_of(c->seq) == LZ_PUSH1 || lz_of(c->seq) == LZ_PUSH2;
  if (noted)
    if (int rc = deposit_or_step(c)) return rc;
  Span tm(c, PIC1DP_IWT_COLLECT_CHARGE, c->timers_on);
  if (!noted)
    if (int rc = deposit_or_step(c)) return rc;
  const bool multi = c->lay.nranks > 1 || c->comm != nullptr;
  if (multi)
    if (int rc = reduce_charge(c)) return rc;
  if (c->lazy_calls) {  // the species sum (one rank) and the scaling: in the launch of the solve_field that follows
    if (int rc = set_owed(c, multi ? Owed::Scale : Owed::SumScale)) return rc;
  } else {
    HIP_TRY(launch_chargeden(c->fa, !multi, c->st));
  }
  return tm.end();
}

// d_E has been written: by the mode-filter solve (the kept modes describe it) or by something else
void pic1dp_host::field_written(pic1dp_ctx *c, bool by_solve) {
  c->field_version++;
  if (by_solve && c->field_solver == 0) c->modes_field_version = c->field_version;
}

// field_solve_electric: the reference's mode-filter solve, optionally followed by
// the finite-difference alternative overwriting E (field_solver = 1)
static int enqueue_field_solve(pic1dp_ctx *c, FieldArgs f, bool with_local, bool from_chargeden) {
  double *hist = f.history;
  if (c->field_solver == 1) f.history = nullptr;
  HIP_TRY(launch_field_solve(f, with_local, from_chargeden, c->st));
  if (c->field_solver == 1)
    HIP_TRY(launch_field_fd(f.chargeden, f.E, hist, f.nx, f.lx, f.dnx, c->st));
  return 0;
}

int pic1dp_hip_set_field_solver(pic1dp_ctx *c, int32_t kind) {
  CHECK_CTX(c);
  if (kind != 0 && kind != 1) return fail(PIC1DP_ERR_ARG, "field solver must be 0 (reference mode filter) or 1 (finite differences)");
  if (kind == 1 && (c->in.nx < 3 || c->in.nx > 4096))
    return fail(PIC1DP_ERR_ARG, "the finite-difference solver needs 3 <= nx <= 4096");
  if (c->owed == Owed::AdoptHalfField && kind != c->field_solver) {  // a half-step field of the OTHER solver waits to be adopted:
    HIP_TRY(hipSetDevice(c->device));                                // the half-step charge is deposited for real instead
    if (int rc = rebuild_half_step_chargeden(c)) return rc;
    if (int rc = set_owed(c, Owed::Nothing)) return rc;
  }
  c->field_solver = kind;
  return 0;
}

int pic1dp_hip_solve_field(pic1dp_ctx *c) {
  CHECK_CTX(c);
  HIP_TRY(hipSetDevice(c->device));
  if (c->seq == Seq::HalfPair) {  // the half-step field is solved already (the pair below): nothing to launch
    if (int rc = set_call_state(c, Seq::HalfPairSolved, Owed::Nothing)) return rc;
    c->call_pair_skips++;
    field_written(c, true);  // from now on field_electric IS the half-step field, as far as anybody can tell
    return 0;
  }
  if (int rc = settle_half_pair(c)) return rc;
  // a noted push has to see the field of its own moment
  if (lz_of(c->seq) == LZ_PUSH1 || lz_of(c->seq) == LZ_PUSH2)
    if (int rc = materialize(c)) return rc;
  Span tm(c, PIC1DP_IWT_FIELD_ELECTRIC, c->timers_on);
  FieldArgs f = c->fa;
  const Owed pending = c->owed;  // what collect_charge left to this launch
  if (int rc = set_owed(c, Owed::Nothing)) return rc;
  // One rank, behind the collect_charge of push(2) whose kernel has predicted the next half-step charge: BOTH fields in
  // one launch, as pic1dp_hip_step solves them -- the next step's push(1), collect_charge, solve_field then launch nothing
  // and a time step through the three call sites is two launches (round 5; three and a copy before)
  if (pending == Owed::SumScale && c->call_pair && c->lazy_calls && c->field_solver == 0 && c->seq == Seq::Clean && pred_usable(c) &&
      c->pred_kind == 2 && c->in.nmode == 1) {
    PairArgs pa{c->d_pred, c->d_Ehn, c->d_mode_h, c->d_cd_h, nullptr, c->pred_kind, c->pred_tab, 0};
    HIP_TRY(launch_field_solve_pair(f, pa, nullptr, c->st));
    field_written(c, true);
    c->pred_version = 0;  // consumed
    c->eh_version = c->state_version;
    c->eh_field_version = c->field_version;
    return tm.end();
  }
  if (pending == Owed::AdoptHalfField) {  // (the fast path above was left by an inspection in between: the field is adopted by copying)
    if (int rc = adopt_half_field(c)) return rc;
  } else if (pending == Owed::PredTiles && c->field_solver == 0) {
    HIP_TRY(launch_field_solve_pred(f, c->d_pred, c->in.nmode, c->st));
  } else if (pending == Owed::PredSums && c->field_solver == 0) {
    HIP_TRY(launch_field_solve_pred_sums(f, c->pred_tab, c->d_pred, c->st));
  } else {
    if (pending == Owed::PredSums) HIP_TRY(launch_pred_chargeden(c->fa, c->pred_tab, c->d_pred, nullptr, c->st));
    if (pending == Owed::PredTiles) HIP_TRY(launch_pred_combine(c->fa, c->d_pred, c->in.nmode, c->st));
    if (int rc = enqueue_field_solve(c, f, pending == Owed::SumScale, pending == Owed::Nothing || pending == Owed::PredSums)) return rc;
  }
  field_written(c, true);
  return tm.end();
}

int pic1dp_hip_push(pic1dp_ctx *c, int32_t irk) {
  CHECK_CTX(c);
  if (irk != 1 && irk != 2) return fail(PIC1DP_ERR_ARG, "irk must be 1 or 2");
  if (int rc = require_loaded_keep_lazy(c)) return rc;
  if (irk == 1 && c->seq == Seq::Clean && lazy_ok(c)) return set_seq(c, Seq::Push1);
  if (irk == 2 && lz_of(c->seq) == LZ_HALF) return set_seq(c, push2_noted(c->seq));
  if (int rc = materialize(c)) return rc;
  return enqueue_push(c, irk, false);
}

static int substep_impl(pic1dp_ctx *c, int irk, bool record) {
  c->cd_kept_mode_only = false;
  if (irk == 2 && optimize_due_any(c)) {
    // src/pic1dp.F90:80-88: push, particle_optimize, collect_charge -- the pushed
    // state has to exist in memory for the host-side optimisation
    if (int rc = enqueue_push(c, irk, false)) return rc;
    if (int rc = pic1dp_hip_particle_optimize(c, irk, nullptr)) return rc;
    if (int rc = enqueue_deposit(c)) return rc;
  } else if (int rc = enqueue_push(c, irk, true)) {
    return rc;
  }
  const bool multi = c->lay.nranks > 1 || c->comm != nullptr;
  const bool fused_xchg = xchg_active(c) && c->field_solver == 0;  // exchange inside the solve's launch
  if (multi && !fused_xchg)
    if (int rc = reduce_charge(c)) return rc;
  Span tm(c, PIC1DP_IWT_FIELD_ELECTRIC, c->timers_on);
  FieldArgs f = c->fa;
  if (record && c->hist_count < kHistCap) f.history = c->d_hist + c->hist_count++;
  if (fused_xchg) {
    HIP_TRY(launch_field_solve_xchg(f, next_xchg_args(c), c->st));
  } else if (int rc = enqueue_field_solve(c, f, !multi, false)) {
    return rc;
  }
  field_written(c, true);
  return tm.end();
}

int pic1dp_hip_substep(pic1dp_ctx *c, int32_t irk) {
  CHECK_CTX(c);
  if (irk != 1 && irk != 2) return fail(PIC1DP_ERR_ARG, "irk must be 1 or 2");
  if (int rc = require_loaded(c)) return rc;
  return substep_impl(c, irk, false);
}

// LDS bytes of the whole-step kernels: E0 tile, Eh tile (full only), rho tile
size_t pic1dp_host::step_lds_bytes(int nx, bool full) {
  const size_t ne = static_cast<size_t>((nx + 2) & ~1);
  return sizeof(double) * ((full ? 2 : 1) * ne + ((static_cast<size_t>(nx) + 2) & ~static_cast<size_t>(1)) +
                           2);  // (+ the drawn chunks' counter, 16-byte slot)
}

static bool step_recompute_ok(const pic1dp_ctx *c) {
  return c->step_mode == 0 && step_lds_bytes(c->in.nx, true) <= PARTICLE_LDS_CAP;
}

// Grid of a marker kernel: `resident` workgroups fill the CUs; with a grid of exactly that size the kernel ends
// when its slowest workgroup does, and the CUs do not all stream at the same rate.  A grid of several times that
// size lets the CUs that finish early take more of the work (tools/ab_one_shapes.sh: k_step_one at 1e8 markers
// 1.290 -> 1.238 ms with four times the resident workgroups) -- as long as a workgroup's share of the markers
// dwarfs what it pays once (staging and flushing tiles of nx cells): about 48 markers per cell, at most x4.
// Only where two workgroups share a CU (while one stages or flushes the other streams; alone on its CU a
// workgroup's turn-over idles it: k_step_sums at nx 4096, 0.935 -> 1.004 ms with twice the grid), and not for
// k_step_full, which measures 1-3 % slower that way (k_step_half 5 % faster; tools/ab_osub.sh).
// PIC1DP_OSUB=n insists on a factor (1: the resident grid).
static int64_t oversubscribed(const pic1dp_ctx *c, int64_t np, int64_t resident, bool allow = true) {
  if (c->bpc_req > 0) return resident;  // a launch shape asked for by hand is taken literally
  int64_t f = c->osub_req;
  if (f <= 0 && !allow) return resident;
  if (f <= 0) {
    const int64_t per_wg = static_cast<int64_t>(48) * c->in.nx;
    f = (np / per_wg + resident / 2) / std::max<int64_t>(resident, 1);
  }
  f = std::max<int64_t>(1, std::min<int64_t>(f, c->osub_req > 0 ? 64 : 4));
  return resident * f;
}

static LaunchCfg step_launch(const pic1dp_ctx *c, int64_t np, bool full) {
  LaunchCfg lc{};
  lc.lds = step_lds_bytes(c->in.nx, full);
  int by_lds = static_cast<int>(kCuLds / (lc.lds + kStaticLds));
  if (by_lds < 1) by_lds = 1;
  // two workgroups of 768 threads per CU (24 waves): measured inside one process
  // (tools/ab_launch.py) best or within 1 % of best from 6.4e6 to 1e8 markers --
  // fewer workgroups mean fewer LDS stagings and half as many global flush atomics
  // as four workgroups of 512, which cost k_step_half 25 % at 6.4e6 and 15 % at 2e7
  int threads = c->threads_req > 0 ? c->threads_req : 768;
  if (c->threads_req <= 0 && by_lds < 2) threads = 1024;
  int bpc = c->threads_req > 0 ? 2048 / threads : (threads == 768 ? 2 : 1);
  if (bpc > by_lds) bpc = by_lds;
  if (c->bpc_req > 0) bpc = c->bpc_req < by_lds ? c->bpc_req : by_lds;
  if (bpc < 1) bpc = 1;
  int64_t blocks = oversubscribed(c, np, static_cast<int64_t>(c->num_cu) * bpc, !full && bpc >= 2);
  const int64_t need = ((np >> 1) + threads - 1) / threads;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  lc.threads = threads;
  lc.blocks = static_cast<int>(blocks);
  return lc;
}

// does output_all follow the step that is being taken?  (src/pic1dp.F90:98-107 evaluated one step ahead)
static bool output_follows_at(const pic1dp_ctx *c, int32_t itime0, double time0) {  // (counters at the start of the step)
  const pic1dp_input &in = c->in;
  if (!(in.output_interval > 0.0)) return false;
  const double t = time0 + in.dt;
  if (itime0 + 1 >= in.ntime_max || t + kSqrtEps >= in.time_max) return true;
  return std::fmod(t + kSqrtEps, in.output_interval) < std::fmod(t + kSqrtEps - in.dt, in.output_interval);
}
static bool output_follows(const pic1dp_ctx *c) { return output_follows_at(c, c->itime, c->time); }

// Does a step that output_all follows take the diagnostics inside its marker kernel (k_step_full<DIAG>)?
// fuse_output 2: always.  1 (what a host that calls output_all at the reference's cadence asks for): only where the step
// is not a predicted one-pass step anyway.  Where it is, k_step_full<DIAG> costs the prediction -- the step after the
// output pays a first-sub-step pass again (k_step_half): 10 steps + output_all at 1e8 markers 10.57 ms against 9.5 for
// ten plain steps --, whereas k_step_one followed by the diagnostics' own pass (k_ptcldist: it changes no marker, the
// prediction stays valid) costs that pass alone (profiles/r05/experiments/diag_bench.log).
static bool diag_in_step(const pic1dp_ctx *c) {
  return c->fuse_output == 2 || (c->fuse_output == 1 && !predict_capable(c));
}

// One pass per step (kernels_step.hip k_step_one) needs: the mode-filter solver (the kept modes must
// describe E), few kept modes, and LDS for E0, Eh, the mode tables and the four accumulators
static bool predict_capable(const pic1dp_ctx *c) {
  return c->predict && c->pred_kind != 0 && c->d_pred && c->field_solver == 0 && step_recompute_ok(c);
}
static size_t pred_doubles(const pic1dp_ctx *c) {
  return c->pred_kind == 2 ? 8 * PRED_SUM_COPIES : static_cast<size_t>(c->in.nspecies) * (1 + 2 * c->in.nmode) * c->in.nx;
}

// the accumulator set the marker kernels deposit into from now on (d_rho_all / d_pred_all hold three)
static void use_accumulators(pic1dp_ctx *c, int idx) {
  c->acc_idx = idx;
  c->d_rho_sp = c->d_rho_all + static_cast<size_t>(idx) * c->rho_set_doubles;
  for (int s = 0; s < c->in.nspecies; ++s) c->sp[s].rho = c->d_rho_sp + static_cast<size_t>(s) * c->in.nx;
  c->fa.rho_sp = c->d_rho_sp;
  if (c->d_pred_all) c->d_pred = c->d_pred_all + static_cast<size_t>(idx) * c->pred_set_doubles;
}

// One launch per time step (kernels.hpp FusedSolve): may the solve of a step be left to the prologue of the next
// step's marker launch?  One rank (no charge sum between the launches), the six-sum prediction of one kept mode, the
// mode-filter solver, and partial chains of an npe-rank order that fit the prologue's scratch.
// And a grid of at most the resident workgroups: EVERY workgroup runs the solve in its prologue, side by side in
// one round; in an oversubscribed grid each round pays it again (1e8 markers / nx 1024, four rounds: 0.962 -> 0.990 ms
// per step, profiles/r04/experiments/ab_fused_solve.log).  What the fusion is worth where it applies: one dependency
// gap and the field launch's start-up, 1-2 us of a step (the solve itself is a chain of dependent round trips either way).
static bool fuse_capable(const pic1dp_ctx *c) {
  const bool multi = c->lay.nranks > 1 || c->comm != nullptr;
  if (!(c->fuse_solve && !multi && c->pred_kind == 2 && c->in.nmode == 1 && c->field_solver == 0 && c->fa.npe <= 32 &&
        predict_capable(c)))
    return false;
  // ... and serial forward sums that are short enough.  The solve costs inside a marker launch what it costs in its own: a
  // row of dependent round trips and the chain in the reference's order (12 cycles a term at the marker kernel's
  // clock).  With a chain of 1024 terms the launch it saves is level or slightly behind (1.25e7 markers / nx 1024:
  // 0.1471 against 0.1460 ms per step); with 192 terms it is 1 % ahead at 6.4e6 markers and 11 % at 2e5
  // (profiles/r04/experiments/ab_fused_solve.log, ab_small_knobs.log).  PIC1DP_FUSE_SOLVE=2 fuses whatever the length.
  // Round 4, later: with the sums through the matrix unit (one-rank order, FieldArgs::chain_mfma: a third of the chain's
  // time) the fused launch is ahead at nx 1024 too (1.25e7 markers: 0.1395-0.1416 against 0.1420-0.1429 ms per step) and
  // level to +0.3 % at nx 4096 (ab_fused_solve_mfma_chain.log; left unfused: its kernel stays the plain stream the profiles
  // price): the length that counts is the chain's cost in terms.
  const int chain_terms = (c->fa.npe == 1 && c->fa.chain_mfma) ? c->in.nx / 3 : c->in.nx / std::max(1, c->fa.npe);
  if (c->fuse_solve != 2 && chain_terms > 1024) return false;
  const bool priv = c->pred_private && c->threads_req <= 0;
  for (int s = 0; s < c->in.nspecies; ++s) {
    if (c->sp[s].np <= 0) continue;
    int64_t resident = 0;
    const LaunchCfg lc = pred_launch(c, c->sp[s].np, priv, &resident);
    // (the first species launched carries the solve; the prologue needs a first and a last wave of its own:
    // lean_forward_sums runs its chains in wave 0 while the last wave adds up the copies of the six sums)
    return lc.blocks <= resident && lc.threads >= 128 && lc.threads % 64 == 0;
  }
  return false;
}

// launch shape of the one-pass kernels (k_step_one, k_step_one<PRIV>, k_step_sums) for np markers; *resident: the
// workgroups that fill the CUs (the grid is that, or a multiple: oversubscribed())
static LaunchCfg pred_launch(const pic1dp_ctx *c, int64_t np, bool priv, int64_t *resident) {
  LaunchCfg lc{};
  lc.lds = priv ? step_one_private_lds_bytes(c->in.nx)
                : (c->pred_kind == 2 ? step_sums_lds_bytes(c->in.nx)
                                     : step_one_lds_bytes(c->in.nx, c->in.nmode));
  bool two = 2 * (lc.lds + kStaticLds) <= kCuLds;  // both workgroups resident: each also holds the static exp table
  int th2 = 768;
  int th1 = 1024;
  if (priv) {  // the private sums' slot stride is a compile-time constant: exactly that many threads
    th2 = th1 = STEP_PRIVATE_THREADS;
    if (STEP_PRIVATE_THREADS > 768) two = false;
  }
  if (c->pred_kind == 2 && !priv) {
    // k_step_sums keeps its registers: four waves per SIMD with the exp-bearing distributions (one
    // workgroup of 1024 per CU), eight with the others, which saturate the memory system with far fewer
    // (tools/ab_sums_shapes.sh: 1e8 markers, Maxwellian, nx 4096: 512 x 1 0.925 ms, 1024 x 1 0.965 ms)
    if (c->in.deltaf && (c->in.iptcldist == 2 || c->in.iptcldist == 3))
      two = false;
    else
      th1 = 512;
  }
  lc.threads = c->threads_req > 0 ? c->threads_req : (two ? th2 : th1);
  const int bpc = c->bpc_req > 0 ? c->bpc_req : (two ? 2 : 1);
  const int64_t need = ((np >> 1) + lc.threads - 1) / lc.threads;
  const int64_t res = static_cast<int64_t>(c->num_cu) * bpc;
  if (resident) *resident = res;
  lc.blocks = static_cast<int>(std::max<int64_t>(1, std::min(oversubscribed(c, np, res, bpc >= 2), need)));
  return lc;
}

// the particle kernel(s) of one sub-step of the whole-step path: E0 = field at the start of the step, Eh = field after
// the first sub-step (full only).
// full = true: the caller has bumped state_version for this step; the state the kernel READS is version - 1
// tail_ok (the step() path on several ranks): the last species' launch may pack / post this rank's charge in its tail
// (kernels.hpp StepTail) -- c->tail_done tells solve_phase what it did
static int step_particles(pic1dp_ctx *c, bool full, const double *E0, const double *Eh, bool diag, bool pred, bool tail_ok) {
  c->tail_done = 0;
  if (pred && (!full || diag || !predict_capable(c))) pred = false;
  const bool priv = c->pred_kind == 2 && c->pred_private && c->threads_req <= 0;  // k_step_one<PRIV>: Eh from its tile
  if (pred && c->pred_kind == 2 && !priv && c->eh_modes == 0) pred = false;  // k_step_sums forms Eh from its kept mode
  if (pred && c->pred_version != 0)  // a prediction nobody used: the accumulators start from zero
    HIP_TRY(hipMemsetAsync(c->d_pred, 0, sizeof(double) * pred_doubles(c), c->st));
  // the diagnostics of output_all inside k_step_full: when asked for, the LDS holds them, and the
  // tuning build of the marker loop is the default one
  if (diag) {
    const size_t need = step_lds_bytes(c->in.nx, true) +
                        step_diag_lds_bytes(c->in.nx, c->in.nx_opd, c->in.nv_opd);
    if (!full || c->in.nx_opd < 1 || c->in.nv_opd < 2 || need > PARTICLE_LDS_CAP) diag = false;
  }
  if (diag)
    if (int rc = diag_buffers(c)) return rc;
  // several ranks, the six sums of one kept mode, the mode-filter solver: the packing of this rank's charge for the sum
  // over ranks rides in the tail of the last marker launch (RCCL: one all-reduce follows; exchange: posted at once)
  int tail_mode = 0, tail_species = -1;
  if (tail_ok && pred && c->tail_on && c->pred_kind == 2 && c->in.nmode == 1 && c->field_solver == 0 &&
      (c->lay.nranks > 1 || c->comm != nullptr)) {
    tail_mode = xchg_active(c) ? 2 : (c->comm != nullptr ? 1 : 0);
    for (int s = 0; s < c->in.nspecies; ++s)
      if (c->sp[s].np > 0) tail_species = s;
    if (tail_species < 0) tail_mode = 0;
  }
  // x, v, w, p of all species against the 256 MiB Infinity Cache
  double state_bytes = 0.0;
  for (int s = 0; s < c->in.nspecies; ++s) state_bytes += 32.0 * static_cast<double>(c->sp[s].np);
  int stream_nt = state_bytes > (full ? c->nt_threshold_full : c->nt_threshold_half) ? 1 : 0;
  if (const char *e = tuning_env("PIC1DP_NT_FORCE")) {  // tuning build only, read per launch (tools/ab_nt.py)
    const int f = std::atoi(e);
    if (f == 0) stream_nt = 0;
    if (f == 1) stream_nt = 1;
    if (f == 2) stream_nt = full ? 0 : 1;
    if (f == 3) stream_nt = full ? 1 : 0;
  }
  for (int s = 0; s < c->in.nspecies; ++s) {
    Species &S = c->sp[s];
    if (S.np <= 0) continue;
    StepArgs a{};
    a.x = S.set[c->cur].x;
    a.v = S.set[c->cur].v;
    a.w = S.set[c->cur].w;
    a.p = S.p;
    a.E0 = E0;
    a.Eh = Eh;
    a.rho = S.rho;
    a.np = S.np;
    a.dt_half = 0.5 * c->in.dt;  // src/pic1dp_interaction.F90:179
    a.dt_full = c->in.dt;        // :192
    a.g = c->grid;
    a.s = S.sc;
    a.iptcldist = c->in.iptcldist;
    a.deltaf = c->in.deltaf;
    a.linear = c->in.linear;
    a.stream_nt = stream_nt;
    // the drawn chunk tail of every whole-step kernel: half a workgroup's chunks, all of them for k_step_full (two passes
    // per step at 1e8 markers: 0.913 -> 0.898 ms with 16/16 against 8/16, profiles/r05/experiments/ab_dyn_tail_other.log)
    a.dyn_tail = (full && !pred) ? c->dyn_tail_full : c->dyn_tail;
    // a species with general divisor constants and an exp-bearing f0 is FP64-issue-bound: its
    // -f0'/f0 at the step-start velocity goes from the first kernel to the second through
    // memory (8 B per marker) instead of being evaluated twice.  Measured at 1e8 markers
    // (tools/ab_pipe_carry.sh): bump-on-tail with T = 1.3, T2 = 0.7, m = 1.1 8.9e10 -> 9.85e10
    // updates/s; two-stream2 (one division fewer per exp pair) 1.05e11 either way, so only
    // bump-on-tail carries.  PIC1DP_CARRY=0 switches it off, 2 also carries for two-stream2.
    const uint64_t read_version = full ? c->state_version - 1 : c->state_version;
    const bool carry2 = c->carry != 0 && c->in.deltaf && !S.sc.pow2 && !S.sc.one_exp &&
                        (c->in.iptcldist == 3 || (c->carry == 2 && c->in.iptcldist == 2));
    if (carry2 && !pred) {
      if (!S.t2) HIP_TRY(hipMalloc(&S.t2, sizeof(double) * static_cast<size_t>(S.nalloc + 2)));
      // the second kernel may only load what the first one stored for these very markers
      if (!full || S.t2_version == read_version) a.t2 = S.t2;
      if (!full) S.t2_version = c->state_version;
    }
    LaunchCfg lc = step_launch(c, S.np, full);
    if (pred && c->fuse_args.on) {  // this launch's prologue solves the previous step's field (first species launched)
      const LaunchCfg fl = pred_launch(c, S.np, priv, nullptr);
      if (fl.threads < 128 || fl.threads % 64 != 0)  // what fuse_capable promised (checked at the launch too)
        return fail(PIC1DP_ERR_STATE, "internal: fused field solve in a launch of %d threads", fl.threads);
      a.fused = c->fuse_args;
      c->fuse_args.on = 0;
      c->fused_solves++;
    }
    if (pred) {  // k_step_one: the full step + the prediction of the next first sub-step's charge
      a.tabA = c->d_tabA;
      a.tabB = c->d_tabB;
      a.pred_kind = c->pred_kind;
      a.pred = c->pred_kind == 2 ? c->d_pred  // six sums, all species together (Z folded in)
                                 : c->d_pred + static_cast<size_t>(s) * (1 + 2 * c->in.nmode) * c->in.nx;
      a.pred_nm = c->in.nmode;
      a.pred_private = priv ? 1 : 0;
      a.fxb = S.fxb;
      if (c->pred_kind == 2) {
        a.eh_re = c->eh_modes == 2 ? c->d_mode_h : c->fa.mode_re;
        a.eh_im = c->eh_modes == 2 ? c->d_mode_h + 1 : c->fa.mode_im;
      }
      // -f0'/f0 at the new velocity is what the NEXT step's recomputation of the half-step state
      // needs: it goes there through memory (16 B per marker and step; k_step_one 1.45 -> 1.33 ms at
      // 1e8 markers, tools/ab_pred.sh).  PIC1DP_CARRY=0: evaluated again instead.
      // Only where -f0'/f0 costs something: two-stream2 and bump-on-tail (two exp and a division);
      // Maxwellian and two-stream1 evaluate it in one or two operations.
      // With the one-exp form of -f0'/f0 (device_math.hpp) an evaluation costs about what its 16 B of carry
      // traffic cost: measured (profiles/r03/experiments/ab_one_exp.log), PIC1DP_CARRY=1 / 0 insists either way.
      const bool exp_bearing = c->in.deltaf && (c->in.iptcldist == 2 || c->in.iptcldist == 3);
      const bool carry_one = c->carry < 0 ? (S.sc.one_exp ? kCarryOneExpDefault : true) : c->carry > 0;
      if (exp_bearing && carry_one) {
        if (!S.t2) HIP_TRY(hipMalloc(&S.t2, sizeof(double) * static_cast<size_t>(S.nalloc + 2)));
        a.t2 = S.t2;
        a.t2_mode = S.t2_version == read_version ? 2 : 1;
        S.t2_version = c->state_version;
      }
      lc = pred_launch(c, S.np, priv, nullptr);
      if (tail_mode != 0 && s == tail_species && !a.fused.on) {
        StepTail &t = a.tail;
        t.mode = tail_mode;
        t.ticket = c->d_ticket;
        t.rho_sp = c->fa.rho_sp;
        t.rho_copies = c->fa.rho_copies;
        t.rho_stride = c->fa.rho_stride;
        t.nspecies = c->in.nspecies;
        t.nx = c->in.nx;
        for (int k = 0; k < 8; ++k) t.Z[k] = c->fa.Z[k];
        t.sums = c->d_pred;
        t.pack = c->d_pack;
        if (tail_mode == 2) {
          c->tail_x = next_xchg_args(c);
          t.x = c->tail_x;
          t.x.ticks = nullptr;  // (the field launch's half of the exchange is the one the attribution times)
        }
        c->tail_done = tail_mode;
        c->tail_launches++;
      }
    }
    if (diag) {  // one workgroup of 1024 threads per CU: grid tiles + histograms in its LDS
      const size_t ntot = dist_len(c->in);
      a.dg = dist_geom(c);
      a.dist_out = c->d_dist + ntot * s;
      a.dist_partial = c->d_diag_part + static_cast<size_t>(6) * diag_max_blocks(c) * s;
      HIP_TRY(hipMemsetAsync(a.dist_out, 0, sizeof(double) * ntot, c->st));
      lc.lds += step_diag_lds_bytes(c->in.nx, c->in.nx_opd, c->in.nv_opd);
      lc.threads = 1024;
      int64_t blocks = c->num_cu;
      const int64_t need = ((S.np >> 1) + lc.threads - 1) / lc.threads;
      lc.blocks = static_cast<int>(std::max<int64_t>(1, std::min(blocks, need)));
      c->diag_blocks[s] = lc.blocks;
      c->diag_stride[s] = 6;       // the kinetic sums, max |p|, max |w|, the fixed-point pass's overflow flag
      // the histograms as 64-bit fixed-point sums where the species' max |p|, max |w| are known from the pass before (as the
      // diagnostics' own pass does, capi_diag.cpp run_diag_pass; a marker beyond them: the collector repeats in doubles)
      a.diag_fx = 0;
      if (c->diag_fx && c->diag_max_p[s] > 0.0 && (c->in.deltaf != 1 || c->diag_max_w[s] > 0.0))
        a.diag_fx = make_dist_scale(S.np, lc.blocks, c->in.deltaf == 1, 2.0 * c->diag_max_p[s],
                                    c->diag_fx_margin_w * c->diag_max_w[s], &a.dscale, lc.threads) ? 1 : 0;
      c->diag_fixed[s] = a.diag_fx != 0;
      if (a.diag_fx) c->diag_fx_passes++;
      c->diag_pending[s] = 1;
      c->diag_version[s] = c->state_version;  // the caller has bumped it for this step already
    }
    const int tag = pred ? kTagStepOne : (full ? kTagStepFull : kTagStepHalf);
    {  // the bytes this instantiation moves per marker: x, v, p (+ w) read; x (+ v) (+ w) written by a full step
      pic1dp_ctx::KernelBytes &kb = c->kbytes[tag];
      kb.rd = 8.0 * (3 + (c->in.deltaf ? 1 : 0));
      kb.wr = full ? 8.0 * (1 + (c->in.linear ? 0 : 1) + (c->in.deltaf ? 1 : 0)) : 0.0;
      kb.carry = 0.0;
      if (a.t2) kb.carry = pred ? (a.t2_mode == 2 ? 16.0 : 8.0) : 8.0;  // k_step_one: 8 read (mode 2) + 8 written
      std::snprintf(kb.name, sizeof kb.name, "%s%s", pred ? (c->pred_kind == 2 ? (priv ? "k_step_one<sums>" : "k_step_sums") : "k_step_one")
                                                          : (full ? (diag ? "k_step_full<DIAG>" : "k_step_full") : "k_step_half"),
                    S.sc.one_exp && c->in.deltaf ? " (one-exp -f0'/f0)" : "");
      if (a.fused.on) std::strncat(kb.name, " + field solve", sizeof kb.name - std::strlen(kb.name) - 1);
    }
    Span tm(c, PIC1DP_IWT_PUSH_PARTICLE, c->timers_on);
    Span ks(c, tag, c->stats_on);
    HIP_TRY(launch_step(a, full, lc, c->st));
    if (int rc = ks.end()) return rc;
    if (int rc = tm.end()) return rc;
  }
  if (c->fuse_args.on) return fail(PIC1DP_ERR_STATE, "internal: a fused field solve found no marker launch to run in");
  if (pred) c->pred_version = c->state_version;
  return 0;
}

// the prediction in d_pred describes the next first sub-step of the markers as they are, and the kept
// modes describe the field as it is
static bool pred_usable(const pic1dp_ctx *c) {
  return c->pred_version != 0 && c->pred_version == c->state_version && c->modes_field_version == c->field_version &&
         predict_capable(c);
}

// the local result of the prediction summed over ranks in d_charge
static int pred_reduce(pic1dp_ctx *c) {
  if (xchg_active(c)) {
    Span sp(c, PIC1DP_IWT_MPIALLREDU, c->timers_on);
    XchgArgs x = next_xchg_args(c);
    x.local_in_charge = 1;
    HIP_TRY(launch_charge_exchange(c->fa, x, c->st));
    return sp.end();
  }
  return allreduce_charge(c);
}

// prediction -> chargeden of the next first sub-step (f.chargeden: field_chargeden, or a scratch vector).
// Tiles: combined locally, summed over ranks, scaled.  Six sums: summed over ranks, then the kept mode's
// content of that charge density -- all the solve looks at (k_pred_chargeden).
// defer (call sites, f = c->fa): leave the scaling of the summed charge to the solve_field that follows (Owed)
static int pred_to_chargeden(pic1dp_ctx *c, const FieldArgs &f, bool defer = false) {
  c->pred_version = 0;  // consumed: the accumulators are zero again afterwards
  const bool multi = c->lay.nranks > 1 || c->comm != nullptr;
  if (c->pred_kind == 2) {
    if (f.chargeden == c->d_chargeden) c->cd_kept_mode_only = true;
    if (defer && !multi && c->field_solver == 0 && f.tab_lds)  // the sums' combination, chargeden and the solve in the
      return set_owed(c, Owed::PredSums);                       // launch of the solve_field that follows
    if (!multi) {
      HIP_TRY(launch_pred_chargeden(f, c->pred_tab, c->d_pred, nullptr, c->st));
      return 0;
    }
    HIP_TRY(launch_pred_to_charge(c->fa, c->d_pred, c->st));
    if (int rc = pred_reduce(c)) return rc;
    HIP_TRY(launch_pred_chargeden(f, c->pred_tab, nullptr, c->d_charge, c->st));
    return 0;
  }
  if (defer && !multi && c->field_solver == 0 && 2 * c->in.nmode <= 256)
    return set_owed(c, Owed::PredTiles);  // all of it in the launch of the solve_field that follows
  HIP_TRY(launch_pred_combine(c->fa, c->d_pred, c->in.nmode, c->st));
  if (multi)
    if (int rc = pred_reduce(c)) return rc;
  if (defer) return set_owed(c, Owed::Scale);
  HIP_TRY(launch_chargeden(f, false, c->st));
  return 0;
}

// step() path: Eh of the NEXT step from the prediction, right after the field of the new state is solved
static int predict_half_field(pic1dp_ctx *c) {
  FieldArgs f = c->fa;
  f.chargeden = c->d_cd_h;
  if (int rc = pred_to_chargeden(c, f)) return rc;
  Span tm(c, PIC1DP_IWT_FIELD_ELECTRIC, c->timers_on);
  f.E = c->d_Ehn;
  f.mode_re = c->d_mode_h;
  f.mode_im = c->d_mode_h + c->in.nmode;
  f.history = nullptr;
  HIP_TRY(launch_field_solve(f, false, true, c->st));
  c->eh_version = c->state_version;
  c->eh_field_version = c->field_version;
  return tm.end();
}

// sub-step of the whole-step path: particle kernel(s), charge, field into Eout
// fused_in: the previous step's marker launch left its solve (field of the state the markers are in, Eh of this step)
// to this launch's prologue; fuse_out: this step's solve is left to the next launch likewise (fused_pending)
static int step_phase(pic1dp_ctx *c, bool full, double *Eout, bool record, bool diag = false, bool pred = false,
                      bool fused_in = false, bool fuse_out = false) {
  if (full) c->state_version++;
  if (fused_in) {
    FusedSolve &fs = c->fuse_args;
    fs = FusedSolve{};
    fs.on = 1;
    fs.f = c->fa;  // rho_sp: the set the previous launch deposited into
    if (c->hist_count < kHistCap) fs.f.history = c->d_hist + c->hist_count++;
    fs.pt = c->pred_tab;
    fs.pred_in = c->d_pred;
    fs.E_h = c->d_Eh;
    fs.mode_h = c->d_mode_h;
    const int read = c->acc_idx, dirty = c->fused_dirty;
    fs.zero_rho = c->d_rho_all + static_cast<size_t>(dirty >= 0 ? dirty : read) * c->rho_set_doubles;
    fs.zero_rho_n = dirty >= 0 ? static_cast<int64_t>(c->rho_set_doubles) : 0;
    fs.zero_pred = dirty >= 0 ? c->d_pred_all + static_cast<size_t>(dirty) * c->pred_set_doubles
                              : c->d_pred_all + static_cast<size_t>((read + 2) % 3) * c->pred_set_doubles;  // (zero already)
    use_accumulators(c, (read + 1) % 3);  // zero: nobody has deposited into it since it was last zeroed
    c->fused_dirty = read;
    c->fused_pending = false;
    // what the dedicated launch would have recorded: E and its kept mode are the new state's, Eh is this step's
    field_written(c, true);
    c->pred_version = 0;
    c->eh_modes = 2;
  }
  if (int rc = step_particles(c, full, c->d_E, c->d_Eh, diag, pred, /*tail_ok=*/full && Eout == c->d_E)) return rc;
  if (fuse_out && pred && c->pred_version == c->state_version) {
    c->fused_pending = true;
    return 0;
  }
  return solve_phase(c, Eout, record, pred);
}

// the solve of a step whose marker launch left it pending, in a launch of its own after all
static int finish_pending_solve(pic1dp_ctx *c) {
  c->fused_pending = false;
  return solve_phase(c, c->d_E, true, true);
}

// charge sum over ranks and field solve(s) behind the marker kernel(s) of a sub-step of the whole-step path
static int solve_phase(pic1dp_ctx *c, double *Eout, bool record, bool pred) {
  const bool multi = c->lay.nranks > 1 || c->comm != nullptr;
  const bool fused_xchg = xchg_active(c) && c->field_solver == 0;  // exchange inside the solve's launch
  // RCCL path of a one-pass step: everything the two charge sums of the step need in one all-reduce
  const bool will_pack = pred && c->pred_version == c->state_version && multi && !fused_xchg && c->comm != nullptr &&
                         !xchg_active(c) && c->field_solver == 0 && 2 * c->in.nmode <= 256 && Eout == c->d_E;
  const int tail_done = c->tail_done;  // what the marker launch's tail has done already (kernels.hpp StepTail)
  c->tail_done = 0;
  if (tail_done == 1 && !will_pack) return fail(PIC1DP_ERR_STATE, "internal: a packed charge nobody reduces");
  if (tail_done == 2 && !(fused_xchg && pred && c->pred_version == c->state_version))
    return fail(PIC1DP_ERR_STATE, "internal: a posted charge nobody waits for");
  if (will_pack) {
    if (tail_done != 1) {  // the species sum and the packing are collect_charge's share of the step (src/pic1dp_interaction.F90:126-127)
      Span pk(c, PIC1DP_IWT_COLLECT_CHARGE, c->timers_on);
      HIP_TRY(launch_charge_pack(c->fa, c->d_pred, c->in.nmode, c->pred_kind, c->d_pack, c->st));
      if (int rc = pk.end()) return rc;
    }
    Span sp(c, PIC1DP_IWT_MPIALLREDU, c->timers_on);
    ncclResult_t r = rccl().AllReduce(c->d_pack, c->d_pack, pack_doubles(c->in.nx, c->in.nmode, c->pred_kind), ncclDouble,
                                      ncclSum, c->comm, c->st);
    if (r != ncclSuccess) return fail(PIC1DP_ERR_COMM, "ncclAllReduce: %s", rccl().GetErrorString(r));
    if (int rc = sp.end()) return rc;
  } else if (multi && !fused_xchg) {
    if (int rc = reduce_charge(c)) return rc;
  }
  Span tm(c, PIC1DP_IWT_FIELD_ELECTRIC, c->timers_on);
  FieldArgs f = c->fa;
  f.E = Eout;
  if (record && c->hist_count < kHistCap) f.history = c->d_hist + c->hist_count++;
  // one-pass step on one rank or with the exchange: both fields (the new state's, and the next step's
  // half-step field from the prediction) in ONE launch
  if (c->fused_dirty >= 0) {  // the set the last fused launch read: no launch follows that would zero it
    HIP_TRY(hipMemsetAsync(c->d_rho_all + static_cast<size_t>(c->fused_dirty) * c->rho_set_doubles, 0,
                           sizeof(double) * c->rho_set_doubles, c->st));
    HIP_TRY(hipMemsetAsync(c->d_pred_all + static_cast<size_t>(c->fused_dirty) * c->pred_set_doubles, 0,
                           sizeof(double) * c->pred_set_doubles, c->st));
    c->fused_dirty = -1;
  }
  const bool pair = pred && c->pred_version == c->state_version && (!multi || fused_xchg || will_pack) &&
                    c->field_solver == 0 && 2 * c->in.nmode <= 256 && Eout == c->d_E;
  if (pair) {
    // (cd_h: the tiles' scratch; with the six sums the kept mode's content of the half-step charge density -- what the call
    // sites adopt into field_chargeden when the host's next push(1), collect_charge, solve_field are served from this solve,
    // which they are after a step() as after the call sites' own pair: round 6's fuzz campaign found the stale copy)
    PairArgs pa{c->d_pred, c->d_Ehn, c->d_mode_h, c->d_cd_h, nullptr, c->pred_kind, c->pred_tab, 0};
    if (will_pack) {  // both charge sums of the step came in ONE all-reduce (pack_doubles)
      pa.pack = c->d_pack;
      HIP_TRY(launch_field_solve_pair(f, pa, nullptr, c->st));
    } else if (fused_xchg) {  // ONE exchange: charge2 and the prediction slices travel together
      XchgArgs x1;
      if (tail_done == 2) {  // ... and this rank's half of it is under way since the marker launch's tail
        x1 = c->tail_x;
        x1.ticks = c->timers_on ? c->xc.ticks : nullptr;
        pa.posted = 1;
      } else {
        x1 = next_xchg_args(c);
      }
      HIP_TRY(launch_field_solve_pair(f, pa, &x1, c->st));
    } else {
      HIP_TRY(launch_field_solve_pair(f, pa, nullptr, c->st));
    }
    field_written(c, true);
    c->pred_version = 0;  // consumed
    c->eh_version = c->state_version;
    c->eh_field_version = c->field_version;
    return tm.end();
  }
  if (fused_xchg) {
    HIP_TRY(launch_field_solve_xchg(f, next_xchg_args(c), c->st));
  } else if (int rc = enqueue_field_solve(c, f, !multi, false)) {
    return rc;
  }
  if (Eout == c->d_E) field_written(c, true);
  return tm.end();
}

int pic1dp_hip_step(pic1dp_ctx *c, int32_t nsteps) {
  CHECK_CTX(c);
  if (nsteps < 0) return fail(PIC1DP_ERR_ARG, "nsteps < 0");
  if (int rc = require_loaded(c)) return rc;
  if (nsteps > 0) c->cd_kept_mode_only = false;  // every step ends with the deposit of the new state
  const bool recompute = step_recompute_ok(c);
  for (int it = 0; it < nsteps; ++it) {
    // a step in which a marker optimisation is due goes through the sub-steps
    if (recompute && !optimize_due_any(c)) {
      // E0 = d_E stays untouched until the second solve overwrites it
      // Eh of this step: predicted by the previous step's kernel (one pass per step), or from a
      // first-sub-step pass over the markers
      const bool pc = predict_capable(c);
      // the host can only call output_all after the last step of this call
      const bool diag = diag_in_step(c) && it == nsteps - 1 && output_follows(c);
      const bool pred = pc && !diag;
      // the previous step left its solve to this step's marker launch (one launch per step, kernels.hpp FusedSolve)
      const bool fused_in = c->fused_pending && pred && fuse_capable(c);
      if (c->fused_pending && !fused_in)
        if (int rc = finish_pending_solve(c)) return rc;
      if (!fused_in) {
        const bool have_eh = pc && c->eh_version == c->state_version && c->eh_field_version == c->field_version;
        if (have_eh) {
          std::swap(c->d_Eh, c->d_Ehn);  // d_Eh: the half-step field of the step being taken
          c->eh_modes = 2;               // its kept modes: d_mode_h
        } else if (int rc = step_phase(c, false, c->d_Eh, false)) {
          return rc;
        } else {
          c->eh_modes = c->field_solver == 0 ? 1 : 0;  // that solve left them in fa.mode_re / mode_im
        }
      }
      // may the NEXT step's launch take this step's solve?  Only if that step will be an ordinary predicted one (the
      // step after an output, an optimisation step or the last step of the call need the field in memory first)
      bool fuse_out = false;
      if (pred && it + 1 < nsteps && fuse_capable(c)) {
        const bool next_diag = diag_in_step(c) && it + 1 == nsteps - 1 && output_follows_at(c, c->itime + 1, c->time + c->in.dt);
        bool due[3];
        optimize_due_at(c, c->time + c->in.dt, due);
        fuse_out = !next_diag && !(due[0] || due[1] || due[2]);
      }
      if (int rc = step_phase(c, true, c->d_E, true, diag, pred, fused_in, fuse_out)) return rc;
      if (pred && !c->fused_pending && c->pred_version == c->state_version)  // not already turned into Eh by the paired solve
        if (int rc = predict_half_field(c)) return rc;
    } else {
      if (c->fused_pending)
        if (int rc = finish_pending_solve(c)) return rc;
      if (int rc = substep_impl(c, 1, false)) return rc;
      HIP_TRY(hipMemcpyAsync(c->d_Eh, c->d_E, sizeof(double) * c->in.nx, hipMemcpyDeviceToDevice, c->st));
      if (int rc = substep_impl(c, 2, true)) return rc;
    }
    c->itime += 1;                  // src/pic1dp.F90:92
    c->time = c->time + c->in.dt;   // :93
  }
  if (c->fused_pending)  // (the last step of a call never leaves its solve pending; a safety net)
    if (int rc = finish_pending_solve(c)) return rc;
  return 0;
}

// ---------------------------------------------------------------------------
// pic1dp_hip_check_state: the relations between the flags of the state machine above (the table of
// invariants) checked at an API boundary -- between two calls of the library every one of them has to hold, whatever the
// calls were.  deep != 0 also looks at device memory (one stream synchronisation and small copies): the accumulator sets
// nobody owes anything to are zero, the tail's ticket is back at zero.  A debugging aid and what the randomised call-
// sequence tests call after EVERY call (tests/test_gpu_fuzz.py); nothing on the hot path calls it.
// ---------------------------------------------------------------------------
int pic1dp_hip_check_state(pic1dp_ctx *c, int32_t deep) {
  CHECK_CTX(c);
#define INVARIANT(cond)                                                                                   \
  do {                                                                                                    \
    if (!(cond))                                                                                          \
      return fail(PIC1DP_ERR_STATE, "state invariant violated: %s (call-site state %d, owed %d)", #cond, \
                  static_cast<int>(c->seq), static_cast<int>(c->owed));                                   \
  } while (0)
  // The call sites' state is one (Seq, Owed) pair written by set_call_state alone, which refuses the pairs that cannot
  // occur (ctx.hpp kCallStateLegal): what used to be a dozen relations between lz, cd_lazy, half_pair and half_solved
  // checked here cannot be violated any more.  What is left relates the state to the CONTEXT it is held in:
  const bool one_rank = c->lay.nranks == 1 && c->comm == nullptr;
  INVARIANT(kCallStateLegal[static_cast<int>(c->seq)][static_cast<int>(c->owed)]);
  INVARIANT(c->seq == Seq::Clean || (c->lazy_calls && c->loaded));    // a push is only noted by the lazy call sites
  INVARIANT(c->owed == Owed::Nothing || c->lazy_calls);
  INVARIANT(c->owed < Owed::SumScale || one_rank);                    // species sum / prediction left to solve_field: one rank
  INVARIANT(c->owed != Owed::PredTiles || c->pred_kind == 1);
  INVARIANT((c->owed != Owed::PredSums && c->owed != Owed::AdoptHalfField && !pair_of(c->seq)) || (c->pred_kind == 2 && c->in.nmode == 1));
  INVARIANT(!pair_of(c->seq) || (c->call_pair && c->eh_version == c->state_version));   // the field in d_Ehn belongs to the state in memory
  INVARIANT(!c->cd_kept_mode_only || c->pred_kind == 2);
  // whole-step path: nothing of a step() is left over between calls
  INVARIANT(!c->fused_pending);
  INVARIANT(c->fuse_args.on == 0);
  INVARIANT(c->fused_dirty == -1);
  INVARIANT(c->tail_done == 0);
  INVARIANT(c->acc_idx >= 0 && c->acc_idx < 3);
  // versions only ever point backwards
  INVARIANT(c->pred_version <= c->state_version);
  INVARIANT(c->eh_version <= c->state_version);
  INVARIANT(c->eh_field_version <= c->field_version);
  INVARIANT(c->modes_field_version <= c->field_version);
  INVARIANT(c->eh_modes >= 0 && c->eh_modes <= 2);
  for (int s = 0; s < c->in.nspecies; ++s) {
    if (static_cast<size_t>(s) < c->diag_version.size()) INVARIANT(c->diag_version[s] <= c->state_version);  // (sized by the first diagnostics call)
    INVARIANT(c->sp[s].t2_version <= c->state_version);
  }
  INVARIANT(!c->charge_pending_pred || c->charge_pending);
  if (!deep) return 0;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->st));
  unsigned ticket = 0;
  HIP_TRY(hipMemcpy(&ticket, c->d_ticket, sizeof ticket, hipMemcpyDeviceToHost));
  INVARIANT(ticket == 0u);
  // accumulators: the sets the marker kernels do not deposit into are zero; the current one holds something only while
  // a collect_charge has left its end to solve_field (Owed::SumScale, PredTiles: deposits; with a usable prediction: the six sums / tiles)
  auto nonzero = [](const std::vector<double> &v, size_t off, size_t n) {
    for (size_t i = 0; i < n; ++i)
      if (v[off + i] != 0.0) return true;  // (NaN != 0: a poisoned accumulator counts)
    return false;
  };
  std::vector<double> h(3 * c->rho_set_doubles);
  HIP_TRY(hipMemcpy(h.data(), c->d_rho_all, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; ++k) {
    const bool nz = nonzero(h, k * c->rho_set_doubles, c->rho_set_doubles);
    if (k != c->acc_idx) INVARIANT(!nz && "a species accumulator set that is not the current one holds deposits");
    else if (c->owed != Owed::SumScale && c->owed != Owed::PredTiles && !c->charge_pending) INVARIANT(!nz && "deposits nobody is going to sum");
  }
  if (c->d_pred_all && c->pred_set_doubles) {
    h.assign(3 * c->pred_set_doubles, 0.0);
    HIP_TRY(hipMemcpy(h.data(), c->d_pred_all, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) {
      const bool nz = nonzero(h, k * c->pred_set_doubles, c->pred_set_doubles);
      if (k != c->acc_idx) INVARIANT(!nz && "a prediction accumulator set that is not the current one holds sums");
      else if (c->pred_version == 0 && c->owed != Owed::PredTiles && c->owed != Owed::PredSums && !c->charge_pending)
        INVARIANT(!nz && "prediction sums that belong to no state");
    }
  }
#undef INVARIANT
  return 0;
}

int pic1dp_hip_set_step_mode(pic1dp_ctx *c, int32_t mode) {
  CHECK_CTX(c);
  if (mode != 0 && mode != 1) return fail(PIC1DP_ERR_ARG, "step mode must be 0 or 1");
  c->step_mode = mode;
  return 0;
}

int pic1dp_hip_predict_kind(pic1dp_ctx *c, int32_t *kind) {
  CHECK_CTX(c);
  if (!kind) return fail(PIC1DP_ERR_ARG, "null argument");
  *kind = predict_capable(c) ? c->pred_kind : 0;
  return 0;
}

int pic1dp_hip_set_output_fusion(pic1dp_ctx *c, int32_t on) {
  CHECK_CTX(c);
  if (on < 0 || on > 2) return fail(PIC1DP_ERR_ARG, "output fusion must be 0 (off), 1 (where it pays) or 2 (always)");
  c->fuse_output = on;
  return 0;
}

// field_chargeden as the reference holds it between the sub-steps, when the collect_charge after a noted push(1)
// was served from the six sums (the kept mode's content only): the half-step state is pushed into memory after all
// (as every inspection of a noted push does) and deposited for real.  One rank only -- on several ranks the
// reduction is a collective that an inspection on one of them must not start: there chargeden keeps the kept
// mode's content (include/pic1dp_hip.h says so).  The field solved from either is the same to rounding: the
// solve only looks at the kept mode.
int pic1dp_host::rebuild_half_step_chargeden(pic1dp_ctx *c) {
  // the flag is cleared only when the vector has actually been rebuilt: on several ranks, or when the
  // markers have left the half-step state through calls outside the sequence, chargeden keeps the kept mode's
  // content and pic1dp_hip_chargeden_state says so
  if (c->lay.nranks > 1 || c->comm != nullptr) return 0;
  if (lz_of(c->seq) != LZ_HALF && lz_of(c->seq) != LZ_PUSH2) return 0;
  if (int rc = settle_half_pair(c)) return rc;
  const bool push2_was_noted = lz_of(c->seq) == LZ_PUSH2;
  if (int rc = enqueue_push(c, 1, false, c->d_E0)) return rc;
  if (int rc = set_seq(c, Seq::Clean)) return rc;  // memory now holds the half-step state (x not yet wrapped): the deposit wraps and stores it
  if (int rc = enqueue_deposit(c)) return rc;
  HIP_TRY(launch_chargeden(c->fa, true, c->st));
  c->cd_kept_mode_only = false;
  // a push(2) that had been noted: memory as the eager calls would have left it (the field it sees, d_E, is the one
  // solve_field wrote after the half step)
  if (push2_was_noted)
    if (int rc = enqueue_push(c, 2, false)) return rc;
  return 0;
}

// ---------------------------------------------------------------------------
// split-phase deposit
// ---------------------------------------------------------------------------
int pic1dp_hip_charge_local(pic1dp_ctx *c, double *charge2) {
  CHECK_CTX(c);
  if (!charge2) return fail(PIC1DP_ERR_ARG, "null array");
  if (int rc = require_loaded_keep_lazy(c)) return rc;
  if (pair_of(c->seq) && c->seq != Seq::Push2PairSolved)  // out of sequence (as in collect_charge): memory as the eager calls leave it
    if (int rc = settle_half_pair(c)) return rc;
  if (c->owed == Owed::AdoptHalfField)   // (left over by a pair settled before its solve_field: void now)
    if (int rc = set_owed(c, Owed::Nothing)) return rc;
  if (c->seq == Seq::Push1 && pred_usable(c)) {  // predicted by the previous step's kernel: no marker pass
    HIP_TRY(hipMemcpyAsync(c->d_E0, c->d_E, sizeof(double) * c->in.nx, hipMemcpyDeviceToDevice, c->st));
    if (int rc = set_seq(c, Seq::Half)) return rc;
    if (c->pred_kind == 2) {  // the six sums in charge2[0..5], zeros behind: the host's sum over ranks sums them
      HIP_TRY(launch_pred_to_charge(c->fa, c->d_pred, c->st));
      c->charge_pending_pred = true;
    } else {
      HIP_TRY(launch_pred_combine(c->fa, c->d_pred, c->in.nmode, c->st));
    }
    c->pred_version = 0;
  } else {
    c->cd_kept_mode_only = false;
    if (int rc = deposit_or_step(c)) return rc;
    HIP_TRY(launch_charge_local(c->fa, c->st));
  }
  HIP_TRY(hipStreamSynchronize(c->st));
  HIP_TRY(hipMemcpy(charge2, c->d_charge, sizeof(double) * c->in.nx, hipMemcpyDeviceToHost));
  c->charge_pending = true;
  return 0;
}

int pic1dp_hip_charge_reduced(pic1dp_ctx *c, const double *charge1) {
  CHECK_CTX(c);
  if (!charge1) return fail(PIC1DP_ERR_ARG, "null array");
  if (!c->charge_pending) return fail(PIC1DP_ERR_STATE, "charge_reduced without charge_local");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(c->d_charge, charge1, sizeof(double) * c->in.nx, hipMemcpyHostToDevice));
  c->charge_pending = false;
  if (c->charge_pending_pred) {  // what came back are the summed prediction sums
    c->charge_pending_pred = false;
    c->cd_kept_mode_only = true;
    HIP_TRY(launch_pred_chargeden(c->fa, c->pred_tab, nullptr, c->d_charge, c->st));
    return 0;
  }
  if (c->lazy_calls) return set_owed(c, Owed::Scale);
  HIP_TRY(launch_chargeden(c->fa, false, c->st));
  return 0;
}

