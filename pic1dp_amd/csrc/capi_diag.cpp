// capi_diag.cpp -- diagnostics of output_all (src/pic1dp_output.F90:100-189, 196-477): the kinetic sums and the
// (x, v) / v histograms, one fused pass per species, cached against state_version.
#include "ctx.hpp"

namespace pic1dp_host {

// One fused pass over a species' markers for output_all: histograms of
// output_ptcldist into d_dist[isp] and the kinetic sums of output_field; results
// stay valid until the markers change (state_version).
size_t dist_len(const pic1dp_input &in) {
  return 3 * static_cast<size_t>(in.nx_opd) * in.nv_opd + 3 * static_cast<size_t>(in.nv_opd);
}

int diag_max_blocks(const pic1dp_ctx *c) { return 2 * c->num_cu; }

int pinned(pic1dp_ctx *c, size_t ndoubles, double **out) {
  if (ndoubles > c->h_pin_doubles) {
    HIP_TRY(hipStreamSynchronize(c->st));  // (nothing in flight into the old buffer)
    (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr;
    c->h_pin_doubles = 0;
    const size_t want = ndoubles + ndoubles / 4 + 64;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_pin), sizeof(double) * want, hipHostMallocDefault));
    c->h_pin_doubles = want;
  }
  *out = c->h_pin;
  return 0;
}

// the histogram geometry with its constant divisors vouched for, formed once per context
const DistGeom &dist_geom(pic1dp_ctx *c) {
  if (!c->dist_geom_ready) {
    c->dist_geom_v = make_dist_geom(c->in.lx, c->in.v_max, c->in.nx_opd, c->in.nv_opd);
    c->dist_geom_ready = true;
  }
  return c->dist_geom_v;
}

// buffers of the marker diagnostics: [nspecies] cached histograms + one for the all-reduced
// copy handed out, per-workgroup partial sums per species
int diag_buffers(pic1dp_ctx *c) {
  const pic1dp_input &in = c->in;
  const int ns = in.nspecies;
  if (in.nx_opd < 1 || in.nv_opd < 2) return fail(PIC1DP_ERR_ARG, "nx_opd >= 1 and nv_opd >= 2 required");
  if (c->diag_version.empty()) {
    c->diag_version.assign(ns, 0);
    c->diag_sums.assign(3 * static_cast<size_t>(ns), 0.0);
    c->diag_pending.assign(ns, 0);
    c->diag_blocks.assign(ns, 0);
    c->diag_stride.assign(ns, 3);
    c->diag_max_p.assign(ns, 0.0);
    c->diag_max_w.assign(ns, 0.0);
    c->diag_fixed.assign(ns, 0);
  }
  if (!c->d_dist) HIP_TRY(hipMalloc(&c->d_dist, sizeof(double) * dist_len(in) * (ns + 1)));
  if (!c->d_diag_part) HIP_TRY(hipMalloc(&c->d_diag_part, sizeof(double) * 6 * diag_max_blocks(c) * ns));
  return 0;
}

// the diagnostics' own pass over species isp (k_ptcldist) into its cached histograms; fixed: 64-bit fixed-point sums where
// the species' max |p|, max |w| are known from the pass before (with a margin: p changes with load / upload / events only,
// w grows by a few per cent between two records), else -- and whenever `fixed` is false -- double sums
static int run_diag_pass(pic1dp_ctx *c, int isp, bool fixed, bool *was_fixed) {
  const pic1dp_input &in = c->in;
  Species &S = c->sp[isp];
  const PSet &A = S.set[c->cur];
  const size_t ntot = dist_len(in);
  double *hist = c->d_dist + ntot * isp;
  double *part_dev = c->d_diag_part + static_cast<size_t>(6) * diag_max_blocks(c) * isp;
  HIP_TRY(hipMemsetAsync(hist, 0, sizeof(double) * ntot, c->st));
  c->diag_blocks[isp] = 0;
  c->diag_stride[isp] = 6;
  *was_fixed = false;
  c->diag_fixed[isp] = 0;
  if (S.np <= 0) return 0;
  double bp = 0.0, bw = 0.0;
  if (fixed && c->diag_fx && c->diag_max_p[isp] > 0.0 && (in.deltaf != 1 || c->diag_max_w[isp] > 0.0)) {
    bp = 2.0 * c->diag_max_p[isp];
    bw = c->diag_fx_margin_w * c->diag_max_w[isp];
  }
  c->diag_blocks[isp] = ptcldist_blocks(S.np, in.nx_opd, in.nv_opd, c->num_cu);
  HIP_TRY(launch_ptcldist(A.x, A.v, S.p, A.w, S.np, dist_geom(c), in.deltaf == 1, bp, bw, hist, part_dev, c->num_cu, c->dyn_tail, c->st,
                          was_fixed));
  c->diag_passes++;
  if (*was_fixed) c->diag_fx_passes++;
  c->diag_fixed[isp] = *was_fixed;
  return 0;
}

int ensure_diag(pic1dp_ctx *c, int isp) {
  const pic1dp_input &in = c->in;
  if (int rc = diag_buffers(c)) return rc;
  Species &S = c->sp[isp];
  double *part_dev = c->d_diag_part + static_cast<size_t>(6) * diag_max_blocks(c) * isp;
  bool was_fixed = false;
  if (c->diag_version[isp] != c->state_version) {  // no pass has seen these markers yet: run one
    if (int rc = run_diag_pass(c, isp, true, &was_fixed)) return rc;
    c->diag_pending[isp] = 1;
    c->diag_version[isp] = c->state_version;
  }
  if (!c->diag_pending[isp]) return 0;
  was_fixed = c->diag_fixed[isp] != 0;  // (the pending pass: the one just run, or k_step_full<DIAG> inside a step)
  // collect: partial kinetic sums of the pass (k_ptcldist, or k_step_full's DIAG variant), workgroup order
  double *sums = &c->diag_sums[3 * static_cast<size_t>(isp)];
  // the pass's partial sums and those of the tail slots (the reference sums the whole local vector, VecSum; slots beyond
  // np live in set 0) through the pinned staging: both transfers enqueued, ONE wait
  const int64_t ntail = S.nalloc - S.np;
  const int tb = ntail > 0 ? static_cast<int>(std::min<int64_t>(kEnergyBlocks, (ntail + 255) / 256)) : 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int blocks = c->diag_blocks[isp], stride = c->diag_stride[isp];
    sums[0] = sums[1] = sums[2] = 0.0;
    double *part = nullptr;
    if (int rc = pinned(c, static_cast<size_t>(stride) * blocks + 3 * static_cast<size_t>(tb) + 8, &part)) return rc;
    if (blocks > 0) HIP_TRY(hipMemcpyAsync(part, part_dev, sizeof(double) * blocks * stride, hipMemcpyDeviceToHost, c->st));
    if (tb > 0) {
      HIP_TRY(launch_energy_sums(S.set[0].v, S.p, in.deltaf ? S.set[0].w : nullptr, S.np, ntail, c->d_scratch, tb,
                                 c->st));
      HIP_TRY(hipMemcpyAsync(part + static_cast<size_t>(stride) * blocks, c->d_scratch, sizeof(double) * tb * 3,
                             hipMemcpyDeviceToHost, c->st));
    }
    if (blocks + tb > 0) HIP_TRY(hipStreamSynchronize(c->st));
    double maxp = 0.0, maxw = 0.0, over = 0.0;
    for (int b = 0; b < blocks; ++b) {
      for (int k = 0; k < 3; ++k) sums[k] += part[b * stride + k];
      if (stride >= 6) {
        maxp = std::max(maxp, part[b * stride + 3]);
        maxw = std::max(maxw, part[b * stride + 4]);
        over = std::max(over, part[b * stride + 5]);
      }
    }
    for (int b = 0; b < tb; ++b)
      for (int k = 0; k < 3; ++k) sums[k] += part[static_cast<size_t>(stride) * blocks + b * 3 + k];
    if (stride >= 6 && blocks > 0) {  // what the next pass of this species may scale its fixed-point sums with
      c->diag_max_p[isp] = maxp;
      c->diag_max_w[isp] = maxw;
    }
    if (!(was_fixed && over > 0.0)) break;
    // a marker beyond the bounds the fixed-point pass was scaled for (it skipped that marker): once more, in doubles
    c->diag_fx_repeats++;
    if (int rc = run_diag_pass(c, isp, false, &was_fixed)) return rc;
  }
  c->diag_pending[isp] = 0;
  return 0;
}


}  // namespace pic1dp_host

extern "C" {

int pic1dp_hip_energy_sums(pic1dp_ctx *c, int32_t isp, double out[3]) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies || !out) return fail(PIC1DP_ERR_ARG, "bad argument");
  if (int rc = require_loaded(c)) return rc;
  if (int rc = ensure_diag(c, isp)) return rc;
  for (int k = 0; k < 3; ++k) out[k] = c->diag_sums[3 * static_cast<size_t>(isp) + k];
  if (!c->in.deltaf) out[2] = out[1];
  return 0;
}

// ---------------------------------------------------------------------------
// diagnostics of output_all
// ---------------------------------------------------------------------------
int pic1dp_hip_output_scalars(pic1dp_ctx *c, double *out, int32_t n) {
  CHECK_CTX(c);
  const int ns = c->in.nspecies;
  if (!out || n != 2 + 3 * ns) return fail(PIC1DP_ERR_ARG, "out must hold 2 + 3*nspecies doubles");
  std::vector<double> sums(3 * ns);
  for (int s = 0; s < ns; ++s)
    if (int rc = pic1dp_hip_energy_sums(c, s, &sums[3 * s])) return rc;
  if (c->comm) {  // VecSum's scalar all-reduce
    double *d = c->d_scratch;
    HIP_TRY(hipMemcpyAsync(d, sums.data(), sizeof(double) * 3 * ns, hipMemcpyHostToDevice, c->st));
    if (int rc = allreduce_doubles(c, d, 3 * ns)) return rc;
    HIP_TRY(hipStreamSynchronize(c->st));
    HIP_TRY(hipMemcpy(sums.data(), d, sizeof(double) * 3 * ns, hipMemcpyDeviceToHost));
  }
  return pic1dp_hip_output_scalars_from(c, sums.data(), out, n);
}

// realbuf of output_field (src/pic1dp_output.F90:117-175) from the kinetic sums over all ranks and int E^2 dx
static void fill_scalars(const pic1dp_ctx *c, const double *sums, double energy, double *out) {
  const int ns = c->in.nspecies;
  out[0] = c->time;
  out[1] = energy;
  const pic1dp_input &in = c->in;
  for (int s = 0; s < ns; ++s) {
    double marker = sums[3 * s], total = sums[3 * s + 1], pert;
    if (in.deltaf == 1) {
      pert = sums[3 * s + 2];
      if (in.linear == 1) total = total + pert;  // :152-155
    } else {                                      // :156-170
      pert = total;
      if (in.iptcldist == 1) {
        pert = pert - 3.0 * in.species_density[s] * in.lx;
      } else if (in.iptcldist == 0) {
        pert = pert - in.species_temperature[s] / in.species_mass[s] * in.species_density[s] * in.lx;
      }
    }
    out[2 + 3 * s] = marker;
    out[3 + 3 * s] = total;
    out[4 + 3 * s] = pert;
  }
}

int pic1dp_hip_output_scalars_from(pic1dp_ctx *c, const double *sums, double *out, int32_t n) {
  CHECK_CTX(c);
  const int ns = c->in.nspecies;
  if (!sums || !out || n != 2 + 3 * ns) return fail(PIC1DP_ERR_ARG, "sums must hold 3*nspecies and out 2 + 3*nspecies doubles");
  double energy = 0.0;
  if (int rc = pic1dp_hip_field_energy(c, &energy)) return rc;
  fill_scalars(c, sums, energy, out);
  return 0;
}

// equilibrium f0(v) as the full-f branch of output_ptcldist normalises it
// (src/pic1dp_output.F90:375-451; note the reference divides by T/m, not sqrt(T/m))
static double output_f0(const pic1dp_input &in, int s, double sv) {
  const double T = in.species_temperature[s], T2 = in.species_temperature2[s], m = in.species_mass[s];
  const double den = in.species_density[s], v0 = in.species_v0[s];
  if (in.iptcldist == 1) return den * (sv * sv) * std::exp(-(sv * sv) / 2.0) / std::sqrt(2.0 * kPi);
  if (in.iptcldist == 2)
    return den * (std::exp(-((sv + v0) * (sv + v0)) / (2.0 * T / m)) + std::exp(-((sv - v0) * (sv - v0)) / (2.0 * T / m))) /
           (std::sqrt(8.0 * kPi) * T / m);
  if (in.iptcldist == 3)
    return den * std::exp(-(sv * sv) / (2.0 * T / m)) / (std::sqrt(2.0 * kPi) * T / m) +
           (1.0 - den) * std::exp(-((sv - v0) * (sv - v0)) / (2.0 * T2 / m)) / (std::sqrt(2.0 * kPi) * T2 / m);
  return den * std::exp(-((sv - v0) * (sv - v0)) / (2.0 * T / m)) / (std::sqrt(2.0 * kPi) * T / m);
}

// what output_ptcldist does with the sums over ranks (src/pic1dp_output.F90:328-331, :361-453): linear total +=
// pertb, scaling by the histogram cell sizes, full-f pertb = total - f0; in place on host arrays
static void finish_ptcldist(const pic1dp_input &in, int isp, double *mxv, double *txv, double *pxv, double *mv, double *tv,
                            double *pv) {
  const int nxo = in.nx_opd, nvo = in.nv_opd;
  const size_t nxv = static_cast<size_t>(nxo) * nvo;
  if (in.linear == 1) {  // :328-331
    for (size_t i = 0; i < nxv; ++i) txv[i] = txv[i] + pxv[i];
    for (int i = 0; i < nvo; ++i) tv[i] = tv[i] + pv[i];
  }
  const double delv_inv = static_cast<double>(nvo - 1) / (2.0 * in.v_max);  // :203-205
  const double delx_inv = static_cast<double>(nxo) / in.lx;
  for (size_t i = 0; i < nxv; ++i) {
    mxv[i] = mxv[i] * delx_inv * delv_inv;
    txv[i] = txv[i] * delx_inv * delv_inv;
  }
  for (int i = 0; i < nvo; ++i) {
    mv[i] = mv[i] * delv_inv;
    tv[i] = tv[i] * delv_inv;
  }
  if (in.deltaf == 1) {
    for (size_t i = 0; i < nxv; ++i) pxv[i] = pxv[i] * delx_inv * delv_inv;
    for (int i = 0; i < nvo; ++i) pv[i] = pv[i] * delv_inv;
  } else {  // :370-453
    for (int iv = 0; iv < nvo; ++iv) {
      const double sv = (static_cast<double>(iv) / static_cast<double>(nvo - 1) * 2.0 - 1.0) * in.v_max;
      const double f0 = output_f0(in, isp, sv);
      for (int ix = 0; ix < nxo; ++ix) pxv[static_cast<size_t>(iv) * nxo + ix] = txv[static_cast<size_t>(iv) * nxo + ix] - f0;
      pv[iv] = tv[iv] - in.lx * f0;
    }
  }
}

int pic1dp_hip_ptcldist(pic1dp_ctx *c, int32_t isp, int32_t finish, double *markr_xv, double *total_xv,
                        double *pertb_xv, double *markr_v, double *total_v, double *pertb_v) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  if (int rc = require_loaded(c)) return rc;
  const pic1dp_input &in = c->in;
  const int nxo = in.nx_opd, nvo = in.nv_opd;
  if (int rc = ensure_diag(c, isp)) return rc;
  const size_t nxv = static_cast<size_t>(nxo) * nvo, ntot = 3 * nxv + 3 * nvo;
  const double *hist = c->d_dist + ntot * isp;
  if (finish && c->comm) {  // reduce a copy: the cached local histograms stay local
    double *red = c->d_dist + ntot * in.nspecies;
    HIP_TRY(hipMemcpyAsync(red, hist, sizeof(double) * ntot, hipMemcpyDeviceToDevice, c->st));
    if (int rc = allreduce_doubles(c, red, ntot)) return rc;
    hist = red;
  } else if (finish && c->lay.nranks > 1) {
    return fail(PIC1DP_ERR_STATE, "nranks > 1 but no communicator: take finish = 0 and reduce the local sums on the host");
  }
  double *h = nullptr;
  if (int rc = pinned(c, ntot, &h)) return rc;
  HIP_TRY(hipMemcpyAsync(h, hist, sizeof(double) * ntot, hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipStreamSynchronize(c->st));
  double *mxv = h, *txv = mxv + nxv, *pxv = txv + nxv, *mv = pxv + nxv, *tv = mv + nvo, *pv = tv + nvo;
  if (finish) finish_ptcldist(in, isp, mxv, txv, pxv, mv, tv, pv);
  auto give = [&](double *dst, const double *src, size_t n) {
    if (dst) std::memcpy(dst, src, sizeof(double) * n);
  };
  give(markr_xv, mxv, nxv);
  give(total_xv, txv, nxv);
  give(pertb_xv, pxv, nxv);
  give(markr_v, mv, nvo);
  give(total_v, tv, nvo);
  give(pertb_v, pv, nvo);
  return 0;
}

int pic1dp_hip_ptcldist_finish(pic1dp_ctx *c, int32_t isp, double *markr_xv, double *total_xv, double *pertb_xv,
                               double *markr_v, double *total_v, double *pertb_v) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  if (!markr_xv || !total_xv || !pertb_xv || !markr_v || !total_v || !pertb_v) return fail(PIC1DP_ERR_ARG, "null array");
  finish_ptcldist(c->in, isp, markr_xv, total_xv, pertb_xv, markr_v, total_v, pertb_v);
  return 0;
}


// Everything output_all writes (src/pic1dp_output.F90:100-189, 196-477) in ONE call and one wait: the diagnostics passes of
// all species, int E^2 dx, and every transfer to the host enqueued behind them on the engine's stream through the pinned
// staging (the separate calls wait three times and more per record: 0.16 ms of host time that is most of what a record
// costs at 6.4e6 markers).  dist: [nspecies][3 nx_opd nv_opd + 3 nv_opd] as output_ptcldist writes them (finished), or null.
// Several ranks: composed of the separate calls, which own the reductions.
int pic1dp_hip_output_all(pic1dp_ctx *c, double *scalars, int32_t nscal, double *E, double *cd, double *re, double *im,
                          double *dist) {
  CHECK_CTX(c);
  const pic1dp_input &in = c->in;
  const int ns = in.nspecies;
  if (!scalars || nscal != 2 + 3 * ns) return fail(PIC1DP_ERR_ARG, "scalars must hold 2 + 3*nspecies doubles");
  const size_t nx = in.nx, nm = in.nmode, ntot = dist_len(in), nxv = static_cast<size_t>(in.nx_opd) * in.nv_opd;
  if (c->comm || c->lay.nranks > 1) {
    if (int rc = pic1dp_hip_output_scalars(c, scalars, nscal)) return rc;
    if (int rc = pic1dp_hip_get_field(c, E, cd, re, im)) return rc;
    for (int s = 0; s < ns && dist; ++s) {
      double *d = dist + ntot * s;
      if (int rc = pic1dp_hip_ptcldist(c, s, 1, d, d + nxv, d + 2 * nxv, d + 3 * nxv, d + 3 * nxv + in.nv_opd,
                                       d + 3 * nxv + 2 * in.nv_opd))
        return rc;
    }
    return 0;
  }
  if (int rc = require_loaded(c)) return rc;
  if (int rc = settle_field_view(c)) return rc;
  if (int rc = materialize_cd(c)) return rc;
  if (cd && c->cd_kept_mode_only)
    if (int rc = rebuild_half_step_chargeden(c)) return rc;
  if (int rc = diag_buffers(c)) return rc;
  // the pinned record: [E | cd | re | im | energy + pad] then per species [partial sums | tail sums | histograms]
  std::vector<size_t> off_part(ns), off_tail(ns), off_hist(ns);
  std::vector<int> tb(ns, 0);
  std::vector<char> fixed(ns, 0);
  size_t off = 2 * nx + 2 * nm + 8;
  for (int s = 0; s < ns; ++s) {  // launch what has not seen these markers yet
    if (c->diag_version[s] != c->state_version) {
      bool was_fixed = false;
      if (int rc = run_diag_pass(c, s, true, &was_fixed)) return rc;
      c->diag_pending[s] = 1;
      c->diag_version[s] = c->state_version;
    }
    fixed[s] = c->diag_pending[s] && c->diag_fixed[s];  // (the pass just run, or k_step_full<DIAG> inside the step before)
    const int64_t ntail = c->sp[s].nalloc - c->sp[s].np;
    tb[s] = c->diag_pending[s] && ntail > 0 ? static_cast<int>(std::min<int64_t>(kEnergyBlocks, (ntail + 255) / 256)) : 0;
    off_part[s] = off;
    off += static_cast<size_t>(6) * diag_max_blocks(c);
    off_tail[s] = off;
    off += static_cast<size_t>(3) * kEnergyBlocks;
    off_hist[s] = off;
    off += ntot;
  }
  double *h = nullptr;
  if (int rc = pinned(c, off, &h)) return rc;
  double *slot = c->d_scratch + kEnergyBlocks * 3;
  HIP_TRY(launch_field_energy(c->d_E, in.nx, in.lx, static_cast<double>(in.nx), slot, c->st));
  // the record is gathered on the device and crosses in ONE transfer (every small copy of its own costs the stream ~10 us)
  if (c->d_rec_doubles < off) {
    if (c->d_rec) HIP_TRY(hipFree(c->d_rec));
    c->d_rec = nullptr, c->d_rec_doubles = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_rec), sizeof(double) * off));
    c->d_rec_doubles = off;
  }
  PackArgs pk{};
  auto seg = [&pk](const double *src, size_t dst, size_t n) {
    if (n == 0) return;
    pk.src[pk.count] = src, pk.dst[pk.count] = static_cast<unsigned>(dst), pk.n[pk.count] = static_cast<unsigned>(n);
    pk.count++;
  };
  seg(c->d_E, 0, nx);
  seg(c->d_chargeden, nx, nx);
  seg(c->d_mode_re, 2 * nx, nm);
  seg(c->d_mode_im, 2 * nx + nm, nm);
  seg(slot, 2 * nx + 2 * nm, 1);
  size_t used = 2 * nx + 2 * nm + 8;
  bool tails = false;
  for (int s = 0; s < ns; ++s) {
    if (c->diag_pending[s]) {
      const int blocks = c->diag_blocks[s], stride = c->diag_stride[s];
      seg(c->d_diag_part + static_cast<size_t>(6) * diag_max_blocks(c) * s, off_part[s], static_cast<size_t>(blocks) * stride);
      if (tb[s] > 0) tails = true;
    }
    if (dist) seg(c->d_dist + ntot * s, off_hist[s], ntot);
    used = off_hist[s] + ntot;
  }
  HIP_TRY(launch_pack_record(pk, c->d_rec, c->st));
  HIP_TRY(hipMemcpyAsync(h, c->d_rec, sizeof(double) * used, hipMemcpyDeviceToHost, c->st));
  if (tails) {  // (unloaded or emptied slots: the reference sums the whole local vector (VecSum); slots beyond np live in set 0)
    for (int s = 0; s < ns; ++s) {
      Species &S = c->sp[s];
      if (!(c->diag_pending[s] && tb[s] > 0)) continue;
      HIP_TRY(launch_energy_sums(S.set[0].v, S.p, in.deltaf ? S.set[0].w : nullptr, S.np, S.nalloc - S.np, c->d_scratch,
                                 tb[s], c->st));
      HIP_TRY(hipMemcpyAsync(h + off_tail[s], c->d_scratch, sizeof(double) * tb[s] * 3, hipMemcpyDeviceToHost, c->st));
    }
  }
  HIP_TRY(hipStreamSynchronize(c->st));
  if (int rc = xchg_check(c)) return rc;
  // a species' kinetic sums from its slots of the record (the pass's per-workgroup partial sums, then the tail slots'), and
  // whether the pass saw a marker beyond its fixed-point bounds
  auto fold = [&](int s) -> bool {
    double *acc = &c->diag_sums[3 * static_cast<size_t>(s)];
    const int blocks = c->diag_blocks[s], stride = c->diag_stride[s];
    const double *part = h + off_part[s], *tail = h + off_tail[s];
    acc[0] = acc[1] = acc[2] = 0.0;
    double maxp = 0.0, maxw = 0.0, over = 0.0;
    for (int b = 0; b < blocks; ++b) {
      for (int k = 0; k < 3; ++k) acc[k] += part[b * stride + k];
      if (stride >= 6) {
        maxp = std::max(maxp, part[b * stride + 3]);
        maxw = std::max(maxw, part[b * stride + 4]);
        over = std::max(over, part[b * stride + 5]);
      }
    }
    for (int b = 0; b < tb[s]; ++b)
      for (int k = 0; k < 3; ++k) acc[k] += tail[b * 3 + k];
    if (stride >= 6 && blocks > 0) {
      c->diag_max_p[s] = maxp;
      c->diag_max_w[s] = maxw;
    }
    return over > 0.0;
  };
  std::vector<double> sums(3 * static_cast<size_t>(ns));
  for (int s = 0; s < ns; ++s) {
    if (c->diag_pending[s]) {
      if (fold(s) && fixed[s]) {
        // a marker beyond the fixed-point bounds: this species once more, in doubles (rare).  The repeat's partial sums and
        // histograms go into THIS species' slots of the record -- everything else in it (the fields, the other species) is
        // still to be read (the repeat used to stage through the start of the same buffer); the tail slots' sums
        // are the ones already there
        c->diag_fx_repeats++;
        bool was_fixed = false;
        if (int rc = run_diag_pass(c, s, false, &was_fixed)) return rc;
        const size_t npart = static_cast<size_t>(c->diag_blocks[s]) * c->diag_stride[s];
        if (npart > static_cast<size_t>(6) * diag_max_blocks(c)) return fail(PIC1DP_ERR_STATE, "diagnostics: more partial sums than their slot holds");
        if (npart > 0)
          HIP_TRY(hipMemcpyAsync(h + off_part[s], c->d_diag_part + static_cast<size_t>(6) * diag_max_blocks(c) * s, sizeof(double) * npart,
                                 hipMemcpyDeviceToHost, c->st));
        if (dist) HIP_TRY(hipMemcpyAsync(h + off_hist[s], c->d_dist + ntot * s, sizeof(double) * ntot, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        (void)fold(s);
      }
      c->diag_pending[s] = 0;
    }
    for (int k = 0; k < 3; ++k) sums[3 * s + k] = c->diag_sums[3 * static_cast<size_t>(s) + k];
    if (!in.deltaf) sums[3 * s + 2] = sums[3 * s + 1];
  }
  fill_scalars(c, sums.data(), h[2 * nx + 2 * nm], scalars);
  if (E) std::memcpy(E, h, sizeof(double) * nx);
  if (cd) std::memcpy(cd, h + nx, sizeof(double) * nx);
  if (re) std::memcpy(re, h + 2 * nx, sizeof(double) * nm);
  if (im) std::memcpy(im, h + 2 * nx + nm, sizeof(double) * nm);
  for (int s = 0; s < ns && dist; ++s) {
    double *d = h + off_hist[s];
    finish_ptcldist(in, s, d, d + nxv, d + 2 * nxv, d + 3 * nxv, d + 3 * nxv + in.nv_opd, d + 3 * nxv + 2 * in.nv_opd);
    std::memcpy(dist + ntot * s, d, sizeof(double) * ntot);
  }
  return 0;
}

}  // extern "C"
