// capi_optimize.cpp -- marker optimisation events (particle_optimize -> particle_merge / particle_remove /
// particle_split, src/pic1dp_particle.F90:356-813).
#include "ctx.hpp"

namespace pic1dp_host {

// ---- marker optimisation (host side, rare; see optimize.hpp) -------------------
// which events are due for the step that is being taken: merge, remove, split
// time0: the time at the start of the step being taken
void optimize_due_at(const pic1dp_ctx *c, double time0, bool due[3]) {
  const pic1dp_input &in = c->in;
  const double t = time0 + in.dt;  // src/pic1dp_particle.F90:742,756,770
  due[0] = c->imerge > 0 && c->imerge <= in.nmerge && t >= in.tmerge[c->imerge - 1];
  due[1] = c->iremove > 0 && c->iremove <= in.nremove && t >= in.tremove[c->iremove - 1];
  due[2] = c->isplit > 0 && c->isplit <= in.nsplit && t >= in.tsplit[c->isplit - 1];
  if (in.deltaf == 0) due[0] = due[1] = due[2] = false;  // :734
}
void optimize_due(const pic1dp_ctx *c, bool due[3]) { optimize_due_at(c, c->time, due); }

bool optimize_due_any(const pic1dp_ctx *c) {
  bool due[3];
  optimize_due(c, due);
  return due[0] || due[1] || due[2];
}


}  // namespace pic1dp_host

extern "C" {

int pic1dp_hip_particle_optimize(pic1dp_ctx *c, int32_t irk, int32_t *flag_optimized) {
  CHECK_CTX(c);
  if (flag_optimized) *flag_optimized = 0;
  if (irk != 1 && irk != 2) return fail(PIC1DP_ERR_ARG, "irk must be 1 or 2");
  bool due[3];
  optimize_due(c, due);
  if (irk != 2 || !(due[0] || due[1] || due[2])) return 0;
  if (int rc = require_loaded(c)) return rc;
  if (c->cur != 0) return fail(PIC1DP_ERR_STATE, "particle_optimize must follow the push of sub-step 2");
  if ((due[1] || due[2]) && !c->rng_ready)
    return fail(PIC1DP_ERR_STATE,
                "particle_remove / particle_split continue the loader's random stream: load the markers with "
                "pic1dp_hip_particle_load");
  const pic1dp_input &in = c->in;
  const int ns = in.nspecies, nb = c->nblk, nv = in.nv;
  for (int s = 0; s < ns; ++s) {
    int64_t sum = 0;
    for (int b = 0; b < nb; ++b) sum += c->blk_np[s][b];
    if (sum != c->sp[s].np) return fail(PIC1DP_ERR_STATE, "marker counts per block unknown (uploaded over several blocks)");
  }
  Span tm(c, PIC1DP_IWT_PARTICLE_OPTIMIZE, c->timers_on);
  c->state_version++;
  HIP_TRY(hipStreamSynchronize(c->st));
  // host copy of every owned block, full allocation (valid markers + tail slots)
  struct Block {
    std::vector<double> a[4];  // x v p w
  };
  std::vector<std::vector<Block>> host(ns, std::vector<Block>(nb));
  for (int s = 0; s < ns; ++s) {
    Species &S = c->sp[s];
    const double *dev[4] = {S.set[0].x, S.set[0].v, S.p, S.set[0].w};
    int64_t voff = 0, toff = S.np;
    for (int b = 0; b < nb; ++b) {
      const int64_t na = c->blk_alloc[b], np = c->blk_np[s][b];
      for (int k = 0; k < 4; ++k) {
        host[s][b].a[k].resize(static_cast<size_t>(na));
        if (int rc = get_range(c, dev[k], voff, host[s][b].a[k].data(), np)) return rc;
        if (int rc = get_range(c, dev[k], toff, host[s][b].a[k].data() + np, na - np)) return rc;
      }
      voff += np;
      toff += na - np;
    }
  }
  const double *times[3] = {in.tmerge, in.tremove, in.tsplit};
  const double *thresholds[3] = {in.thshmerge, in.thshremove, in.thshsplit};
  int *counters[3] = {&c->imerge, &c->iremove, &c->isplit};
  (void)times;
  std::vector<double> hist(static_cast<size_t>(ns) * nv), local(nv);
  for (int kind = 0; kind < 3; ++kind) {
    if (!due[kind]) continue;
    // particle_compute_dist_pertb_abs_v: block by block, summed in block order,
    // then over processes (MPI_Allreduce, :392)
    for (int s = 0; s < ns; ++s) {
      double *h = &hist[static_cast<size_t>(s) * nv];
      for (int b = 0; b < nb; ++b) {
        std::fill(local.begin(), local.end(), 0.0);
        opt_histogram(in, c->blk_np[s][b], host[s][b].a[1].data(), host[s][b].a[3].data(), local.data());
        for (int i = 0; i < nv; ++i) h[i] = b == 0 ? local[i] : h[i] + local[i];
      }
    }
    if (c->lay.nranks > 1 || c->comm) {
      double *d = c->d_scratch;
      if (static_cast<size_t>(ns) * nv > static_cast<size_t>(kEnergyBlocks) * 3)
        return fail(PIC1DP_ERR_ARG, "nv too large for the reduction scratch");
      HIP_TRY(hipMemcpy(d, hist.data(), sizeof(double) * ns * nv, hipMemcpyHostToDevice));
      if (int rc = allreduce_doubles(c, d, static_cast<size_t>(ns) * nv)) return rc;
      HIP_TRY(hipStreamSynchronize(c->st));
      HIP_TRY(hipMemcpy(hist.data(), d, sizeof(double) * ns * nv, hipMemcpyDeviceToHost));
    }
    const double th = thresholds[kind][*counters[kind] - 1];
    for (int b = 0; b < nb; ++b)
      for (int s = 0; s < ns; ++s) {
        Block &B = host[s][b];
        const double *h = &hist[static_cast<size_t>(s) * nv];
        int64_t &np = c->blk_np[s][b];
        if (kind == 0)
          opt_merge(in, th, h, np, B.a[0].data(), B.a[1].data(), B.a[2].data(), B.a[3].data());
        else if (kind == 1)
          opt_remove(in, th, h, c->blk_rng[b], np, B.a[0].data(), B.a[1].data(), B.a[2].data(), B.a[3].data());
        else
          opt_split(in, th, h, c->blk_rng[b], c->blk_alloc[b], np, B.a[0].data(), B.a[1].data(), B.a[2].data(),
                    B.a[3].data());
      }
    *counters[kind] += 1;
  }
  // back to the device: valid markers of all blocks packed first, tails behind
  for (int s = 0; s < ns; ++s) {
    Species &S = c->sp[s];
    S.np = 0;
    for (int b = 0; b < nb; ++b) S.np += c->blk_np[s][b];
    double *dev[4] = {S.set[0].x, S.set[0].v, S.p, S.set[0].w};
    int64_t voff = 0, toff = S.np;
    for (int b = 0; b < nb; ++b) {
      const int64_t na = c->blk_alloc[b], np = c->blk_np[s][b];
      for (int k = 0; k < 4; ++k) {
        if (int rc = put_range(c, dev[k], voff, host[s][b].a[k].data(), np)) return rc;
        if (int rc = put_range(c, dev[k], toff, host[s][b].a[k].data() + np, na - np)) return rc;
      }
      voff += np;
      toff += na - np;
    }
  }
  if (flag_optimized) *flag_optimized = 1;
  return tm.end();
}


}  // extern "C"
