// capi_optimize.cpp -- marker optimisation events (particle_optimize -> particle_merge / particle_remove /
// particle_split, src/pic1dp_particle.F90:356-813).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>

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


namespace {
struct OptWorker;
void delete_opt_worker(void *w);
}  // namespace

namespace pic1dp_host {
void optimize_release(pic1dp_ctx *c) {
  for (void *w : c->opt_workers) delete_opt_worker(w);
  c->opt_workers.clear();
}
}  // namespace pic1dp_host

namespace {

// RAII for the event's device scratch: grows when a block needs more than the ones before it, lives for the event
// (one hipMalloc / hipFree pair per worker and event, not per block: round 5)
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { (void)hipFree(p); }
  hipError_t alloc(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes ? bytes + bytes / 8 : 16;
    const hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
};

// pinned host staging of a worker (grow-only): pageable copies on a worker's stream cost ~20 ms PER CALL here whatever
// their size (16 blocks of 2.5 MB of keys: 335 ms, against 0.9 ms for the same blocks' 1-byte flags;
// profiles/r05/experiments/opt_event_bench_16_blocks.log), pinned ones what their bytes cost
struct HostPin {
  void *p = nullptr;
  size_t cap = 0;
  ~HostPin() { (void)hipHostFree(p); }
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 64;
    const hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
};

// what one block worker of an event owns; created when an event first needs it, kept in the context until destroy()
// (streams, pinned and device allocations cost milliseconds each and synchronise the device: sixteen workers setting
// themselves up per event cost more than the walks they shared at 1e7 markers)
struct OptWorker {
  hipStream_t st = nullptr;
  DevBuf d_key, d_lists, d_holes;
  HostPin pin_up, pin_down;
  ~OptWorker() {
    if (st) (void)hipStreamDestroy(st);
  }
};

void delete_opt_worker(void *w) { delete static_cast<OptWorker *>(w); }

// The event with the markers staying on the device (kernels_opt.hip): per block the |delta f|(v) histogram in the
// reference's order of additions, one small key per marker to the host, the sequential walk there (optimize.cpp
// plan_*: visiting order, swap-with-last, merge partners, the block's random stream), and what the walk decided back:
// moves, merge pairs, the split parents and their velocity offsets.  4 B per marker over PCIe for a merge, 8 B
// (typeremove 2) or 1 B (typeremove 1) for a remove, 1 B for a split, plus the lists -- instead of 64 B.  Bit for bit
// what opt_merge / opt_remove / opt_split leave (tests/test_gpu_optimize.py, unchanged).
// The walk is what an event costs (round 4: 137-157 ms at 1e7 markers), and the reference's own blocks are independent
// -- one rank each in the reference, own random stream, own markers (src/pic1dp_particle.F90:411-746 run per rank) --, so
// the blocks of an event are walked side by side on host threads (round 5; PIC1DP_OPT_THREADS, default one per block up
// to the host's cores and 16), each with a stream and scratch of its own; the species of a block stay in order (they
// share the block's random stream).  Same decisions, same bits, whatever the number of threads.
int optimize_on_device(pic1dp_ctx *c, const bool due[3]) {
  const pic1dp_input &in = c->in;
  const int ns = in.nspecies, nb = c->nblk, nv = in.nv;
  if (int rc = ensure_second_set(c)) return rc;  // the re-packing target
  // PIC1DP_OPT_TIMING=1: wall clock of the event's phases to stderr (tools/opt_event_bench.py)
  const char *te = tuning_env("PIC1DP_OPT_TIMING");
  const bool timing = te && std::atoi(te) != 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
  double t_hist = 0.0, t_blocks = 0.0;
  int threads_used = 1;
  const OptGrid grid{in.lx, in.v_max, in.nx, nv};
  // the layout the event starts from: valid markers of the owned blocks packed first, their tail slots behind
  std::vector<std::vector<OptBlock>> ob(ns, std::vector<OptBlock>(nb));
  for (int s = 0; s < ns; ++s) {
    Species &S = c->sp[s];
    int64_t voff = 0, toff = S.np;
    for (int b = 0; b < nb; ++b) {
      const int64_t na = c->blk_alloc[b], np = c->blk_np[s][b];
      ob[s][b] = OptBlock{S.set[0].x, S.set[0].v, S.p, S.set[0].w, voff, np, toff};
      voff += np;
      toff += na - np;
    }
  }
  DevBuf d_hist, d_local, d_sort;
  HIP_TRY(d_hist.alloc(sizeof(double) * ns * nv));
  HIP_TRY(d_local.alloc(sizeof(double) * nv));
  const double *thresholds[3] = {in.thshmerge, in.thshremove, in.thshsplit};
  int *counters[3] = {&c->imerge, &c->iremove, &c->isplit};
  std::vector<double> hist(static_cast<size_t>(ns) * nv), local(nv);
  auto to_device = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
    c->opt_pcie_bytes += static_cast<int64_t>(bytes);
    return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  auto to_host = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
    c->opt_pcie_bytes += static_cast<int64_t>(bytes);
    return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess;
  };
  for (int kind = 0; kind < 3; ++kind) {
    if (!due[kind]) continue;
    // particle_compute_dist_pertb_abs_v: block by block, summed in block order, then over processes (:356-403)
    auto t_phase = now();
    for (int s = 0; s < ns; ++s) {
      double *h = &hist[static_cast<size_t>(s) * nv];
      for (int b = 0; b < nb; ++b) {
        HIP_TRY(d_sort.alloc(opt_hist_scratch_bytes(c->blk_np[s][b], nv)));
        HIP_TRY(opt_hist_block(ob[s][b], grid, c->blk_np[s][b], d_local.as<double>(), d_sort.p, c->st));
        HIP_TRY(to_host(local.data(), d_local.p, sizeof(double) * nv));
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
    HIP_TRY(to_device(d_hist.p, hist.data(), sizeof(double) * ns * nv));
    const double th = thresholds[kind][*counters[kind] - 1];
    t_hist += ms_since(t_phase);
    t_phase = now();
    // one reference block (all its species, in order): keys -> walk -> lists -> apply, on the worker's stream
    auto do_block = [&](int b, hipStream_t st, DevBuf &d_key, DevBuf &d_lists, DevBuf &d_holes, HostPin &pin_up, HostPin &pin_down,
                        int64_t &pcie, double (&tph)[3]) -> int {
      // through the worker's pinned staging: up (lists to the device; every call drains the stream, so the staging is free
      // again when it returns) and down (keys: *host points into the staging until the next to_host)
      auto to_device = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        pcie += static_cast<int64_t>(bytes);
        if (!bytes) return hipSuccess;
        hipError_t e = pin_up.reserve(bytes);
        if (e != hipSuccess) return e;
        std::memcpy(pin_up.p, src, bytes);
        e = hipMemcpyAsync(dst, pin_up.p, bytes, hipMemcpyHostToDevice, st);
        return e != hipSuccess ? e : hipStreamSynchronize(st);
      };
      auto to_host = [&](const void **host, const void *src, size_t bytes) -> hipError_t {
        pcie += static_cast<int64_t>(bytes);
        hipError_t e = pin_down.reserve(bytes);
        if (e != hipSuccess) return e;
        *host = pin_down.p;
        if (!bytes) return hipSuccess;
        e = hipMemcpyAsync(pin_down.p, src, bytes, hipMemcpyDeviceToHost, st);
        return e != hipSuccess ? e : hipStreamSynchronize(st);
      };
      for (int s = 0; s < ns; ++s) {
        const OptBlock &B = ob[s][b];
        const double *h = &hist[static_cast<size_t>(s) * nv];
        const double *dh = d_hist.as<double>() + static_cast<size_t>(s) * nv;
        const double peak = *std::max_element(h, h + nv), limit = peak * th;
        int64_t &np = c->blk_np[s][b];
        if (np <= 0) continue;
        if (c->blk_alloc[b] >= (static_cast<int64_t>(1) << 32)) return fail(PIC1DP_ERR_ARG, "a block of 2^32 slots or more");
        // the moved markers' ids travel; the holes they move into are found on the device (opt_holes)
        auto upload_moves = [&](const OptMoves &m, uint32_t *&d_pos, uint32_t *&d_id, size_t extra, char *&d_extra) -> int {
          const size_t nm = m.id.size();
          HIP_TRY(d_lists.alloc(sizeof(uint32_t) * 2 * (nm + 2) + extra + 64));
          d_pos = d_lists.as<uint32_t>();
          d_id = d_pos + nm;
          d_extra = reinterpret_cast<char *>(d_lists.p) + ((sizeof(uint32_t) * 2 * nm + 15) & ~static_cast<size_t>(15));
          HIP_TRY(to_device(d_id, m.id.data(), sizeof(uint32_t) * nm));
          return 0;
        };
        if (kind == 0) {  // particle_merge
          auto t0 = now();
          const void *keys = nullptr;
          HIP_TRY(d_key.alloc(sizeof(uint32_t) * np));
          HIP_TRY(opt_merge_keys(B, grid, dh, limit, np, d_key.as<uint32_t>(), st));
          HIP_TRY(to_host(&keys, d_key.p, sizeof(uint32_t) * np));
          tph[0] += ms_since(t0), t0 = now();
          MergePlan plan;
          plan_merge(static_cast<const uint32_t *>(keys), np, static_cast<size_t>(in.nx) * nv * 2, plan);
          tph[1] += ms_since(t0), t0 = now();
          const size_t npairs = plan.dst.size();
          uint32_t *d_pos, *d_id;
          char *d_extra;
          if (int rc = upload_moves(plan.moves, d_pos, d_id, sizeof(uint32_t) * 2 * (npairs + 2) + sizeof(double) * 4 * npairs + 32, d_extra)) return rc;
          double *d_scr = reinterpret_cast<double *>(d_extra);     // (16-byte aligned)
          uint32_t *d_dst = reinterpret_cast<uint32_t *>(d_scr + 4 * npairs), *d_k = d_dst + npairs;
          HIP_TRY(to_device(d_dst, plan.dst.data(), sizeof(uint32_t) * npairs));
          HIP_TRY(to_device(d_k, plan.idk.data(), sizeof(uint32_t) * npairs));
          HIP_TRY(d_holes.alloc(opt_holes_scratch_bytes(plan.np_new)));
          HIP_TRY(opt_holes(d_k, static_cast<int64_t>(npairs), nullptr, plan.np_new, static_cast<int64_t>(plan.moves.id.size()), d_pos,
                            d_holes.p, st));
          HIP_TRY(opt_merge_apply(B, grid, dh, limit, d_dst, d_k, static_cast<int64_t>(npairs), d_pos, d_id,
                                  static_cast<int64_t>(plan.moves.id.size()), plan.moves.ghost, plan.np_new, d_scr, st));
          HIP_TRY(hipStreamSynchronize(st));
          tph[2] += ms_since(t0);
          np = plan.np_new;
        } else if (kind == 1) {  // particle_remove
          const bool by_threshold = in.typeremove == 1;
          auto t0 = now();
          const void *vals = nullptr;
          HIP_TRY(d_key.alloc((by_threshold ? sizeof(uint8_t) : sizeof(double)) * np));
          HIP_TRY(opt_remove_vals(B, grid, dh, peak, limit, by_threshold ? 1 : 0, np, d_key.as<uint8_t>(), d_key.as<double>(), st));
          HIP_TRY(to_host(&vals, d_key.p, (by_threshold ? sizeof(uint8_t) : sizeof(double)) * np));
          tph[0] += ms_since(t0), t0 = now();
          RemovePlan plan;
          plan_remove(in, by_threshold ? static_cast<const uint8_t *>(vals) : nullptr,
                      by_threshold ? nullptr : static_cast<const double *>(vals), c->blk_rng[b], np, plan);
          tph[1] += ms_since(t0), t0 = now();
          uint32_t *d_pos, *d_id;
          char *d_extra;
          if (int rc = upload_moves(plan.moves, d_pos, d_id, sizeof(uint32_t) * (plan.gone_bits.size() + 4), d_extra)) return rc;
          uint32_t *d_bits = reinterpret_cast<uint32_t *>(d_extra);
          HIP_TRY(to_device(d_bits, plan.gone_bits.data(), sizeof(uint32_t) * plan.gone_bits.size()));
          HIP_TRY(d_holes.alloc(opt_holes_scratch_bytes(plan.np_new)));
          HIP_TRY(opt_holes(nullptr, 0, d_bits, plan.np_new, static_cast<int64_t>(plan.moves.id.size()), d_pos, d_holes.p, st));
          HIP_TRY(opt_remove_apply(B, grid, dh, peak, limit, by_threshold ? 1 : 0, 1.0 - in.remove_frac, d_pos, d_id,
                                   static_cast<int64_t>(plan.moves.id.size()), plan.moves.ghost, plan.np_new, st));
          HIP_TRY(hipStreamSynchronize(st));
          tph[2] += ms_since(t0);
          np = plan.np_new;
        } else {  // particle_split
          auto t0 = now();
          const void *flag = nullptr;
          HIP_TRY(d_key.alloc(sizeof(uint8_t) * np));
          HIP_TRY(opt_split_flags(B, grid, dh, limit, np, d_key.as<uint8_t>(), st));
          HIP_TRY(to_host(&flag, d_key.p, sizeof(uint8_t) * np));
          tph[0] += ms_since(t0), t0 = now();
          SplitPlan plan;
          plan_split(in, static_cast<const uint8_t *>(flag), c->blk_rng[b], c->blk_alloc[b], np, plan);
          tph[1] += ms_since(t0), t0 = now();
          const size_t nsp = plan.ks.size();
          HIP_TRY(d_lists.alloc(sizeof(double) * plan.dv.size() + sizeof(uint32_t) * nsp + 64));
          double *d_dv = d_lists.as<double>();
          uint32_t *d_ks = reinterpret_cast<uint32_t *>(d_dv + plan.dv.size());
          HIP_TRY(to_device(d_ks, plan.ks.data(), sizeof(uint32_t) * nsp));
          HIP_TRY(to_device(d_dv, plan.dv.data(), sizeof(double) * plan.dv.size()));
          HIP_TRY(opt_split_apply(B, np, d_ks, d_dv, static_cast<int64_t>(nsp), in.split_ngroup, in.deltaf, st));
          HIP_TRY(hipStreamSynchronize(st));
          tph[2] += ms_since(t0);
          np = plan.np_new;
        }
        HIP_TRY(hipStreamSynchronize(st));  // (the lists are reused next)
      }
      return 0;
    };
    // the blocks side by side: workers draw block numbers from a counter; every worker owns a stream and its scratch
    int nthreads = std::min<int>(nb, std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    if (const char *e = std::getenv("PIC1DP_OPT_THREADS")) nthreads = std::max(1, std::min(nb, std::atoi(e)));
    std::atomic<int> next{0};
    std::atomic<int64_t> pcie_total{0};
    std::mutex err_mu;
    double tph_all[3] = {0.0, 0.0, 0.0};   // summed over the workers: keys to the host | the walks | lists back + apply
    int err_rc = 0;
    std::string err_msg;
    while (static_cast<int>(c->opt_workers.size()) < nthreads) {
      OptWorker *w = new OptWorker();
      c->opt_workers.push_back(w);
      HIP_TRY(hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking));
    }
    auto worker = [&](int t) {
      OptWorker &w = *static_cast<OptWorker *>(c->opt_workers[t]);
      int64_t pcie = 0;
      int rc = 0;
      double tph[3] = {0.0, 0.0, 0.0};
      const hipError_t e = hipSetDevice(c->device);
      if (e != hipSuccess) rc = fail(PIC1DP_ERR_HIP, "optimisation worker: %s", hipGetErrorString(e));
      for (int b = next.fetch_add(1); rc == 0 && b < nb; b = next.fetch_add(1))
        rc = do_block(b, w.st, w.d_key, w.d_lists, w.d_holes, w.pin_up, w.pin_down, pcie, tph);
      (void)hipStreamSynchronize(w.st);
      pcie_total += pcie;
      {
        std::lock_guard<std::mutex> lk(err_mu);
        for (int k = 0; k < 3; ++k) tph_all[k] += tph[k];
      }
      if (rc != 0) {  // the message is thread-local: hand it to the caller's thread
        std::lock_guard<std::mutex> lk(err_mu);
        if (err_rc == 0) {
          err_rc = rc;
          err_msg = pic1dp_hip_last_error();
        }
        next.store(nb);  // the others stop drawing
      }
    };
    if (nthreads <= 1) {
      worker(0);
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker, t);
      for (auto &t : pool) t.join();
    }
    c->opt_pcie_bytes += pcie_total.load();
    if (err_rc != 0) return fail(err_rc, "%s", err_msg.c_str());
    t_blocks += ms_since(t_phase);
    threads_used = nthreads;
    if (timing)
      std::fprintf(stderr, "pic1dp:   summed over the workers: keys to the host %.1f ms, the walks %.1f ms, lists back + apply %.1f ms\n",
                   tph_all[0], tph_all[1], tph_all[2]);
    *counters[kind] += 1;
  }
  // re-pack into the other slab -- valid markers of all blocks first, their tail slots behind, as the layout wants
  // them -- and make it the species' storage
  const auto t_repack = now();
  for (int s = 0; s < ns; ++s) {
    Species &S = c->sp[s];
    const int64_t as = slab_array_stride(S.nalloc + 2);
    double *nx_ = S.slab[1], *nv_ = S.slab[1] + as, *nw_ = S.slab[1] + 2 * as, *np_ = S.slab[1] + 3 * as;
    S.np = 0;
    for (int b = 0; b < nb; ++b) S.np += c->blk_np[s][b];
    int64_t voff = 0, toff = S.np;
    for (int b = 0; b < nb; ++b) {
      const int64_t na = c->blk_alloc[b], np = c->blk_np[s][b];
      HIP_TRY(opt_copy_segment(ob[s][b], 0, np, nx_, nv_, np_, nw_, voff, c->st));
      HIP_TRY(opt_copy_segment(ob[s][b], np, na - np, nx_, nv_, np_, nw_, toff, c->st));
      voff += np;
      toff += na - np;
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    std::swap(S.slab[0], S.slab[1]);
    S.set[0].x = S.slab[0];
    S.set[0].v = S.slab[0] + as;
    S.set[0].w = S.slab[0] + 2 * as;
    S.p = S.slab[0] + 3 * as;
    S.set[1].x = S.slab[1];
    S.set[1].v = in.linear == 1 ? S.set[0].v : S.slab[1] + as;
    S.set[1].w = in.deltaf == 0 ? S.set[0].w : S.slab[1] + 2 * as;
  }
  if (timing)
    std::fprintf(stderr, "pic1dp: optimisation event: |delta f|(v) %.1f ms, %d blocks on %d host threads %.1f ms, re-pack %.1f ms\n",
                 t_hist, nb, threads_used, t_blocks, ms_since(t_repack));
  return 0;
}

}  // namespace

extern "C" {

// the event on host copies of the owned blocks (PIC1DP_OPT_HOST=1): every marker crosses PCIe twice, 64 B in all
static int optimize_on_host(pic1dp_ctx *c, const bool due[3]) {
  const pic1dp_input &in = c->in;
  const int ns = in.nspecies, nb = c->nblk, nv = in.nv;
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
  for (int s = 0; s < ns; ++s) c->opt_pcie_bytes += 2 * 32 * c->sp[s].nalloc;
  return 0;
}

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
  const int ns = in.nspecies, nb = c->nblk;
  for (int s = 0; s < ns; ++s) {
    int64_t sum = 0;
    for (int b = 0; b < nb; ++b) sum += c->blk_np[s][b];
    if (sum != c->sp[s].np) return fail(PIC1DP_ERR_STATE, "marker counts per block unknown (uploaded over several blocks)");
  }
  Span tm(c, PIC1DP_IWT_PARTICLE_OPTIMIZE, c->timers_on);
  c->state_version++;
  HIP_TRY(hipStreamSynchronize(c->st));
  const char *eh = std::getenv("PIC1DP_OPT_HOST");  // (read per event: rare)
  bool on_host = eh && std::atoi(eh) != 0;
  // the device path names positions inside a block with 32 bits (keys, move lists, bit masks): a block of 2^32 slots or
  // more -- 137 GB of markers in one reference block -- takes the host copies, which count in 64 bits
  for (int b = 0; b < nb; ++b)
    if (c->blk_alloc[b] >= (int64_t{1} << 32) - 2) on_host = true;
  if (int rc = on_host ? optimize_on_host(c, due) : optimize_on_device(c, due)) return rc;
  std::fill(c->diag_max_p.begin(), c->diag_max_p.end(), 0.0);  // (merged / rescaled / split weights: the fixed-point
  std::fill(c->diag_max_w.begin(), c->diag_max_w.end(), 0.0);  //  diagnostics' bounds are void)
  if (flag_optimized) *flag_optimized = 1;
  return tm.end();
}

}  // extern "C"
