// optcheck.cpp -- the planners of the GPU marker optimisation (optimize.hpp plan_*) against the routines they restate
// (opt_merge / opt_remove / opt_split), on the host: random markers, the event applied both ways -- the routine on the
// arrays; the walk over the keys followed by a host statement of what kernels_opt.hip does with its decisions -- and
// the slots compared bit for bit.  Test support (libpic1dp_probe.so), no GPU needed.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "check_values.hpp"
#include "optimize.hpp"

namespace pic1dp {

namespace {

struct Grid {
  const pic1dp_input &in;
  const double *hist;
  double at(double v, int &cell) const {  // (kernels_opt.hip opt_df)
    const int last = in.nv - 1;
    const double pos = (v + in.v_max) / (in.v_max * 2.0) * static_cast<double>(last);
    const double fl = std::floor(pos);
    if (fl < 0.0) {
      cell = 0;
      return hist[0];
    }
    if (fl >= static_cast<double>(last)) {
      cell = last;
      return hist[last];
    }
    const int c = static_cast<int>(fl);
    cell = c;
    const double left = 1.0 - (pos - static_cast<double>(c));
    return hist[c] * left + hist[c + 1] * (1.0 - left);
  }
};
double wrap(double x, double lx) {
  double xx = std::fmod(x, lx);
  if (xx < 0.0) xx = xx + lx;
  return xx;
}
struct Arrays {
  std::vector<double> x, v, p, w;
};
// tail markers into the holes (ascending), then the ghost slot (kernels_opt.hip opt_holes, k_opt_moves, k_opt_ghost)
void apply_moves(Arrays &a, const std::vector<uint8_t> &gone, const OptMoves &m, int64_t np_new) {
  size_t t = 0;
  for (int64_t pos = 0; pos < np_new; ++pos) {
    if (!gone[pos]) continue;
    const uint32_t id = m.id[t++];
    a.x[pos] = a.x[id], a.v[pos] = a.v[id], a.p[pos] = a.p[id], a.w[pos] = a.w[id];
  }
  if (m.ghost >= 0 && m.ghost != np_new) {
    const int64_t id = m.ghost;
    a.x[np_new] = a.x[id], a.v[np_new] = a.v[id], a.p[np_new] = a.p[id], a.w[np_new] = a.w[id];
  }
}

}  // namespace

// kind 0 merge, 1 remove (in.typeremove), 2 split; returns the number of slots (of nalloc) that differ in any array
// between the two ways, or -1 when the marker counts differ
int64_t host_optimize_check(const pic1dp_input &in, int kind, double threshold, uint64_t seed, int64_t np0, int64_t nalloc,
                            int64_t *np_after) {
  Arrays a;
  for (auto *arr : {&a.x, &a.v, &a.p, &a.w}) arr->resize(static_cast<size_t>(nalloc));
  for (int64_t i = 0; i < nalloc; ++i) {
    const double u0 = check_uniform(seed, 4 * i), u1 = check_uniform(seed, 4 * i + 1), u2 = check_uniform(seed, 4 * i + 2),
                 u3 = check_uniform(seed, 4 * i + 3);
    a.x[i] = (u0 * 3.0 - 1.0) * in.lx;          // some outside the box: the merge wraps
    a.v[i] = (u1 * 2.2 - 1.1) * in.v_max;       // some beyond v_max
    a.p[i] = 0.5 + u2;
    a.w[i] = (u3 - 0.5) * 1e-3 * (1.0 + 50.0 * std::exp(-(a.v[i] - 3.0) * (a.v[i] - 3.0)));
  }
  std::vector<double> hist(static_cast<size_t>(in.nv), 0.0);
  opt_histogram(in, np0, a.v.data(), a.w.data(), hist.data());
  const double peak = *std::max_element(hist.begin(), hist.end()), limit = peak * threshold;
  const Grid grid{in, hist.data()};
  // ---- the routine itself
  Arrays r = a;
  int64_t np_r = np0;
  Multirand rng_r, rng_p;
  rng_r.init(in.multirand_al_int, 1, 0, 1, true);
  rng_p = rng_r;
  if (kind == 0) opt_merge(in, threshold, hist.data(), np_r, r.x.data(), r.v.data(), r.p.data(), r.w.data());
  if (kind == 1) opt_remove(in, threshold, hist.data(), rng_r, np_r, r.x.data(), r.v.data(), r.p.data(), r.w.data());
  if (kind == 2) opt_split(in, threshold, hist.data(), rng_r, nalloc, np_r, r.x.data(), r.v.data(), r.p.data(), r.w.data());
  // ---- keys -> plan -> apply
  Arrays d = a;
  int64_t np_d = np0;
  if (kind == 0) {
    std::vector<uint32_t> keys(static_cast<size_t>(np0));
    for (int64_t i = 0; i < np0; ++i) {
      int vc;
      keys[i] = 0xFFFFFFFFu;
      if (!(grid.at(d.v[i], vc) >= limit)) {
        const double xx = wrap(d.x[i], in.lx);
        int xc = static_cast<int>(std::floor(xx / in.lx * static_cast<double>(in.nx)));
        if (xc >= in.nx) xc = in.nx - 1;
        if (xc < 0) xc = 0;
        keys[i] = (static_cast<uint32_t>(xc) * in.nv + vc) * 2u + (d.w[i] > 0.0 ? 1u : 0u);
      }
    }
    MergePlan plan;
    plan_merge(keys.data(), np0, static_cast<size_t>(in.nx) * in.nv * 2, plan);
    const size_t npairs = plan.dst.size();
    std::vector<double> save(4 * npairs);
    for (size_t t = 0; t < npairs; ++t) {  // k_opt_merge_save
      const uint32_t k = plan.idk[t];
      save[4 * t] = wrap(d.x[k], in.lx), save[4 * t + 1] = d.v[k], save[4 * t + 2] = d.p[k], save[4 * t + 3] = d.w[k];
    }
    std::vector<uint8_t> gone(static_cast<size_t>(std::max<int64_t>(plan.np_new, 1)), 0);
    for (uint32_t k : plan.idk)
      if (k < plan.np_new) gone[k] = 1;
    apply_moves(d, gone, plan.moves, plan.np_new);
    const int64_t nwrap = plan.np_new + (plan.moves.ghost >= 0 ? 1 : 0);
    for (int64_t i = 0; i < nwrap; ++i) {  // k_opt_merge_wrap
      int vc;
      if (!(grid.at(d.v[i], vc) >= limit)) d.x[i] = wrap(d.x[i], in.lx);
    }
    for (size_t t = 0; t < npairs; ++t) {  // k_opt_merge_combine
      const uint32_t o = plan.dst[t];
      const double xj = d.x[o], wj = d.w[o], wsum = wj + save[4 * t + 3];
      d.x[o] = (wj * xj + save[4 * t + 3] * save[4 * t]) / wsum;
      d.v[o] = (wj * d.v[o] + save[4 * t + 3] * save[4 * t + 1]) / wsum;
      d.p[o] = d.p[o] + save[4 * t + 2];
      d.w[o] = wsum;
    }
    np_d = plan.np_new;
  } else if (kind == 1) {
    const bool by_threshold = in.typeremove == 1;
    std::vector<uint8_t> skip(static_cast<size_t>(np0));
    std::vector<double> df(static_cast<size_t>(np0));
    for (int64_t i = 0; i < np0; ++i) {
      int vc;
      const double f = grid.at(d.v[i], vc);
      skip[i] = f >= limit ? 1 : 0;
      df[i] = f / peak;
    }
    RemovePlan plan;
    plan_remove(in, by_threshold ? skip.data() : nullptr, by_threshold ? nullptr : df.data(), rng_p, np0, plan);
    std::vector<uint8_t> gone(static_cast<size_t>(std::max<int64_t>(plan.np_new, 1)), 0);
    for (int64_t i = 0; i < plan.np_new; ++i) gone[i] = (plan.gone_bits[i >> 5] >> (i & 31)) & 1u;
    apply_moves(d, gone, plan.moves, plan.np_new);
    const double keep_scale = 1.0 - in.remove_frac;
    for (int64_t i = 0; i < plan.np_new; ++i) {  // k_opt_remove_scale
      int vc;
      double f = grid.at(d.v[i], vc);
      if (by_threshold) {
        if (f >= limit) continue;
        d.p[i] = d.p[i] / keep_scale, d.w[i] = d.w[i] / keep_scale;
      } else {
        f = f / peak;
        d.p[i] = d.p[i] / f, d.w[i] = d.w[i] / f;
      }
    }
    np_d = plan.np_new;
  } else {
    std::vector<uint8_t> flag(static_cast<size_t>(np0));
    for (int64_t i = 0; i < np0; ++i) {
      int vc;
      flag[i] = grid.at(d.v[i], vc) <= limit ? 0 : 1;
    }
    SplitPlan plan;
    plan_split(in, flag.data(), rng_p, nalloc, np0, plan);
    const int ng = in.split_ngroup;
    const int64_t children = 2 * static_cast<int64_t>(ng) - 1;
    const double share = static_cast<double>(ng) * 2.0;
    for (size_t t = 0; t < plan.ks.size(); ++t) {  // k_opt_split_apply
      const int64_t k = plan.ks[t];
      const double xk = d.x[k], vk = d.v[k], pk = d.p[k] / share, wk = d.w[k] / share;
      for (int g = 0; g < ng; ++g) {
        const int64_t plus = np0 + static_cast<int64_t>(t) * children + 2 * g, minus = g == ng - 1 ? k : plus + 1;
        const double dv = plan.dv[t * ng + g];
        d.x[plus] = xk, d.v[plus] = vk + dv, d.p[plus] = pk;
        d.x[minus] = xk, d.v[minus] = vk - dv, d.p[minus] = pk;
        if (in.deltaf == 1) d.w[plus] = wk, d.w[minus] = wk;
      }
    }
    np_d = plan.np_new;
  }
  if (np_after) *np_after = np_r;
  if (np_d != np_r) return -1;
  if (rng_p.next() != rng_r.next()) return -2;  // the two walks consumed the random stream alike
  int64_t bad = 0;
  for (int64_t i = 0; i < nalloc; ++i) {
    // (slots beyond the new count: the routine leaves the wrapped position of a marker it looked at and dropped in its
    // own slot; the device leaves that slot alone -- positions there are never looked at again, weights and velocities are)
    const bool cmp_x = i < np_d;
    if ((cmp_x && std::memcmp(&r.x[i], &d.x[i], 8)) || std::memcmp(&r.v[i], &d.v[i], 8) || std::memcmp(&r.p[i], &d.p[i], 8) ||
        std::memcmp(&r.w[i], &d.w[i], 8))
      ++bad;
  }
  return bad;
}

}  // namespace pic1dp
