// optimize.cpp -- see optimize.hpp.  Host code, compiled without FMA contraction.
#include "optimize.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace pic1dp {

namespace {

// position of velocity v on the nv-point grid over [-v_max, v_max] and the
// |delta f| there: linear interpolation inside, end values outside
// (src/pic1dp_particle.F90:449-463, repeated at :552-566 and :660-674)
struct VGrid {
  const pic1dp_input &in;
  const double *hist;
  double peak;
  VGrid(const pic1dp_input &i, const double *h) : in(i), hist(h), peak(*std::max_element(h, h + i.nv)) {}
  double at(double v, int &cell) const {
    const int last = in.nv - 1;
    const double pos = (v + in.v_max) / (in.v_max * 2.0) * static_cast<double>(last);
    const int c = static_cast<int>(std::floor(pos));
    if (c < 0) {
      cell = 0;
      return hist[0];
    }
    if (c >= last) {
      cell = last;
      return hist[last];
    }
    cell = c;
    const double left = 1.0 - (pos - static_cast<double>(c));
    return hist[c] * left + hist[c + 1] * (1.0 - left);
  }
};

// "remove marker k": the last valid marker takes its place (and will be visited
// next), the count drops by one (:495-503, :579-588)
inline bool drop(int64_t k, int64_t &np, double *x, double *v, double *p, double *w) {
  const int64_t last = np - 1;
  const bool moved = k < last;
  if (moved) {
    x[k] = x[last];
    v[k] = v[last];
    p[k] = p[last];
    w[k] = w[last];
  }
  np = last;
  return moved;
}

}  // namespace

void opt_histogram(const pic1dp_input &in, int64_t np, const double *v, const double *w, double *hist) {
  const double span = in.v_max * 2.0, top = static_cast<double>(in.nv - 1);
  for (int64_t k = 0; k < np; ++k) {
    if (std::fabs(v[k]) >= in.v_max) continue;
    const double pos = (v[k] + in.v_max) / span * top;
    const int c = static_cast<int>(std::floor(pos));
    const double left = 1.0 - (pos - static_cast<double>(c));
    const double a = std::fabs(w[k]);
    hist[c] = hist[c] + left * a;
    hist[c + 1] = hist[c + 1] + (1.0 - left) * a;
  }
}

void opt_merge(const pic1dp_input &in, double threshold, const double *hist, int64_t &np, double *x,
               double *v, double *p, double *w) {
  const VGrid grid(in, hist);
  const double limit = grid.peak * threshold;
  const int nx = in.nx, nv = in.nv;
  // one waiting marker per (x cell, v cell, sign of w); -1 = empty.  The stored
  // index is NOT updated when that slot is later overwritten by a moved marker:
  // the reference does not either, and results depend on it.
  std::vector<int64_t> waiting(static_cast<size_t>(nx) * nv * 2, -1);
  for (int64_t k = 0; k < np; ++k) {
    int vc;
    if (grid.at(v[k], vc) >= limit) continue;  // important marker: leave alone
    double xx = std::fmod(x[k], in.lx);
    if (xx < 0.0) xx = xx + in.lx;
    x[k] = xx;
    int xc = static_cast<int>(std::floor(xx / in.lx * static_cast<double>(nx)));
    if (xc >= nx) xc = nx - 1;  // memory safety (xx == lx)
    int64_t &slot = waiting[(static_cast<size_t>(xc) * nv + vc) * 2 + (w[k] > 0.0 ? 1 : 0)];
    if (slot < 0) {
      slot = k;
      continue;
    }
    const int64_t j = slot;  // merge k into j, weighting positions and velocities by w
    const double wsum = w[j] + w[k];
    x[j] = (w[j] * x[j] + w[k] * x[k]) / wsum;
    v[j] = (w[j] * v[j] + w[k] * v[k]) / wsum;
    p[j] = p[j] + p[k];
    w[j] = wsum;
    slot = -1;
    if (drop(k, np, x, v, p, w)) --k;  // look at the marker moved into k next
  }
}

void opt_remove(const pic1dp_input &in, double threshold, const double *hist, Multirand &rng, int64_t &np,
                double *x, double *v, double *p, double *w) {
  const VGrid grid(in, hist);
  const double limit = grid.peak * threshold;
  const bool by_threshold = in.typeremove == 1;
  const double keep_scale = 1.0 - in.remove_frac;
  for (int64_t k = 0; k < np; ++k) {
    int vc;
    double df = grid.at(v[k], vc);
    if (by_threshold && df >= limit) continue;
    df = df / grid.peak;
    const double dice = rng.real();
    const bool out = by_threshold ? dice < in.remove_frac : dice > df;
    if (out) {
      if (drop(k, np, x, v, p, w)) --k;
    } else if (by_threshold) {  // survivors carry the removed weight
      p[k] = p[k] / keep_scale;
      w[k] = w[k] / keep_scale;
    } else {
      p[k] = p[k] / df;
      w[k] = w[k] / df;
    }
  }
}

void opt_split(const pic1dp_input &in, double threshold, const double *hist, Multirand &rng,
               int64_t nalloc, int64_t &np, double *x, double *v, double *p, double *w) {
  const int ng = in.split_ngroup;
  const int64_t children = 2 * static_cast<int64_t>(ng) - 1;  // new slots per split marker
  if (nalloc - np < children) return;
  const VGrid grid(in, hist);
  const double limit = grid.peak * threshold;
  const double share = static_cast<double>(ng) * 2.0;
  std::vector<double> dv(ng);
  int64_t added = 0;
  const int64_t parents = np;
  for (int64_t k = 0; k < parents; ++k) {
    if (nalloc - (parents + added) < children) break;
    int vc;
    if (grid.at(v[k], vc) <= limit) continue;  // only resonant (important) markers split
    rng.fill_gaussian(dv.data(), ng);
    for (double &d : dv) d = d * 2.0 * in.v_max / static_cast<double>(in.nv) * in.split_dv_sig_frac;
    const double xk = x[k], vk = v[k], pk = p[k] / share, wk = w[k] / share;
    // ng pairs at v +- dv; the last "minus" copy replaces the parent itself
    for (int g = 0; g < ng; ++g) {
      const int64_t plus = parents + added + 2 * g;
      const int64_t minus = g == ng - 1 ? k : plus + 1;
      x[plus] = xk;
      v[plus] = vk + dv[g];
      p[plus] = pk;
      x[minus] = xk;
      v[minus] = vk - dv[g];
      p[minus] = pk;
      if (in.deltaf == 1) {
        w[plus] = wk;
        w[minus] = wk;
      }
    }
    added += children;
  }
  np = parents + added;
}

namespace {

// the final arrangement against the original one: position pos holds marker id_at[pos]; a position that does not hold
// its own marker any more is a hole that a marker from beyond the new count has moved into
void collect_moves(const std::vector<int64_t> &id_at, int64_t np_new, OptMoves &m) {
  m.id.clear();
  for (int64_t pos = 0; pos < np_new; ++pos)
    if (id_at[pos] != pos) m.id.push_back(static_cast<uint32_t>(id_at[pos]));
}

}  // namespace

void plan_merge(const uint32_t *keys, int64_t np, size_t nslots, MergePlan &plan) {
  plan = MergePlan{};
  std::vector<uint32_t> key_at(keys, keys + np);
  std::vector<int64_t> id_at(static_cast<size_t>(np));
  for (int64_t i = 0; i < np; ++i) id_at[i] = i;
  std::vector<int64_t> waiting(nslots, -1);  // position of the marker waiting in a slot (opt_merge)
  for (int64_t k = 0; k < np; ++k) {
    const uint32_t key = key_at[k];
    if (key == 0xFFFFFFFFu) continue;
    int64_t &slot = waiting[key];
    if (slot < 0) {
      slot = k;
      continue;
    }
    const int64_t j = slot;
    plan.dst.push_back(static_cast<uint32_t>(j));
    plan.idk.push_back(static_cast<uint32_t>(id_at[k]));
    slot = -1;
    const int64_t last = np - 1;
    if (k < last) {  // the last valid marker takes the place of k and is looked at next
      key_at[k] = key_at[last];
      id_at[k] = id_at[last];
      np = last;
      --k;
    } else {
      np = last;
      plan.moves.ghost = id_at[k];  // (the loop ends here)
    }
  }
  plan.np_new = np;
  const int64_t ghost = plan.moves.ghost;
  collect_moves(id_at, np, plan.moves);
  plan.moves.ghost = ghost;
}

void plan_remove(const pic1dp_input &in, const uint8_t *skip, const double *df, Multirand &rng, int64_t np, RemovePlan &plan) {
  plan = RemovePlan{};
  const bool by_threshold = in.typeremove == 1;
  std::vector<uint8_t> skip_at;
  std::vector<double> df_at;
  if (by_threshold)
    skip_at.assign(skip, skip + np);
  else
    df_at.assign(df, df + np);
  std::vector<int64_t> id_at(static_cast<size_t>(np));
  for (int64_t i = 0; i < np; ++i) id_at[i] = i;
  for (int64_t k = 0; k < np; ++k) {
    if (by_threshold && skip_at[k]) continue;
    const double dice = rng.real();
    const bool out = by_threshold ? dice < in.remove_frac : dice > df_at[k];
    if (!out) continue;  // (the survivor's weights are rescaled on the device)
    const int64_t last = np - 1;
    if (k < last) {
      if (by_threshold)
        skip_at[k] = skip_at[last];
      else
        df_at[k] = df_at[last];
      id_at[k] = id_at[last];
      np = last;
      --k;
    } else {
      np = last;
      plan.moves.ghost = id_at[k];  // (the loop ends here)
    }
  }
  plan.np_new = np;
  const int64_t ghost = plan.moves.ghost;
  collect_moves(id_at, np, plan.moves);
  plan.moves.ghost = ghost;
  plan.gone_bits.assign(static_cast<size_t>((np + 31) / 32), 0u);
  for (int64_t pos = 0; pos < np; ++pos)
    if (id_at[pos] != pos) plan.gone_bits[pos >> 5] |= 1u << (pos & 31);
}

void plan_split(const pic1dp_input &in, const uint8_t *flag, Multirand &rng, int64_t nalloc, int64_t np, SplitPlan &plan) {
  plan = SplitPlan{};
  plan.np_new = np;
  const int ng = in.split_ngroup;
  const int64_t children = 2 * static_cast<int64_t>(ng) - 1;
  if (nalloc - np < children) return;
  std::vector<double> dv(ng);
  int64_t added = 0;
  const int64_t parents = np;
  for (int64_t k = 0; k < parents; ++k) {
    if (nalloc - (parents + added) < children) break;
    if (!flag[k]) continue;
    rng.fill_gaussian(dv.data(), ng);
    for (double &d : dv) d = d * 2.0 * in.v_max / static_cast<double>(in.nv) * in.split_dv_sig_frac;
    plan.ks.push_back(static_cast<uint32_t>(k));
    plan.dv.insert(plan.dv.end(), dv.begin(), dv.end());
    added += children;
  }
  plan.np_new = parents + added;
}

}  // namespace pic1dp
