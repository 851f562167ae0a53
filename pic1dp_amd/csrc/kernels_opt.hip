// kernels_opt.hip -- marker optimisation events on the GPU (particle_compute_dist_pertb_abs_v, particle_merge,
// particle_remove, particle_split: src/pic1dp_particle.F90:356-746).  gfx950, wave64, -ffp-contract=off.
//
// The reference's routines are sequential by construction: markers are visited in storage order, a removed marker is
// overwritten by the last one and visited again, merge bins remember storage indices, remove and split consume the
// rank's random stream in visiting order.  What IS sequential about them, though, needs one small key per marker, not
// the markers: the host walks the keys (optimize.cpp plan_*) and sends back who merges into whom, who moves where and
// which random numbers the new markers get; everything that touches the 32 B of a marker stays here.
//   |delta f|(v)   k_opt_hist_items -> stable radix sort by bin (rocPRIM) -> k_opt_hist_fold: every bin summed in
//                  storage order by one lane -- the reference's order of additions, hence its bits
//   keys           k_opt_merge_keys (x cell, v cell, sign of w | important), k_opt_remove_vals (|delta f|/peak or the
//                  skip flag), k_opt_split_flags
//   apply          k_opt_moves (tail markers into the holes: sources lie beyond the new count, destinations below
//                  it -- disjoint, in place), k_opt_merge_save / _wrap / _combine, k_opt_remove_scale, k_opt_split_apply,
//                  k_opt_copy_segment (re-packing)
#include <hip/hip_runtime.h>

#include <cstring>  // (rocPRIM's texture iterator calls the host memset)

#include <rocprim/rocprim.hpp>

#include "kernels.hpp"

namespace pic1dp {

namespace {

constexpr int OPT_THREADS = 256;

// block-local marker i of a reference rank block -> offset inside the species' (tiled) arrays
__device__ __forceinline__ int64_t opt_at(const OptBlock &b, int64_t i) {
  return tidx(i < b.nvalid0 ? b.voff + i : b.toff + (i - b.nvalid0));
}

// position of velocity v on the nv-point grid over [-v_max, v_max] and the |delta f| there: linear interpolation
// inside, end values outside (src/pic1dp_particle.F90:449-463, repeated at :552-566 and :660-674) -- the operations of
// optimize.cpp VGrid::at, one for one
__device__ __forceinline__ double opt_df(const OptGrid &g, const double *hist, double v, int &cell) {
  const int last = g.nv - 1;
  const double pos = (v + g.v_max) / (g.v_max * 2.0) * static_cast<double>(last);
  const double fl = floor(pos);
  // (the comparison on the double: a velocity far outside must not overflow the conversion)
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

// x = mod(x, lx); if (x < 0) x = x + lx   (:473-475)
__device__ __forceinline__ double opt_wrap(double x, double lx) {
  double xx = fmod(x, lx);
  if (xx < 0.0) xx = xx + lx;
  return xx;
}

// ---- |delta f|(v) ----------------------------------------------------------
// two items per marker, in storage order: (bin c, left |w|), (bin c + 1, (1 - left) |w|); |v| >= v_max: the dump bin nv
__global__ void __launch_bounds__(OPT_THREADS) k_opt_hist_items(const OptBlock b, const OptGrid g, int64_t np, uint32_t *keys,
                                                               double *vals) {
  const double span = g.v_max * 2.0, top = static_cast<double>(g.nv - 1);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = opt_at(b, i);
    const double v = b.v[o], w = b.w[o];
    uint32_t k0 = g.nv, k1 = g.nv;
    double a0 = 0.0, a1 = 0.0;
    if (!(fabs(v) >= g.v_max)) {
      const double pos = (v + g.v_max) / span * top;
      const double fl = floor(pos);
      const int c = static_cast<int>(fl);
      const double left = 1.0 - (pos - static_cast<double>(c));
      const double a = fabs(w);
      k0 = static_cast<uint32_t>(c);
      k1 = static_cast<uint32_t>(c + 1);
      a0 = left * a;
      a1 = (1.0 - left) * a;
    }
    keys[2 * i] = k0;
    vals[2 * i] = a0;
    keys[2 * i + 1] = k1;
    vals[2 * i + 1] = a1;
  }
}

// bin t: its items lie together after the stable sort, in storage order -- summed from zero, one after the other
__global__ void __launch_bounds__(OPT_THREADS) k_opt_hist_fold(const uint32_t *keys, const double *vals, int64_t n, int nv,
                                                              double *hist) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv) return;
  auto lower = [&](uint32_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t a = lower(static_cast<uint32_t>(t)), e = lower(static_cast<uint32_t>(t) + 1);
  double h = 0.0;
  int64_t i = a;
  for (; i + 8 <= e; i += 8) {  // eight loads in flight ahead of their dependent additions
    double u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = vals[i + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) h = h + u[k];
  }
  for (; i < e; ++i) h = h + vals[i];
  hist[t] = h;
}

// ---- keys ------------------------------------------------------------------
// merge: (x cell, v cell, sign of w) of a marker that may merge; OPT_KEY_IMPORTANT for one that is left alone (:466-485)
__global__ void __launch_bounds__(OPT_THREADS) k_opt_merge_keys(const OptBlock b, const OptGrid g, const double *hist, double limit,
                                                               int64_t np, uint32_t *keys) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = opt_at(b, i);
    int vc;
    uint32_t key = OPT_KEY_IMPORTANT;
    if (!(opt_df(g, hist, b.v[o], vc) >= limit)) {
      const double xx = opt_wrap(b.x[o], g.lx);
      int xc = static_cast<int>(floor(xx / g.lx * static_cast<double>(g.nx)));
      if (xc >= g.nx) xc = g.nx - 1;  // memory safety (xx == lx), as optimize.cpp
      if (xc < 0) xc = 0;             // (NaN)
      key = (static_cast<uint32_t>(xc) * g.nv + vc) * 2u + (b.w[o] > 0.0 ? 1u : 0u);
    }
    keys[i] = key;
  }
}

// remove: the skip flag (typeremove 1: |delta f| >= limit), and |delta f| / peak (typeremove 2: the dice is compared
// with it) (:552-571)
__global__ void __launch_bounds__(OPT_THREADS) k_opt_remove_vals(const OptBlock b, const OptGrid g, const double *hist, double peak,
                                                                double limit, int by_threshold, int64_t np, uint8_t *skip,
                                                                double *df_out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int vc;
    double df = opt_df(g, hist, b.v[opt_at(b, i)], vc);
    if (by_threshold)
      skip[i] = df >= limit ? 1 : 0;
    else
      df_out[i] = df / peak;
  }
}

// split: only resonant (important) markers split (:676)
__global__ void __launch_bounds__(OPT_THREADS) k_opt_split_flags(const OptBlock b, const OptGrid g, const double *hist, double limit,
                                                                int64_t np, uint8_t *flag) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int vc;
    flag[i] = opt_df(g, hist, b.v[opt_at(b, i)], vc) <= limit ? 0 : 1;
  }
}

// ---- apply -----------------------------------------------------------------
// The holes: positions below the new count whose marker is gone (merged away / removed), ascending -- where the
// markers from beyond the new count move to.  The device knows who is gone (the merge pairs' idk; for a remove a bit
// per marker from the host), so the positions need not travel: mark, then a stable selection of the marked positions.
__global__ void __launch_bounds__(OPT_THREADS) k_opt_mark_ids(const uint32_t *ids, int64_t n, int64_t np_new, uint8_t *gone) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x)
    if (ids[t] < np_new) gone[ids[t]] = 1;
}
__global__ void __launch_bounds__(OPT_THREADS) k_opt_mark_bits(const uint32_t *bits, int64_t np_new, uint8_t *gone) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np_new;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    gone[i] = (bits[i >> 5] >> (i & 31)) & 1u;
}

// marker at block-local `id` (beyond the new count) -> position `pos` (below it)
__global__ void __launch_bounds__(OPT_THREADS) k_opt_moves(const OptBlock b, const uint32_t *pos, const uint32_t *id, int64_t n) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t s = opt_at(b, id[t]), d = opt_at(b, pos[t]);
    b.x[d] = b.x[s];
    b.v[d] = b.v[s];
    b.p[d] = b.p[s];
    b.w[d] = b.w[s];
  }
}

// the slot just beyond the new count keeps the marker the walk looked at last (optimize.hpp OptMoves::ghost)
__global__ void k_opt_ghost(const OptBlock b, int64_t id, int64_t pos) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t s = opt_at(b, id), d = opt_at(b, pos);
    b.x[d] = b.x[s];
    b.v[d] = b.v[s];
    b.p[d] = b.p[s];
    b.w[d] = b.w[s];
  }
}

// merge k into j, weighting positions and velocities by w (:486-494); both were wrapped when they were visited.
// In two parts around the moves: what k brings is set aside first (its slot may be the hole a tail marker moves
// into) ...
__global__ void __launch_bounds__(OPT_THREADS) k_opt_merge_save(const OptBlock b, double lx, const uint32_t *idk, int64_t n,
                                                               double *out) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t ok = opt_at(b, idk[t]);
    out[4 * t] = opt_wrap(b.x[ok], lx);
    out[4 * t + 1] = b.v[ok];
    out[4 * t + 2] = b.p[ok];
    out[4 * t + 3] = b.w[ok];
  }
}
// ... every surviving marker that was visited as unimportant holds its wrapped position (:473-476) ...
__global__ void __launch_bounds__(OPT_THREADS) k_opt_merge_wrap(const OptBlock b, const OptGrid g, const double *hist, double limit,
                                                               int64_t np_new) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np_new;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = opt_at(b, i);
    int vc;
    if (!(opt_df(g, hist, b.v[o], vc) >= limit)) b.x[o] = opt_wrap(b.x[o], g.lx);
  }
}
// ... and j, at its final position dst, takes k in
__global__ void __launch_bounds__(OPT_THREADS) k_opt_merge_combine(const OptBlock b, const uint32_t *dst, const double *in, int64_t n) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = opt_at(b, dst[t]);
    const double xj = b.x[o], wj = b.w[o];
    const double xk = in[4 * t], vk = in[4 * t + 1], pk = in[4 * t + 2], wk = in[4 * t + 3];
    const double wsum = wj + wk;
    b.x[o] = (wj * xj + wk * xk) / wsum;
    b.v[o] = (wj * b.v[o] + wk * vk) / wsum;
    b.p[o] = b.p[o] + pk;
    b.w[o] = wsum;
  }
}

// survivors that took part in the draw carry the removed weight (:590-600)
__global__ void __launch_bounds__(OPT_THREADS) k_opt_remove_scale(const OptBlock b, const OptGrid g, const double *hist, double peak,
                                                                 double limit, int by_threshold, double keep_scale,
                                                                 int64_t np_new) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np_new;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t o = opt_at(b, i);
    int vc;
    double df = opt_df(g, hist, b.v[o], vc);
    if (by_threshold) {
      if (df >= limit) continue;
      b.p[o] = b.p[o] / keep_scale;
      b.w[o] = b.w[o] / keep_scale;
    } else {
      df = df / peak;
      b.p[o] = b.p[o] / df;
      b.w[o] = b.w[o] / df;
    }
  }
}

// parent ks[t] -> ng pairs at v +- dv; the last "minus" copy replaces the parent itself (:690-712)
__global__ void __launch_bounds__(OPT_THREADS) k_opt_split_apply(const OptBlock b, int64_t parents, const uint32_t *ks, const double *dv,
                                                                int64_t nsplit, int ng, int deltaf) {
  const int64_t children = 2 * static_cast<int64_t>(ng) - 1;
  const double share = static_cast<double>(ng) * 2.0;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < nsplit;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t k = ks[t], ok = opt_at(b, k);
    const double xk = b.x[ok], vk = b.v[ok], pk = b.p[ok] / share, wk = b.w[ok] / share;
    for (int g = 0; g < ng; ++g) {
      const int64_t plus = parents + t * children + 2 * g;
      const int64_t minus = g == ng - 1 ? k : plus + 1;
      const int64_t op = opt_at(b, plus), om = opt_at(b, minus);
      const double d = dv[t * ng + g];
      b.x[op] = xk;
      b.v[op] = vk + d;
      b.p[op] = pk;
      b.x[om] = xk;
      b.v[om] = vk - d;
      b.p[om] = pk;
      if (deltaf == 1) {
        b.w[op] = wk;
        b.w[om] = wk;
      }
    }
  }
}

// block-local markers [i0, i0 + n) of one block -> markers [doff, doff + n) of another set of arrays
__global__ void __launch_bounds__(OPT_THREADS) k_opt_copy_segment(const OptBlock b, int64_t i0, int64_t n, double *dx, double *dv,
                                                                 double *dp, double *dw, int64_t doff) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t s = opt_at(b, i0 + t), d = tidx(doff + t);
    dx[d] = b.x[s];
    dv[d] = b.v[s];
    dp[d] = b.p[s];
    dw[d] = b.w[s];
  }
}

int opt_blocks(int64_t n) {
  const int64_t need = (n + OPT_THREADS - 1) / OPT_THREADS;
  return static_cast<int>(need < 1 ? 1 : (need > 4096 ? 4096 : need));
}

}  // namespace

// Device scratch of opt_hist_block / opt_holes comes from the caller (grow-only, one allocation per worker and event):
// hipMalloc / hipFree per block would synchronise the whole device under the feet of the other blocks' workers.
namespace {
size_t align256(size_t n) { return (n + 255) & ~static_cast<size_t>(255); }
int hist_sort_bits(int nv) {
  int bits = 1;
  while ((1u << bits) < static_cast<unsigned>(nv + 1)) ++bits;
  return bits;
}
}  // namespace

size_t opt_hist_scratch_bytes(int64_t np, int nv) {
  if (np <= 0) return 0;
  const size_t n = 2 * static_cast<size_t>(np);
  size_t tmp_bytes = 0;
  uint32_t *k = nullptr;
  double *v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k, k, v, v, n, 0, hist_sort_bits(nv), static_cast<hipStream_t>(nullptr));
  return align256(sizeof(uint32_t) * 2 * n) + align256(sizeof(double) * 2 * n) + align256(tmp_bytes ? tmp_bytes : 16);
}

hipError_t opt_hist_block(const OptBlock &b, const OptGrid &g, int64_t np, double *hist, void *scratch, hipStream_t st) {
  if (np <= 0) {  // an empty block: zeros, and -- as every other path of this function -- the stream waited for
    hipError_t e0 = hipMemsetAsync(hist, 0, sizeof(double) * g.nv, st);
    return e0 != hipSuccess ? e0 : hipStreamSynchronize(st);
  }
  const size_t n = 2 * static_cast<size_t>(np);
  char *base = static_cast<char *>(scratch);
  uint32_t *keys = reinterpret_cast<uint32_t *>(base);
  double *vals = reinterpret_cast<double *>(base + align256(sizeof(uint32_t) * 2 * n));
  void *tmp = base + align256(sizeof(uint32_t) * 2 * n) + align256(sizeof(double) * 2 * n);
  hipLaunchKernelGGL(k_opt_hist_items, dim3(opt_blocks(np)), dim3(OPT_THREADS), 0, st, b, g, np, keys, vals);
  hipError_t e = hipGetLastError();
  const int bits = hist_sort_bits(g.nv);
  size_t tmp_bytes = 0;
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys + n, vals, vals + n, n, 0, bits, st);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys + n, vals, vals + n, n, 0, bits, st);  // stable
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_opt_hist_fold, dim3((g.nv + OPT_THREADS - 1) / OPT_THREADS), dim3(OPT_THREADS), 0, st, keys + n, vals + n,
                       static_cast<int64_t>(n), g.nv, hist);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return e;
}

hipError_t opt_merge_keys(const OptBlock &b, const OptGrid &g, const double *hist, double limit, int64_t np, uint32_t *keys,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_opt_merge_keys, dim3(opt_blocks(np)), dim3(OPT_THREADS), 0, st, b, g, hist, limit, np, keys);
  return hipGetLastError();
}
hipError_t opt_remove_vals(const OptBlock &b, const OptGrid &g, const double *hist, double peak, double limit, int by_threshold,
                           int64_t np, uint8_t *skip, double *df, hipStream_t st) {
  hipLaunchKernelGGL(k_opt_remove_vals, dim3(opt_blocks(np)), dim3(OPT_THREADS), 0, st, b, g, hist, peak, limit, by_threshold, np,
                     skip, df);
  return hipGetLastError();
}
hipError_t opt_split_flags(const OptBlock &b, const OptGrid &g, const double *hist, double limit, int64_t np, uint8_t *flag,
                           hipStream_t st) {
  hipLaunchKernelGGL(k_opt_split_flags, dim3(opt_blocks(np)), dim3(OPT_THREADS), 0, st, b, g, hist, limit, np, flag);
  return hipGetLastError();
}
size_t opt_holes_scratch_bytes(int64_t np_new) {
  if (np_new <= 0) return 0;
  size_t tmp_bytes = 0;
  rocprim::counting_iterator<uint32_t> positions(0);
  uint8_t *gone = nullptr;
  uint32_t *holes = nullptr;
  unsigned *count = nullptr;
  (void)rocprim::select(nullptr, tmp_bytes, positions, gone, holes, count, static_cast<size_t>(np_new), static_cast<hipStream_t>(nullptr));
  return align256(static_cast<size_t>(np_new)) + 256 + align256(tmp_bytes ? tmp_bytes : 16);
}

hipError_t opt_holes(const uint32_t *gone_ids, int64_t ngone, const uint32_t *gone_bits, int64_t np_new, int64_t nholes,
                     uint32_t *holes, void *scratch, hipStream_t st) {
  if (nholes <= 0 || np_new <= 0) return hipSuccess;
  char *base = static_cast<char *>(scratch);
  uint8_t *gone = reinterpret_cast<uint8_t *>(base);
  unsigned *count = reinterpret_cast<unsigned *>(base + align256(static_cast<size_t>(np_new)));
  void *tmp = base + align256(static_cast<size_t>(np_new)) + 256;
  hipError_t e = hipSuccess;
  if (gone_bits) {
    hipLaunchKernelGGL(k_opt_mark_bits, dim3(opt_blocks(np_new)), dim3(OPT_THREADS), 0, st, gone_bits, np_new, gone);
    e = hipGetLastError();
  } else {
    e = hipMemsetAsync(gone, 0, static_cast<size_t>(np_new), st);
    if (e == hipSuccess && ngone > 0) {
      hipLaunchKernelGGL(k_opt_mark_ids, dim3(opt_blocks(ngone)), dim3(OPT_THREADS), 0, st, gone_ids, ngone, np_new, gone);
      e = hipGetLastError();
    }
  }
  size_t tmp_bytes = 0;
  rocprim::counting_iterator<uint32_t> positions(0);
  if (e == hipSuccess) e = rocprim::select(nullptr, tmp_bytes, positions, gone, holes, count, static_cast<size_t>(np_new), st);
  if (e == hipSuccess) e = rocprim::select(tmp, tmp_bytes, positions, gone, holes, count, static_cast<size_t>(np_new), st);
  unsigned found = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&found, count, sizeof found, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess && static_cast<int64_t>(found) != nholes) return hipErrorAssert;  // the walk and the marks disagree
  return e;
}

hipError_t opt_moves(const OptBlock &b, const uint32_t *pos, const uint32_t *id, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_opt_moves, dim3(opt_blocks(n)), dim3(OPT_THREADS), 0, st, b, pos, id, n);
  return hipGetLastError();
}
hipError_t opt_merge_apply(const OptBlock &b, const OptGrid &g, const double *hist, double limit, const uint32_t *dst,
                           const uint32_t *idk, int64_t npairs, const uint32_t *move_pos, const uint32_t *move_id, int64_t nmoves,
                           int64_t ghost, int64_t np_new, double *scratch, hipStream_t st) {
  if (npairs > 0) hipLaunchKernelGGL(k_opt_merge_save, dim3(opt_blocks(npairs)), dim3(OPT_THREADS), 0, st, b, g.lx, idk, npairs, scratch);
  if (hipError_t e = opt_moves(b, move_pos, move_id, nmoves, st); e != hipSuccess) return e;
  if (ghost >= 0 && ghost != np_new) hipLaunchKernelGGL(k_opt_ghost, dim3(1), dim3(64), 0, st, b, ghost, np_new);
  const int64_t nwrap = np_new + (ghost >= 0 ? 1 : 0);  // (the marker looked at last was wrapped before it merged)
  if (nwrap > 0) hipLaunchKernelGGL(k_opt_merge_wrap, dim3(opt_blocks(nwrap)), dim3(OPT_THREADS), 0, st, b, g, hist, limit, nwrap);
  if (npairs > 0) hipLaunchKernelGGL(k_opt_merge_combine, dim3(opt_blocks(npairs)), dim3(OPT_THREADS), 0, st, b, dst, scratch, npairs);
  return hipGetLastError();
}
hipError_t opt_remove_apply(const OptBlock &b, const OptGrid &g, const double *hist, double peak, double limit, int by_threshold,
                            double keep_scale, const uint32_t *move_pos, const uint32_t *move_id, int64_t nmoves, int64_t ghost,
                            int64_t np_new, hipStream_t st) {
  if (hipError_t e = opt_moves(b, move_pos, move_id, nmoves, st); e != hipSuccess) return e;
  if (ghost >= 0 && ghost != np_new) hipLaunchKernelGGL(k_opt_ghost, dim3(1), dim3(64), 0, st, b, ghost, np_new);
  if (np_new > 0)
    hipLaunchKernelGGL(k_opt_remove_scale, dim3(opt_blocks(np_new)), dim3(OPT_THREADS), 0, st, b, g, hist, peak, limit, by_threshold,
                       keep_scale, np_new);
  return hipGetLastError();
}
hipError_t opt_split_apply(const OptBlock &b, int64_t parents, const uint32_t *ks, const double *dv, int64_t nsplit, int ng, int deltaf,
                           hipStream_t st) {
  if (nsplit <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_opt_split_apply, dim3(opt_blocks(nsplit)), dim3(OPT_THREADS), 0, st, b, parents, ks, dv, nsplit, ng, deltaf);
  return hipGetLastError();
}
hipError_t opt_copy_segment(const OptBlock &b, int64_t i0, int64_t n, double *dx, double *dv, double *dp, double *dw, int64_t doff,
                            hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_opt_copy_segment, dim3(opt_blocks(n)), dim3(OPT_THREADS), 0, st, b, i0, n, dx, dv, dp, dw, doff);
  return hipGetLastError();
}

}  // namespace pic1dp
