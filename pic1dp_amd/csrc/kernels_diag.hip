// kernels_diag.hip -- off the timed path: the fused diagnostics pass of output_all (k_ptcldist), kinetic sums of
// tail slots, cell indices per marker (parity tests), and the copies between contiguous host-side buffers and the
// tiled marker slabs.  gfx950, wave64.
#include "device_diag.hpp"

#include <algorithm>
#include <cmath>
#include "device_math.hpp"

namespace pic1dp {

// ---------------------------------------------------------------------------
// diagnostics
// ---------------------------------------------------------------------------
namespace {

// sum v^2, v^2 p, v^2 w (src/pic1dp_output.F90:126-151): per-workgroup partials,
// the host adds them in workgroup order
__global__ void __launch_bounds__(256)
k_energy_sums(const double *v, const double *p, const double *w, int64_t i0, int64_t n, double *partial) {
  __shared__ double scr[16];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride) {
    const int64_t i = tidx(i0 + k);
    const double v2 = v[i] * v[i];
    s0 += v2;
    s1 += v2 * p[i];
    if (w) s2 += v2 * w[i];
  }
  const double t0 = block_sum(s0, scr);
  const double t1 = block_sum(s1, scr);
  const double t2 = block_sum(s2, scr);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = t0;
    partial[blockIdx.x * 3 + 1] = t1;
    partial[blockIdx.x * 3 + 2] = t2;
  }
}

__global__ void __launch_bounds__(256)
k_cell_indices(const double *x, int64_t np, const GridConst g, int32_t *ixo,
               unsigned long long *count) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < np; i += stride) {
    int ix;
    double wl;
    locate(x[tidx(i)], g, ix, wl);
    if (ixo) ixo[i] = ix;
    if (count) atomicAdd(&count[ix], 1ULL);
  }
}

}  // namespace

namespace {

// One pass over a species for everything output_all needs from the markers:
// * the (x,v) and v histograms of output_ptcldist, src/pic1dp_output.F90:239-315:
//   4-point bilinear weights on an nx_opd x nv_opd grid, markers with
//   |v| >= v_max skipped (:241);
// * the kinetic sums of output_field, sum v^2, v^2 p, v^2 w over ALL markers
//   (:126-151), as per-workgroup partials the host adds in workgroup order.
// LDS = true keeps a private copy of the histograms per workgroup
// (3*(nxo*nvo)+3*nvo doubles) and flushes it with global atomics.  There the v
// histograms are not accumulated marker by marker (64 hot bins: the LDS atomics
// of a wave collide) but formed once per workgroup as the row sums of its (x,v)
// histograms -- the same numbers in exact arithmetic, (sx + (1-sx))*sv = sv, and
// within rounding (<= 1e-15 relative per term) of the separate accumulation.
// LDS = false (grids too large for 160 KiB) adds everything straight to memory.
// The per-marker part and the finish are shared with the DIAG variant of k_step_full.
//
// Round 5 (profiles/r05/sq_counters_diag.json: 0.85 ms at 1e8 markers = 3.8 TB/s for 32 B per marker): the histogram copy
// leaves room for ONE workgroup of 1024 threads per CU, and with one marker per thread and 8-byte loads those sixteen
// waves had 32 KB of loads in flight per CU -- by Little's law ~4 TB/s at the latency the memory system has under load;
// the LDS pipe (twelve FP64 atomics at random bins per marker) was ~70 % busy BEHIND that, not the first limit.  Hence:
// marker PAIRS per thread (16-byte loads, the marker kernels' access shape), the NEXT trip's loads issued before this
// trip's atomics (twice the bytes in flight again), non-temporal loads once the state outgrows the Infinity Cache, the
// two divisions by constants without the hardware's division sequence (bit for bit the same quotients), and the three
// planes interleaved so that a corner's three atomics share one address computation (device_diag.hpp).
// FX: the LDS copy as 64-bit fixed-point sums (device_diag.hpp DistScale): the atomics at 1.9x the rate.
template <bool LDS, bool DELTAF, bool NT, bool FX>
__global__ void __launch_bounds__(1024)
k_ptcldist(const double *x, const double *v, const double *p, const double *w, int64_t np, const DistGeom dg,
           double *out, double *partial, const DistScale fx, int dyn_tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntot = 3 * dg.nxo * dg.nvo + 3 * dg.nvo;
  DistBins b{LDS ? reinterpret_cast<double *>(smem) : out, dg.nxo * dg.nvo, dg.nvo};
  double *scr = reinterpret_cast<double *>(smem) + (LDS ? ntot : 0);  // [16]
  unsigned *sDraw = reinterpret_cast<unsigned *>(scr + 16);            // the drawn chunks' counter
  if constexpr (LDS)
    for (int i = threadIdx.x; i < ntot; i += blockDim.x) b.h[i] = 0.0;
  if (threadIdx.x == 0) *sDraw = 0u;
  __syncthreads();
  const int64_t npair = np >> 1;
  const double2 *x2 = reinterpret_cast<const double2 *>(x), *v2 = reinterpret_cast<const double2 *>(v);
  const double2 *p2 = reinterpret_cast<const double2 *>(p), *w2 = reinterpret_cast<const double2 *>(w);
  DistSums sm;
  // the workgroup's rows of pairs, the last dyn_tail / 16 of them drawn chunk by chunk by its waves (device_math.hpp
  // pair_rows): with one workgroup of sixteen waves per CU, the waves that are done would idle a quarter of the CU each
  const PairRows rows = pair_rows(npair, dyn_tail);
  int k = 0;
  int64_t j = rows.first + threadIdx.x;
  bool have = rows.dealt > 0 || draw_chunk(rows, sDraw, j);
  double2 X = make_double2(0.0, 0.0), V = X, P = X, W = X;
  if (have && j < npair) {
    const int64_t o = tidx2(j);
    X = ld2t<NT>(x2 + o), V = ld2t<NT>(v2 + o), P = ld2t<NT>(p2 + o);
    if constexpr (DELTAF) W = ld2t<NT>(w2 + o);
  }
  while (have) {
    int64_t jn = j + rows.stride;
    bool have_n = true;
    if (++k >= rows.dealt) have_n = draw_chunk(rows, sDraw, jn);
    double2 Xn = make_double2(0.0, 0.0), Vn = Xn, Pn = Xn, Wn = Xn;
    if (have_n && jn < npair) {  // the next trip's loads are under way while this trip's atomics run
      const int64_t o = tidx2(jn);
      Xn = ld2t<NT>(x2 + o), Vn = ld2t<NT>(v2 + o), Pn = ld2t<NT>(p2 + o);
      if constexpr (DELTAF) Wn = ld2t<NT>(w2 + o);
    }
    if (j < npair) {
      ptcldist_one<LDS, DELTAF, FX>(X.x, V.x, P.x, W.x, dg, b, sm, &fx);
      ptcldist_one<LDS, DELTAF, FX>(X.y, V.y, P.y, W.y, dg, b, sm, &fx);
    }
    X = Xn, V = Vn, P = Pn, W = Wn;
    j = jn;
    have = have_n;
  }
  if ((np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(np - 1);
    ptcldist_one<LDS, DELTAF, FX>(x[i], v[i], p[i], DELTAF ? w[i] : 0.0, dg, b, sm, &fx);
  }
  ptcldist_finish<LDS, DELTAF, FX, 6>(dg, b, sm, scr, out, partial, &fx);
}

}  // namespace

int ptcldist_blocks(int64_t np, int nxo, int nvo, int num_cu) {
  const size_t bytes = sizeof(double) * (3 * static_cast<size_t>(nxo) * nvo + 3 * static_cast<size_t>(nvo));
  const bool lds = bytes <= 150 * 1024;
  int64_t blocks = lds ? num_cu : static_cast<int64_t>(num_cu) * 2;
  const int64_t need = ((np >> 1) + 1023) / 1024;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  return static_cast<int>(blocks);
}

// fixed-point sums: a workgroup adds at most its share of the markers into one bin; weights <= 1
bool make_dist_scale(int64_t np, int blocks, bool deltaf, double bound_p, double bound_w, DistScale *out, int threads) {
  DistScale fx{};
  bool use_fx = blocks > 0 && bound_p > 0.0 && (!deltaf || bound_w > 0.0) && std::isfinite(bound_p) && std::isfinite(bound_w);
  if (use_fx) {
    const double th = static_cast<double>(threads);
    const double per_wg = 2.0 * th * std::ceil(static_cast<double>((np >> 1) + 1) / (static_cast<double>(blocks) * th)) + 2.0;
    const int e_n = static_cast<int>(std::ceil(std::log2(per_wg))) + 1;  // 2^e_n > the terms a bin can receive
    const double bounds[3] = {1.0, bound_p, deltaf ? bound_w : 1.0};
    for (int k = 0; k < 3; ++k) {
      int eb;
      (void)std::frexp(bounds[k], &eb);               // bounds[k] < 2^eb
      const int mag = std::min(62 - e_n, 50);         // |term * 2^e| < 2^mag: below 2^51 for to_fixed, and the sums below 2^62
      const int e = mag - eb;
      if (e < -900 || e > 900) use_fx = false;        // (a bound no scale can serve)
      fx.sc[k] = std::ldexp(1.0, e);
      fx.inv[k] = std::ldexp(1.0, -e);
      fx.bound[k] = std::ldexp(1.0, eb);
    }
  }
  *out = fx;
  return use_fx;
}

// bound_p / bound_w: max |p|, max |w| the markers are known not to exceed (with the caller's margin), or <= 0: unknown --
// the pass then sums in doubles.  partial: [blocks][6] = the kinetic sums, max |p|, max |w|, overflow flag per workgroup
hipError_t launch_ptcldist(const double *x, const double *v, const double *p, const double *w,
                           int64_t np, const DistGeom &dg, bool deltaf, double bound_p, double bound_w,
                           double *out, double *partial, int num_cu, int dyn_tail, hipStream_t st, bool *fixed_point) {
  const int nxo = dg.nxo, nvo = dg.nvo;
  const size_t hist = sizeof(double) * (3 * static_cast<size_t>(nxo) * nvo + 3 * static_cast<size_t>(nvo));
  const bool lds = hist <= 150 * 1024;
  const size_t bytes = (lds ? hist : 0) + 18 * sizeof(double);  // + block_sum scratch + the drawn chunks' counter
  const int threads = 1024;
  const int blocks = ptcldist_blocks(np, nxo, nvo, num_cu);
  // x, v, p, w against the 256 MiB Infinity Cache: beyond it the pass streams (PIC1DP_DIAG_NT=0 / 1 insists)
  bool nt = 32.0 * static_cast<double>(np) > 288.0 * 1048576.0;
  if (const char *e = tuning_env("PIC1DP_DIAG_NT")) nt = std::atoi(e) != 0;
  DistScale fx{};
  const bool use_fx = lds && make_dist_scale(np, blocks, deltaf, bound_p, bound_w, &fx);
  if (fixed_point) *fixed_point = use_fx;
  auto go = [&](auto kern) -> hipError_t {
    if (lds && bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(threads), bytes, st, x, v, p, w, np, dg, out,
                       partial, fx, dyn_tail);
    return hipGetLastError();
  };
  if (use_fx) {
    if (nt) return deltaf ? go(k_ptcldist<true, true, true, true>) : go(k_ptcldist<true, false, true, true>);
    return deltaf ? go(k_ptcldist<true, true, false, true>) : go(k_ptcldist<true, false, false, true>);
  }
  if (lds) {
    if (nt) return deltaf ? go(k_ptcldist<true, true, true, false>) : go(k_ptcldist<true, false, true, false>);
    return deltaf ? go(k_ptcldist<true, true, false, false>) : go(k_ptcldist<true, false, false, false>);
  }
  if (nt) return deltaf ? go(k_ptcldist<false, true, true, false>) : go(k_ptcldist<false, false, true, false>);
  return deltaf ? go(k_ptcldist<false, true, false, false>) : go(k_ptcldist<false, false, false, false>);
}

namespace {
__global__ void __launch_bounds__(256) k_pack_record(const PackArgs a, double *out) {
  const int seg = blockIdx.y;
  if (seg >= a.count) return;
  const double *src = a.src[seg];
  double *dst = out + a.dst[seg];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[seg]; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace
hipError_t launch_pack_record(const PackArgs &a, double *out, hipStream_t st) {
  if (a.count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_record, dim3(16, a.count), dim3(256), 0, st, a, out);
  return hipGetLastError();
}

hipError_t launch_energy_sums(const double *v, const double *p, const double *w, int64_t i0, int64_t n,
                              double *partial, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(k_energy_sums, dim3(blocks), dim3(256), 0, st, v, p, w, i0, n, partial);
  return hipGetLastError();
}

namespace {

// host arrays are contiguous, marker arrays tiled: the two meet in these kernels
__global__ void __launch_bounds__(256) k_tile_scatter(double *arr, int64_t i0, const double *src, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    arr[tidx(i0 + k)] = src[k];
}
__global__ void __launch_bounds__(256) k_tile_gather(const double *arr, int64_t i0, double *dst, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    dst[k] = arr[tidx(i0 + k)];
}
__global__ void __launch_bounds__(256) k_tile_copy(double *dst, const double *src, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    dst[tidx(k)] = src[tidx(k)];
}

int copy_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return static_cast<int>(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

hipError_t launch_tile_scatter(double *arr, int64_t i0, const double *src, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_scatter, dim3(copy_blocks(n)), dim3(256), 0, st, arr, i0, src, n);
  return hipGetLastError();
}
hipError_t launch_tile_gather(const double *arr, int64_t i0, double *dst, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_gather, dim3(copy_blocks(n)), dim3(256), 0, st, arr, i0, dst, n);
  return hipGetLastError();
}
hipError_t launch_tile_copy(double *dst, const double *src, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_copy, dim3(copy_blocks(n)), dim3(256), 0, st, dst, src, n);
  return hipGetLastError();
}

hipError_t launch_cell_indices(const double *x, int64_t np, const GridConst &g, int32_t *ix,
                               unsigned long long *count, hipStream_t st) {
  int blocks = static_cast<int>((np + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_cell_indices, dim3(blocks), dim3(256), 0, st, x, np, g, ix, count);
  return hipGetLastError();
}

}  // namespace pic1dp
