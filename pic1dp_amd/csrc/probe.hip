// probe.hip -- MEASUREMENT code, not product: libpic1dp_probe.so.  Streaming-rate probes with the marker kernels'
// access shapes (bench.py's second roofline denominator, tools/*.py), and array evaluations of the device functions
// the marker kernels call (the exact divisions, the table-driven exp, -f0'/f0 in both forms) for the parity tests.
// Built by pic1dp_amd/build.py next to libpic1dp_hip.so from the same device headers; loaded by bench.py, tools/
// and tests/ only (pic1dp_amd/probe.py) -- the product library exports none of this.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/pic1dp_probe.h"
#include "check_values.hpp"
#include "device_math.hpp"

namespace pic1dp {

namespace {

__global__ void k_div_check(GridConst g, uint64_t seed, int64_t n, unsigned long long *bad) {
  GridConst gf = g;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = div_check_value(seed, i, g.lx, g.nx);
    const double a = div_lx(x, gf), b = x / g.lx;
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(bad, 1ULL);
  }
}

}  // namespace

namespace {

// bandwidth probe with the access pattern of the particle kernels: NR input
// streams and NW output streams of doubles, 16 B per lane, grid-stride
struct ProbeArgs {
  const double2 *in[8];
  double2 *out[4];
  int64_t npair;
};

template <int NR, int NW, int VARIANT>
__global__ void __launch_bounds__(1024) k_stream_probe(const ProbeArgs a) {
  // VARIANT 0: plain loads/stores; 1: non-temporal; 2: plain, two pairs per lane per trip
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 acc = make_double2(0.0, 0.0);
  constexpr int U = VARIANT == 2 ? 2 : 1;
  for (int64_t j0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j0 < a.npair; j0 += U * stride) {
    double2 s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = make_double2(0.0, 0.0);
      const int64_t j = j0 + u * stride;
      if (j < a.npair) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          double2 t;
          if constexpr (VARIANT == 1) {
            t.x = __builtin_nontemporal_load(&a.in[k][j].x);
            t.y = __builtin_nontemporal_load(&a.in[k][j].y);
          } else {
            t = a.in[k][j];
          }
          s[u].x += t.x;
          s[u].y += t.y;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = j0 + u * stride;
      if (j < a.npair) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          if constexpr (VARIANT == 1) {
            __builtin_nontemporal_store(s[u].x + k, &a.out[k][j].x);
            __builtin_nontemporal_store(s[u].y - k, &a.out[k][j].y);
          } else {
            a.out[k][j] = make_double2(s[u].x + k, s[u].y - k);
          }
        }
      }
      if constexpr (NW == 0) {
        acc.x += s[u].x;
        acc.y += s[u].y;
      }
    }
  }
  if constexpr (NW == 0) {
    if (acc.x == 1.2345e300 && acc.y == -1.2345e300) a.out[0][0] = acc;  // keeps the loads alive
  }
}

template <int NR, int NW>
hipError_t launch_probe_v(const ProbeArgs &a, int variant, int blocks, int threads, hipStream_t st) {
  switch (variant) {
    case 1: hipLaunchKernelGGL((k_stream_probe<NR, NW, 1>), dim3(blocks), dim3(threads), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_stream_probe<NR, NW, 2>), dim3(blocks), dim3(threads), 0, st, a); break;
    default: hipLaunchKernelGGL((k_stream_probe<NR, NW, 0>), dim3(blocks), dim3(threads), 0, st, a); break;
  }
  return hipGetLastError();
}

template <int NR>
hipError_t launch_probe_nr(const ProbeArgs &a, int nw, int variant, int blocks, int threads, hipStream_t st) {
  switch (nw) {
    case 0: return launch_probe_v<NR, 0>(a, variant, blocks, threads, st);
    case 1: return launch_probe_v<NR, 1>(a, variant, blocks, threads, st);
    case 3: return launch_probe_v<NR, 3>(a, variant, blocks, threads, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

namespace {

// Layout probe (tuning only): the traffic of k_step_full -- four arrays read, three
// of them written back in place, 16 B per lane, non-temporal -- over ONE slab, with
// the four arrays either apart by `step` double2 (SoA, TILED = false) or interleaved
// in tiles of 2^lt2 pairs: [x tile | v tile | w tile | p tile] (TILED = true).
template <bool TILED, bool WRITE, bool WG_PER_TILE>
__global__ void __launch_bounds__(1024) k_layout_probe(double2 *base, int64_t step, int lt2, int64_t npair) {
  const int64_t mask = (static_cast<int64_t>(1) << lt2) - 1;
  double2 acc = make_double2(0.0, 0.0);
  auto body = [&](int64_t j) {
    const int64_t o = TILED ? (((j >> lt2) << (lt2 + 2)) + (j & mask)) : j;
    const int64_t d = TILED ? (static_cast<int64_t>(1) << lt2) : step;
    const double2 a = ld2t<true>(base + o), b = ld2t<true>(base + o + d), c = ld2t<true>(base + o + 2 * d),
                  e = ld2t<true>(base + o + 3 * d);
    const double sx = a.x + b.x + c.x + e.x, sy = a.y + b.y + c.y + e.y;
    if constexpr (WRITE) {
      st2t<true>(base + o, sx * 0.25, sy * 0.25);
      st2t<true>(base + o + d, sx * 0.125, sy * 0.125);
      st2t<true>(base + o + 2 * d, sx * 0.0625, sy * 0.0625);
    } else {
      acc.x += sx;
      acc.y += sy;
    }
  };
  if constexpr (WG_PER_TILE) {  // a workgroup walks whole tiles: [x|v|w|p] of one tile, then its next tile
    const int64_t tp = static_cast<int64_t>(1) << lt2, ntile = (npair + tp - 1) >> lt2;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x)
      for (int64_t l = threadIdx.x; l < tp; l += blockDim.x)
        if ((t << lt2) + l < npair) body((t << lt2) + l);
  } else {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair; j += stride) body(j);
  }
  if constexpr (!WRITE) {
    if (acc.x == 1.2345e300 && acc.y == -1.2345e300) base[0] = acc;  // keeps the loads alive
  }
}

}  // namespace

// variant: bit 0 tiled, bit 1 read-only (k_step_half's shape), bit 2 one workgroup per tile
hipError_t launch_layout_probe(double *base, int64_t step_doubles, int log2_tile, int64_t n, int variant, int blocks,
                               int threads, hipStream_t st) {
  double2 *b2 = reinterpret_cast<double2 *>(base);
  const int64_t s2 = step_doubles >> 1, np = n >> 1;
  const int lt2 = log2_tile - 1;
#define PIC1DP_LP(T, W, G) hipLaunchKernelGGL((k_layout_probe<T, W, G>), dim3(blocks), dim3(threads), 0, st, b2, s2, lt2, np)
  switch (variant & 7) {
    case 0: PIC1DP_LP(false, true, false); break;
    case 1: PIC1DP_LP(true, true, false); break;
    case 2: PIC1DP_LP(false, false, false); break;
    case 3: PIC1DP_LP(true, false, false); break;
    case 5: PIC1DP_LP(true, true, true); break;
    case 7: PIC1DP_LP(true, false, true); break;
    default: return hipErrorInvalidValue;
  }
#undef PIC1DP_LP
  return hipGetLastError();
}

hipError_t launch_stream_probe(double *const *in, int nr, double *const *out, int nw, int64_t n,
                               int blocks, int threads, int variant, hipStream_t st) {
  ProbeArgs a{};
  for (int k = 0; k < nr && k < 8; ++k) a.in[k] = reinterpret_cast<const double2 *>(in[k]);
  for (int k = 0; k < 4; ++k) a.out[k] = reinterpret_cast<double2 *>(out[k < nw ? k : 0]);
  a.npair = n >> 1;
  switch (nr) {
    case 1: return launch_probe_nr<1>(a, nw, variant, blocks, threads, st);
    case 4: return launch_probe_nr<4>(a, nw, variant, blocks, threads, st);
    case 7: return launch_probe_nr<7>(a, nw, variant, blocks, threads, st);
    default: return hipErrorInvalidValue;
  }
}

namespace {

__global__ void k_divc_check(double c, double rc, uint64_t seed, int64_t n, unsigned long long *bad) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double a = divc_check_value(seed, i);
    const double q = div_const(a, c, rc, 1), b = a / c;
    if (__double_as_longlong(q) != __double_as_longlong(b)) atomicAdd(bad, 1ULL);
  }
}

// the push's transcendental on its own (tests bound it against libm)
__global__ void __launch_bounds__(256) k_exp_array(const double *x, double *y, int64_t n) {
  exp_table_init();
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = pexp(x[i]);
}

// -f0'/f0 on its own, as the marker kernels evaluate it: FORM 0 the reference's operation order (with the exact
// constant divisions), FORM 1 the one-exp form
template <int DIST, int POW2>
__global__ void __launch_bounds__(256) k_dlnf0_array(const SpeciesConst c, const double *v, double *y, int64_t n) {
  exp_table_init();
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    DivTrue dv;
    y[i] = dlnf0<DIST, POW2>(v[i], c, dv);
  }
}

template <int DIST>
hipError_t launch_dlnf0_array(const SpeciesConst &c, const double *v, double *y, int64_t n, hipStream_t st) {
  if (c.unit)
    hipLaunchKernelGGL((k_dlnf0_array<DIST, 2>), dim3(1024), dim3(256), 0, st, c, v, y, n);
  else if (c.pow2)
    hipLaunchKernelGGL((k_dlnf0_array<DIST, 1>), dim3(1024), dim3(256), 0, st, c, v, y, n);
  else
    hipLaunchKernelGGL((k_dlnf0_array<DIST, 0>), dim3(1024), dim3(256), 0, st, c, v, y, n);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_exp_array(const double *x, double *y, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_exp_array, dim3(1024), dim3(256), 0, st, x, y, n);
  return hipGetLastError();
}

hipError_t launch_divc_check(double c, uint64_t seed, int64_t n, unsigned long long *bad, hipStream_t st) {
  hipLaunchKernelGGL(k_divc_check, dim3(2048), dim3(256), 0, st, c, 1.0 / c, seed, n, bad);
  return hipGetLastError();
}

hipError_t launch_div_check(const GridConst &g, uint64_t seed, int64_t n, unsigned long long *bad,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_div_check, dim3(2048), dim3(256), 0, st, g, seed, n, bad);
  return hipGetLastError();
}

}  // namespace pic1dp

// ===========================================================================
// C ABI (include/pic1dp_probe.h)
// ===========================================================================
using namespace pic1dp;

namespace {

thread_local std::string g_perr;

int pfail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_perr = buf;
  return 1;
}

#define PROBE_TRY(expr)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return pfail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

int num_cu(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256;
  return prop.multiProcessorCount;
}

// device buffers freed on every return path
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};
struct Events {
  hipEvent_t a = nullptr, b = nullptr;
  ~Events() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

std::vector<void *> g_keep;  // slabs a layout probe was asked to leave allocated (until pic1dp_probe_release)

}  // namespace

extern "C" {

const char *pic1dp_probe_last_error(void) { return g_perr.c_str(); }

int pic1dp_probe_stream(int32_t device, int32_t nread, int32_t nwrite, int64_t n, int32_t reps, int32_t blocks,
                        int32_t threads, int32_t variant, double *gbytes_per_s) {
  if (!gbytes_per_s || n < 2 || reps < 1) return pfail("bad argument");
  if ((nread != 1 && nread != 4 && nread != 7) || (nwrite != 0 && nwrite != 1 && nwrite != 3))
    return pfail("nread must be 1, 4 or 7 and nwrite 0, 1 or 3");
  PROBE_TRY(hipSetDevice(device));
  if (threads <= 0) threads = 512;  // the sub-step kernels' shape: four workgroups of 512 per CU
  if (blocks <= 0) blocks = num_cu(device) * (2048 / threads);
  const int nbuf = nread + (nwrite > 0 ? nwrite : 1);
  DevBuf base;
  PROBE_TRY(hipMalloc(&base.p, sizeof(double) * static_cast<size_t>(n) * nbuf));
  PROBE_TRY(hipMemset(base.p, 0, sizeof(double) * static_cast<size_t>(n) * nbuf));
  double *b = static_cast<double *>(base.p);
  double *in[8] = {nullptr}, *out[4] = {nullptr};
  for (int k = 0; k < nread; ++k) in[k] = b + static_cast<size_t>(k) * n;
  for (int k = 0; k < (nwrite > 0 ? nwrite : 1); ++k) out[k] = b + static_cast<size_t>(nread + k) * n;
  Events ev;
  PROBE_TRY(hipEventCreate(&ev.a));
  PROBE_TRY(hipEventCreate(&ev.b));
  PROBE_TRY(launch_stream_probe(in, nread, out, nwrite, n, blocks, threads, variant, nullptr));  // warm-up
  PROBE_TRY(hipEventRecord(ev.a, nullptr));
  for (int r = 0; r < reps; ++r) PROBE_TRY(launch_stream_probe(in, nread, out, nwrite, n, blocks, threads, variant, nullptr));
  PROBE_TRY(hipEventRecord(ev.b, nullptr));
  PROBE_TRY(hipEventSynchronize(ev.b));
  float ms = 0.f;
  PROBE_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
  *gbytes_per_s = 8.0 * static_cast<double>(n) * (nread + nwrite) * reps / (ms * 1e-3) / 1e9;
  return 0;
}

int pic1dp_probe_layout(int32_t device, int64_t n, int32_t log2_tile, int64_t stagger_bytes, int32_t reps, int32_t keep,
                        int32_t blocks, int32_t threads, double ms[6]) {
  if (!ms || n < 2 || reps < 1 || log2_tile < 2 || log2_tile > 24) return pfail("bad argument");
  PROBE_TRY(hipSetDevice(device));
  if (threads <= 0) threads = 768;  // the whole-step kernels' shape: two workgroups of 768 per CU
  if (blocks <= 0) blocks = num_cu(device) * 2;
  const int64_t tile = static_cast<int64_t>(1) << log2_tile;
  n = n / tile * tile;
  if (n < tile) return pfail("n below one tile");
  const size_t unit = static_cast<size_t>(2) << 20;
  const size_t stride = (sizeof(double) * static_cast<size_t>(n) + unit - 1) / unit * unit + static_cast<size_t>(stagger_bytes);
  DevBuf base;
  PROBE_TRY(hipMalloc(&base.p, 4 * stride));
  PROBE_TRY(hipMemset(base.p, 0, 4 * stride));
  Events ev;
  PROBE_TRY(hipEventCreate(&ev.a));
  PROBE_TRY(hipEventCreate(&ev.b));
  // ms[]: SoA r/w, tiled r/w, SoA read-only, tiled read-only, tiled r/w one workgroup per tile, the same read-only
  const int variants[6] = {0, 1, 2, 3, 5, 7};
  double *b = static_cast<double *>(base.p);
  for (int k = 0; k < 6; ++k) {
    for (int r = 0; r < 3; ++r)
      PROBE_TRY(launch_layout_probe(b, static_cast<int64_t>(stride / sizeof(double)), log2_tile, n, variants[k], blocks, threads, nullptr));
    PROBE_TRY(hipEventRecord(ev.a, nullptr));
    for (int r = 0; r < reps; ++r)
      PROBE_TRY(launch_layout_probe(b, static_cast<int64_t>(stride / sizeof(double)), log2_tile, n, variants[k], blocks, threads, nullptr));
    PROBE_TRY(hipEventRecord(ev.b, nullptr));
    PROBE_TRY(hipEventSynchronize(ev.b));
    float t = 0.f;
    PROBE_TRY(hipEventElapsedTime(&t, ev.a, ev.b));
    ms[k] = t / reps;
  }
  if (keep) {
    g_keep.push_back(base.p);
    base.p = nullptr;
  }
  return 0;
}

int pic1dp_probe_release(void) {
  for (void *p : g_keep) (void)hipFree(p);
  g_keep.clear();
  return 0;
}

int pic1dp_probe_div_lx(int32_t device, double lx, int32_t nx, int64_t n, uint64_t seed, int64_t *mismatches) {
  if (!mismatches || !(lx > 0.0) || nx < 1 || n < 0) return pfail("bad argument");
  PROBE_TRY(hipSetDevice(device));
  DevBuf d;
  PROBE_TRY(hipMalloc(&d.p, sizeof(unsigned long long)));
  PROBE_TRY(hipMemset(d.p, 0, sizeof(unsigned long long)));
  GridConst g{};
  g.lx = lx;
  g.dnx = static_cast<double>(nx);
  g.nx = nx;
  g.rlx = 1.0 / lx;
  PROBE_TRY(launch_div_check(g, seed, n, static_cast<unsigned long long *>(d.p), nullptr));
  PROBE_TRY(hipDeviceSynchronize());
  unsigned long long h = 0;
  PROBE_TRY(hipMemcpy(&h, d.p, sizeof h, hipMemcpyDeviceToHost));
  *mismatches = static_cast<int64_t>(h);
  return 0;
}

int pic1dp_probe_div_const(int32_t device, double divisor, int64_t n, uint64_t seed, int64_t *mismatches) {
  if (!mismatches || n < 0 || !(divisor != 0.0)) return pfail("bad argument");
  PROBE_TRY(hipSetDevice(device));
  DevBuf d;
  PROBE_TRY(hipMalloc(&d.p, sizeof(unsigned long long)));
  PROBE_TRY(hipMemset(d.p, 0, sizeof(unsigned long long)));
  PROBE_TRY(launch_divc_check(divisor, seed, n, static_cast<unsigned long long *>(d.p), nullptr));
  PROBE_TRY(hipDeviceSynchronize());
  unsigned long long h = 0;
  PROBE_TRY(hipMemcpy(&h, d.p, sizeof h, hipMemcpyDeviceToHost));
  *mismatches = static_cast<int64_t>(h);
  return 0;
}

int pic1dp_probe_host_div_lx(double lx, int32_t nx, int64_t n, uint64_t seed, int64_t *mismatches) {
  if (!mismatches || !(lx > 0.0) || nx < 1 || n < 0) return pfail("bad argument");
  *mismatches = host_div_check(lx, nx, seed, n);
  return 0;
}

int pic1dp_probe_host_div_const(double divisor, int64_t n, uint64_t seed, int64_t *mismatches) {
  if (!mismatches || n < 0 || !(divisor != 0.0)) return pfail("bad argument");
  *mismatches = host_divc_check(divisor, seed, n);
  return 0;
}

int pic1dp_probe_host_optimize(int32_t kind, int32_t typeremove, int32_t nx, int32_t nv, int32_t split_ngroup, double threshold,
                               uint64_t seed, int64_t np, int64_t nalloc, int64_t *mismatches, int64_t *np_after) {
  if (!mismatches || kind < 0 || kind > 2 || nx < 2 || nv < 2 || np < 0 || nalloc < np || split_ngroup < 1)
    return pfail("bad argument");
  pic1dp_input in{};
  in.lx = 17.45, in.v_max = 8.0, in.nx = nx, in.nv = nv, in.deltaf = 1;
  in.typeremove = typeremove, in.remove_frac = 0.7, in.split_ngroup = split_ngroup, in.split_dv_sig_frac = 0.1;
  in.multirand_al_int = 3;
  *mismatches = pic1dp::host_optimize_check(in, kind, threshold, seed, np, nalloc, np_after);
  return 0;
}

int pic1dp_probe_exp(int32_t device, const double *x, double *y, int64_t n) {
  if (!x || !y || n < 0) return pfail("bad argument");
  if (n == 0) return 0;
  PROBE_TRY(hipSetDevice(device));
  DevBuf d;
  PROBE_TRY(hipMalloc(&d.p, sizeof(double) * 2 * static_cast<size_t>(n)));
  double *dx = static_cast<double *>(d.p);
  PROBE_TRY(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
  PROBE_TRY(launch_exp_array(dx, dx + n, n, nullptr));
  PROBE_TRY(hipDeviceSynchronize());
  PROBE_TRY(hipMemcpy(y, dx + n, sizeof(double) * n, hipMemcpyDeviceToHost));
  return 0;
}

int pic1dp_probe_species_const(const pic1dp_probe_species *sp, int32_t *pow2, int32_t *unit, int32_t *fastc, int32_t *one_exp,
                               double f[7]) {
  if (!sp) return pfail("bad argument");
  const SpeciesConst c = make_species_const(SpeciesInput{sp->iptcldist, sp->charge, sp->mass, sp->temperature,
                                                         sp->temperature2, sp->density, sp->v0}, 0);
  if (pow2) *pow2 = c.pow2;
  if (unit) *unit = c.unit;
  if (fastc) *fastc = c.fastc;
  if (one_exp) *one_exp = c.one_exp;
  if (f) {
    const double v[7] = {c.fq2, c.fq1, c.fq0, c.fm1, c.fm0, c.fd1, c.fd0};
    for (int k = 0; k < 7; ++k) f[k] = v[k];
  }
  return 0;
}

int pic1dp_probe_dlnf0(int32_t device, const pic1dp_probe_species *sp, int32_t form, const double *v, double *y, int64_t n) {
  if (!sp || !v || !y || n < 0 || (form != 0 && form != 1)) return pfail("bad argument");
  if (sp->iptcldist < 0 || sp->iptcldist > 3) return pfail("iptcldist out of range");
  if (form == 1 && sp->iptcldist != 2 && sp->iptcldist != 3) return pfail("the one-exp form exists for iptcldist 2 and 3");
  if (n == 0) return 0;
  const SpeciesConst c = make_species_const(SpeciesInput{sp->iptcldist, sp->charge, sp->mass, sp->temperature,
                                                         sp->temperature2, sp->density, sp->v0}, 0);
  PROBE_TRY(hipSetDevice(device));
  DevBuf d;
  PROBE_TRY(hipMalloc(&d.p, sizeof(double) * 2 * static_cast<size_t>(n)));
  double *dv = static_cast<double *>(d.p);
  PROBE_TRY(hipMemcpy(dv, v, sizeof(double) * n, hipMemcpyHostToDevice));
  hipError_t e = hipErrorInvalidValue;
  switch (sp->iptcldist + (form ? 2 : 0)) {
    case 0: e = launch_dlnf0_array<0>(c, dv, dv + n, n, nullptr); break;
    case 1: e = launch_dlnf0_array<1>(c, dv, dv + n, n, nullptr); break;
    case 2: e = launch_dlnf0_array<2>(c, dv, dv + n, n, nullptr); break;
    case 3: e = launch_dlnf0_array<3>(c, dv, dv + n, n, nullptr); break;
    case 4: e = launch_dlnf0_array<DIST_TS2_ONE_EXP>(c, dv, dv + n, n, nullptr); break;
    case 5: e = launch_dlnf0_array<DIST_BUMP_ONE_EXP>(c, dv, dv + n, n, nullptr); break;
  }
  PROBE_TRY(e);
  PROBE_TRY(hipDeviceSynchronize());
  PROBE_TRY(hipMemcpy(y, dv + n, sizeof(double) * n, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
