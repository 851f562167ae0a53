// kernels_push.hip -- the sub-step kernels behind the drop-in call sites: k_push (gather + push, optionally with
// the fused wrap + deposit) and k_deposit (stand-alone wrap + deposit).  gfx950, wave64; see kernels_step.hip for
// the whole-step kernels and DESIGN.md for the numbers.
#include "device_math.hpp"

namespace pic1dp {

namespace {

template <int DIST, int MODE, int POW2, bool IRK2, bool FUSED>
__global__ void __launch_bounds__(1024) k_push(const PushArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exp_table_init();
  double *sE = reinterpret_cast<double *>(smem);
  const int nx = a.g.nx;
  double *sR0 = sE + ((nx + 2) & ~1);
  for (int i = threadIdx.x; i < nx; i += blockDim.x) sE[i] = a.E[i];
  if constexpr (FUSED) zero_rho(sR0, a.g);
  if (threadIdx.x == 0) sE[nx] = a.E[0];
  __syncthreads();
  double *sR = sR0;

  constexpr bool HAS_W = (MODE != MODE_FULLF);
  constexpr bool PUSH_V = (MODE != MODE_DF_LIN);
  const int64_t npair = a.np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const double2 *sx2 = reinterpret_cast<const double2 *>(a.src.x);
  const double2 *sv2 = reinterpret_cast<const double2 *>(a.src.v);
  const double2 *sw2 = reinterpret_cast<const double2 *>(a.src.w);
  const double2 *bx2 = reinterpret_cast<const double2 *>(a.base.x);
  const double2 *bv2 = reinterpret_cast<const double2 *>(a.base.v);
  const double2 *bw2 = reinterpret_cast<const double2 *>(a.base.w);
  const double2 *p2 = reinterpret_cast<const double2 *>(a.p);
  double2 *dx2 = reinterpret_cast<double2 *>(a.dst.x);
  double2 *dv2 = reinterpret_cast<double2 *>(a.dst.v);
  double2 *dw2 = reinterpret_cast<double2 *>(a.dst.w);

  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair;
       j += stride) {
    const int64_t o = tidx2(j);
    const double2 X = ld2(sx2 + o), V = ld2(sv2 + o);
    double2 W = make_double2(0.0, 0.0), P = make_double2(0.0, 0.0);
    if constexpr (HAS_W) W = ld2(sw2 + o);
    if constexpr (MODE != MODE_FULLF || FUSED) P = ld2(p2 + o);
    double2 XB = X, VB = V, WB = W;
    if constexpr (IRK2) {
      XB = ld2(bx2 + o);
      if constexpr (PUSH_V) VB = ld2(bv2 + o);
      if constexpr (HAS_W) WB = ld2(bw2 + o);
    }
    One o0 = push_one<DIST, MODE, POW2>(X.x, V.x, W.x, P.x, XB.x, VB.x, WB.x, sE, a.dt, a.g, a.s);
    One o1 = push_one<DIST, MODE, POW2>(X.y, V.y, W.y, P.y, XB.y, VB.y, WB.y, sE, a.dt, a.g, a.s);
    if constexpr (FUSED) {
      o0.x = deposit_one(o0.x, HAS_W ? o0.w : P.x, sR, a.g);
      o1.x = deposit_one(o1.x, HAS_W ? o1.w : P.y, sR, a.g);
    }
    st2(dx2 + o, o0.x, o1.x);
    if constexpr (PUSH_V) st2(dv2 + o, o0.v, o1.v);
    if constexpr (HAS_W) st2(dw2 + o, o0.w, o1.w);
  }
  // odd tail marker
  if ((a.np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(a.np - 1);
    const double x = a.src.x[i], v = a.src.v[i];
    const double w = HAS_W ? a.src.w[i] : 0.0;
    const double p = a.p[i];
    double xb = x, vb = v, wb = w;
    if constexpr (IRK2) {
      xb = a.base.x[i];
      if constexpr (PUSH_V) vb = a.base.v[i];
      if constexpr (HAS_W) wb = a.base.w[i];
    }
    One o = push_one<DIST, MODE, POW2>(x, v, w, p, xb, vb, wb, sE, a.dt, a.g, a.s);
    if constexpr (FUSED) o.x = deposit_one(o.x, HAS_W ? o.w : p, sR, a.g);
    a.dst.x[i] = o.x;
    if constexpr (PUSH_V) a.dst.v[i] = o.v;
    if constexpr (HAS_W) a.dst.w[i] = o.w;
  }
  if constexpr (FUSED) {
    __syncthreads();
    flush_rho(sR0, a.rho, a.g);
  }
}

// stand-alone wrap + deposit (interaction_collect_charge loop :96-114)
__global__ void __launch_bounds__(1024)
k_deposit(double *x, const double *q, double *rho, int64_t np, const GridConst g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sR0 = reinterpret_cast<double *>(smem);
  zero_rho(sR0, g);
  __syncthreads();
  double *sR = sR0;
  const int64_t npair = np >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  double2 *x2 = reinterpret_cast<double2 *>(x);
  const double2 *q2 = reinterpret_cast<const double2 *>(q);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < npair;
       j += stride) {
    const int64_t o = tidx2(j);
    double2 X = ld2(x2 + o);
    const double2 Q = ld2(q2 + o);
    X.x = deposit_one(X.x, Q.x, sR, g);
    X.y = deposit_one(X.y, Q.y, sR, g);
    st2(x2 + o, X.x, X.y);
  }
  if ((np & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = tidx(np - 1);
    x[i] = deposit_one(x[i], q[i], sR, g);
  }
  __syncthreads();
  flush_rho(sR0, rho, g);
}

template <int DIST, int MODE, int POW2, bool IRK2, bool FUSED>
hipError_t launch_push_t(const PushArgs &a, const LaunchCfg &lc, hipStream_t st) {
  auto kern = k_push<DIST, MODE, POW2, IRK2, FUSED>;
  static bool big_lds_ok = false;  // opt in once to > 64 KiB of dynamic LDS (nx >= 4096)
  if (lc.lds > 64 * 1024 && !big_lds_ok) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PARTICLE_LDS_CAP);
    if (e != hipSuccess) return e;
    big_lds_ok = true;
  }
  hipLaunchKernelGGL(kern, dim3(lc.blocks), dim3(lc.threads), lc.lds, st, a);
  return hipGetLastError();
}

template <int DIST, int MODE, int POW2>
hipError_t launch_push_dm(const PushArgs &a, bool fused, const LaunchCfg &lc, hipStream_t st) {
  const bool irk2 = a.irk == 2;
  if (irk2) {
    return fused ? launch_push_t<DIST, MODE, POW2, true, true>(a, lc, st)
                 : launch_push_t<DIST, MODE, POW2, true, false>(a, lc, st);
  }
  return fused ? launch_push_t<DIST, MODE, POW2, false, true>(a, lc, st)
               : launch_push_t<DIST, MODE, POW2, false, false>(a, lc, st);
}

template <int DIST>
hipError_t launch_push_d(const PushArgs &a, bool fused, const LaunchCfg &lc, hipStream_t st) {
  const int mode = a.deltaf ? (a.linear ? MODE_DF_LIN : MODE_DF_NL) : MODE_FULLF;
  const bool pow2 = a.s.pow2 != 0;
  switch (mode) {
    case MODE_DF_NL:
      return pow2 ? launch_push_dm<DIST, MODE_DF_NL, true>(a, fused, lc, st)
                  : launch_push_dm<DIST, MODE_DF_NL, false>(a, fused, lc, st);
    case MODE_DF_LIN:
      return pow2 ? launch_push_dm<DIST, MODE_DF_LIN, true>(a, fused, lc, st)
                  : launch_push_dm<DIST, MODE_DF_LIN, false>(a, fused, lc, st);
    default:
      // full-f evaluates no f0 derivative (one instantiation serves all DIST)
      // but still divides by the mass in the v push
      return pow2 ? launch_push_dm<0, MODE_FULLF, true>(a, fused, lc, st)
                  : launch_push_dm<0, MODE_FULLF, false>(a, fused, lc, st);
  }
}

}  // namespace

hipError_t launch_push(const PushArgs &a, bool fused_deposit, const LaunchCfg &lc,
                       hipStream_t st) {
  switch (a.iptcldist) {
    case 1: return launch_push_d<1>(a, fused_deposit, lc, st);
    case 2: return a.s.one_exp ? launch_push_d<DIST_TS2_ONE_EXP>(a, fused_deposit, lc, st) : launch_push_d<2>(a, fused_deposit, lc, st);
    case 3: return a.s.one_exp ? launch_push_d<DIST_BUMP_ONE_EXP>(a, fused_deposit, lc, st) : launch_push_d<3>(a, fused_deposit, lc, st);
    default: return launch_push_d<0>(a, fused_deposit, lc, st);
  }
}

hipError_t launch_deposit(double *x, const double *q, double *rho, int64_t np, const GridConst &g,
                          const LaunchCfg &lc, hipStream_t st) {
  static bool big_lds_ok = false;
  if (lc.lds > 64 * 1024 && !big_lds_ok) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_deposit),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    big_lds_ok = true;
  }
  hipLaunchKernelGGL(k_deposit, dim3(lc.blocks), dim3(lc.threads), lc.lds, st, x, q, rho, np, g);
  return hipGetLastError();
}

}  // namespace pic1dp
