// device_field.hpp -- the pieces of the mode-filter solve (src/pic1dp_field.F90:218-257) that more than one
// translation unit runs: the field kernels (kernels_field.hip) and the whole-step marker kernels whose prologue
// solves the field of the previous step itself (kernels_step.hip, FUSED).  Same functions, hence the same bits.
#pragma once
#include "device_math.hpp"

namespace pic1dp {
namespace {

#ifndef PIC1DP_CHAIN_W
#define PIC1DP_CHAIN_W 16
#endif
constexpr int CHAIN_W = PIC1DP_CHAIN_W;  // prefetch depth of the serial mode sums

// chargeden = charge1*nx/lx (- Z*n0 per species for full-f)
// src/pic1dp_interaction.F90:138-148
__device__ __forceinline__ double chargeden_from(const FieldArgs &f, double charge1) {
  double cd = charge1 * f.dnx / f.lx;
  if (!f.deltaf)
    for (int s = 0; s < f.nspecies; ++s) cd = cd - f.Z[s] * f.n0[s];
  return cd;
}

// ---- prediction as six sums (k_step_sums) ----
// The forward sums sum_c fre[c] cd_h[c], sum_c fim[c] cd_h[c] of the NEXT first sub-step's charge density
// from the six sums K (summed over species with Z, and over ranks) and the kept mode (re, im) of the field the
// markers were just advanced to (derivation at k_step_sums); PredTab: what the host knows of the tables
__device__ __forceinline__ void pred_forward_sums(const FieldArgs &f, const PredTab &pt, const double *K, double re, double im,
                                                  double &acc_c, double &acc_s) {
  double off = 0.0;
  if (!f.deltaf)
    for (int s = 0; s < f.nspecies; ++s) off = off + f.Z[s] * f.n0[s];  // chargeden -= Z n0, :142-148
  acc_c = 0.5 * (K[0] + re * K[1] + im * K[2]) * f.dnx / f.lx - off * pt.sum_fre;
  acc_s = 0.5 * (K[3] + re * K[4] + im * K[5]) * f.dnx / f.lx - off * pt.sum_fim;
}

// sum of prod[0..nx) in ascending order, one lane, bit-identical to the sequential loop.
// One wave issues this whole chain, so every instruction counts (a wave64
// VALU or LDS instruction occupies its SIMD for 4 cycles whatever the exec
// mask): two register batches in ping-pong, no copies between them, and
// 16-byte LDS loads when the row is aligned.  16 dependent adds per batch
// cover the LDS round trip of the next one.
template <int W = CHAIN_W>
__device__ __forceinline__ double chain_sum_lds(const double *prod, int nx) {
  double acc = 0.0;
  int ix = 0;
  if (nx > 0 && (reinterpret_cast<uintptr_t>(prod) & 15) != 0) {  // a row block that starts on an odd element
    acc = acc + prod[0];
    ++prod;
    --nx;
  }
  if ((reinterpret_cast<uintptr_t>(prod) & 15) == 0) {
    double A[W], B[W];
    const int nb = nx / W;
    auto load = [](double (&r)[W], const double *q) {
#pragma unroll
      for (int k = 0; k < W; k += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(q + k);
        r[k] = t.x;
        r[k + 1] = t.y;
      }
    };
    if (nb > 0) load(A, prod);
    int b = 0;
    for (; b + 2 <= nb; b += 2) {
      load(B, prod + (b + 1) * W);
#pragma unroll
      for (int k = 0; k < W; ++k) acc = acc + A[k];
      if (b + 2 < nb) load(A, prod + (b + 2) * W);
#pragma unroll
      for (int k = 0; k < W; ++k) acc = acc + B[k];
    }
    if (b < nb) {
#pragma unroll
      for (int k = 0; k < W; ++k) acc = acc + A[k];
    }
    ix = nb * W;
  }
  for (; ix < nx; ++ix) acc = acc + prod[ix];
  return acc;
}

// ---- the forward sums in the order of an npe-rank reference run (f.npe > 1) ----
// Under MPI-AIJ (the operators have PETSC_DECIDE row blocks, src/pic1dp_global.F90:96-133; `make run` starts four
// ranks) MatMultTranspose forms every rank's contribution from its own block of n/npe + (rank < n%npe) rows --
// ascending, from zero -- and the reverse scatter adds the contributions into the owner's entry: the owner's
// own first, then the other ranks in rank order (the tests' CPU statement of the solve takes the same order).  The serial
// chain of the one-rank order is therefore npe chains of nx/npe terms that run side by side in the lanes of a
// wave (a wave64 instruction costs the same for one lane as for sixty-four), and a combine of npe terms:
// nx = 1024 over 8 ranks 5.8 -> ~1 us.  Same products, the reference's N-rank grouping: bit for bit what the
// CPU arithmetic of an npe-rank solve gives for the same chargeden.
__device__ __forceinline__ void rank_block(int n, int npe, int r, int &lo, int &len) {
  const int q = n / npe, rem = n - q * npe;
  lo = r * q + (r < rem ? r : rem);
  len = q + (r < rem ? 1 : 0);
}
// rank whose block is the k-th to be added into an entry owned by `owner`
__device__ __forceinline__ int combine_rank(int k, int owner) { return k == 0 ? owner : (k <= owner ? k - 1 : k); }
// owner of entry m of a vector of nm entries split PETSC_DECIDE over npe ranks (src/pic1dp_field.F90:86-88)
__device__ __forceinline__ int entry_owner(int m, int nm, int npe) {
  const int q = nm / npe, rem = nm - q * npe;
  if (m < rem * (q + 1)) return m / (q + 1);
  return q > 0 ? rem + (m - rem * (q + 1)) / q : npe - 1;
}
// One row of the inverse transform (:251-256) in the order of the npe-rank products: MatMult, and MatMultAdd onto
// its result, each take the columns of the row's own rank first (MPI-AIJ's diagonal block: the PETSC_DECIDE block
// of the nmode entries that rank holds), then every other column, both ascending.  One rank, or nmode <= 2: the
// plain ascending sum.  sMode: [re(nm) | im(nm)]; tables mode-major.
__device__ __forceinline__ double inverse_row(const FieldArgs &f, int ix, const double *sMode) {
  const int nx = f.nx, nm = f.nmode;
  int own = 0, len = nm;
  if (f.npe > 1) rank_block(nm, f.npe, entry_owner(ix, nx, f.npe), own, len);
  const int end = own + len;
  double s = 0.0;
  for (int m = own; m < end; ++m) s = s + f.fre[static_cast<size_t>(m) * nx + ix] * sMode[m];
  for (int m = 0; m < own; ++m) s = s + f.fre[static_cast<size_t>(m) * nx + ix] * sMode[m];
  for (int m = end; m < nm; ++m) s = s + f.fre[static_cast<size_t>(m) * nx + ix] * sMode[m];
  for (int m = own; m < end; ++m) s = s + f.fim[static_cast<size_t>(m) * nx + ix] * sMode[nm + m];
  for (int m = 0; m < own; ++m) s = s + f.fim[static_cast<size_t>(m) * nx + ix] * sMode[nm + m];
  for (int m = end; m < nm; ++m) s = s + f.fim[static_cast<size_t>(m) * nx + ix] * sMode[nm + m];
  return s * 2.0;
}
// sum of prod[lo .. lo + len) ascending, from zero (eight loads in flight ahead of their dependent adds)
template <class P>
__device__ __forceinline__ double chain_partial(P term, int lo, int len) {
  double acc = 0.0;
  int k = 0;
  for (; k + 8 <= len; k += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = term(lo + k + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = acc + t[u];
  }
  for (; k < len; ++k) acc = acc + term(lo + k);
  return acc;
}
// one forward sum of nx terms in the npe-rank order, by one thread (where the partial chains do not run in
// parallel): the same additions in the same grouping
template <class P>
__device__ __forceinline__ double ranks_sum_serial(P term, int nx, int npe, int owner) {
  double tot = 0.0;
  for (int k = 0; k < npe; ++k) {
    int lo, len;
    rank_block(nx, npe, combine_rank(k, owner), lo, len);
    const double part = chain_partial(term, lo, len);
    tot = k == 0 ? part : tot + part;
  }
  return tot;
}

// The two forward sums of ONE kept mode from its products in LDS (sPc = fre * chargeden -> mode_im, sPs = fim *
// chargeden -> mode_re, :231-240), for the lean kernels: threads 0 and 1 return the sums (cos table, -sin table).
// One-rank order: two serial chains.  npe-rank order (f.npe > 1): the 2 npe partial chains side by side -- thread t:
// table t & 1, block t >> 1 (the one mode entry is owned by rank 0: plain rank order) --, then threads 0 and 1 add
// the partials.  Called by every thread (the npe-rank form meets at a barrier); `beside` runs between the chains'
// start and the meeting, on threads that do not carry a chain where there are any.
// Up to sixteen serial sums side by side -- row r = rows[r stride .. + n), each summed in ascending order, every addition
// rounded: the reference's one-rank order -- through the FP64 matrix unit.  v_mfma_f64_4x4x4f64 computes, for four
// independent 4x4 blocks, D = C + A B with the four k-steps as four consecutive fused multiply-adds (tools/
// mfma_probe.hip: in ascending k, one rounding per step): with B = 1 in every element each step is an exactly-rounded
// addition, D[i][*] = (((C + A[i][0]) + A[i][1]) + A[i][2]) + A[i][3] -- four terms of the sequential sum per
// instruction at the instruction's four passes (16 cycles) instead of four dependent v_add_f64 (~12 cycles each), and
// sixteen sums (four blocks of four rows) for the price of one.  Operand layout on gfx950 (the probe): lane l holds,
// of block (l / 4) % 4, A[i = l % 4][k = l / 16] and B[k = l / 16][j = l % 4]; D / C [i = l / 16][j = l % 4].  Row r of
// the caller is row r % 4 of block r / 4: its sum comes out in lane chain_mfma_lane(r).
// That this reproduces the sequential sums BIT FOR BIT on the device at hand is not assumed: create() runs
// launch_chain_selftest and FieldArgs::chain_mfma stays 0 (the chain of additions in one lane) unless it does.
// Called by all 64 lanes of one wave.
__device__ __forceinline__ int chain_mfma_lane(int r) { return 16 * (r & 3) + 4 * (r >> 2); }
__device__ __forceinline__ double chain_rows_mfma(const double *rows, int stride, int nrows, int n) {
  const int lane = threadIdx.x & 63;
  const int k = lane >> 4;
  const int ra = ((lane >> 2) & 3) * 4 + (lane & 3);   // the row this lane's A element belongs to
  const int rd = ((lane >> 2) & 3) * 4 + (lane >> 4);  // the row this lane's D element (the running sum) belongs to
  const double *src = rows + (ra < nrows ? ra : 0) * stride;
  double acc = 0.0;
  const int nb = n >> 2;
  constexpr int U = 8;  // terms of eight instructions per batch; the NEXT batch's LDS reads are under way while this
                        // batch's dependent instructions run (round 5: read-then-run left the LDS round trip, about as long
                        // as the eight instructions themselves, exposed in every batch -- profiles/r05/experiments/
                        // stamps_prologue_solve.log: 4.6 us for 1024 terms)
  int t = 0;
  if (nb >= U) {
    double a[U], b[U];  // two batches in ping-pong; the scheduling fences keep a batch's reads IN FRONT of the other's instructions
#define PIC1DP_CHAIN_READ(r, t0)                                       \
  _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = src[4 * ((t0) + u) + k]; \
  __builtin_amdgcn_sched_barrier(0)
#define PIC1DP_CHAIN_RUN(r) \
  _Pragma("unroll") for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(r[u], 1.0, acc, 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0)
    PIC1DP_CHAIN_READ(a, 0);
    for (; t + 3 * U <= nb; t += 2 * U) {  // (a holds batch t; two more full batches follow)
      PIC1DP_CHAIN_READ(b, t + U);
      PIC1DP_CHAIN_RUN(a);
      PIC1DP_CHAIN_READ(a, t + 2 * U);
      PIC1DP_CHAIN_RUN(b);
    }
    if (t + 2 * U <= nb) {
      PIC1DP_CHAIN_READ(b, t + U);
      PIC1DP_CHAIN_RUN(a);
      PIC1DP_CHAIN_RUN(b);
      t += 2 * U;
    } else {
      PIC1DP_CHAIN_RUN(a);
      t += U;
    }
#undef PIC1DP_CHAIN_READ
#undef PIC1DP_CHAIN_RUN
  }
  for (; t < nb; ++t) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(src[4 * t + k], 1.0, acc, 0, 0, 0);
  const double *dsrc = rows + (rd < nrows ? rd : 0) * stride;  // the last n % 4 terms, by the result's row
  for (int i = 4 * nb; i < n; ++i) acc = acc + dsrc[i];
  return acc;
}

// W: register batch of the chains (the marker kernels, held to 80 VGPRs, take 8)
template <int W = CHAIN_W, class F>
__device__ __forceinline__ double lean_forward_sums(const FieldArgs &f, const double *sPc, const double *sPs, double *sPart,
                                                    F beside) {
  const int nx = f.nx, npe = f.npe;
  double acc = 0.0;
  if (npe > 1 && 2 * npe <= static_cast<int>(blockDim.x)) {
    if (threadIdx.x < 2 * npe) {
      const double *prod = (threadIdx.x & 1) ? sPs : sPc;
      int lo, len;
      rank_block(nx, npe, threadIdx.x >> 1, lo, len);
      sPart[(threadIdx.x & 1) * npe + (threadIdx.x >> 1)] = chain_sum_lds<W>(prod + lo, len);
    }
    beside();
    __syncthreads();
    if (threadIdx.x < 2) {
      const double *part = sPart + threadIdx.x * npe;
      acc = part[0];
      for (int k = 1; k < npe; ++k) acc = acc + part[k];
    }
  } else if (npe == 1 && f.chain_mfma != 0 && nx >= 16) {
    if (threadIdx.x < 64) {  // the first wave, all of it: the matrix unit wants every lane
      const double d = chain_rows_mfma(sPc, static_cast<int>(sPs - sPc), 2, nx);  // row 0 cos -> lane 0, row 1 -sin -> lane 16
      const double ds = __shfl(d, chain_mfma_lane(1), 64);
      acc = threadIdx.x == 1 ? ds : d;
    }
    beside();
  } else {
    if (threadIdx.x < 2) {
      const double *prod = threadIdx.x ? sPs : sPc;
      acc = npe > 1 ? ranks_sum_serial([prod](int i) { return prod[i]; }, nx, npe, 0) : chain_sum_lds<W>(prod, nx);
    }
    beside();
  }
  return acc;
}


}  // namespace
}  // namespace pic1dp
