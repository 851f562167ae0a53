// kernels_field.hip -- the grid side of the hot path: species sum and scaling of the charge, the mode-filter
// partial-DFT solve (src/pic1dp_field.F90:218-257) in its one-workgroup and wide forms, the paired solves of a
// one-pass step, the one-hop charge exchange, the opt-in finite-difference solver.  gfx950, wave64.
#include "device_field.hpp"
#include "device_math.hpp"
#include "device_xchg.hpp"

namespace pic1dp {

// ---------------------------------------------------------------------------
// field kernels (nx <= a few thousand: one workgroup, latency-bound, tiny)
// ---------------------------------------------------------------------------
namespace {

constexpr int FIELD_THREADS = 256;
// charge2(:) = charge2(:) + charge1(:)*Z over species, from 0
// (src/pic1dp_interaction.F90:81,126-127); accumulators are re-zeroed
__device__ __forceinline__ double charge_local_one(const FieldArgs &f, int ix) {
  double c2 = 0.0;
  for (int s = 0; s < f.nspecies; ++s) {
    double *r = f.rho_sp + static_cast<size_t>(s) * f.nx + ix;
    double c1 = *r;
    *r = 0.0;
    for (int g = 1; g < f.rho_copies; ++g) {  // the copies the workgroups flushed into
      c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
      r[static_cast<size_t>(g) * f.rho_stride] = 0.0;
    }
    c2 = c2 + c1 * f.Z[s];
  }
  f.charge[ix] = c2;
  return c2;
}

__global__ void __launch_bounds__(FIELD_THREADS) k_charge_local(const FieldArgs f) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) charge_local_one(f, ix);
}

template <bool WITH_LOCAL>
__global__ void __launch_bounds__(FIELD_THREADS) k_chargeden(const FieldArgs f) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x)
    f.chargeden[ix] = chargeden_from(f, WITH_LOCAL ? charge_local_one(f, ix) : f.charge[ix]);
}

// k_step_one's prediction turned into this rank's charge2 of the next step's first sub-step:
//   charge2_h = sum_s Z_s * (R0_s + sum_m re_m RA_sm + im_m RB_sm),   re / im = the kept modes of the
// field the markers were just advanced to.  The accumulators are consumed (re-zeroed).  The caller
// reduces f.charge over ranks and scales it (k_chargeden<false>) like any other charge2.
__global__ void __launch_bounds__(FIELD_THREADS) k_pred_combine(const FieldArgs f, double *pred, int nm_pred) {
  const int nx = f.nx, np1 = 1 + 2 * nm_pred;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    for (int s = 0; s < f.nspecies; ++s) {
      double *r = pred + static_cast<size_t>(s) * np1 * nx + ix;
      double c1 = r[0];
      r[0] = 0.0;
      for (int m = 0; m < nm_pred; ++m) {
        double *ra = r + static_cast<size_t>(1 + m) * nx, *rb = r + static_cast<size_t>(1 + nm_pred + m) * nx;
        c1 = c1 + f.mode_re[m] * *ra;
        c1 = c1 + f.mode_im[m] * *rb;
        *ra = 0.0;
        *rb = 0.0;
      }
      c2 = c2 + c1 * f.Z[s];
    }
    f.charge[ix] = c2;
  }
}

// RCCL path of a one-pass step: everything this rank contributes to the two charge sums of the step,
// packed for ONE all-reduce: pack[0] = charge2 of the new state, pack[1 + k] = sum_s Z_s * (R0, RA_m, RB_m)_s
// (the combination with the kept modes is linear, so the species sum and the sum over ranks commute with it).
__global__ void __launch_bounds__(FIELD_THREADS) k_charge_pack(const FieldArgs f, double *pred, int nm_pred, double *pack) {
  const int nx = f.nx, np1 = 1 + 2 * nm_pred;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    pack[ix] = charge_local_one(f, ix);
    for (int k = 0; k < np1; ++k) {
      double c2 = 0.0;
      for (int s = 0; s < f.nspecies; ++s) {
        double *r = pred + (static_cast<size_t>(s) * np1 + k) * nx + ix;
        c2 = c2 + *r * f.Z[s];
        *r = 0.0;
      }
      pack[static_cast<size_t>(1 + k) * nx + ix] = c2;
    }
  }
}

// The six sums arrive in PRED_SUM_COPIES copies (kernels.hpp: workgroup b of the marker kernel adds into copy
// b % PRED_SUM_COPIES -- six addresses shared by every workgroup serialise their atomics at the L2 when the
// workgroups finish together: 29 us of a 140 us kernel at 1e7 markers).  Sum k over the copies, in copy order, all
// loads in flight at once; the copies are re-zeroed (consumed).
__device__ __forceinline__ double pred_sum_take(double *pred, int k) {
  double t[PRED_SUM_COPIES];
#pragma unroll
  for (int c = 0; c < PRED_SUM_COPIES; ++c) t[c] = pred[c * 8 + k];
#pragma unroll
  for (int c = 0; c < PRED_SUM_COPIES; ++c) pred[c * 8 + k] = 0.0;
  double acc = t[0];
#pragma unroll
  for (int c = 1; c < PRED_SUM_COPIES; ++c) acc = acc + t[c];
  return acc;
}

// Call-site path: collect_charge after a noted push(1).  The host will call solve_field next, which works
// from field_chargeden -- so chargeden gets the kept mode's content of the half-step charge density,
//     cd[c] = alpha fre[c] + beta fim[c]   with   sum fre cd = acc_c,  sum fim cd = acc_s,
// from which the ordinary solve reproduces the predicted Eh (to rounding).  What the filter drops is absent
// from this chargeden; nothing in the reference driver reads chargeden between the sub-steps.
// One rank: pred = this rank's copies of the six sums (consumed), K null.  Several ranks: pred null, K the six sums
// already summed over ranks.
__global__ void __launch_bounds__(FIELD_THREADS) k_pred_chargeden(const FieldArgs f, const PredTab pt, double *pred,
                                                                 const double *K) {
  __shared__ double sab[2];
  __shared__ double sK[8];
  if (threadIdx.x < 8) sK[threadIdx.x] = pred ? pred_sum_take(pred, threadIdx.x) : K[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    double ac, as;
    pred_forward_sums(f, pt, sK, f.mode_re[0], f.mode_im[0], ac, as);
    const double det = pt.g11 * pt.g22 - pt.g12 * pt.g12;
    sab[0] = (ac * pt.g22 - as * pt.g12) / det;
    sab[1] = (as * pt.g11 - ac * pt.g12) / det;
  }
  __syncthreads();
  const double alpha = sab[0], beta = sab[1];
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) f.chargeden[ix] = alpha * f.fre[ix] + beta * f.fim[ix];
}

// k_pred_chargeden and the mode-filter solve in one launch (call sites, one rank: collect_charge after a noted
// push(1) leaves both to the solve_field that follows)
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve_pred_sums(const FieldArgs f, const PredTab pt, double *pred);

// this rank's six sums into the head of f.charge (rest zero) for a reduction over ranks (call-site path)
__global__ void __launch_bounds__(FIELD_THREADS) k_pred_to_charge(const FieldArgs f, double *pred) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) f.charge[ix] = 0.0;
  __syncthreads();
  if (threadIdx.x < 8) {
    const double k = pred_sum_take(pred, threadIdx.x);
    if (threadIdx.x < 6) f.charge[threadIdx.x] = k;
  }
}

// k_charge_pack for the six sums: pack[0..nx) = charge2 of the new state, pack[nx..nx+8) = the sums (+ pad)
__global__ void __launch_bounds__(FIELD_THREADS) k_charge_pack_sums(const FieldArgs f, double *pred, double *pack) {
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) pack[ix] = charge_local_one(f, ix);
  if (threadIdx.x < 8) pack[f.nx + threadIdx.x] = pred_sum_take(pred, threadIdx.x);
}

// field_solve_electric, src/pic1dp_field.F90:231-257, with the one-rank PETSc
// summation order: forward sums run over ascending ix in ONE thread per
// (mode, re/im) so the result is bit-identical to the sequential CPU loop.
// chargeden into sCD (and memory) from: the all-reduced charge (neither flag), the
// raw species deposits (WITH_LOCAL, one rank), or field_chargeden itself (FROM_CD)
template <bool WITH_LOCAL, bool FROM_CD>
__device__ __forceinline__ void solve_fill_chargeden(const FieldArgs &f, double *sCD) {
  const int nx = f.nx;
  // four grid points per thread per trip, all loads issued before the first use
  // (one memory round trip instead of four for nx = 1024)
  constexpr int U = 4;
  for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
    double c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      c[u] = 0.0;
      if (ix < nx) {
        if constexpr (FROM_CD) {
          c[u] = f.chargeden[ix];
        } else if constexpr (WITH_LOCAL) {  // src/pic1dp_interaction.F90:126-127
          for (int sp = 0; sp < f.nspecies; ++sp) {
            const double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            double c1 = *r;
            for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
            c[u] = c[u] + c1 * f.Z[sp];
          }
        } else {
          c[u] = f.charge[ix];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      if (ix < nx) {
        double cd = c[u];
        if constexpr (!FROM_CD) {
          if constexpr (WITH_LOCAL) {
            for (int sp = 0; sp < f.nspecies; ++sp)
              for (int g = 0; g < f.rho_copies; ++g)
                f.rho_sp[static_cast<size_t>(g) * f.rho_stride + static_cast<size_t>(sp) * nx + ix] = 0.0;
            f.charge[ix] = c[u];
          }
          cd = chargeden_from(f, c[u]);
          f.chargeden[ix] = cd;
        }
        sCD[ix] = cd;
      }
    }
  }
}

// the solve proper: chargeden in sCD -> mode_re/im, E (+ field energy)
// TREE: the forward sums as workgroup reductions instead of the reference's serial ascending-ix chains.
// Only for a field that has no reference order to keep: the half-step field predicted by k_step_one, whose
// charge already differs from a marker-by-marker deposit by rounding (0.6 us instead of 5.8 us at nx = 1024).
template <bool TREE = false>
__device__ __forceinline__ void solve_body(const FieldArgs &f, double *sCD, double *sMode, double *sScr,
                                           double *sTab) {
  const int nx = f.nx, nm = f.nmode;
  constexpr int U = 4;
  if constexpr (TREE) {
    for (int c = 0; c < 2 * nm; ++c) {  // chain c -> mode c>>1, (c&1 ? cos-table : -sin-table)
      const int m = c >> 1;
      const bool use_cos = c & 1;
      const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
      double part = 0.0;
      for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) part = part + tab[ix] * sCD[ix];
      const double acc = block_sum(part, sScr);
      if (threadIdx.x == 0) {
        if (use_cos) {
          const double im = acc * f.sc_im * f.grad_inv[m];
          sMode[nm + m] = im;
          f.mode_im[m] = im;
        } else {
          const double re = acc * f.sc_re * f.grad_inv[m];
          sMode[m] = re;
          f.mode_re[m] = re;
        }
      }
    }
    __syncthreads();
  } else {
  // forward partial DFT.  Every term table[ix]*chargeden[ix] is rounded on its
  // own in the reference too (no FMA), so the products are formed by all threads
  // at once (coalesced table reads) and only the additions run serially, in the
  // reference's ascending-ix order.
  if (f.tab_lds) {
    const int n = nm * nx;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double cd = sCD[i % nx];
      sTab[i] = f.fre[i] * cd;
      sTab[n + i] = f.fim[i] * cd;
    }
    __syncthreads();
  }
  // npe-rank order with the partial chains side by side: chain c = 2 m + (cos ? 1 : 0), partial (c, k) = the k-th
  // block to be added; the partials meet in sCD (free once the products are formed) when they fit there
  const bool ranks = f.npe > 1;
  const bool ranks_par = ranks && f.tab_lds && 2 * nm * f.npe <= nx;
  if (ranks_par) {
    for (int pc = threadIdx.x; pc < 2 * nm * f.npe; pc += blockDim.x) {
      const int c = pc / f.npe, k = pc - c * f.npe, m = c >> 1;
      const double *prod = sTab + ((c & 1) ? 0 : nm * nx) + m * nx;
      int lo, len;
      rank_block(nx, f.npe, combine_rank(k, entry_owner(m, nm, f.npe)), lo, len);
      sCD[pc] = chain_sum_lds(prod + lo, len);
    }
    __syncthreads();
  }
  // one-rank order, up to eight kept modes, on a device whose matrix unit reproduces the sequential sums: all 2 nm sums
  // side by side through it (device_field.hpp chain_rows_mfma); rows of sTab: m the cos products of mode m, nm + m the -sin ones
  const bool mfma_rows = !ranks && f.tab_lds && f.chain_mfma != 0 && 2 * nm <= 16 && nx >= 16;
  double mf_acc = 0.0;
  if (mfma_rows && threadIdx.x < 64) {
    const double d = chain_rows_mfma(sTab, nx, 2 * nm, nx);
    const int t = static_cast<int>(threadIdx.x) < 2 * nm ? static_cast<int>(threadIdx.x) : 0;
    mf_acc = __shfl(d, chain_mfma_lane((t & 1) ? (t >> 1) : nm + (t >> 1)), 64);
  }
  // thread t -> mode t>>1, (t&1 ? cos-table : -sin-table)
  if (threadIdx.x < 2 * nm) {
    const int m = threadIdx.x >> 1;
    const bool use_cos = threadIdx.x & 1;
    double acc = 0.0;
    int ix = 0;
    if (ranks_par) {
      acc = sCD[threadIdx.x * f.npe];
      for (int k = 1; k < f.npe; ++k) acc = acc + sCD[threadIdx.x * f.npe + k];
    } else if (ranks) {
      const int owner = entry_owner(m, nm, f.npe);
      if (f.tab_lds) {
        const double *prod = sTab + (use_cos ? 0 : nm * nx) + m * nx;
        acc = ranks_sum_serial([prod](int i) { return prod[i]; }, nx, f.npe, owner);
      } else {
        const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
        acc = ranks_sum_serial([tab, sCD](int i) { return tab[i] * sCD[i]; }, nx, f.npe, owner);
      }
    } else if (mfma_rows) {
      acc = mf_acc;
    } else if (f.tab_lds) {
      const double *prod = sTab + (use_cos ? 0 : nm * nx) + m * nx;
      acc = chain_sum_lds(prod, nx);
    } else {
      const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
      for (; ix + 8 <= nx; ix += 8) {
        double t[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          t[k] = tab[ix + k];
          r[k] = sCD[ix + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = acc + t[k] * r[k];
      }
      for (; ix < nx; ++ix) acc = acc + tab[ix] * sCD[ix];
    }
    // :234/:239 VecScale by -1/nx resp. 1/nx, then :243-247 times 1/k
    if (use_cos) {
      const double im = acc * f.sc_im * f.grad_inv[m];
      sMode[nm + m] = im;
      f.mode_im[m] = im;
    } else {
      const double re = acc * f.sc_re * f.grad_inv[m];
      sMode[m] = re;
      f.mode_re[m] = re;
    }
  }
  __syncthreads();
  }  // !TREE

  // inverse: E = 2*(Fre*mode_re + Fim*mode_im), :251-256 (mode order: inverse_row)
  double e2 = 0.0;
  if (nm == 1) {  // the usual case: both table reads of four grid points in flight together
    for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
      double tr[U], ti[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * FIELD_THREADS;
        tr[u] = ix < nx ? f.fre[ix] : 0.0;
        ti[u] = ix < nx ? f.fim[ix] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * FIELD_THREADS;
        if (ix < nx) {
          double s = 0.0;
          s = s + tr[u] * sMode[0];
          s = s + ti[u] * sMode[1];
          const double e = s * 2.0;
          f.E[ix] = e;
          e2 += e * e;
        }
      }
    }
  } else {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      const double e = inverse_row(f, ix, sMode);
      f.E[ix] = e;
      e2 += e * e;
    }
  }
  if (f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

template <bool WITH_LOCAL, bool FROM_CD>
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve(const FieldArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);       // [nx]
  double *sMode = sCD + f.nx;                           // [2*nmode]: re then im
  double *sScr = sMode + 2 * f.nmode;                   // [16]
  double *sTab = sScr + 16;                             // [2][nmode][nx] when tab_lds
  solve_fill_chargeden<WITH_LOCAL, FROM_CD>(f, sCD);
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
}

// Call sites, one rank: collect_charge after a noted push(1) whose charge k_step_one has predicted, and the
// solve_field that follows, in one launch -- k_pred_combine (with the kept modes of the field as it is), the
// scaling, the solve.  The prediction accumulators are consumed.
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve_pred(const FieldArgs f, double *pred, int nm_pred) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  const int nx = f.nx, np1 = 1 + 2 * nm_pred;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    for (int s = 0; s < f.nspecies; ++s) {
      double *r = pred + static_cast<size_t>(s) * np1 * nx + ix;
      double c1 = r[0];
      r[0] = 0.0;
      for (int m = 0; m < nm_pred; ++m) {
        double *ra = r + static_cast<size_t>(1 + m) * nx, *rb = r + static_cast<size_t>(1 + nm_pred + m) * nx;
        c1 = c1 + f.mode_re[m] * *ra;
        c1 = c1 + f.mode_im[m] * *rb;
        *ra = 0.0;
        *rb = 0.0;
      }
      c2 = c2 + c1 * f.Z[s];
    }
    f.charge[ix] = c2;
    const double cd = chargeden_from(f, c2);
    f.chargeden[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();  // every thread has read mode_re / mode_im before solve_body overwrites them
  solve_body(f, sCD, sMode, sScr, sTab);
}

__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve_pred_sums(const FieldArgs f, const PredTab pt, double *pred) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  __shared__ double sab[2];
  __shared__ double sK[8];
  if (threadIdx.x < 8) sK[threadIdx.x] = pred_sum_take(pred, threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    double ac, as;
    pred_forward_sums(f, pt, sK, f.mode_re[0], f.mode_im[0], ac, as);
    const double det = pt.g11 * pt.g22 - pt.g12 * pt.g12;
    sab[0] = (ac * pt.g22 - as * pt.g12) / det;
    sab[1] = (as * pt.g11 - ac * pt.g12) / det;
  }
  __syncthreads();  // (every thread is past its reads of mode_re / mode_im before solve_body overwrites them: only thread 0 read)
  const double alpha = sab[0], beta = sab[1];
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) {
    const double cd = alpha * f.fre[ix] + beta * f.fim[ix];
    f.chargeden[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
}

// ---------------------------------------------------------------------------
// One-hop charge exchange (replaces MPI_Allreduce, src/pic1dp_interaction.F90:130-135,
// for N processes = N GPUs of one node).  Every rank owns an exchange
// area (fine-grained device memory, mapped into every peer through hipIpc handles):
//     flags[2][XCHG_MAX_RANKS]   epoch of the last charge rank q delivered, per parity
//     slots[2][nranks][nx]       the charge2 vectors, one slot per source rank
// Exchange number e (1, 2, ...), parity e & 1:
//   1. charge2 = sum_s rho_s * Z_s (accumulators re-zeroed), stored into slot [rank]
//      of EVERY rank's area (system-scope stores: over xGMI for the peers),
//   2. every storing wave drains its stores (system-scope release fence), the
//      workgroup meets, then one lane per destination stores the flag e,
//   3. lane q of the first wave polls flag q of the OWN area until it reads e
//      (bounded by a wall-clock limit: on expiry the error word is set and the
//      kernel goes on, so the grid always drains),
//   4. charge1[ix] = slots[0][ix] + slots[1][ix] + ... in rank order: the same
//      additions in the same order on every GPU, so charge1 -- and with it E and the
//      marker trajectories -- are bit-identical on all ranks and from run to run,
//      which RCCL's choice of algorithm does not promise.
// Two parities suffice: a rank can start exchange e+2 (same parity as e) only after
// every peer has flagged e+1, which a peer does after it has finished reading e.
// ---------------------------------------------------------------------------
// (the two halves, exchange_post and exchange_wait_sum: device_xchg.hpp -- a one-pass marker launch's tail can run the
// first one itself, kernels.hpp StepTail)

// n values per rank (n <= x.vstride), this rank's in sV -- every thread has filled the elements
// threadIdx.x + k * blockDim.x and only ever touches those -- summed over ranks in rank order, in place.
// posted: the marker launch's tail has stored this rank's values and flag already; only the wait and the sum are left
__device__ __forceinline__ void exchange_vectors(const XchgArgs &x, double *sV, int n, bool posted = false) {
  const long long t_in = x.ticks ? wall_clock64() : 0;
  if (!posted) exchange_post(x, sV, n);
  exchange_wait_sum(x, sV, n);
  if (x.ticks && threadIdx.x == 0) {  // what of this launch was the exchange (the bench's attribution splits it from the solve)
    x.ticks[0] += static_cast<unsigned long long>(wall_clock64() - t_in);
    x.ticks[1] += 1;
  }
}

__device__ __forceinline__ void exchange_charge(const FieldArgs &f, const XchgArgs &x, double *sC) {
  const int nx = f.nx;
  // this rank's charge2: from the species accumulators, or already formed in f.charge (k_pred_combine)
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sC[ix] = x.local_in_charge ? f.charge[ix] : charge_local_one(f, ix);
  exchange_vectors(x, sC, nx);
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) f.charge[ix] = sC[ix];
}

// exchange only: charge1 into field charge (collect_charge call site, many-mode solve)
__global__ void __launch_bounds__(FIELD_THREADS) k_charge_exchange(const FieldArgs f, const XchgArgs x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  exchange_charge(f, x, reinterpret_cast<double *>(smem));
}

// local charge -> exchange -> chargeden -> solve: one launch per sub-step
__global__ void __launch_bounds__(FIELD_THREADS) k_field_solve_xchg(const FieldArgs f, const XchgArgs x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  exchange_charge(f, x, sCD);
  for (int ix = threadIdx.x; ix < f.nx; ix += blockDim.x) {  // own elements again
    const double cd = chargeden_from(f, sCD[ix]);
    f.chargeden[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
}

// One launch for both fields of the one-pass-per-step scheme (k_step_one): the field of the new
// state from its deposited charge (as k_field_solve / k_field_solve_xchg), then -- with the kept modes
// just found -- the predicted charge of the next first sub-step (as k_pred_combine), summed over
// ranks when XCHG, scaled, and solved into the half-step field of the NEXT step.
// SRC: 0 one rank (charges from the local accumulators), 1 one-hop exchange (two exchanges inside this
// launch), 2 packed (pa.pack holds the rank-summed charge2 and Z-weighted prediction slices, k_charge_pack +
// one all-reduce)
template <int SRC>
__global__ void __launch_bounds__(FIELD_THREADS)
k_field_solve_pair(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);
  double *sMode = sCD + f.nx;
  double *sScr = sMode + 2 * f.nmode;
  double *sTab = sScr + 16;
  const int nx = f.nx, nm = f.nmode, np1 = 1 + 2 * nm;
  // SRC 1: [charge2 | Z-weighted prediction slices] of this rank, then of all ranks, behind the solve's tiles
  double *sV = sTab + (f.tab_lds ? 2 * static_cast<size_t>(nm) * nx : 0);
  const double *pk = pa.pack;  // SRC 2: the same slices, all-reduced in memory
  if constexpr (SRC == 1) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      sV[ix] = charge_local_one(f, ix);
      for (int k = 0; k < np1; ++k) {
        double c2 = 0.0;
        for (int s = 0; s < f.nspecies; ++s) {
          double *r = pa.pred + (static_cast<size_t>(s) * np1 + k) * nx + ix;
          c2 = c2 + *r * f.Z[s];
          *r = 0.0;
        }
        sV[static_cast<size_t>(1 + k) * nx + ix] = c2;
      }
    }
    // element i of the packed vector belongs to thread i % blockDim; with nx a multiple of blockDim that is
    // the thread that wrote it -- otherwise meet first
    __syncthreads();
    exchange_vectors(x1, sV, (1 + np1) * nx);
    __syncthreads();
    pk = sV;
  }
  if constexpr (SRC != 0) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      const double c = pk[ix];
      f.charge[ix] = c;
      const double cd = chargeden_from(f, c);
      f.chargeden[ix] = cd;
      sCD[ix] = cd;
    }
  } else {
    solve_fill_chargeden<true, false>(f, sCD);
  }
  __syncthreads();
  solve_body(f, sCD, sMode, sScr, sTab);
  __syncthreads();  // E, mode_re/im (also in sMode) are final; sCD and sTab are free again
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    if constexpr (SRC != 0) {
      const double *r = pk + nx + ix;  // Z-weighted, summed over species and ranks
      c2 = r[0];
      for (int m = 0; m < nm; ++m) {
        c2 = c2 + sMode[m] * r[static_cast<size_t>(1 + m) * nx];
        c2 = c2 + sMode[nm + m] * r[static_cast<size_t>(1 + nm + m) * nx];
      }
    } else {
      for (int s = 0; s < f.nspecies; ++s) {
        double *r = pa.pred + static_cast<size_t>(s) * np1 * nx + ix;
        double c1 = r[0];
        r[0] = 0.0;
        for (int m = 0; m < nm; ++m) {
          double *ra = r + static_cast<size_t>(1 + m) * nx, *rb = r + static_cast<size_t>(1 + nm + m) * nx;
          c1 = c1 + sMode[m] * *ra;
          c1 = c1 + sMode[nm + m] * *rb;
          *ra = 0.0;
          *rb = 0.0;
        }
        c2 = c2 + c1 * f.Z[s];
      }
    }
    const double cd = chargeden_from(f, c2);
    pa.cd_h[ix] = cd;
    sCD[ix] = cd;
  }
  __syncthreads();
  FieldArgs g = f;
  g.E = pa.E_h;
  g.mode_re = pa.mode_h;
  g.mode_im = pa.mode_h + nm;
  g.history = nullptr;
  solve_body<true>(g, sCD, sMode, sScr, sTab);
}

// k_field_solve_pair for the usual case -- ONE kept mode, tables that fit the LDS -- with everything that does
// not wait for the serial sums moved in front of them.  The launch is latency-bound (one workgroup; at 1e7
// markers per GPU it is 7-10 % of the time step), and k_field_solve_pair spends it in a row of dependent
// round trips: charge, tables, [chain], prediction tiles, two workgroup reductions, inverse.  Here every
// thread issues all its loads at once (charge, the three prediction slices, both tables), forms the chain's
// products AND the prediction's forward sums before the chain runs -- the predicted charge density is
// cd_h = g0 + re ga + im gb with g0 = chargeden(R0), ga = RA nx/lx, gb = RB nx/lx, so its projections are
// S0 + re Sa + im Sb with six sums that need no mode: wave reductions, no barrier -- and after the chain one
// thread combines them; both inverse transforms then run in one loop.  The field of the new state: the same
// products in the same order as k_field_solve (bit for bit).  Eh: regrouped sums, as before (TREE).
template <int SRC>
__global__ void __launch_bounds__(FIELD_THREADS)
k_field_solve_pair1(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nx = f.nx;
  const int ne = (nx + 1) & ~1;
  double *sPc = reinterpret_cast<double *>(smem);  // [ne] fre * chargeden   (16-byte aligned rows for the chain)
  double *sPs = sPc + ne;                           // [ne] fim * chargeden
  double *sW = sPs + ne;                            // [FIELD_THREADS / 64][6] wave partials of the six sums
  double *sMode = sW + (FIELD_THREADS / 64) * 6;    // re, im, then the six sums of the workgroup
  double *sScr = sMode + 8;                         // [16]
  double *sPart = sScr + 16;                        // [2 npe] partial chains of the npe-rank order
  double *sV = sPart + ((2 * f.npe + 1) & ~1);      // SRC 1: [charge2 | R0 | RA | RB] of this rank, then of all
  const double *pk = pa.pack;                       // SRC 2: the same, all-reduced in memory
  const size_t np1 = 3;
  if constexpr (SRC == 1) {
    for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
      sV[ix] = charge_local_one(f, ix);
      for (size_t k = 0; k < np1; ++k) {
        double c2 = 0.0;
        for (int sp = 0; sp < f.nspecies; ++sp) {
          double *r = pa.pred + (static_cast<size_t>(sp) * np1 + k) * nx + ix;
          c2 = c2 + *r * f.Z[sp];
          *r = 0.0;
        }
        sV[(1 + k) * nx + ix] = c2;
      }
    }
    __syncthreads();
    exchange_vectors(x1, sV, 4 * nx);
    __syncthreads();
    pk = sV;
  }
  double off = 0.0;
  if (!f.deltaf)
    for (int sp = 0; sp < f.nspecies; ++sp) off = off + f.Z[sp] * f.n0[sp];
  double s0c = 0.0, sac = 0.0, sbc = 0.0, s0s = 0.0, sas = 0.0, sbs = 0.0;
  constexpr int U = 4;
  const double ginv = f.grad_inv[0];  // off the critical path behind the chain
  double tr[U], ti[U];                // the tables of the last trip stay in registers for the inverse
  for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
    double c[U], r0[U], ra[U], rb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // all loads of the trip in flight together
      const int ix = base + u * FIELD_THREADS;
      c[u] = r0[u] = ra[u] = rb[u] = tr[u] = ti[u] = 0.0;
      if (ix < nx) {
        tr[u] = f.fre[ix];
        ti[u] = f.fim[ix];
        if constexpr (SRC != 0) {
          c[u] = pk[ix];
          r0[u] = pk[nx + ix];
          ra[u] = pk[2 * static_cast<size_t>(nx) + ix];
          rb[u] = pk[3 * static_cast<size_t>(nx) + ix];
        } else {
          for (int sp = 0; sp < f.nspecies; ++sp) {  // src/pic1dp_interaction.F90:126-127
            double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            double c1 = *r;
            for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
            c[u] = c[u] + c1 * f.Z[sp];
            const double *q = pa.pred + static_cast<size_t>(sp) * np1 * nx + ix;
            r0[u] = r0[u] + q[0] * f.Z[sp];
            ra[u] = ra[u] + q[nx] * f.Z[sp];
            rb[u] = rb[u] + q[2 * static_cast<size_t>(nx)] * f.Z[sp];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      if (ix < nx) {
        if constexpr (SRC == 0) {  // accumulators consumed: zero for the next kernels
          for (int sp = 0; sp < f.nspecies; ++sp) {
            double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            for (int g = 0; g < f.rho_copies; ++g) r[static_cast<size_t>(g) * f.rho_stride] = 0.0;
            double *q = pa.pred + static_cast<size_t>(sp) * np1 * nx + ix;
            q[0] = 0.0;
            q[nx] = 0.0;
            q[2 * static_cast<size_t>(nx)] = 0.0;
          }
        }
        f.charge[ix] = c[u];
        const double cd = chargeden_from(f, c[u]);  // :138-148
        f.chargeden[ix] = cd;
        sPc[ix] = tr[u] * cd;
        sPs[ix] = ti[u] * cd;
        const double g0 = r0[u] * f.dnx / f.lx - off, ga = ra[u] * f.dnx / f.lx, gb = rb[u] * f.dnx / f.lx;
        s0c += tr[u] * g0;
        sac += tr[u] * ga;
        sbc += tr[u] * gb;
        s0s += ti[u] * g0;
        sas += ti[u] * ga;
        sbs += ti[u] * gb;
      }
    }
  }
  {
    double v[6] = {s0c, sac, sbc, s0s, sas, sbs};
#pragma unroll
    for (int k = 0; k < 6; ++k)
      for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    if ((threadIdx.x & 63) == 0)
      for (int k = 0; k < 6; ++k) sW[(threadIdx.x >> 6) * 6 + k] = v[k];
  }
  __syncthreads();
  // the forward sums (:231-240): thread 0 the cos table -> im, thread 1 the -sin table -> re; beside them the last
  // wave adds up the wave partials of the six sums
  {
    const double acc = lean_forward_sums(f, sPc, sPs, sPart, [&]() {
      const int k = static_cast<int>(threadIdx.x) - (FIELD_THREADS - 64);
      if (k >= 0 && k < 6) {
        double t = 0.0;
        for (int w = 0; w < FIELD_THREADS / 64; ++w) t += sW[w * 6 + k];
        sMode[2 + k] = t;
      }
    });
    if (threadIdx.x == 0) {
      const double im = acc * f.sc_im * ginv;
      sMode[1] = im;
      f.mode_im[0] = im;
    } else if (threadIdx.x == 1) {
      const double re = acc * f.sc_re * ginv;
      sMode[0] = re;
      f.mode_re[0] = re;
    }
  }
  __syncthreads();
  // the kept mode of the next step's half-step field (every thread for itself), both inverse transforms (:251-257)
  double e2 = 0.0;
  const double re = sMode[0], im = sMode[1];
  const double ac = sMode[2] + re * sMode[3] + im * sMode[4], as = sMode[5] + re * sMode[6] + im * sMode[7];
  const double im_h = ac * f.sc_im * ginv, re_h = as * f.sc_re * ginv;
  if (threadIdx.x == 0) {
    pa.mode_h[0] = re_h;
    pa.mode_h[1] = im_h;
  }
  const bool one_trip = nx <= U * FIELD_THREADS;
  for (int base = threadIdx.x; base < nx; base += U * FIELD_THREADS) {
    if (!one_trip) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * FIELD_THREADS;
        tr[u] = ix < nx ? f.fre[ix] : 0.0;
        ti[u] = ix < nx ? f.fim[ix] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * FIELD_THREADS;
      if (ix < nx) {
        double a = 0.0;
        a = a + tr[u] * re;
        a = a + ti[u] * im;
        const double e = a * 2.0;
        f.E[ix] = e;
        e2 += e * e;
        double b = 0.0;
        b = b + tr[u] * re_h;
        b = b + ti[u] * im_h;
        pa.E_h[ix] = b * 2.0;
      }
    }
  }
  if (f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

// The pair solve for the six-sum prediction (k_step_one<PRIV>, k_step_sums; one kept mode): the field of the new state
// from its deposited charge in the reference's order, then -- with the kept mode just found -- the half-step field of the
// NEXT step from the six sums: no second forward transform, the sums ARE the projections (pred_forward_sums), only the
// inverse (:251-257).  SRC: 0 one rank (charge from the local accumulators, sums from pa.pred), 1 one-hop exchange
// (charge2 and the six sums travel together, one exchange of nx + 8 doubles), 2 packed (pa.pack holds both, already
// all-reduced).  Laid out like k_field_solve_pair1 and launched with as many threads as the grid has cells (up to 1024: at
// nx = 4096 every thread owns four cells and has all its loads in flight at once -- with 256 threads each of the kernel's
// loops is four dependent round trips): charge -> products, [chain | the six sums fetched beside it], both inverse
// transforms in one loop.
template <int SRC>
__global__ void __launch_bounds__(1024)
k_field_solve_pair_sums1(const FieldArgs f, const XchgArgs x1, const PairArgs pa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nx = f.nx;
  const int ne = (nx + 1) & ~1;
  double *sPc = reinterpret_cast<double *>(smem);  // [ne] fre * chargeden
  double *sPs = sPc + ne;                           // [ne] fim * chargeden
  double *sMode = sPs + ne;                         // re, im, then the six sums
  double *sScr = sMode + 8;                         // [16]
  double *sPart = sScr + 16;                        // [2 npe] partial chains of the npe-rank order
  double *sV = sPart + ((2 * f.npe + 1) & ~1);      // SRC 1: [charge2 | six sums | pad]
  const double *pk = pa.pack;
  if constexpr (SRC == 1) {
    if (!pa.posted) {  // (posted: the marker launch's tail has formed and stored this rank's vector, StepTail mode 2)
      for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sV[ix] = charge_local_one(f, ix);
      if (threadIdx.x < 8) sV[nx + threadIdx.x] = pred_sum_take(pa.pred, threadIdx.x);
      __syncthreads();
    }
    exchange_vectors(x1, sV, nx + 8, pa.posted != 0);
    __syncthreads();
    pk = sV;
  }
  const double ginv = f.grad_inv[0];
  constexpr int U = 4;
  const int T = blockDim.x;
  double tr[U], ti[U];
  for (int base = threadIdx.x; base < nx; base += U * T) {
    double c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * T;
      c[u] = tr[u] = ti[u] = 0.0;
      if (ix < nx) {
        tr[u] = f.fre[ix];
        ti[u] = f.fim[ix];
        if constexpr (SRC != 0) {
          c[u] = pk[ix];
        } else {
          for (int sp = 0; sp < f.nspecies; ++sp) {  // src/pic1dp_interaction.F90:126-127
            const double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            double c1 = *r;
            for (int g = 1; g < f.rho_copies; ++g) c1 = c1 + r[static_cast<size_t>(g) * f.rho_stride];
            c[u] = c[u] + c1 * f.Z[sp];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * T;
      if (ix < nx) {
        if constexpr (SRC == 0)
          for (int sp = 0; sp < f.nspecies; ++sp) {
            double *r = f.rho_sp + static_cast<size_t>(sp) * nx + ix;
            for (int g = 0; g < f.rho_copies; ++g) r[static_cast<size_t>(g) * f.rho_stride] = 0.0;
          }
        f.charge[ix] = c[u];
        const double cd = chargeden_from(f, c[u]);  // :138-148
        f.chargeden[ix] = cd;
        sPc[ix] = tr[u] * cd;
        sPs[ix] = ti[u] * cd;
      }
    }
  }
  __syncthreads();
  {  // the forward sums (:231-240); beside them the last wave fetches the six sums (+ pad) of this step
    const double acc = lean_forward_sums(f, sPc, sPs, sPart, [&]() {
      const int k = static_cast<int>(threadIdx.x) - (static_cast<int>(blockDim.x) - 64);
      if (k >= 0 && k < 8) {
        if constexpr (SRC == 0) {
          const double t = pred_sum_take(pa.pred, k);
          if (k < 6) sMode[2 + k] = t;
        } else {
          if (k < 6) sMode[2 + k] = pk[nx + k];
        }
      }
    });
    if (threadIdx.x == 0) {
      const double im = acc * f.sc_im * ginv;
      sMode[1] = im;
      f.mode_im[0] = im;
    } else if (threadIdx.x == 1) {
      const double re = acc * f.sc_re * ginv;
      sMode[0] = re;
      f.mode_re[0] = re;
    }
  }
  __syncthreads();
  const double re = sMode[0], im = sMode[1];
  double ac, as;
  pred_forward_sums(f, pa.pt, sMode + 2, re, im, ac, as);
  const double im_h = ac * f.sc_im * ginv, re_h = as * f.sc_re * ginv;  // :234, :239, :243-247
  if (threadIdx.x == 0) {
    pa.mode_h[0] = re_h;
    pa.mode_h[1] = im_h;
  }
  // call sites: the kept mode's content of the half-step charge density, as k_pred_chargeden forms it -- what
  // field_chargeden has to hold once the host has called solve_field for the half step (adopt_half_field)
  double alpha = 0.0, beta = 0.0;
  if (pa.cd_h) {
    const double det = pa.pt.g11 * pa.pt.g22 - pa.pt.g12 * pa.pt.g12;
    alpha = (ac * pa.pt.g22 - as * pa.pt.g12) / det;
    beta = (as * pa.pt.g11 - ac * pa.pt.g12) / det;
  }
  double e2 = 0.0;
  const bool one_trip = nx <= U * T;
  for (int base = threadIdx.x; base < nx; base += U * T) {
    if (!one_trip) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ix = base + u * T;
        tr[u] = ix < nx ? f.fre[ix] : 0.0;
        ti[u] = ix < nx ? f.fim[ix] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ix = base + u * T;
      if (ix < nx) {  // both inverse transforms, :251-257
        double a = 0.0;
        a = a + tr[u] * re;
        a = a + ti[u] * im;
        const double e = a * 2.0;
        f.E[ix] = e;
        e2 += e * e;
        double b = 0.0;
        b = b + tr[u] * re_h;
        b = b + ti[u] * im_h;
        pa.E_h[ix] = b * 2.0;
        if (pa.cd_h) pa.cd_h[ix] = alpha * tr[u] + beta * ti[u];
      }
    }
  }
  if (f.history) {  // int E^2 dx, src/pic1dp_output.F90:120-124
    const double tot = block_sum(e2, sScr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(tot);
      *f.history = nrm * nrm * f.lx / f.dnx;
    }
  }
}

// Many kept modes (2*nmode > FIELD_THREADS, up to the full spectrum nmode = nx/2,
// SURVEY N4): the same arithmetic in the same order, spread over workgroups.
// The reference's operators are then O(nx^2) dense matrices exactly as here
// (doc/formulation.tex:288-290); one thread still owns one serial sum.
constexpr int WIDE_THREADS = 64;

// forward sums: chain t -> mode t>>1, (t&1 ? cos-table : -sin-table), ascending ix
__global__ void __launch_bounds__(WIDE_THREADS) k_field_modes_wide(const FieldArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sCD = reinterpret_cast<double *>(smem);  // [nx]
  const int nx = f.nx, nm = f.nmode;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) sCD[ix] = f.chargeden[ix];
  __syncthreads();
  const int chain = blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= 2 * nm) return;
  const int m = chain >> 1;
  const bool use_cos = chain & 1;
  const double *tab = (use_cos ? f.fre : f.fim) + static_cast<size_t>(m) * nx;
  double acc = 0.0;
  int ix = 0;
  if (f.npe > 1) {  // the order of an npe-rank run: row blocks from zero, added owner first (see rank_block)
    acc = ranks_sum_serial([tab, sCD](int i) { return tab[i] * sCD[i]; }, nx, f.npe, entry_owner(m, nm, f.npe));
    ix = nx;
  }
  for (; ix + 8 <= nx; ix += 8) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tab[ix + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = acc + t[k] * sCD[ix + k];
  }
  for (; ix < nx; ++ix) acc = acc + tab[ix] * sCD[ix];
  if (use_cos)
    f.mode_im[m] = acc * f.sc_im * f.grad_inv[m];
  else
    f.mode_re[m] = acc * f.sc_re * f.grad_inv[m];
}

// inverse: one grid point per thread, serial over the modes in the row's order (inverse_row, :251-256)
__global__ void __launch_bounds__(WIDE_THREADS) k_field_inverse_wide(const FieldArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sMode = reinterpret_cast<double *>(smem);  // [2*nmode]: re then im
  const int nx = f.nx, nm = f.nmode;
  for (int m = threadIdx.x; m < nm; m += blockDim.x) {
    sMode[m] = f.mode_re[m];
    sMode[nm + m] = f.mode_im[m];
  }
  __syncthreads();
  const int ix = blockIdx.x * blockDim.x + threadIdx.x;
  if (ix >= nx) return;
  f.E[ix] = inverse_row(f, ix, sMode);
}

__global__ void __launch_bounds__(FIELD_THREADS)
k_field_energy(const double *E, int nx, double lx, double dnx, double *out) {
  __shared__ double scr[16];
  double e2 = 0.0;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) e2 += E[ix] * E[ix];
  const double tot = block_sum(e2, scr);
  if (threadIdx.x == 0) {
    const double nrm = sqrt(tot);
    *out = nrm * nrm * lx / dnx;
  }
}

// ---------------------------------------------------------------------------
// Opt-in ALTERNATIVE field solve (NOT the reference's algorithm, which is the
// mode-filtered partial DFT above): second-order finite differences
// keeping every mode,
//     (phi[i-1] - 2 phi[i] + phi[i+1]) / h^2 = -(rho[i] - <rho>),
//     E[i] = -(phi[i+1] - phi[i-1]) / (2 h),      periodic, gauge phi[0] = 0.
// The nx-1 unknowns form a tridiagonal system, solved by parallel cyclic
// reduction held in LDS (ceil(log2(nx-1)) sweeps, every row eliminated against
// its neighbours at distance 1, 2, 4, ...).  One workgroup; nx <= 4096.
// ---------------------------------------------------------------------------
constexpr int FD_THREADS = 1024;
constexpr int FD_MAX_NX = 4096;
constexpr int FD_PER_THREAD = FD_MAX_NX / FD_THREADS;

__global__ void __launch_bounds__(FD_THREADS)
k_field_fd(const double *chargeden, double *E, double *history, int nx, double lx, double dnx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double scr[16];
  __shared__ double s_mean;
  double *A = reinterpret_cast<double *>(smem), *B = A + nx, *Cc = B + nx, *D = Cc + nx;
  const int n = nx - 1;
  const double h = lx / dnx;
  double part = 0.0;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) part += chargeden[ix];
  const double tot = block_sum(part, scr);
  if (threadIdx.x == 0) s_mean = tot / dnx;
  __syncthreads();
  const double mean = s_mean;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    A[j] = j == 0 ? 0.0 : -1.0;
    B[j] = 2.0;
    Cc[j] = j == n - 1 ? 0.0 : -1.0;
    D[j] = h * h * (chargeden[j + 1] - mean);
  }
  __syncthreads();
  for (int s = 1; s < n; s <<= 1) {
    double na[FD_PER_THREAD], nb[FD_PER_THREAD], nc[FD_PER_THREAD], nd[FD_PER_THREAD];
    int k = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x, ++k) {
      const int lo = j - s, hi = j + s;
      double a = A[j], b = B[j], c = Cc[j], d = D[j];
      double a2 = 0.0, c2 = 0.0;
      if (lo >= 0) {
        const double al = -a / B[lo];
        a2 = al * A[lo];
        b += al * Cc[lo];
        d += al * D[lo];
      }
      if (hi < n) {
        const double ga = -c / B[hi];
        c2 = ga * Cc[hi];
        b += ga * A[hi];
        d += ga * D[hi];
      }
      na[k] = a2;
      nb[k] = b;
      nc[k] = c2;
      nd[k] = d;
    }
    __syncthreads();
    k = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x, ++k) {
      A[j] = na[k];
      B[j] = nb[k];
      Cc[j] = nc[k];
      D[j] = nd[k];
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < n; j += blockDim.x) D[j] = D[j] / B[j];  // phi[j+1]
  __syncthreads();
  double e2 = 0.0;
  for (int i = threadIdx.x; i < nx; i += blockDim.x) {
    const int ip = i + 1 == nx ? 0 : i + 1, im = i == 0 ? nx - 1 : i - 1;
    const double pp = ip == 0 ? 0.0 : D[ip - 1], pm = im == 0 ? 0.0 : D[im - 1];
    const double e = -(pp - pm) / (2.0 * h);
    E[i] = e;
    e2 += e * e;
  }
  if (history) {
    const double t2 = block_sum(e2, scr);
    if (threadIdx.x == 0) {
      const double nrm = sqrt(t2);
      *history = nrm * nrm * lx / dnx;
    }
  }
}

}  // namespace

// the serial sums two ways (kernels.hpp launch_chain_selftest): one wave
__global__ void __launch_bounds__(64) k_chain_selftest(const double *v, int nrows, int n, double *out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *rows = reinterpret_cast<double *>(smem);
  const int stride = (n + 1) & ~1;
  for (int i = threadIdx.x; i < nrows * n; i += 64) rows[(i / n) * stride + i % n] = v[i];
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < nrows) out[threadIdx.x] = chain_sum_lds(rows + threadIdx.x * stride, n);
  const double d = chain_rows_mfma(rows, stride, nrows, n);
  for (int r = 0; r < nrows; ++r)
    if (static_cast<int>(threadIdx.x) == chain_mfma_lane(r)) out[16 + r] = d;
}
hipError_t launch_chain_selftest(const double *v, int nrows, int n, double *out, hipStream_t st) {
  const size_t lds = sizeof(double) * nrows * ((n + 1) & ~1);
  if (lds > 64 * 1024) {  // opt in to > 64 KiB of dynamic LDS, as launch_step_kernel does
    if (lds > 160 * 1024 - 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_selftest),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_chain_selftest, dim3(1), dim3(64), lds, st, v, nrows, n, out);
  return hipGetLastError();
}

hipError_t launch_field_fd(const double *chargeden, double *E, double *history, int nx, double lx,
                           double dnx, hipStream_t st) {
  if (nx < 3 || nx > FD_MAX_NX) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * 4 * static_cast<size_t>(nx);
  static bool big_lds_ok = false;
  if (lds > 64 * 1024 && !big_lds_ok) {
    // the kernel also holds 136 B of static LDS: leave room for it
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_fd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    big_lds_ok = true;
  }
  hipLaunchKernelGGL(k_field_fd, dim3(1), dim3(FD_THREADS), lds, st, chargeden, E, history, nx, lx, dnx);
  return hipGetLastError();
}

hipError_t launch_charge_local(const FieldArgs &f, hipStream_t st) {
  hipLaunchKernelGGL(k_charge_local, dim3(1), dim3(FIELD_THREADS), 0, st, f);
  return hipGetLastError();
}

hipError_t launch_chargeden(const FieldArgs &f, bool with_local, hipStream_t st) {
  if (with_local) {
    hipLaunchKernelGGL(k_chargeden<true>, dim3(1), dim3(FIELD_THREADS), 0, st, f);
  } else {
    hipLaunchKernelGGL(k_chargeden<false>, dim3(1), dim3(FIELD_THREADS), 0, st, f);
  }
  return hipGetLastError();
}

hipError_t launch_field_solve(const FieldArgs &f, bool with_local, bool from_chargeden,
                              hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) {  // many modes: chargeden, forward, inverse, energy
    if (!from_chargeden) {
      hipError_t e = launch_chargeden(f, with_local, st);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_field_modes_wide, dim3((2 * f.nmode + WIDE_THREADS - 1) / WIDE_THREADS),
                       dim3(WIDE_THREADS), sizeof(double) * f.nx, st, f);
    hipLaunchKernelGGL(k_field_inverse_wide, dim3((f.nx + WIDE_THREADS - 1) / WIDE_THREADS), dim3(WIDE_THREADS),
                       sizeof(double) * 2 * f.nmode, st, f);
    if (f.history) hipLaunchKernelGGL(k_field_energy, dim3(1), dim3(FIELD_THREADS), 0, st, f.E, f.nx, f.lx, f.dnx, f.history);
    return hipGetLastError();
  }
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  if (from_chargeden) {
    hipLaunchKernelGGL((k_field_solve<false, true>), dim3(1), dim3(FIELD_THREADS), lds, st, f);
  } else if (with_local) {
    hipLaunchKernelGGL((k_field_solve<true, false>), dim3(1), dim3(FIELD_THREADS), lds, st, f);
  } else {
    hipLaunchKernelGGL((k_field_solve<false, false>), dim3(1), dim3(FIELD_THREADS), lds, st, f);
  }
  return hipGetLastError();
}

hipError_t launch_pred_combine(const FieldArgs &f, double *pred, int nm_pred, hipStream_t st) {
  hipLaunchKernelGGL(k_pred_combine, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred, nm_pred);
  return hipGetLastError();
}

hipError_t launch_charge_exchange(const FieldArgs &f, const XchgArgs &x, hipStream_t st) {
  hipLaunchKernelGGL(k_charge_exchange, dim3(1), dim3(FIELD_THREADS), sizeof(double) * f.nx, st, f, x);
  return hipGetLastError();
}

hipError_t launch_field_solve_xchg(const FieldArgs &f, const XchgArgs &x, hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) {  // many modes: exchange, then the wide kernels from the summed charge
    hipError_t e = launch_charge_exchange(f, x, st);
    if (e != hipSuccess) return e;
    return launch_field_solve(f, false, false, st);
  }
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  hipLaunchKernelGGL(k_field_solve_xchg, dim3(1), dim3(FIELD_THREADS), lds, st, f, x);
  return hipGetLastError();
}

hipError_t launch_field_solve_pair(const FieldArgs &f, const PairArgs &pa, const XchgArgs *x1, hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) return hipErrorInvalidValue;
  size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                 (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  const XchgArgs none{};
  if (pa.kind != 2 && f.nmode == 1 && f.tab_lds) {  // the lean kernel of the usual case
    const size_t ne = (static_cast<size_t>(f.nx) + 1) & ~static_cast<size_t>(1);
    size_t l1 = sizeof(double) * (2 * ne + (FIELD_THREADS / 64) * 6 + 8 + 16 + ((2 * static_cast<size_t>(f.npe) + 1) & ~static_cast<size_t>(1)));
    if (x1) {
      l1 += sizeof(double) * 4 * static_cast<size_t>(f.nx);
      if (l1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair1<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair1<1>, dim3(1), dim3(FIELD_THREADS), l1, st, f, *x1, pa);
    } else if (pa.pack) {
      hipLaunchKernelGGL(k_field_solve_pair1<2>, dim3(1), dim3(FIELD_THREADS), l1, st, f, none, pa);
    } else {
      hipLaunchKernelGGL(k_field_solve_pair1<0>, dim3(1), dim3(FIELD_THREADS), l1, st, f, none, pa);
    }
    return hipGetLastError();
  }
  if (pa.kind == 2) {  // the six sums of ONE kept mode
    if (f.nmode != 1) return hipErrorInvalidValue;
    const size_t ne = (static_cast<size_t>(f.nx) + 1) & ~static_cast<size_t>(1);
    size_t l1 = sizeof(double) * (2 * ne + 8 + 16 + ((2 * static_cast<size_t>(f.npe) + 1) & ~static_cast<size_t>(1)));
    const int threads = f.nx > 2048 ? 1024 : (f.nx > 1024 ? 512 : FIELD_THREADS);
    if (x1) {
      l1 += sizeof(double) * pack_doubles(f.nx, 1, 2);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums1<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_field_solve_pair_sums1<1>, dim3(1), dim3(threads), l1, st, f, *x1, pa);
    } else if (pa.pack) {
      if (l1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums1<2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair_sums1<2>, dim3(1), dim3(threads), l1, st, f, none, pa);
    } else {
      if (l1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair_sums1<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_field_solve_pair_sums1<0>, dim3(1), dim3(threads), l1, st, f, none, pa);
    }
    return hipGetLastError();
  }
  if (x1) {
    lds += sizeof(double) * (2 + 2 * static_cast<size_t>(f.nmode)) * f.nx;  // the packed vector
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_solve_pair<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_field_solve_pair<1>, dim3(1), dim3(FIELD_THREADS), lds, st, f, *x1, pa);
  } else if (pa.pack) {
    hipLaunchKernelGGL(k_field_solve_pair<2>, dim3(1), dim3(FIELD_THREADS), lds, st, f, none, pa);
  } else {
    hipLaunchKernelGGL(k_field_solve_pair<0>, dim3(1), dim3(FIELD_THREADS), lds, st, f, none, pa);
  }
  return hipGetLastError();
}

hipError_t launch_field_solve_pred(const FieldArgs &f, double *pred, int nm_pred, hipStream_t st) {
  if (2 * f.nmode > FIELD_THREADS) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  hipLaunchKernelGGL(k_field_solve_pred, dim3(1), dim3(FIELD_THREADS), lds, st, f, pred, nm_pred);
  return hipGetLastError();
}

hipError_t launch_field_solve_pred_sums(const FieldArgs &f, const PredTab &pt, double *pred, hipStream_t st) {
  if (f.nmode != 1) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * (static_cast<size_t>(f.nx) + 2 * f.nmode + 16 +
                                       (f.tab_lds ? 2 * static_cast<size_t>(f.nmode) * f.nx : 0));
  hipLaunchKernelGGL(k_field_solve_pred_sums, dim3(1), dim3(FIELD_THREADS), lds, st, f, pt, pred);
  return hipGetLastError();
}

hipError_t launch_charge_pack(const FieldArgs &f, double *pred, int nm_pred, int kind, double *pack, hipStream_t st) {
  if (kind == 2)
    hipLaunchKernelGGL(k_charge_pack_sums, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred, pack);
  else
    hipLaunchKernelGGL(k_charge_pack, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred, nm_pred, pack);
  return hipGetLastError();
}

hipError_t launch_pred_chargeden(const FieldArgs &f, const PredTab &pt, double *pred, const double *K, hipStream_t st) {
  if (f.nmode != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pred_chargeden, dim3(1), dim3(FIELD_THREADS), 0, st, f, pt, pred, K);
  return hipGetLastError();
}

hipError_t launch_pred_to_charge(const FieldArgs &f, double *pred, hipStream_t st) {
  hipLaunchKernelGGL(k_pred_to_charge, dim3(1), dim3(FIELD_THREADS), 0, st, f, pred);
  return hipGetLastError();
}

hipError_t launch_field_energy(const double *E, int nx, double lx, double dnx, double *out,
                               hipStream_t st) {
  hipLaunchKernelGGL(k_field_energy, dim3(1), dim3(FIELD_THREADS), 0, st, E, nx, lx, dnx, out);
  return hipGetLastError();
}

}  // namespace pic1dp
