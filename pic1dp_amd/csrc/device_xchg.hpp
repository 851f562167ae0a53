// device_xchg.hpp -- the two halves of the one-hop charge exchange (kernels_field.hip has the protocol) and the
// TAIL of a one-pass marker launch on several ranks (kernels.hpp StepTail).  Shared by kernels_field.hip (the whole
// exchange, or its second half when the marker launch has posted) and kernels_step.hip (the tail).
#pragma once
#include "device_math.hpp"

namespace pic1dp {
namespace {

#define PIC1DP_SYS __HIP_MEMORY_SCOPE_SYSTEM

// First half: this rank's n values (sV; every thread has filled the elements threadIdx.x + k * blockDim.x, or a
// workgroup barrier lies between the filling and this call) stored into slot [rank] of EVERY rank's area with
// system-scope stores (one xGMI hop for the peers), the stores drained, the workgroup met, then one lane per
// destination stores the flag = epoch.
__device__ __forceinline__ void exchange_post(const XchgArgs &x, const double *sV, int n) {
  const int nr = x.nranks, par = static_cast<int>(x.epoch & 1);
  for (int k = 0; k < nr; ++k) {
    int q = x.rank + k;  // start with the own area, then the peers in ring order
    if (q >= nr) q -= nr;
    double *dst = x.slots[q] + (static_cast<size_t>(par) * nr + x.rank) * x.vstride;
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(dst + i, sV[i], __ATOMIC_RELAXED, PIC1DP_SYS);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this wave's stores have landed
  __syncthreads();
  if (threadIdx.x < nr)
    __hip_atomic_store(x.flags[threadIdx.x] + par * XCHG_MAX_RANKS + x.rank, x.epoch, __ATOMIC_RELEASE, PIC1DP_SYS);
}

// Second half: lane q polls flag q of the OWN area until it reads the epoch (bounded by a wall-clock limit: on expiry
// the host-visible error word is set and the kernel goes on, so the grid always drains); then sV[i] = the slots'
// values added in rank order -- the same additions in the same order on every GPU.
__device__ __forceinline__ void exchange_wait_sum(const XchgArgs &x, double *sV, int n) {
  const int nr = x.nranks, par = static_cast<int>(x.epoch & 1);
  if (threadIdx.x < nr) {
    const int q = threadIdx.x;
    const unsigned long long *fl = x.flags[x.rank] + par * XCHG_MAX_RANKS + q;
    const long long t0 = wall_clock64();
    // a run that already timed out once does not wait again: its remaining launches drain at once
    const long long limit = __hip_atomic_load(x.err, __ATOMIC_RELAXED, PIC1DP_SYS) ? 0 : x.timeout_ticks;
    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, PIC1DP_SYS) < x.epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > limit) {  // give up: report, never hang
        __hip_atomic_store(x.err, (x.epoch << 8) | static_cast<unsigned long long>(q + 1), __ATOMIC_RELAXED, PIC1DP_SYS);
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  const double *mine = x.slots[x.rank] + static_cast<size_t>(par) * nr * x.vstride;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double t[XCHG_MAX_RANKS];
#pragma unroll
    for (int q = 0; q < XCHG_MAX_RANKS; ++q)
      t[q] = q < nr ? __hip_atomic_load(mine + static_cast<size_t>(q) * x.vstride + i, __ATOMIC_RELAXED, PIC1DP_SYS) : 0.0;
    double sum = t[0];
#pragma unroll
    for (int q = 1; q < XCHG_MAX_RANKS; ++q)
      if (q < nr) sum = sum + t[q];
    sV[i] = sum;
  }
}

// ---------------------------------------------------------------------------
// TAIL of a one-pass marker launch on several ranks (kernels.hpp StepTail; k_step_one<PRIV>, k_step_sums).
// What used to be a launch of its own between the marker kernel and the sum over ranks -- k_charge_pack_sums: this
// rank's charge2 = sum_s Z_s rho_s (src/pic1dp_interaction.F90:126-127) and its six sums, accumulators re-zeroed --
// is done by the LAST workgroup of the marker launch to finish:
//   every wave waits for its own atomics (the flush of the rho tile, the six sums: agent-scope, performed at the memory
//   side), the workgroup meets, one lane draws a ticket (agent-scope returning add); the workgroup that draws the last
//   ticket knows every other workgroup's atomics have been performed and reads the accumulators with agent-scope
//   (sc1) loads -- the "agent atomics on both sides, signalled by the last arriver" hand-off of the CDNA4 notes.
// mode 1 (RCCL): the packed vector [charge2 | six sums | pad] goes to t.pack, the all-reduce follows on the stream.
// mode 2 (one-hop exchange): it goes straight into every rank's exchange slots (exchange_post); the field launch
//   that follows only waits for the peers' flags, adds in rank order and solves (exchange_wait_sum) -- the stores'
//   flight over xGMI overlaps the launch boundary.
// The additions are charge_local_one's / pred_sum_take's, in their order: the same bits as the separate launch gives
// for the same accumulators.  sV: LDS scratch of nx + 8 doubles (the field tiles, which nobody reads any more).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void step_tail(const StepTail &t, double *sV) {
  // (no static LDS of its own: the marker kernels' dynamic tiles may take all but the exp table's 1 KiB of the CU's 160)
  int *s_last = reinterpret_cast<int *>(sV);
  // Ordering.  What is handed over are agent-scope atomics ON BOTH SIDES (relaxed RMWs by the producers, agent-scope loads
  // by the last arriver) signalled through one unsharded counter whose returned value names the last arriver -- one of the
  // forms the CDNA4 notes list as valid on gfx950 ("8-B agent atomics both sides", MI355X_MICROARCH.md, Valid forms):
  // a no-return atomic is performed at the memory side, beyond the per-XCD L2s, once the issuing wave's vmcnt has counted
  // it, so "every wave waits for vmcnt(0), the workgroup meets, one lane adds to the counter" orders them before the
  // ticket.  That is an ISA-level argument, not one the HIP memory model makes: the model's form is a RELEASE on the
  // ticket add.  Built and measured in round 6 (__ATOMIC_ACQ_REL on the add): the compiler's agent-scope
  // release is buffer_wbl2 sc1 -- a write-back of the XCD L2's dirty lines, i.e. of the marker stores this very kernel
  // has just made -- once per workgroup: the 8-way share's marker launch 123.5 -> 142.2 us, the step 0.1293 -> 0.1475 ms
  // (profiles/r06/experiments/ab_tail_order.log), 7.2x -> 6.4x of the strong-scaling budget for an ordering the atomics
  // do not need (nothing handed over here is a plain store).  Hence relaxed + the explicit wait; tools/tail_soak.py and
  // the bit-identity tests of the tail (tests/test_gpu_exchange.py, test_gpu_one_pass.py) are the evidence it holds.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics have been performed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = prev + 1u == gridDim.x ? 1 : 0;
  }
  __syncthreads();  // (the ticket's lane has its answer: the barrier is behind the returned add)
  const int last = *s_last;
  if (!last) return;
  __syncthreads();  // every thread has read the word before the scratch is written over
  const int nx = t.nx;
  for (int ix = threadIdx.x; ix < nx; ix += blockDim.x) {
    double c2 = 0.0;
    for (int s = 0; s < t.nspecies; ++s) {  // charge_local_one's additions, in its order
      double *r = t.rho_sp + static_cast<size_t>(s) * nx + ix;
      double c1 = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(r, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int g = 1; g < t.rho_copies; ++g) {
        double *rg = r + static_cast<size_t>(g) * t.rho_stride;
        c1 = c1 + __hip_atomic_load(rg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rg, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      c2 = c2 + c1 * t.Z[s];
    }
    if (t.mode == 1)
      t.pack[ix] = c2;
    else
      sV[ix] = c2;
  }
  if (threadIdx.x < 8) {  // pred_sum_take's additions: the copies in copy order
    const int k = threadIdx.x;
    double v[PRED_SUM_COPIES];
#pragma unroll
    for (int c = 0; c < PRED_SUM_COPIES; ++c) v[c] = __hip_atomic_load(t.sums + c * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int c = 0; c < PRED_SUM_COPIES; ++c) __hip_atomic_store(t.sums + c * 8 + k, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double acc = v[0];
#pragma unroll
    for (int c = 1; c < PRED_SUM_COPIES; ++c) acc = acc + v[c];
    if (t.mode == 1)
      t.pack[nx + k] = acc;
    else
      sV[nx + k] = acc;
  }
  if (threadIdx.x == 0) __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
  if (t.mode == 2) {
    __syncthreads();  // element i of the packed vector belongs to thread i % blockDim in the exchange
    exchange_post(t.x, sV, nx + 8);
  }
}

}  // namespace
}  // namespace pic1dp
