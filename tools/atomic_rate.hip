// atomic_rate.hip -- how fast can waves draw work from shared counters on MI355X?  (tools/, measurement only)
// Every wave's lane 0 performs `iters` DEPENDENT fetch-adds (the returned value feeds the next address computation)
// on one of `naddr` counters, 128 B apart; device (agent) scope, or -- counters chosen by the XCD the wave runs on
// -- workgroup scope (the atomic executes in the XCD's L2).  Prints latency (one wave) and throughput (all CUs).
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o tools/bin/atomic_rate && tools/bin/atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t err_ = (e);                                                            \
    if (err_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(err_), __LINE__); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

template <int SCOPE, bool BY_XCC>
__global__ void k_draw(unsigned long long *cnt, int naddr, int iters, unsigned long long *sink) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;
  int a = (BY_XCC ? xcc : wave) % naddr;
  unsigned long long acc = 0;
  if (lane == 0) {
    for (int i = 0; i < iters; ++i) {
      const unsigned long long r = __hip_atomic_fetch_add(cnt + 16 * a, 1ull, __ATOMIC_RELAXED, SCOPE);
      acc += r;
      if (!BY_XCC) a = (a + static_cast<int>(r & 0)) % naddr;  // dependence on the returned value
    }
    sink[wave] = acc;
  }
}

int main() {
  unsigned long long *cnt, *sink;
  const int blocks = 512, threads = 768, waves = blocks * threads / 64;
  CHECK(hipMalloc(&cnt, 128 * 4096));
  CHECK(hipMalloc(&sink, sizeof(unsigned long long) * waves));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](const char *what, auto kern, int nb, int nt, int naddr, int iters) -> int {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipMemset(cnt, 0, 128 * 4096));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(nb), dim3(nt), 0, 0, cnt, naddr, iters, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double n = static_cast<double>(nb) * nt / 64 * iters;
    std::printf("%-58s %8.1f us  %10.0f atomics  %8.2f ns/atomic overall  %8.2f ns per atomic and address\n", what,
                best * 1e3, n, best * 1e6 / n, best * 1e6 / n * naddr);
    return 0;
  };
  constexpr int DEV = __HIP_MEMORY_SCOPE_AGENT, WG = __HIP_MEMORY_SCOPE_WORKGROUP;
  run("latency: 1 wave, 1000 dependent, device scope", k_draw<DEV, false>, 1, 64, 1, 1000);
  run("latency: 1 wave, 1000 dependent, workgroup scope (L2)", k_draw<WG, true>, 1, 64, 8, 1000);
  for (int iters : {4, 16}) {
    std::printf("-- %d waves, %d draws each\n", waves, iters);
    run("device scope, 1 counter", k_draw<DEV, false>, blocks, threads, 1, iters);
    run("device scope, 8 counters (by wave)", k_draw<DEV, false>, blocks, threads, 8, iters);
    run("device scope, 8 counters (by XCD)", k_draw<DEV, true>, blocks, threads, 8, iters);
    run("device scope, 64 counters (by wave)", k_draw<DEV, false>, blocks, threads, 64, iters);
    run("device scope, 512 counters (by wave)", k_draw<DEV, false>, blocks, threads, 512, iters);
    run("workgroup scope = XCD's L2, 8 counters (by XCD)", k_draw<WG, true>, blocks, threads, 8, iters);
  }
  // are the L2-local counters exact?  every XCD's counter must hold the draws of the waves that ran on it
  CHECK(hipMemset(cnt, 0, 128 * 4096));
  hipLaunchKernelGGL((k_draw<WG, true>), dim3(blocks), dim3(threads), 0, 0, cnt, 8, 16, sink);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(16 * 8);
  CHECK(hipMemcpy(h.data(), cnt, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  unsigned long long tot = 0;
  std::printf("L2-local counters after %d draws:", waves * 16);
  for (int x = 0; x < 8; ++x) {
    std::printf(" %llu", h[16 * x]);
    tot += h[16 * x];
  }
  std::printf("  sum %llu (%s)\n", tot, tot == static_cast<unsigned long long>(waves) * 16 ? "exact" : "LOST UPDATES");
  return 0;
}
