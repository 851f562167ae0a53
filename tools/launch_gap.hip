// launch_gap.hip -- what a DEPENDENT kernel launch costs on the GPU's side (tools/, measurement only): a chain of
// tiny kernels in one stream (each waits for the one before: the in-order queue's barrier), the same chain captured
// in a HIP graph, and a chain that alternates a wide kernel (all CUs, some streaming) with a one-workgroup kernel --
// the shape of a time step (marker kernel, field kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/launch_gap.hip -o tools/bin/launch_gap && tools/bin/launch_gap
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t err_ = (e);                                                            \
    if (err_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(err_), __LINE__); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

__global__ void k_tiny(double *p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0;
}
__global__ void k_wide(double2 *a, long n) {  // read + write n double2
  for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * blockDim.x) {
    double2 t = a[i];
    t.x += 1.0;
    a[i] = t;
  }
}

int main() {
  double *p;
  double2 *a;
  const long n = 4 << 20;  // 64 MiB
  CHECK(hipMalloc(&p, 4096));
  CHECK(hipMalloc(&a, sizeof(double2) * n));
  CHECK(hipMemset(p, 0, 4096));
  CHECK(hipMemset(a, 0, sizeof(double2) * n));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int N = 2000;
  auto wall = [&](auto fn) -> double {
    fn();  // warm
    (void)hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    fn();
    (void)hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  double us = wall([&]() {
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
    return 0;
  });
  std::printf("stream, %d tiny dependent kernels          : %.2f us per launch\n", N, us / N);
  // the same chain as a graph
  hipGraph_t g;
  hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
  CHECK(hipStreamEndCapture(st, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  us = wall([&]() { return hipGraphLaunch(ge, st); });
  std::printf("graph,  %d tiny dependent kernels          : %.2f us per launch\n", N, us / N);
  // wide + tiny alternating
  const int M = 500;
  double us_wide = wall([&]() {
    for (int i = 0; i < M; ++i) hipLaunchKernelGGL(k_wide, dim3(512), dim3(768), 0, st, a, n);
    return 0;
  });
  double us_pair = wall([&]() {
    for (int i = 0; i < M; ++i) {
      hipLaunchKernelGGL(k_wide, dim3(512), dim3(768), 0, st, a, n);
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
    }
    return 0;
  });
  std::printf("stream, wide kernel alone                     : %.2f us per launch\n", us_wide / M);
  std::printf("stream, wide + tiny                           : %.2f us per pair  (the tiny one adds %.2f us)\n", us_pair / M,
              (us_pair - us_wide) / M);
  hipGraph_t g2;
  hipGraphExec_t ge2;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < M; ++i) {
    hipLaunchKernelGGL(k_wide, dim3(512), dim3(768), 0, st, a, n);
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
  }
  CHECK(hipStreamEndCapture(st, &g2));
  CHECK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  us = wall([&]() { return hipGraphLaunch(ge2, st); });
  std::printf("graph,  wide + tiny                           : %.2f us per pair\n", us / M);
  return 0;
}
