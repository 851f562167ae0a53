// lds_atomic_rate.hip -- what does an LDS atomic cost on MI355X?  (tools/, measurement only; round 5: the histograms of
// output_ptcldist are twelve FP64 LDS atomics at random bins per marker, and they bound every form of an output step)
// One workgroup of 1024 threads per CU, a 64 KiB LDS table (a power of two near the histogram copy's 96 KiB); every thread performs
// `iters` x 12 atomics on it.  Operand types: f64 add (ds_add_f64), u64 add (ds_add_u64), f32 add, u32 add.
// Address patterns: "stride" lane l -> word (base + l) (conflict-free, consecutive), "random" a per-lane LCG over the
// whole table (what the histograms see), "random16" random, but the 16 lanes of a group on 16 distinct bank pairs
// (random row, column = lane: what a conflict-free scatter would cost).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_rate.hip -o tools/bin/lds_atomic_rate && tools/bin/lds_atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t err_ = (e);                                                            \
    if (err_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(err_), __LINE__); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

template <class T>
__device__ __forceinline__ void lds_add(T *p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// PATTERN 0 stride, 1 random, 2 random16
template <class T, int PATTERN>
__global__ void __launch_bounds__(1024) k_rate(int iters, int nwords, T *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *tab = reinterpret_cast<T *>(smem);
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) tab[i] = T(0);
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const int lane16 = threadIdx.x & 15;
  const T one = T(1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      s = s * 1664525u + 1013904223u;
      int a;
      if (PATTERN == 0)
        a = (threadIdx.x + (it * 12 + k) * 1024) & (nwords - 1);
      else if (PATTERN == 1)
        a = static_cast<int>((s >> 8) & static_cast<unsigned>(nwords - 1));
      else
        a = static_cast<int>(((s >> 8) & static_cast<unsigned>(nwords / 16 - 1)) * 16 + lane16);
      lds_add(tab + a, one);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = tab[0] + tab[nwords - 1];
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  void *sink;
  CHECK(hipMalloc(&sink, 8 * 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 400;
  const size_t lds = 64 * 1024;
  // (a small helper to name the sink's type per kernel)
#define RUN(T, P, name)                                                                                              \
  do {                                                                                                               \
    auto kern = k_rate<T, P>;                                                                                        \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    float best = 1e30f;                                                                                              \
    for (int rep = 0; rep < 4; ++rep) {                                                                              \
      CHECK(hipDeviceSynchronize());                                                                                 \
      CHECK(hipEventRecord(e0));                                                                                     \
      hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), lds, 0, iters, static_cast<int>(lds / sizeof(T)),              \
                         reinterpret_cast<T *>(sink));                                                               \
      CHECK(hipEventRecord(e1));                                                                                     \
      CHECK(hipEventSynchronize(e1));                                                                                \
      float ms;                                                                                                      \
      CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                       \
      if (ms < best) best = ms;                                                                                      \
    }                                                                                                                \
    const double wi = 16.0 * iters * 12;                                                                             \
    std::printf("%-44s %8.1f us  %7.2f ns per wave-instruction and CU  = %5.2f lanes per ns and CU\n", name, best * 1e3, \
                best * 1e6 / wi, 64.0 / (best * 1e6 / wi));                                                          \
  } while (0)
  std::printf("%d CUs, one workgroup of 1024 threads each, %d x 12 atomics per thread on a 64 KiB LDS table\n", cus, iters);
  RUN(double, 0, "f64 add, consecutive lanes (conflict-free)");
  RUN(double, 2, "f64 add, random rows, lane = bank pair");
  RUN(double, 1, "f64 add, random words");
  RUN(unsigned long long, 0, "u64 add, consecutive lanes");
  RUN(unsigned long long, 2, "u64 add, random rows, lane = bank pair");
  RUN(unsigned long long, 1, "u64 add, random words");
  RUN(float, 0, "f32 add, consecutive lanes");
  RUN(float, 1, "f32 add, random words");
  RUN(unsigned, 0, "u32 add, consecutive lanes");
  RUN(unsigned, 1, "u32 add, random words");
  return 0;
}
