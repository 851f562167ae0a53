// check_values.hpp -- generated operands of the exact-division checks (host and device draw the same values)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace pic1dp {
namespace {

// test positions for div_lx: uniform over three periods, cell boundaries and
// their neighbours (where a wrong last bit would change the cell index), raw
// bit patterns over a wide exponent range, small and large magnitudes
__host__ __device__ inline double div_check_value(uint64_t seed, int64_t i, double lx, int nx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * static_cast<uint64_t>(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  const double u = static_cast<double>(z >> 11) * 0x1p-53;  // [0,1)
  const int kind = static_cast<int>(z & 7);
  if (kind <= 2) return lx * (u * 3.0 - 1.0);
  if (kind == 3 || kind == 4) {
    const int k = static_cast<int>((z >> 3) % static_cast<uint64_t>(nx + 1));
    double x = lx * static_cast<double>(k) / static_cast<double>(nx);
    const int steps = static_cast<int>((z >> 40) & 7) - 3;  // -3..4 ulps around the boundary
    union { double d; int64_t b; } c;
    c.d = x;
    if (x != 0.0) c.b += steps;
    return c.d;
  }
  if (kind == 5) {
    union { double d; uint64_t b; } c;
    const uint64_t e = 1023 - 400 + (z >> 12) % 800;
    c.b = (z & 0x800FFFFFFFFFFFFFULL) | (e << 52);
    return c.d;
  }
  if (kind == 6) return lx * u * 0x1p-30;
  return lx * (u - 0.5) * 1e6;
}

// a uniform deviate in [0, 1) per (seed, index) (splitmix64)
__host__ __device__ inline double check_uniform(uint64_t seed, int64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * static_cast<uint64_t>(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return static_cast<double>(z >> 11) * 0x1p-53;
}

// dividends for the div_const check: random sign, exponent in [-300, 300],
// random significand -- every 16th one from the edges (0...0k, 1...1k) where
// rounding decisions are closest
__host__ __device__ inline double divc_check_value(uint64_t seed, int64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(i + 1);  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  uint64_t mant = z & 0xFFFFFFFFFFFFFull;
  if ((i & 15) == 0) mant = (z & 0x3FF) | ((z >> 10) & 1 ? 0xFFFFFFFFFFC00ull : 0ull);
  const uint64_t expo = 1023 - 300 + (z >> 52) % 601;
  const uint64_t bits = (z & 0x8000000000000000ull) | (expo << 52) | mant;
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double(static_cast<long long>(bits));
#else
  double d;
  std::memcpy(&d, &bits, 8);
  return d;
#endif
}

}  // namespace
}  // namespace pic1dp
