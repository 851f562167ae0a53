// multirand.hpp -- host-side random number generator of the native particle
// loader.  Reproduces the streams of the reference's multirand module
// (src/multirand.F90) so that a run started here begins from the very markers
// the reference would load: KISS64 (:921-945), MT19937-64 (:952-997),
// SuperKISS64 (:1004-1039), the [0,1] real conversion (:49), the constant /
// clock / urandom seeding (:244-351), warm-up (:373-381), the known-answer
// self-test (:390-553) and the polar Gaussian (:838-872).
#pragma once
#include <cstdint>
#include <vector>

namespace pic1dp {

class Multirand {
 public:
  enum Engine { KISS64 = 1, MT19937_64 = 2, SUPERKISS64 = 3 };
  enum Status { OK = 0, SELFTEST_FAILED = 1, WOULD_HANG = 2, IO_ERROR = 3 };

  Multirand();

  // multirand_init(al_int, seed_type, mype, warmup, selftest)
  Status init(int al_int, int seed_type, int mype, int warmup, bool selftest);
  // multirand_selftest: true when every known answer is reproduced
  bool selftest(int al_int);

  uint64_t next();                         // multirand_int64 (as unsigned bits)
  double real() { return to_real(next()); }  // multirand_real64
  void fill_real(double *a, int64_t n);    // multirand_real_array64
  void fill_gaussian(double *a, int64_t n);  // multirand_gaussian_array64

  static double to_real(uint64_t bits) {
    // INT2REAL64: signed value / (2**64-1 rounded to double) + 0.5
    return static_cast<double>(static_cast<int64_t>(bits)) / 18446744073709551615.0 + 0.5;
  }

 private:
  static constexpr int kStateWords = 20635;
  static constexpr int kLag = 20632;  // SuperKISS lag table; carry, cng, xs follow
  void default_seeds(int al_int);
  uint64_t kiss();
  uint64_t mt();
  uint64_t superkiss();
  void refill_mt();
  void refill_superkiss();

  std::vector<uint64_t> q_;
  int pos_ = 0;
  int engine_ = SUPERKISS64;
  bool gauss_held_ = false;
  double gauss_val_ = 0.0;
};

}  // namespace pic1dp
