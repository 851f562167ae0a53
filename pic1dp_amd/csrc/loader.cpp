// loader.cpp -- see loader.hpp.  Host code; compiled without FMA contraction so
// that the weights equal the reference's plain-double arithmetic.
#include "loader.hpp"

#include <algorithm>
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1
#endif
#include <cmath>
#include <math.h>
#include <functional>
#include <thread>
#include <vector>

namespace pic1dp {

namespace {
constexpr double kPi = 3.14159265358979323846264;  // PETSC_PI

// run fn(lo, hi) over [0, n) split across threads
void parallel_ranges(int64_t n, int nthreads, const std::function<void(int64_t, int64_t)> &fn) {
  nthreads = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(nthreads, n / 65536)));
  if (nthreads == 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t chunk = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    const int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
    if (lo < hi) pool.emplace_back(fn, lo, hi);
  }
  for (auto &th : pool) th.join();
}
}  // namespace

int64_t block_alloc(int64_t nglobal, int rank, int size) {
  return nglobal / size + (nglobal % size > rank ? 1 : 0);
}

int64_t block_np(const pic1dp_input &in, int isp, int mype, int npe) {
  const int64_t spare = in.nparticle_max - in.species_nparticle_init[isp];
  int64_t unload = spare / npe;
  if (mype == 0) unload += spare % npe;
  return block_alloc(in.nparticle_max, mype, npe) - unload;
}

void load_block_species(const pic1dp_input &in, int isp, Multirand &g, int64_t n, double *x,
                        double *v, double *p, double *w, int nthreads) {
  const double T = in.species_temperature[isp], T2 = in.species_temperature2[isp];
  const double m = in.species_mass[isp], den = in.species_density[isp];
  const double v0 = in.species_v0[isp];
  const double ninit = static_cast<double>(in.species_nparticle_init[isp]);
  const double lx = in.lx, vmax = in.v_max;

  // ---- velocities and equilibrium weights p = f0/g  (:172-219)
  if (in.imarker == 1) {
    g.fill_gaussian(v, n);
    const double sigma = std::sqrt(T / m);
    const double pconst = den * lx / ninit;
    parallel_ranges(n, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        v[i] = v[i] * sigma + v0;
        p[i] = pconst;
      }
    });
  } else {
    g.fill_real(v, n);
    const int dist = in.iptcldist;
    // marker density prefactor n*lx*2*vmax/N (bump-on-tail carries its own
    // densities inside the bracket, :198)
    const double pref = (dist == 3 ? 1.0 : den) * lx * 2.0 * vmax / ninit;
    const double a1 = 2.0 * T / m, a2 = 2.0 * T2 / m;
    const double g1 = std::sqrt(2.0 * kPi * T / m), g2 = std::sqrt(2.0 * kPi * T2 / m);
    const double g8 = std::sqrt(8.0 * kPi * T / m), gs = std::sqrt(2.0 * kPi);
    const double beam = 1.0 - den;
    parallel_ranges(n, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        const double vi = (v[i] - 0.5) * 2.0 * vmax;  // :181
        v[i] = vi;
        double f;
        switch (dist) {
          case 1: {  // :183-186
            const double q = vi * vi;
            f = pref * q * std::exp(-q / 2.0) / gs;
            break;
          }
          case 2: {  // :188-196
            const double up = vi + v0, um = vi - v0;
            f = pref * (std::exp(-(up * up) / a1) + std::exp(-(um * um) / a1)) / g8;
            break;
          }
          case 3: {  // :198-209
            const double um = vi - v0;
            f = pref * (den * std::exp(-(vi * vi) / a1) / g1 + beam * std::exp(-(um * um) / a2) / g2);
            break;
          }
          default: {  // :211-217
            const double um = vi - v0;
            f = pref * std::exp(-(um * um) / a1) / g1;
          }
        }
        p[i] = f;
      }
    });
  }

  // ---- positions, uniform in [0, lx]  (:222-223), then the perturbation
  // w = sum_modes [a cos(k x) + b sin(k x)] * p * shape(v), shape == 1 (:225-237)
  // and for a nonlinear run p += w (:260-264)
  g.fill_real(x, n);
  const int nim = in.init_nmode;
  std::vector<double> kk(nim);
  for (int j = 0; j < nim; ++j) kk[j] = 2.0 * kPi / lx * static_cast<double>(in.init_mode[j]);
  const bool nonlinear = in.linear == 0;
  parallel_ranges(n, nthreads, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const double xi = x[i] * lx;
      x[i] = xi;
      double amp = 0.0;
      for (int j = 0; j < nim; ++j) {
        // one sincos call per argument: GCC (so gfortran -O3, the reference's
        // compiler) merges the cos/sin pair of :228-231 into sincos, and glibc's
        // sincos differs from sin()/cos() in the last bit for ~0.05% of arguments
        double sn, cs;
        ::sincos(kk[j] * xi, &sn, &cs);
        amp = amp + in.init_mode_cos[j] * cs + in.init_mode_sin[j] * sn;
      }
      const double wi = amp * p[i] * 1.0;
      w[i] = wi;
      if (nonlinear) p[i] = p[i] + wi;
    }
  });
}

}  // namespace pic1dp
