// capi.cpp -- context, input, transfers, field access, timers and knobs behind the C ABI of include/pic1dp_hip.h (the hot
// path and the state machine of the lazy call sites: capi_step.cpp; the other units: ctx.hpp).
//
// One context = one process = one GPU = one HIP stream.  Every compute entry
// point only enqueues kernels (and at most one RCCL all-reduce) on that stream;
// nothing in the time loop synchronises with the host.
#include <algorithm>

#include "ctx.hpp"

namespace pic1dp_host {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

}  // namespace pic1dp_host

namespace {

int validate(const pic1dp_input &in, const pic1dp_layout &lay) {
  if (in.abi_version != PIC1DP_ABI_VERSION)
    return fail(PIC1DP_ERR_ARG, "abi_version %d != %d", in.abi_version, PIC1DP_ABI_VERSION);
  if (in.nspecies < 1 || in.nspecies > PIC1DP_MAX_SPECIES) return fail(PIC1DP_ERR_ARG, "nspecies out of range");
  if (in.nmode < 1 || in.nmode > PIC1DP_MAX_MODES) return fail(PIC1DP_ERR_ARG, "nmode out of range");
  if (in.init_nmode < 0 || in.init_nmode > PIC1DP_MAX_INIT_MODES) return fail(PIC1DP_ERR_ARG, "init_nmode out of range");
  if (in.nx < 2 || in.nx > 8192) return fail(PIC1DP_ERR_ARG, "nx must be in [2, 8192] (LDS-resident grid)");
  if (in.iptclshape != 4)
    return fail(PIC1DP_ERR_ARG, "only iptclshape = 4 is built (the PETSc shape-matrix variants 1-3 are out of scope)");
  if (in.iptcldist < 0 || in.iptcldist > 3) return fail(PIC1DP_ERR_ARG, "iptcldist out of range");
  if (in.deltaf != 0 && in.deltaf != 1) return fail(PIC1DP_ERR_ARG, "deltaf must be 0 or 1");
  if (in.linear != 0 && in.linear != 1) return fail(PIC1DP_ERR_ARG, "linear must be 0 or 1");
  // the two checks of input_init, src/pic1dp_input.F90:292-307
  if (in.iptcldist >= 1 && in.imarker == 1)
    return fail(PIC1DP_ERR_ARG, "case of input_iptcldist >= 1 and input_imarker = 1 not implemented yet");
  if (in.linear == 1 && in.deltaf == 0)
    return fail(PIC1DP_ERR_ARG, "case of input_linear = 1 and input_deltaf = 0 not implemented yet");
  if (in.imarker != 1 && in.imarker != 2) return fail(PIC1DP_ERR_ARG, "imarker must be 1 or 2");
  if (!(in.lx > 0.0) || !(in.dt > 0.0) || !(in.v_max > 0.0)) return fail(PIC1DP_ERR_ARG, "lx, dt, v_max must be positive");
  if (in.nparticle_max < 1) return fail(PIC1DP_ERR_ARG, "nparticle_max must be positive");
  for (int s = 0; s < in.nspecies; ++s) {
    if (in.species_nparticle_init[s] < 1 || in.species_nparticle_init[s] > in.nparticle_max)
      return fail(PIC1DP_ERR_ARG, "species_nparticle_init out of range");
    if (!(in.species_mass[s] > 0.0)) return fail(PIC1DP_ERR_ARG, "species_mass must be positive");
  }
  for (int m = 0; m < in.nmode; ++m)
    if (in.modes[m] < 1) return fail(PIC1DP_ERR_ARG, "modes must be >= 1");
  if (in.nmerge < 0 || in.nmerge > PIC1DP_MAX_OPT || in.nremove < 0 || in.nremove > PIC1DP_MAX_OPT ||
      in.nsplit < 0 || in.nsplit > PIC1DP_MAX_OPT)
    return fail(PIC1DP_ERR_ARG, "nmerge, nremove, nsplit must be in [0, %d]", PIC1DP_MAX_OPT);
  if (in.nmerge + in.nremove + in.nsplit > 0) {
    if (in.nv < 2) return fail(PIC1DP_ERR_ARG, "marker optimisation needs nv >= 2");
    if (in.typeremove != 1 && in.typeremove != 2) return fail(PIC1DP_ERR_ARG, "typeremove must be 1 or 2");
    if (in.nsplit > 0 && in.split_ngroup < 1) return fail(PIC1DP_ERR_ARG, "split_ngroup must be >= 1");
  }
  if (lay.nranks < 1 || lay.rank < 0 || lay.rank >= lay.nranks) return fail(PIC1DP_ERR_ARG, "bad rank/nranks");
  const int npe = lay.npe > 0 ? lay.npe : lay.nranks;
  if (npe % lay.nranks) return fail(PIC1DP_ERR_ARG, "npe must be a multiple of nranks");
  // (the field kernels hold the partial sums of the npe-rank summation order in LDS: 2 npe doubles beside their tiles)
  if (npe > 1024) return fail(PIC1DP_ERR_ARG, "npe must be at most 1024 reference ranks");
  return 0;
}

}  // namespace

namespace pic1dp_host {

// second particle set of the RK ping-pong (sub-step kernels only): a slab of the same
// geometry as the first (its p tiles stay unused)
int ensure_second_set(pic1dp_ctx *c) {
  for (Species &S : c->sp) {
    if (S.slab[1]) continue;
    HIP_TRY(hipMalloc(&S.slab[1], sizeof(double) * static_cast<size_t>(slab_doubles(S.nalloc + 2))));
    const int64_t as = slab_array_stride(S.nalloc + 2);
    S.set[1].x = S.slab[1];
    S.set[1].v = c->in.linear == 1 ? S.set[0].v : S.slab[1] + as;      // v is never pushed in a linear run
    S.set[1].w = c->in.deltaf == 0 ? S.set[0].w : S.slab[1] + 2 * as;  // w is not evolved in a full-f run
  }
  return 0;
}

}  // namespace pic1dp_host

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int pic1dp_hip_abi_version(void) { return PIC1DP_ABI_VERSION; }
int pic1dp_hip_tuning_build(void) {
#ifdef PIC1DP_TUNING
  return 1;
#else
  return 0;
#endif
}

const char *pic1dp_hip_last_error(void) { return g_err.c_str(); }

int pic1dp_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int pic1dp_hip_input_defaults(pic1dp_input *in) {
  if (!in) return fail(PIC1DP_ERR_ARG, "null input");
  std::memset(in, 0, sizeof *in);
  in->abi_version = PIC1DP_ABI_VERSION;
  in->ntime_max = 900000;
  in->time_max = 500.0;
  in->linear = 0;
  in->lx = 2.0 * 3.1415926535897932384626 / 0.36;
  in->iptcldist = 3;
  in->nspecies = 1;
  in->species_charge[0] = -1.0;
  in->species_mass[0] = 1.0;
  in->species_temperature[0] = 1.0;
  in->species_temperature2[0] = 1.0;
  in->species_density[0] = 0.9;
  in->species_v0[0] = 5.0;
  in->nmode = 1;
  in->modes[0] = 1;
  in->init_nmode = 1;
  in->init_mode[0] = 1;
  in->init_mode_cos[0] = 0.0;
  in->init_mode_sin[0] = 1e-5;
  in->deltaf = 1;
  in->dt = 0.05;
  in->nparticle_max = 6400000;
  in->species_nparticle_init[0] = 6400000;
  in->imarker = 2;
  in->v_max = 8.0;
  in->nx = 192;
  in->nv = 128;
  in->iptclshape = 4;
  in->multirand_al_int = 3;
  in->multirand_seed_type = 1;  // the shipped input uses 3 (/dev/urandom)
  in->multirand_warmup = 5;
  in->multirand_selftest = 1;
  in->output_interval = 0.5;
  in->nx_opd = 64;
  in->nv_opd = 64;
  // marker optimisation: disabled (src/pic1dp_input.F90:146,162,188).  Whoever
  // raises a count fills the time / threshold lists (the reference derives them
  // with implied-do formulas of the count, :149-158,165-180,191-200; the Python
  // and Fortran hosts do the same).
  in->nmerge = in->nremove = in->nsplit = 0;
  in->typeremove = 2;
  in->remove_frac = 0.9;
  in->split_ngroup = 5;
  in->split_dv_sig_frac = 0.1;
  return 0;
}

int pic1dp_hip_input_size(void) { return static_cast<int>(sizeof(pic1dp_input)); }

int pic1dp_hip_input_validate(const pic1dp_input *in, const pic1dp_layout *layout) {
  if (!in) return fail(PIC1DP_ERR_ARG, "null input");
  pic1dp_layout one{0, 1, 1, -1};
  return validate(*in, layout ? *layout : one);
}

int pic1dp_hip_block_sizes(const pic1dp_input *in, int32_t isp, int32_t mype, int32_t npe,
                           int64_t *nalloc, int64_t *np) {
  if (!in) return fail(PIC1DP_ERR_ARG, "null input");
  if (npe < 1 || mype < 0 || mype >= npe || isp < 0 || isp >= in->nspecies)
    return fail(PIC1DP_ERR_ARG, "bad block / species index");
  if (nalloc) *nalloc = block_alloc(in->nparticle_max, mype, npe);
  if (np) *np = block_np(*in, isp, mype, npe);
  return 0;
}

static int init_block_rng(const pic1dp_input &in, int mype, Multirand &g) {
  Multirand::Status st = g.init(in.multirand_al_int, in.multirand_seed_type, mype, in.multirand_warmup,
                                in.multirand_selftest != 0);
  if (st == Multirand::WOULD_HANG)
    return fail(PIC1DP_ERR_RNG,
                "multirand: al_int=3 with seed_type 1|2 needs selftest on (the reference spins forever at "
                "src/multirand.F90:346-348)");
  if (st == Multirand::SELFTEST_FAILED)
    return fail(PIC1DP_ERR_RNG, "multirand self-test: generator gives an unexpected sequence");
  if (st == Multirand::IO_ERROR) return fail(PIC1DP_ERR_RNG, "multirand: short read from /dev/urandom");
  return 0;
}

static int load_threads() {
  int nthreads = static_cast<int>(std::thread::hardware_concurrency());
  if (const char *e = std::getenv("PIC1DP_LOAD_THREADS")) nthreads = std::atoi(e);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  return nthreads;
}

int pic1dp_hip_host_particle_load(const pic1dp_input *in, int32_t mype, int32_t npe, double *x,
                                  double *v, double *p, double *w, int64_t nalloc) {
  if (!in || !x || !v || !p || !w) return fail(PIC1DP_ERR_ARG, "null argument");
  pic1dp_layout one{0, 1, 1, -1};
  if (int rc = validate(*in, one)) return rc;
  if (npe < 1 || mype < 0 || mype >= npe) return fail(PIC1DP_ERR_ARG, "bad block index");
  if (nalloc != block_alloc(in->nparticle_max, mype, npe))
    return fail(PIC1DP_ERR_ARG, "nalloc does not match the block's PETSC_DECIDE size");
  Multirand g;
  if (int rc = init_block_rng(*in, mype, g)) return rc;
  const int nthreads = load_threads();
  for (int s = 0; s < in->nspecies; ++s)
    load_block_species(*in, s, g, nalloc, x + s * nalloc, v + s * nalloc, p + s * nalloc, w + s * nalloc,
                       nthreads);
  return 0;
}

int pic1dp_hip_host_multirand_int64(int32_t al_int, int32_t seed_type, int32_t mype, int32_t warmup,
                                    int32_t selftest, int64_t *out, int64_t n) {
  if (!out || n < 0) return fail(PIC1DP_ERR_ARG, "bad output buffer");
  pic1dp_input in{};
  in.multirand_al_int = al_int;
  in.multirand_seed_type = seed_type;
  in.multirand_warmup = warmup;
  in.multirand_selftest = selftest;
  Multirand g;
  if (int rc = init_block_rng(in, mype, g)) return rc;
  for (int64_t i = 0; i < n; ++i) out[i] = static_cast<int64_t>(g.next());
  return 0;
}

// The create()-time self-test behind FieldArgs::chain_mfma (device_field.hpp chain_rows_mfma): do the serial sums come out
// of the FP64 matrix unit bit for bit as the host's additions one after the other?  Two data sets of sixteen rows of 1031
// values: A mixed signs, exponents in [-20, 20]; B the corners -- rows whose partial sums nearly cancel (a term, then its
// negative one ulp off, then small terms), rows of subnormal terms, rows mixing the smallest normals with subnormals, rows
// whose sums cross 2^-1022 in both directions (where a matrix unit's denormal handling would show).
// Returns 1 the matrix unit agrees in every bit of both sets, 0 it does not, -1 the test could not run (any HIP failure:
// an optional optimisation must not fail create(); the error state is cleared), -2 the ONE-LANE chain differs from the
// host on set A (fatal for the caller: that is the arithmetic the solve's bit-identity rests on).
static int chain_selftest(pic1dp_ctx *c) {
  constexpr int kn = 1031, kr = 16;
  std::vector<double> v(2 * kr * kn);
  uint64_t z = 0x9E3779B97F4A7C15ull;
  auto next = [&z]() {  // splitmix64
    z += 0x9E3779B97F4A7C15ull;
    uint64_t y = z;
    y = (y ^ (y >> 30)) * 0xBF58476D1CE4E5B9ull;
    y = (y ^ (y >> 27)) * 0x94D049BB133111EBull;
    return y ^ (y >> 31);
  };
  auto make = [](uint64_t sign, uint64_t biased_exp, uint64_t frac) {
    const uint64_t bits = (sign << 63) | (biased_exp << 52) | (frac & 0xFFFFFFFFFFFFFull);
    double x;
    std::memcpy(&x, &bits, 8);
    return x;
  };
  for (int i = 0; i < kr * kn; ++i) {  // A: sign, exponent in [-20, 20], random significand
    const uint64_t y = next();
    v[i] = make(y >> 63, 1023 - 20 + (y >> 52) % 41, y);
  }
  for (int r = 0; r < kr; ++r) {
    double *row = v.data() + static_cast<size_t>(kr + r) * kn;
    for (int i = 0; i < kn; ++i) {
      const uint64_t y = next();
      switch (r & 3) {
        case 0:  // near cancellation: x, -(x one ulp off), then a term 2^-40 .. 2^-60 of x
          if (i % 3 == 0)
            row[i] = make(y >> 63, 1023 + (y >> 52) % 7, y);
          else if (i % 3 == 1)
            row[i] = -std::nextafter(row[i - 1], (y & 1) ? 4e300 : 0.0);
          else
            row[i] = std::ldexp(row[i - 2], -40 - static_cast<int>((y >> 52) % 21));
          break;
        case 1:  // subnormal terms only
          row[i] = make(y >> 63, 0, y);
          break;
        case 2:  // the smallest normals among subnormals
          row[i] = make(y >> 63, (y >> 52) % 3, y);
          break;
        default:  // partial sums that cross the normal / subnormal border both ways
          row[i] = (i & 1) ? -make(0, 1 + (y >> 52) % 2, y) : make(0, 1 + (y >> 53) % 2, y >> 1);
          break;
      }
    }
  }
  double want[2 * kr];
  for (int r = 0; r < 2 * kr; ++r) {
    volatile double t = 0.0;  // (one rounding per addition whatever the compiler would like)
    for (int i = 0; i < kn; ++i) t = t + v[static_cast<size_t>(r) * kn + i];
    want[r] = t;
  }
  double *d_v = nullptr;
  double got[2][32];
  hipError_t err = hipMalloc(&d_v, sizeof(double) * (2 * kr * kn + 64));
  if (err == hipSuccess) err = hipMemcpy(d_v, v.data(), sizeof(double) * 2 * kr * kn, hipMemcpyHostToDevice);
  for (int set = 0; set < 2 && err == hipSuccess; ++set)
    err = launch_chain_selftest(d_v + static_cast<size_t>(set) * kr * kn, kr, kn, d_v + 2 * kr * kn + 32 * set, c->st);
  if (err == hipSuccess) err = hipStreamSynchronize(c->st);
  if (err == hipSuccess) err = hipMemcpy(got, d_v + 2 * kr * kn, sizeof got, hipMemcpyDeviceToHost);
  (void)hipFree(d_v);
  if (err != hipSuccess) {
    (void)hipGetLastError();  // cleared: the next call must not trip over this one's error
    return -1;
  }
  if (std::memcmp(got[0], want, sizeof(double) * kr) != 0) return -2;
  const bool a_ok = std::memcmp(got[0] + 16, want, sizeof(double) * kr) == 0;
  const bool b_ok = std::memcmp(got[1] + 16, want + kr, sizeof(double) * kr) == 0;
  if (const char *dbg = tuning_env("PIC1DP_CHAIN_SELFTEST_VERBOSE"))
    if (std::atoi(dbg) != 0)
      std::fprintf(stderr, "pic1dp: chain self-test: matrix unit set A %s, set B (cancellation, subnormals) %s; one-lane chain set B %s\n",
                   a_ok ? "identical" : "DIFFERS", b_ok ? "identical" : "DIFFERS",
                   std::memcmp(got[1], want + kr, sizeof(double) * kr) == 0 ? "identical" : "DIFFERS");
  return a_ok && b_ok ? 1 : 0;
}

int pic1dp_hip_create(const pic1dp_input *in, const pic1dp_layout *layout, pic1dp_ctx **out) {
  if (!in || !layout || !out) return fail(PIC1DP_ERR_ARG, "null argument");
  *out = nullptr;
  if (int rc = validate(*in, *layout)) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    (void)hipGetLastError();
    return fail(PIC1DP_ERR_NODEVICE, "no HIP device visible: this engine has no CPU path");
  }
  pic1dp_ctx *c = new pic1dp_ctx();
  c->in = *in;
  c->lay = *layout;
  if (c->lay.npe <= 0) c->lay.npe = c->lay.nranks;
  c->device = layout->device >= 0 ? layout->device : layout->rank % ndev;
  if (c->device >= ndev) {
    delete c;
    return fail(PIC1DP_ERR_ARG, "device %d not present (%d visible)", layout->device, ndev);
  }
#define HIP_TRY_C(expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      int rc_ = fail(PIC1DP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
      pic1dp_hip_destroy(c);                                                             \
      return rc_;                                                                        \
    }                                                                                    \
  } while (0)
  HIP_TRY_C(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIP_TRY_C(hipGetDeviceProperties(&prop, c->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = prop.gcnArchName;
    pic1dp_hip_destroy(c);
    return fail(PIC1DP_ERR_NODEVICE, "device arch %s: the kernels are built for gfx950 only", arch.c_str());
  }
  c->num_cu = prop.multiProcessorCount;
  HIP_TRY_C(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));

  const int npe = c->lay.npe;
  c->nblk = npe / c->lay.nranks;
  c->blk0 = c->lay.rank * c->nblk;
  c->blk_alloc.resize(c->nblk);
  c->blk_np.assign(in->nspecies, std::vector<int64_t>(c->nblk, 0));
  for (int b = 0; b < c->nblk; ++b) {
    c->blk_alloc[b] = block_alloc(in->nparticle_max, c->blk0 + b, npe);
    for (int s = 0; s < in->nspecies; ++s) c->blk_np[s][b] = block_np(*in, s, c->blk0 + b, npe);
  }
  c->blk_rng.resize(c->nblk);
  // particle_init, src/pic1dp_particle.F90:73-87
  c->imerge = in->nmerge > 0 ? 1 : 0;
  c->iremove = in->nremove > 0 ? 1 : 0;
  c->isplit = in->nsplit > 0 ? 1 : 0;
  const int nx = in->nx, nm = in->nmode, ns = in->nspecies;
  c->grid.lx = in->lx;
  c->grid.dnx = static_cast<double>(nx);
  c->grid.dt_full = in->dt;
  c->grid.nx = nx;
  c->grid.rlx = 1.0 / in->lx;
  if (const char *e = tuning_env("PIC1DP_OSUB")) c->osub_req = std::max(0, std::atoi(e));
  if (const char *e = tuning_env("PIC1DP_DYN_TAIL")) c->dyn_tail = c->dyn_tail_full = std::max(0, std::min(16, std::atoi(e)));
  if (const char *e = tuning_env("PIC1DP_DYN_TAIL_FULL")) c->dyn_tail_full = std::max(0, std::min(16, std::atoi(e)));
  if (const char *e = tuning_env("PIC1DP_NT_THRESHOLD_MB"))
    c->nt_threshold_half = c->nt_threshold_full = std::atof(e) * 1048576.0;
  if (const char *e = tuning_env("PIC1DP_NT_THRESHOLD_FULL_MB")) c->nt_threshold_full = std::atof(e) * 1048576.0;

  // particle storage: valid markers of the owned blocks packed first, block
  // tails (allocated but unloaded slots) behind them
  c->sp.resize(ns);
  int64_t nalloc = 0;
  for (int b = 0; b < c->nblk; ++b) nalloc += block_alloc(in->nparticle_max, c->blk0 + b, npe);
  // species charge accumulators in gcopies copies: workgroup b adds its LDS tile into copy b % gcopies,
  // so an address receives 1/gcopies of the flush atomics; the field kernels add the copies up.
  // Measured (tools/fresh_and_flush.sh): the flush into ONE copy costs 4.8 % of a step at 6.4e6
  // markers / nx 192, 1.5 % at 1e7 / 256, 0.9 % at 1e8 / 1024.  Eight copies (tools/ab_global_copies.sh)
  // win back 2 % of the step at nx 192, nothing at nx 256, and LOSE 4 % at 1.25e7 markers / nx 1024: the
  // one-workgroup field kernel then reads and re-zeroes 8 x nx words on the critical path.  So: eight
  // copies for small grids only.  PIC1DP_RHO_GLOBAL_COPIES overrides.
  c->grid.gcopies = nx <= 256 ? 8 : 1;
  if (const char *e = tuning_env("PIC1DP_RHO_GLOBAL_COPIES")) {
    const int k = std::atoi(e);
    if (k >= 1 && k <= 64 && (k & (k - 1)) == 0) c->grid.gcopies = k;
  }
  c->grid.gstride = ns * nx;
  const size_t rho_doubles = static_cast<size_t>(c->grid.gcopies) * ns * nx;
  c->rho_set_doubles = rho_doubles;
  HIP_TRY_C(hipMalloc(&c->d_rho_all, sizeof(double) * 3 * rho_doubles));
  HIP_TRY_C(hipMemsetAsync(c->d_rho_all, 0, sizeof(double) * 3 * rho_doubles, c->st));
  c->d_rho_sp = c->d_rho_all;
  if (const char *e = std::getenv("PIC1DP_FUSE_SOLVE")) c->fuse_solve = std::max(0, std::min(2, std::atoi(e)));
  if (const char *e = std::getenv("PIC1DP_TAIL")) c->tail_on = std::atoi(e) != 0;
  if (const char *e = std::getenv("PIC1DP_CALL_PAIR")) c->call_pair = std::atoi(e) != 0;
  if (const char *e = std::getenv("PIC1DP_DIAG_FX")) c->diag_fx = std::atoi(e) != 0;
  if (const char *e = std::getenv("PIC1DP_DIAG_FX_MARGIN")) c->diag_fx_margin_w = std::atof(e);
  HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&c->d_ticket), 64));
  HIP_TRY_C(hipMemsetAsync(c->d_ticket, 0, 64, c->st));
  for (int s = 0; s < ns; ++s) {
    Species &S = c->sp[s];
    S.nalloc = nalloc;
    S.np = 0;
    for (int b = 0; b < c->nblk; ++b) S.np += block_np(*in, s, c->blk0 + b, npe);
    S.sc = make_species_const(SpeciesInput{in->iptcldist, in->species_charge[s], in->species_mass[s],
                                           in->species_temperature[s], in->species_temperature2[s],
                                           in->species_density[s], in->species_v0[s]}, s);
    S.rho = c->d_rho_sp + static_cast<size_t>(s) * nx;
    // x, v, w, p of a species interleaved in tiles in ONE slab (kernels.hpp: 32 B per
    // marker, 9e9 markers in 288 GB); the slab of the RK ping-pong set is allocated on
    // the first sub-step call that needs it (ensure_second_set) -- pic1dp_hip_step
    // never does
    HIP_TRY_C(hipMalloc(&S.slab[0], sizeof(double) * static_cast<size_t>(slab_doubles(nalloc + 2))));
    HIP_TRY_C(hipMalloc(&S.fxb, 4 * sizeof(double)));   // two bounds, a 64-bit count (kernel_stats 13), one word spare
    HIP_TRY_C(hipMemsetAsync(S.fxb, 0, 4 * sizeof(double), c->st));
    const int64_t as = slab_array_stride(nalloc + 2);
    S.set[0].x = S.slab[0];
    S.set[0].v = S.slab[0] + as;
    S.set[0].w = S.slab[0] + 2 * as;
    S.p = S.slab[0] + 3 * as;
  }

  // field storage and the operators of field_init (src/pic1dp_field.F90:158-210),
  // evaluated on the host with libm like the reference, stored mode-major
  HIP_TRY_C(hipMalloc(&c->d_charge, sizeof(double) * nx));
  HIP_TRY_C(hipMalloc(&c->d_chargeden, sizeof(double) * nx));
  HIP_TRY_C(hipMalloc(&c->d_E, sizeof(double) * nx));
  HIP_TRY_C(hipMalloc(&c->d_Eh, sizeof(double) * nx));
  HIP_TRY_C(hipMemsetAsync(c->d_Eh, 0, sizeof(double) * nx, c->st));
  HIP_TRY_C(hipMalloc(&c->d_E0, sizeof(double) * nx));
  HIP_TRY_C(hipMalloc(&c->d_rho_dummy, sizeof(double) * rho_doubles));
  if (const char *e = std::getenv("PIC1DP_LAZY_CALLS")) c->lazy_calls = std::atoi(e) != 0;
  if (const char *e = tuning_env("PIC1DP_CARRY")) c->carry = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("PIC1DP_PREDICT")) c->predict = std::atoi(e);
  HIP_TRY_C(hipMalloc(&c->d_mode_re, sizeof(double) * nm));
  HIP_TRY_C(hipMalloc(&c->d_mode_im, sizeof(double) * nm));
  HIP_TRY_C(hipMalloc(&c->d_fre, sizeof(double) * nm * nx));
  HIP_TRY_C(hipMalloc(&c->d_fim, sizeof(double) * nm * nx));
  HIP_TRY_C(hipMalloc(&c->d_ginv, sizeof(double) * nm));
  HIP_TRY_C(hipMalloc(&c->d_hist, sizeof(double) * kHistCap));
  HIP_TRY_C(hipMalloc(&c->d_scratch, sizeof(double) * (kEnergyBlocks * 3 + 16)));
  HIP_TRY_C(hipMemsetAsync(c->d_charge, 0, sizeof(double) * nx, c->st));
  HIP_TRY_C(hipMemsetAsync(c->d_chargeden, 0, sizeof(double) * nx, c->st));
  HIP_TRY_C(hipMemsetAsync(c->d_E, 0, sizeof(double) * nx, c->st));
  HIP_TRY_C(hipMemsetAsync(c->d_mode_re, 0, sizeof(double) * nm, c->st));
  HIP_TRY_C(hipMemsetAsync(c->d_mode_im, 0, sizeof(double) * nm, c->st));
  {
    std::vector<double> fre(static_cast<size_t>(nm) * nx), fim(static_cast<size_t>(nm) * nx), gi(nm);
    for (int m = 0; m < nm; ++m) {
      const double mode = static_cast<double>(in->modes[m]);
      gi[m] = 1.0 / (2.0 * kPi / in->lx * mode);  // :166
      // two loops, plain cos() and plain sin(), as the reference's two fills
      // (:186-189, :194-197): a paired sincos can differ in the last bit
      double (*volatile cos_fn)(double) = std::cos;
      double (*volatile sin_fn)(double) = std::sin;
      for (int ix = 0; ix < nx; ++ix) {
        const double th = 2.0 * kPi / static_cast<double>(nx) * mode * static_cast<double>(ix);  // :188
        fre[static_cast<size_t>(m) * nx + ix] = cos_fn(th);
      }
      for (int ix = 0; ix < nx; ++ix) {
        const double th = 2.0 * kPi / static_cast<double>(nx) * mode * static_cast<double>(ix);  // :196
        fim[static_cast<size_t>(m) * nx + ix] = -sin_fn(th);
      }
    }
    // one pass per step: with prediction tiles where they fit the LDS (k_step_one), as six sums for larger
    // grids with one kept mode (k_step_sums); PIC1DP_PRED_KIND=1|2|3 insists on tiles | sums | sums in registers (tests)
    const bool private_fits = 2 * (step_one_private_lds_bytes(nx) + kStaticLds) <= kCuLds;
    // The prediction tiles cost 2 + 4 nm LDS atomics at random cells per marker.  Measured against the two passes
    // (profiles/r04/experiments/ab_kept_modes.log, 1e8 markers / nx 1024, ms per step): two kept modes 1.23 against 1.46,
    // three 1.49 against 1.45, four 2.05 (nx 512) against 1.46.  With the tiles as fixed-point sums (round 6: kernels_step.hip
    // FxTiles) two kept modes run 1.06 and three 1.17 ms against the two passes' 1.40 (profiles/r06/experiments/ab_fx_tiles.log,
    // ab_nm3.log): the tiles serve up to three kept modes, four and more take the two passes.
    if (nm <= PRED_MAX_MODES && step_one_lds_bytes(nx, nm) <= PARTICLE_LDS_CAP)
      c->pred_kind = 1;
    // (the six sums travel in the head of an nx-vector on the call-site path: nx >= 8)
    else if (nm == 1 && nx >= 8 && step_sums_lds_bytes(nx) <= PARTICLE_LDS_CAP)
      c->pred_kind = 2;
    // One kept mode: the six sums in thread-private LDS slots (k_step_one<PRIV>) beat the tiles, whose six atomics per
    // marker at random cells pay ~3x in bank conflicts: -2 % at 1e8 markers / nx 1024 (profiles/r03/experiments/
    // ab_private_sums.log), and on the small grids too once the step is timed without per-kernel events in the stream:
    // 6.4e6 / nx 192 83.8 -> 76.6 us per step, 1e7 / nx 256 123.7 -> 116.5 (profiles/r04/experiments/ab_fused_solve.log)
    // -- wherever the slots of two workgroups fit (nx >= 8: the sums travel in the head of an nx-vector on the
    // call-site path).
    if (c->pred_kind == 1 && nm == 1 && nx >= 8 && private_fits) c->pred_kind = 2;
    bool sums_in_registers = false;
    if (const char *e = std::getenv("PIC1DP_PRED_KIND")) {
      const int k = std::atoi(e);
      if ((k == 2 || k == 3) && nm == 1 && nx >= 8 && step_sums_lds_bytes(nx) <= PARTICLE_LDS_CAP) c->pred_kind = 2;
      if (k == 3) sums_in_registers = true;  // k_step_sums also where the private slots would fit (tests: the large-grid kernel at a small grid)
      // 1: the tiles wherever they fit (else the choice above stands)
      if (k == 1 && nm <= PRED_MAX_MODES && step_one_lds_bytes(nx, nm) <= PARTICLE_LDS_CAP) c->pred_kind = 1;
    }
    if (c->pred_kind == 2 && private_fits && !sums_in_registers) c->pred_private = 1;
    if (c->pred_kind) {  // the tables: E = 2*(cos re + (-sin) im) (src/pic1dp_field.F90:251-257)
      std::vector<double> ta(fre.size()), tb(fim.size());
      for (size_t i = 0; i < fre.size(); ++i) ta[i] = 2.0 * fre[i], tb[i] = 2.0 * fim[i];
      if (c->pred_kind == 2) {
        PredTab &pt = c->pred_tab;
        for (int ix = 0; ix < nx; ++ix) {
          pt.sum_fre += fre[ix], pt.sum_fim += fim[ix];
          pt.g11 += fre[ix] * fre[ix], pt.g22 += fim[ix] * fim[ix], pt.g12 += fre[ix] * fim[ix];
        }
      }
      const size_t pred_doubles = c->pred_kind == 2 ? 8 * PRED_SUM_COPIES : static_cast<size_t>(ns) * (1 + 2 * nm) * nx;
      HIP_TRY_C(hipMalloc(&c->d_tabA, sizeof(double) * nm * nx));
      HIP_TRY_C(hipMalloc(&c->d_tabB, sizeof(double) * nm * nx));
      c->pred_set_doubles = pred_doubles;
      HIP_TRY_C(hipMalloc(&c->d_pred_all, sizeof(double) * 3 * pred_doubles));
      c->d_pred = c->d_pred_all;
      HIP_TRY_C(hipMalloc(&c->d_cd_h, sizeof(double) * nx));
      HIP_TRY_C(hipMalloc(&c->d_Ehn, sizeof(double) * nx));
      HIP_TRY_C(hipMalloc(&c->d_pack, sizeof(double) * pack_doubles(nx, nm, c->pred_kind)));
      HIP_TRY_C(hipMalloc(&c->d_mode_h, sizeof(double) * 2 * nm));
      HIP_TRY_C(hipMemcpy(c->d_tabA, ta.data(), sizeof(double) * nm * nx, hipMemcpyHostToDevice));
      HIP_TRY_C(hipMemcpy(c->d_tabB, tb.data(), sizeof(double) * nm * nx, hipMemcpyHostToDevice));
      HIP_TRY_C(hipMemset(c->d_pred_all, 0, sizeof(double) * 3 * pred_doubles));
    }
    HIP_TRY_C(hipMemcpy(c->d_fre, fre.data(), sizeof(double) * nm * nx, hipMemcpyHostToDevice));
    HIP_TRY_C(hipMemcpy(c->d_fim, fim.data(), sizeof(double) * nm * nx, hipMemcpyHostToDevice));
    HIP_TRY_C(hipMemcpy(c->d_ginv, gi.data(), sizeof(double) * nm, hipMemcpyHostToDevice));
  }
  FieldArgs &f = c->fa;
  f.rho_sp = c->d_rho_sp;
  f.rho_copies = c->grid.gcopies;
  f.rho_stride = c->grid.gstride;
  f.charge = c->d_charge;
  f.chargeden = c->d_chargeden;
  f.E = c->d_E;
  f.mode_re = c->d_mode_re;
  f.mode_im = c->d_mode_im;
  f.fre = c->d_fre;
  f.fim = c->d_fim;
  f.grad_inv = c->d_ginv;
  f.history = nullptr;
  f.nx = nx;
  f.nmode = nm;
  f.nspecies = ns;
  f.deltaf = in->deltaf;
  f.npe = c->lay.npe;  // the summation order of the reference run being reproduced (PIC1DP_FIELD_ONE_RANK_ORDER=1: tests)
  if (const char *e = tuning_env("PIC1DP_FIELD_ONE_RANK_ORDER"))
    if (std::atoi(e) != 0) f.npe = 1;
  f.tab_lds = (static_cast<size_t>(2) * nm * nx * sizeof(double) <= 96 * 1024) ? 1 : 0;
  f.lx = in->lx;
  f.dnx = static_cast<double>(nx);
  f.sc_re = 1.0 / static_cast<double>(nx);    // src/pic1dp_field.F90:239
  f.sc_im = -1.0 / static_cast<double>(nx);   // :234
  // The serial forward sums through the FP64 matrix unit -- only if this device gives the sequential sums bit for bit
  // that way (device_field.hpp chain_rows_mfma): sixteen rows of 1031 values of mixed sign and magnitude against the
  // host's additions one after the other.  PIC1DP_CHAIN_MFMA=0 keeps the chain of additions in one lane.
  f.chain_mfma = 0;
  {
    const char *e = std::getenv("PIC1DP_CHAIN_MFMA");
    if (!e || std::atoi(e) != 0) {
      const int verdict = chain_selftest(c);
      if (verdict == -2) {  // (the lane's own chain: the arithmetic every bit-identity claim of the solve rests on)
        pic1dp_hip_destroy(c);
        return fail(PIC1DP_ERR_HIP, "the device's serial sum differs from the host's sequential additions");
      }
      f.chain_mfma = verdict == 1 ? 1 : 0;  // -1 (the test itself could not run): the optimisation is simply off
      c->chain_selftest = verdict;
      if (e && std::atoi(e) > 0 && f.chain_mfma == 0) {
        pic1dp_hip_destroy(c);
        return fail(PIC1DP_ERR_HIP, "PIC1DP_CHAIN_MFMA asked for, but the matrix unit does not give the sequential sums on this device");
      }
    }
  }
  for (int s = 0; s < ns; ++s) {
    f.Z[s] = in->species_charge[s];
    f.n0[s] = in->species_density[s];
  }
  HIP_TRY_C(hipStreamSynchronize(c->st));
#undef HIP_TRY_C
  *out = c;
  return 0;
}

int pic1dp_hip_destroy(pic1dp_ctx *c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->st) (void)hipStreamSynchronize(c->st);
  comm_release(c);
  optimize_release(c);
  for (auto &S : c->sp) {
    (void)hipFree(S.slab[0]);
    (void)hipFree(S.slab[1]);
    (void)hipFree(S.t2);
    (void)hipFree(S.fxb);
  }
  double *bufs[] = {c->d_rho_all, c->d_charge, c->d_chargeden, c->d_E,   c->d_mode_re, c->d_mode_im,
                    c->d_fre,    c->d_fim,    c->d_ginv,      c->d_hist, c->d_scratch, c->d_dist, c->d_Eh, c->d_diag_part, c->d_E0, c->d_rho_dummy, c->d_stage, c->d_tabA, c->d_tabB, c->d_pred_all, c->d_cd_h, c->d_mode_h, c->d_Ehn, c->d_pack};
  for (double *b : bufs) (void)hipFree(b);
  (void)hipFree(c->d_ticket);
  (void)hipFree(c->d_rec);
  (void)hipHostFree(c->h_pin);
  for (auto &e : c->evpool) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
  return 0;
}

// Host arrays are contiguous, marker arrays tiled (kernels.hpp): every transfer goes
// through a contiguous device staging buffer and a scatter / gather kernel, in chunks.
constexpr int64_t kStageDoubles = static_cast<int64_t>(8) << 20;  // 64 MiB

static int ensure_stage(pic1dp_ctx *c) {
  if (!c->d_stage) HIP_TRY(hipMalloc(&c->d_stage, sizeof(double) * kStageDoubles));
  return 0;
}

// host[0, cnt) -> markers [off, off + cnt) of the array starting at arr; returns
// with the host buffer free for reuse
}  // extern "C"
int pic1dp_host::put_range(pic1dp_ctx *c, double *arr, int64_t off, const double *host, int64_t cnt) {
  if (cnt <= 0) return 0;
  if (int rc = ensure_stage(c)) return rc;
  for (int64_t done = 0; done < cnt; done += kStageDoubles) {
    const int64_t n = std::min(kStageDoubles, cnt - done);
    HIP_TRY(hipMemcpyAsync(c->d_stage, host + done, sizeof(double) * n, hipMemcpyHostToDevice, c->st));
    HIP_TRY(launch_tile_scatter(arr, off + done, c->d_stage, n, c->st));
  }
  HIP_TRY(hipStreamSynchronize(c->st));
  return 0;
}
extern "C" {

// markers [off, off + cnt) of the array starting at arr -> host[0, cnt)
}  // extern "C"
int pic1dp_host::get_range(pic1dp_ctx *c, const double *arr, int64_t off, double *host, int64_t cnt) {
  if (cnt <= 0) return 0;
  if (int rc = ensure_stage(c)) return rc;
  for (int64_t done = 0; done < cnt; done += kStageDoubles) {
    const int64_t n = std::min(kStageDoubles, cnt - done);
    HIP_TRY(launch_tile_gather(arr, off + done, c->d_stage, n, c->st));
    HIP_TRY(hipMemcpyAsync(host + done, c->d_stage, sizeof(double) * n, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
  }
  return 0;
}
extern "C" {

int pic1dp_hip_local_sizes(pic1dp_ctx *c, int32_t isp, int64_t *nalloc, int64_t *np) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  if (nalloc) *nalloc = c->sp[isp].nalloc;
  if (np) *np = c->sp[isp].np;
  return 0;
}


// Bounds for the prediction tiles' fixed-point sums (kernels_step.hip FxTiles) from markers the host holds: b[0] >= |q|
// (w, or p in a full-f run), b[1] >= |c| = dt/2 |p - w| |f0'/f0|(v) |Z/m| -- the latter from max |p| + max |w| and a bound on
// |f0'/f0| over the velocities markers can reach (src/pic1dp_interaction.F90:274-326: a blend of v / (T/m) and
// (v - v0) / (T2/m); two-stream1's v - 2/v only away from v = 0).  They need not be tight, nor even hold: a term beyond
// 16x its bound takes the kernel's double path, and the kernels raise the bounds to what they meet.
static void fx_bounds_of(const pic1dp_input &in, int isp, const double *p, const double *w, int64_t n, double b[2]) {
  double maxp = 0.0, maxw = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double ap = std::fabs(p[i]), aw = std::fabs(w[i]);
    if (ap > maxp) maxp = ap;   // (NaN compares false)
    if (aw > maxw) maxw = aw;
  }
  const double m = std::fabs(in.species_mass[isp]), tm = std::fabs(in.species_temperature[isp]) / m,
               tm2 = std::fabs(in.species_temperature2[isp]) / m, v0 = std::fabs(in.species_v0[isp]);
  const double vm = 1.5 * in.v_max + v0;
  double d = vm / tm;
  if (in.iptcldist == 1) d = vm + 200.0;
  if (in.iptcldist == 2) d = (vm + v0) / tm;
  if (in.iptcldist == 3) d = std::max(vm / tm, (vm + v0) / tm2);
  const double cb = 0.5 * in.dt * (maxp + maxw) * d * std::fabs(in.species_charge[isp]) / m;
  b[0] = std::max(b[0], in.deltaf ? maxw : maxp);
  if (std::isfinite(cb)) b[1] = std::max(b[1], cb);
}

int pic1dp_hip_set_seed_offset(pic1dp_ctx *c, int32_t offset) {
  CHECK_CTX(c);
  if (offset < 0) return fail(PIC1DP_ERR_ARG, "seed offset < 0");
  c->seed_offset = offset;
  return 0;
}

int pic1dp_hip_particle_load(pic1dp_ctx *c) {
  CHECK_CTX(c);
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = materialize_cd(c)) return rc;  // deposits of the old markers are consumed, not mixed with the new ones
  HIP_TRY(hipStreamSynchronize(c->st));  // kernels of an earlier run may still be writing the arrays
  if (int rc = set_call_state(c, Seq::Clean, Owed::Nothing)) return rc;  // a noted push of markers that are about to be replaced is
                                                                         // void, and so is a half-step field waiting to be adopted
  c->state_version++;
  const pic1dp_input &in = c->in;
  const int npe = c->lay.npe, ns = in.nspecies;
  const int nthreads = load_threads();
  int64_t max_alloc = 0;
  for (int b = 0; b < c->nblk; ++b)
    max_alloc = std::max<int64_t>(max_alloc, block_alloc(in.nparticle_max, c->blk0 + b, npe));
  double *stage = nullptr;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&stage), sizeof(double) * 4 * static_cast<size_t>(max_alloc), 0));
  double *hx = stage, *hv = stage + max_alloc, *hp = stage + 2 * max_alloc, *hw = stage + 3 * max_alloc;
  // a (re)load restores the loader's marker counts and optimisation counters
  for (int s = 0; s < ns; ++s) {
    c->sp[s].np = 0;
    for (int b = 0; b < c->nblk; ++b) {
      c->blk_np[s][b] = block_np(in, s, c->blk0 + b, npe);
      c->sp[s].np += c->blk_np[s][b];
    }
  }
  c->imerge = in.nmerge > 0 ? 1 : 0;
  c->iremove = in.nremove > 0 ? 1 : 0;
  c->isplit = in.nsplit > 0 ? 1 : 0;
  c->rng_ready = false;
  std::vector<int64_t> voff(ns, 0), toff(ns);
  for (int s = 0; s < ns; ++s) toff[s] = c->sp[s].np;
  std::vector<double> fxb(2 * static_cast<size_t>(ns), 0.0);
  int rc = 0;
  for (int b = 0; b < c->nblk && !rc; ++b) {
    const int mype = c->blk0 + b;
    const int64_t n = block_alloc(in.nparticle_max, mype, npe);
    Multirand g;
    if ((rc = init_block_rng(in, mype + c->seed_offset, g)) != 0) break;
    for (int s = 0; s < ns && !rc; ++s) {
      Species &S = c->sp[s];
      load_block_species(in, s, g, n, hx, hv, hp, hw, nthreads);
      const int64_t np = block_np(in, s, mype, npe), nt = n - np;
      fx_bounds_of(in, s, hp, hw, np, &fxb[2 * static_cast<size_t>(s)]);
      struct {
        double *d;
        const double *h;
      } arr[4] = {{S.set[0].x, hx}, {S.set[0].v, hv}, {S.p, hp}, {S.set[0].w, hw}};
      for (auto &a : arr) {
        if ((rc = put_range(c, a.d, voff[s], a.h, np)) != 0) break;
        if ((rc = put_range(c, a.d, toff[s], a.h + np, nt)) != 0) break;
      }
      voff[s] += np;
      toff[s] += nt;
    }
    if (!rc) c->blk_rng[b] = g;  // remove / split continue this block's stream
  }
  (void)hipHostFree(stage);
  if (rc) return rc;
  for (int s = 0; s < ns; ++s) {
    HIP_TRY(hipMemcpy(c->sp[s].fxb, &fxb[2 * static_cast<size_t>(s)], 2 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(c->sp[s].fxb + 2, 0, 2 * sizeof(double)));   // (the count of terms past the bounds starts over)
  }
  c->rng_ready = true;
  std::fill(c->diag_max_p.begin(), c->diag_max_p.end(), 0.0);  // (new markers: the fixed-point diagnostics' bounds are void)
  std::fill(c->diag_max_w.begin(), c->diag_max_w.end(), 0.0);
  c->cur = 0;
  c->loaded = true;
  c->itime = 0;
  c->time = 0.0;
  c->hist_count = 0;
  return 0;
}

int pic1dp_hip_particles_upload(pic1dp_ctx *c, int32_t isp, const double *x, const double *v,
                                const double *p, const double *w, int64_t n, int64_t np) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  Species &S = c->sp[isp];
  if (!x || !v || !p || !w) return fail(PIC1DP_ERR_ARG, "null array");
  if (n != S.nalloc) return fail(PIC1DP_ERR_ARG, "n = %lld but this process owns %lld slots", (long long)n, (long long)S.nalloc);
  if (np < 0 || np > n) return fail(PIC1DP_ERR_ARG, "np out of range");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = materialize_cd(c)) return rc;   // (first: what collect_charge left pending is settled before the sequence is left)
  if (c->loaded)
    if (int rc = materialize(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->st));
  c->state_version++;
  // an upload (re)starts from set 0 for every species: slots beyond np live there
  if (c->cur != 0) {
    for (Species &T : c->sp) {
      if (&T == &S) continue;
      HIP_TRY(launch_tile_copy(T.set[0].x, T.set[1].x, T.np, c->st));
      if (T.set[1].v != T.set[0].v) HIP_TRY(launch_tile_copy(T.set[0].v, T.set[1].v, T.np, c->st));
      if (T.set[1].w != T.set[0].w) HIP_TRY(launch_tile_copy(T.set[0].w, T.set[1].w, T.np, c->st));
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    c->cur = 0;
  }
  PSet &A = S.set[0];
  if (int rc = put_range(c, A.x, 0, x, n)) return rc;
  if (int rc = put_range(c, A.v, 0, v, n)) return rc;
  if (int rc = put_range(c, A.w, 0, w, n)) return rc;
  if (int rc = put_range(c, S.p, 0, p, n)) return rc;
  S.np = np;
  {  // (the prediction tiles' fixed-point bounds start over with these markers)
    double b[2] = {0.0, 0.0};
    fx_bounds_of(c->in, isp, p, w, np, b);
    HIP_TRY(hipMemcpy(S.fxb, b, sizeof b, hipMemcpyHostToDevice));
  }
  if (c->nblk == 1) c->blk_np[isp][0] = np;
  c->rng_ready = false;  // the host's loader owns the random stream now
  std::fill(c->diag_max_p.begin(), c->diag_max_p.end(), 0.0);
  std::fill(c->diag_max_w.begin(), c->diag_max_w.end(), 0.0);
  c->loaded = true;
  return 0;
}

int pic1dp_hip_particles_download(pic1dp_ctx *c, int32_t isp, double *x, double *v, double *p,
                                  double *w, int64_t n) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  Species &S = c->sp[isp];
  if (n != S.nalloc) return fail(PIC1DP_ERR_ARG, "n = %lld but this process owns %lld slots", (long long)n, (long long)S.nalloc);
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = materialize(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->st));
  const PSet &A = S.set[c->cur];
  const int64_t np = S.np;
  // slots beyond np are only defined in the set they were uploaded to (set 0)
  auto pull = [&](double *h, const double *cur, const double *first) -> int {
    if (!h) return 0;
    if (int rc = get_range(c, cur, 0, h, np)) return rc;
    return get_range(c, first, np, h + np, n - np);
  };
  if (int rc = pull(x, A.x, S.set[0].x)) return rc;
  if (int rc = pull(v, A.v, S.set[0].v)) return rc;
  if (int rc = pull(w, A.w, S.set[0].w)) return rc;
  if (int rc = pull(p, S.p, S.p)) return rc;
  return 0;
}

int pic1dp_hip_particles_download_bak(pic1dp_ctx *c, int32_t isp, double *xb, double *vb, double *wb,
                                      int64_t n) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  Species &S = c->sp[isp];
  if (n != S.nalloc) return fail(PIC1DP_ERR_ARG, "n does not match the owned slots");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = materialize(c)) return rc;
  if (!S.slab[1]) return fail(PIC1DP_ERR_STATE, "no RK backup exists before the first push / substep call");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->st));
  const PSet &B = S.set[1 - c->cur];
  if (xb)
    if (int rc = get_range(c, B.x, 0, xb, S.np)) return rc;
  if (vb)
    if (int rc = get_range(c, B.v, 0, vb, S.np)) return rc;
  if (wb)
    if (int rc = get_range(c, B.w, 0, wb, S.np)) return rc;
  return 0;
}

int pic1dp_hip_get_field_half(pic1dp_ctx *c, double *electric_half) {
  CHECK_CTX(c);
  if (!electric_half) return fail(PIC1DP_ERR_ARG, "null array");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->st));
  HIP_TRY(hipMemcpy(electric_half, c->d_Eh, sizeof(double) * c->in.nx, hipMemcpyDeviceToHost));
  return 0;
}

int pic1dp_hip_sync(pic1dp_ctx *c) {
  CHECK_CTX(c);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->st));
  return xchg_check(c);
}

int pic1dp_hip_get_time(pic1dp_ctx *c, int32_t *itime, double *time) {
  CHECK_CTX(c);
  if (itime) *itime = c->itime;
  if (time) *time = c->time;
  return 0;
}

int pic1dp_hip_set_time(pic1dp_ctx *c, int32_t itime, double time) {
  CHECK_CTX(c);
  c->itime = itime;
  c->time = time;
  return 0;
}

int pic1dp_hip_check_termination(pic1dp_ctx *c, int32_t *flag) {
  CHECK_CTX(c);
  if (!flag) return fail(PIC1DP_ERR_ARG, "null flag");
  *flag = (c->itime >= c->in.ntime_max || c->time + kSqrtEps >= c->in.time_max) ? 1 : 0;
  return 0;
}

int pic1dp_hip_output_due(pic1dp_ctx *c, int32_t itermination, int32_t *flag) {
  CHECK_CTX(c);
  if (!flag) return fail(PIC1DP_ERR_ARG, "null flag");
  const double a = std::fmod(c->time + kSqrtEps, c->in.output_interval);
  const double b = std::fmod(c->time + kSqrtEps - c->in.dt, c->in.output_interval);
  *flag = (a < b || itermination == 1) ? 1 : 0;
  return 0;
}

int pic1dp_hip_steps_to_output(pic1dp_ctx *c, int32_t *nsteps) {
  CHECK_CTX(c);
  if (!nsteps) return fail(PIC1DP_ERR_ARG, "null output");
  const pic1dp_input &in = c->in;
  int32_t it = c->itime, n = 0;
  double t = c->time;
  // src/pic1dp.F90:78-109 walked ahead: step, check_termination, the cadence test
  while (!(it >= in.ntime_max || t + kSqrtEps >= in.time_max)) {
    it += 1;
    t = t + in.dt;
    n += 1;
    if (it >= in.ntime_max || t + kSqrtEps >= in.time_max) break;  // the last step always writes
    if (in.output_interval > 0.0 &&
        std::fmod(t + kSqrtEps, in.output_interval) < std::fmod(t + kSqrtEps - in.dt, in.output_interval))
      break;
    if (n == 0x7fffffff) break;
  }
  *nsteps = n;
  return 0;
}

// ---------------------------------------------------------------------------
// field access
// ---------------------------------------------------------------------------
int pic1dp_hip_chargeden_state(pic1dp_ctx *c, int32_t *kept_mode_only) {
  CHECK_CTX(c);
  if (!kept_mode_only) return fail(PIC1DP_ERR_ARG, "null output");
  *kept_mode_only = c->cd_kept_mode_only ? 1 : 0;
  return 0;
}

int pic1dp_hip_get_field(pic1dp_ctx *c, double *E, double *cd, double *re, double *im) {
  CHECK_CTX(c);
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = settle_field_view(c)) return rc;
  if (int rc = materialize_cd(c)) return rc;
  if (cd && c->cd_kept_mode_only)
    if (int rc = rebuild_half_step_chargeden(c)) return rc;
  // the four small vectors through the pinned staging, on the engine's stream, ONE wait (four synchronous copies from
  // pageable memory cost ~55 us: tools/output_host_cost.py)
  const size_t nx = c->in.nx, nm = c->in.nmode;
  double *h = nullptr;
  if (int rc = pinned(c, 2 * nx + 2 * nm, &h)) return rc;
  if (E) HIP_TRY(hipMemcpyAsync(h, c->d_E, sizeof(double) * nx, hipMemcpyDeviceToHost, c->st));
  if (cd) HIP_TRY(hipMemcpyAsync(h + nx, c->d_chargeden, sizeof(double) * nx, hipMemcpyDeviceToHost, c->st));
  if (re) HIP_TRY(hipMemcpyAsync(h + 2 * nx, c->d_mode_re, sizeof(double) * nm, hipMemcpyDeviceToHost, c->st));
  if (im) HIP_TRY(hipMemcpyAsync(h + 2 * nx + nm, c->d_mode_im, sizeof(double) * nm, hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipStreamSynchronize(c->st));
  if (int rc = xchg_check(c)) return rc;
  if (E) std::memcpy(E, h, sizeof(double) * nx);
  if (cd) std::memcpy(cd, h + nx, sizeof(double) * nx);
  if (re) std::memcpy(re, h + 2 * nx, sizeof(double) * nm);
  if (im) std::memcpy(im, h + 2 * nx + nm, sizeof(double) * nm);
  return 0;
}

int pic1dp_hip_set_electric(pic1dp_ctx *c, const double *E) {
  CHECK_CTX(c);
  if (!E) return fail(PIC1DP_ERR_ARG, "null array");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = settle_half_pair(c)) return rc;   // (d_E0 keeps the step-start field a noted push(1) saw)
  if (lz_of(c->seq) == LZ_PUSH1 || lz_of(c->seq) == LZ_PUSH2)
    if (int rc = materialize(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->st));
  HIP_TRY(hipMemcpy(c->d_E, E, sizeof(double) * c->in.nx, hipMemcpyHostToDevice));
  field_written(c, false);
  return 0;
}

int pic1dp_hip_set_chargeden(pic1dp_ctx *c, const double *cd) {
  CHECK_CTX(c);
  if (!cd) return fail(PIC1DP_ERR_ARG, "null array");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = materialize_cd(c)) return rc;  // pending deposits are consumed, then overwritten
  // a half-step field waiting to be adopted, or adopted as far as the host can tell but not yet copied (Seq::*PairSolved): memory
  // first becomes what the eager calls leave -- the adoption writes field_chargeden too --, then the host's vector rules
  if (int rc = settle_half_pair(c)) return rc;
  if (c->owed == Owed::AdoptHalfField)
    if (int rc = set_owed(c, Owed::Nothing)) return rc;
  HIP_TRY(hipStreamSynchronize(c->st));
  HIP_TRY(hipMemcpy(c->d_chargeden, cd, sizeof(double) * c->in.nx, hipMemcpyHostToDevice));
  c->cd_kept_mode_only = false;
  return 0;
}

int pic1dp_hip_field_energy(pic1dp_ctx *c, double *energy) {
  CHECK_CTX(c);
  if (!energy) return fail(PIC1DP_ERR_ARG, "null output");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = settle_field_view(c)) return rc;
  double *slot = c->d_scratch + kEnergyBlocks * 3;
  HIP_TRY(launch_field_energy(c->d_E, c->in.nx, c->in.lx, static_cast<double>(c->in.nx), slot, c->st));
  double *h = nullptr;
  if (int rc = pinned(c, 1, &h)) return rc;
  HIP_TRY(hipMemcpyAsync(h, slot, sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipStreamSynchronize(c->st));
  if (int rc = xchg_check(c)) return rc;
  *energy = h[0];
  return 0;
}

int pic1dp_hip_energy_history(pic1dp_ctx *c, double *energy, int64_t max, int64_t *count) {
  CHECK_CTX(c);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->st));
  if (int rc = xchg_check(c)) return rc;
  int64_t n = c->hist_count < max ? c->hist_count : max;
  if (n < 0) n = 0;
  if (energy && n > 0) HIP_TRY(hipMemcpy(energy, c->d_hist, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (count) *count = c->hist_count;
  return 0;
}

int pic1dp_hip_energy_history_reset(pic1dp_ctx *c) {
  CHECK_CTX(c);
  c->hist_count = 0;
  return 0;
}

int pic1dp_hip_cell_indices(pic1dp_ctx *c, int32_t isp, int32_t *ix, int64_t *count) {
  CHECK_CTX(c);
  if (isp < 0 || isp >= c->in.nspecies) return fail(PIC1DP_ERR_ARG, "bad species index");
  if (int rc = require_loaded(c)) return rc;
  Species &S = c->sp[isp];
  int32_t *d_ix = nullptr;
  unsigned long long *d_cnt = nullptr;
  if (ix) HIP_TRY(hipMalloc(&d_ix, sizeof(int32_t) * static_cast<size_t>(S.np + 1)));
  if (count) {
    HIP_TRY(hipMalloc(&d_cnt, sizeof(unsigned long long) * c->in.nx));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * c->in.nx, c->st));
  }
  hipError_t e = launch_cell_indices(S.set[c->cur].x, S.np, c->grid, d_ix, d_cnt, c->st);
  if (e == hipSuccess) e = hipStreamSynchronize(c->st);
  if (e == hipSuccess && ix) e = hipMemcpy(ix, d_ix, sizeof(int32_t) * S.np, hipMemcpyDeviceToHost);
  if (e == hipSuccess && count) e = hipMemcpy(count, d_cnt, sizeof(int64_t) * c->in.nx, hipMemcpyDeviceToHost);
  (void)hipFree(d_ix);
  (void)hipFree(d_cnt);
  HIP_TRY(e);
  return 0;
}

// ---------------------------------------------------------------------------
// timers and knobs
// ---------------------------------------------------------------------------
int pic1dp_hip_timers_enable(pic1dp_ctx *c, int32_t on) {
  CHECK_CTX(c);
  if (on < 0) return fail(PIC1DP_ERR_ARG, "timers_enable: 0 off, 1 every launch, n >= 2 every n-th launch of a timer");
  if (int rc = ev_resolve(c)) return rc;
  c->timers_on = on != 0;
  c->timer_every = on > 1 ? on : 1;
  return 0;
}

int pic1dp_hip_timer_ms(pic1dp_ctx *c, int32_t iwt, double *ms) {
  CHECK_CTX(c);
  if (iwt < 0 || iwt >= 100 || !ms) return fail(PIC1DP_ERR_ARG, "bad timer id");
  if (int rc = ev_resolve(c)) return rc;
  *ms = c->acc_ms[iwt];
  // sampled timers: the spans that were timed stand for the ones that were only counted -- apart from the first block of
  // the timer id (first launches, the run's first step), which is always timed and enters as it is
  const int64_t later = c->span_seen[iwt] - c->head_n[iwt];
  if (c->acc_n[iwt] > 0 && later > c->acc_n[iwt]) *ms *= static_cast<double>(later) / static_cast<double>(c->acc_n[iwt]);
  *ms += c->head_ms[iwt];
  // (no later block timed yet -- a short run, or a timer id with few launches: the first block's mean stands for them)
  if (c->acc_n[iwt] == 0 && c->head_n[iwt] > 0 && later > 0)
    *ms += c->head_ms[iwt] * static_cast<double>(later) / static_cast<double>(c->head_n[iwt]);
  return 0;
}

int pic1dp_hip_timers_reset(pic1dp_ctx *c) {
  CHECK_CTX(c);
  if (int rc = ev_resolve(c)) return rc;
  for (int i = 0; i < kNumTags; ++i) {
    c->acc_ms[i] = 0.0;
    c->acc_n[i] = 0;
    c->span_seen[i] = 0;
    c->head_ms[i] = 0.0;
    c->head_n[i] = 0;
  }
  return 0;
}

int pic1dp_hip_set_launch(pic1dp_ctx *c, int32_t threads, int32_t bpc) {
  CHECK_CTX(c);
  if (threads < 0 || threads > 1024 || (threads % 64)) return fail(PIC1DP_ERR_ARG, "threads must be a multiple of 64, <= 1024");
  if (bpc < 0 || bpc > 32) return fail(PIC1DP_ERR_ARG, "blocks_per_cu out of range");
  c->threads_req = threads;
  c->bpc_req = bpc;
  return 0;
}

int pic1dp_hip_get_stream(pic1dp_ctx *c, void **stream) {
  CHECK_CTX(c);
  if (!stream) return fail(PIC1DP_ERR_ARG, "null output");
  *stream = reinterpret_cast<void *>(c->st);
  return 0;
}

int pic1dp_hip_kernel_stats_enable(pic1dp_ctx *c, int32_t on) {
  CHECK_CTX(c);
  c->stats_on = on != 0;
  return 0;
}

int pic1dp_hip_kernel_stats(pic1dp_ctx *c, int32_t which, double *ms, int64_t *launches) {
  CHECK_CTX(c);
  if (which < 0 || which > 13) return fail(PIC1DP_ERR_ARG, "which must be 0..13");
  if (which == 13) {  // prediction tiles: terms that went past the fixed-point sums (beyond 16x the bound); *ms: the first species' bound on |q|
    HIP_TRY(hipStreamSynchronize(c->st));
    int64_t n = 0;
    double b0 = 0.0;
    for (size_t s = 0; s < c->sp.size(); ++s) {
      double h[3] = {0.0, 0.0, 0.0};
      HIP_TRY(hipMemcpy(h, c->sp[s].fxb, sizeof h, hipMemcpyDeviceToHost));
      int64_t k;
      std::memcpy(&k, &h[2], sizeof k);
      n += k;
      if (s == 0) b0 = h[0];
    }
    if (launches) *launches = n;
    if (ms) *ms = b0;
    return PIC1DP_OK;
  }
  if (which == 12) {  // diagnostics passes with 64-bit fixed-point histogram sums; *ms: of them, repeated in doubles (overflow)
    if (ms) *ms = static_cast<double>(c->diag_fx_repeats);
    if (launches) *launches = c->diag_fx_passes;
    return 0;
  }
  if (which == 11) {  // call sites: solve_field calls of a half step that launched nothing (the pair solve before them had it)
    if (ms) *ms = 0.0;
    if (launches) *launches = c->call_pair_skips;
    return 0;
  }
  if (which == 10) {  // marker launches whose last workgroup packed / posted this rank's charge (kernels.hpp StepTail)
    if (ms) *ms = 0.0;
    if (launches) *launches = c->tail_launches;
    return 0;
  }
  if (which == 9) {  // how the serial forward sums of the one-rank order run: 0 chains of additions, 1 the matrix unit
    if (ms) *ms = static_cast<double>(c->chain_selftest);  // create()'s self-test: 1 identical, 0 differs, -1 could not run
    if (launches) *launches = c->fa.chain_mfma;
    return 0;
  }
  if (which == 5 || which == 7 || which == 8) {  // counts: separate diagnostics passes (k_ptcldist), field solves inside
                                                 // marker launches, bytes marker optimisation events moved over PCIe
    if (ms) *ms = 0.0;
    if (launches) *launches = which == 5 ? c->diag_passes : (which == 7 ? c->fused_solves : c->opt_pcie_bytes);
    return 0;
  }
  if (int rc = ev_resolve(c)) return rc;
  const int tag = which == 6 ? kTagStepOne : kTagFused + which;
  if (ms) *ms = c->acc_ms[tag];
  if (launches) *launches = c->acc_n[tag];
  return 0;
}

int pic1dp_hip_kernel_bytes(pic1dp_ctx *c, int32_t which, double *read_bytes, double *written_bytes, double *carry_bytes,
                            char *name, int32_t name_len) {
  CHECK_CTX(c);
  if (which < 0 || which > 6 || which == 5) return fail(PIC1DP_ERR_ARG, "which must be 0..4 or 6");
  const pic1dp_ctx::KernelBytes &kb = c->kbytes[which == 6 ? kTagStepOne : kTagFused + which];
  if (read_bytes) *read_bytes = kb.rd;
  if (written_bytes) *written_bytes = kb.wr;
  if (carry_bytes) *carry_bytes = kb.carry;
  if (name && name_len > 0) std::snprintf(name, static_cast<size_t>(name_len), "%s", kb.name);
  return 0;
}

}  // extern "C"
