/*
 * pic1dp_oracle.c -- CPU restatement of the PIC1D-PETSc time-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pic1dp_oracle.h for the rules and the parity
 * status: RNG pinned by the reference module, push/deposit/solve unpinned).
 *
 * Every function cites the reference lines it restates.  Expressions keep the
 * reference's Fortran evaluation order (left to right for equal precedence,
 * "**" binds tighter than unary minus, integer operands are converted to
 * double where they meet a double).  Build: gcc -O2 -ffp-contract=off.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1 /* sincos */
#endif
#include "pic1dp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846264 /* PETSC_PI */
#define ORC_SQRT_EPS 1.490116119384766e-08 /* PETSC_SQRT_MACHINE_EPSILON */

/* ======================================================================
 * multirand -- src/multirand.F90
 * ====================================================================== */

static inline uint64_t xorshl(uint64_t x, int s) { return x ^ (x << s); }
static inline uint64_t xorshr(uint64_t x, int s) { return x ^ (x >> s); }

/* George Marsaglia's KISS64, src/multirand.F90:921-945 */
static uint64_t kiss64(orc_multirand *g) {
  uint64_t *q = g->seeds;
  uint64_t x = q[0], t = (x << 58) + q[3];
  if ((x >> 63) == (t >> 63))
    q[3] = (x >> 6) + (x >> 63);
  else
    q[3] = (x >> 6) - ((x + t) >> 63) + 1;
  q[0] = x + t;
  q[1] = xorshl(q[1], 13);
  q[1] = xorshr(q[1], 17);
  q[1] = xorshl(q[1], 43);
  q[2] = 6906969069ULL * q[2] + 1234567ULL;
  return q[0] + q[1] + q[2];
}

/* MT19937-64, src/multirand.F90:952-997 */
static uint64_t mt19937_64(orc_multirand *g) {
  enum { NN = 312, MM = 156 };
  const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL;
  const uint64_t mag01[2] = {0ULL, 0xB5026F5AA96619E9ULL};
  uint64_t *mt = g->seeds, x;
  if (g->iseed >= NN) {
    int i;
    for (i = 0; i < NN - MM; i++) {
      x = (mt[i] & UM) | (mt[i + 1] & LM);
      mt[i] = mt[i + MM] ^ (x >> 1) ^ mag01[x & 1ULL];
    }
    for (; i < NN - 1; i++) {
      x = (mt[i] & UM) | (mt[i + 1] & LM);
      mt[i] = mt[i + (MM - NN)] ^ (x >> 1) ^ mag01[x & 1ULL];
    }
    x = (mt[NN - 1] & UM) | (mt[0] & LM);
    mt[NN - 1] = mt[MM - 1] ^ (x >> 1) ^ mag01[x & 1ULL];
    g->iseed = 0;
  }
  x = mt[g->iseed++];
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}

/* SuperKISS64, src/multirand.F90:1004-1039 */
static uint64_t superkiss64(orc_multirand *g) {
  enum { NN = 20632, ICARRY = NN, IXCNG = NN + 1, IXS = NN + 2 };
  uint64_t *q = g->seeds;
  if (g->iseed >= NN) {
    uint64_t carry = q[ICARRY];
    for (int i = 0; i < NN; i++) {
      uint64_t h = carry & 1ULL;
      uint64_t z = ((q[i] << 41) >> 1) + ((q[i] << 39) >> 1) + (carry >> 1);
      carry = (q[i] >> 23) + (q[i] >> 25) + (z >> 63);
      q[i] = ~((z << 1) + h);
    }
    q[ICARRY] = carry;
    g->iseed = 0;
  }
  q[IXCNG] = q[IXCNG] * 6906969069ULL + 123ULL;
  q[IXS] = xorshl(q[IXS], 13);
  q[IXS] = xorshr(q[IXS], 17);
  q[IXS] = xorshl(q[IXS], 43);
  return q[g->iseed++] + q[IXCNG] + q[IXS];
}

int64_t orc_multirand_int64(orc_multirand *g) {
  uint64_t r;
  if (g->al_int == 2)
    r = mt19937_64(g);
  else if (g->al_int == 3)
    r = superkiss64(g);
  else
    r = kiss64(g);
  return (int64_t)r;
}

/* INT2REAL64, src/multirand.F90:49: i / (2**64-1 as double = 2**64) + 0.5 */
static inline double int2real64(int64_t i) {
  return (double)i / 18446744073709551615.0 + 0.5;
}

double orc_multirand_real64(orc_multirand *g) {
  return int2real64(orc_multirand_int64(g));
}

void orc_multirand_int_array64(orc_multirand *g, int64_t *a, int64_t n) {
  for (int64_t i = 0; i < n; i++) a[i] = orc_multirand_int64(g);
}

/* src/multirand.F90:664-707 (exclude0/exclude1 absent on the path) */
void orc_multirand_real_array64(orc_multirand *g, double *a, int64_t n) {
  for (int64_t i = 0; i < n; i++) a[i] = int2real64(orc_multirand_int64(g));
}

/* Marsaglia polar method, src/multirand.F90:838-872 */
void orc_multirand_gaussian_array64(orc_multirand *g, double *a, int64_t n) {
  const double max64 = 9223372036854775807.0;
  int64_t lo = 0;
  if (g->gaussian64buf_filled && n > 0) {
    a[lo++] = g->gaussian64buf;
    g->gaussian64buf_filled = 0;
  }
  for (int64_t i = lo; i < n; i += 2) {
    double x, y, w;
    do {
      x = (double)orc_multirand_int64(g) / max64;
      y = (double)orc_multirand_int64(g) / max64;
      w = x * x + y * y;
    } while (!(w > 0.0 && w < 1.0));
    w = sqrt((-2.0 * log(w)) / w);
    a[i] = x * w;
    if (i + 1 < n) {
      a[i + 1] = y * w;
    } else {
      g->gaussian64buf = y * w;
      g->gaussian64buf_filled = 1;
    }
  }
}

orc_multirand *orc_multirand_new(void) {
  return (orc_multirand *)calloc(1, sizeof(orc_multirand));
}
void orc_multirand_free(orc_multirand *g) { free(g); }

/* default seeds of the self-test, src/multirand.F90:475-518 */
void orc_multirand_default_seeds(orc_multirand *g, int al_int) {
  uint64_t *q = g->seeds;
  g->al_int = al_int;
  if (al_int == 2) {
    q[0] = 5489ULL;
    for (int i = 1; i < 312; i++)
      q[i] = 6364136223846793005ULL * xorshr(q[i - 1], 62) + (uint64_t)i;
    g->iseed = 312;
  } else if (al_int == 3) {
    q[20632] = 36243678541ULL;
    q[20633] = 12367890123456ULL;
    q[20634] = 521288629546311ULL;
    for (int i = 0; i < 20632; i++) {
      q[20633] = q[20633] * 6906969069ULL + 123ULL;
      q[20634] = xorshl(q[20634], 13);
      q[20634] = xorshr(q[20634], 17);
      q[20634] = xorshl(q[20634], 43);
      q[i] = q[20633] + q[20634];
    }
    g->iseed = 20632;
  } else {
    q[0] = 1234567890987654321ULL;
    q[1] = 362436362436362436ULL;
    q[2] = 1066149217761810ULL;
    q[3] = 123456123456123456ULL;
  }
}

/* KAT vectors transcribed (data) from src/multirand.F90:396-425 */
static const int64_t kat_kiss64[10] = {
    8932985056925012148LL,  5710300428094272059LL,  -104233206776033023LL,
    -4143107803135683366LL, 542381058189297533LL,   -4244931820854714191LL,
    6853720724624422285LL,  -767542866500872268LL,  -257204313086867125LL,
    8128797625455304420LL};
static const int64_t kat_mt_head[10] = {
    -3932459287431434586LL, 4620546740167642908LL, -5337173792191653896LL,
    -983805426561117294LL,  355488278567739596LL,  7469126240319926998LL,
    4635995468481642529LL,  418970542659199878LL,  -8842573084457035060LL,
    6358044926049913402LL};
static const int64_t kat_mt_tail[10] = {
    -7948593974297132281LL, 1921007855220546564LL, 7643484074408755248LL,
    -7128315020423208677LL, 1370093900783164344LL, 6776537281339823025LL,
    3450492372588984223LL,  -9045729527952115285LL, 7896519943553875907LL,
    -4143300141377237606LL};
static const int64_t kat_sk_head[10] = {
    6140839658375754198LL,  -95225469143006167LL,  -9148462456964506707LL,
    3912874252778582253LL,  6801212277726928591LL, -809575511391043410LL,
    -397286769868273005LL,  4963780769400405858LL, 2406624640673457322LL,
    1246843699883922102LL};
static const int64_t kat_sk_tail[10] = {
    -1387224431860786161LL, -8846516422183390713LL, 8111357788999165247LL,
    444070776306226770LL,   -7730678117654887867LL, -296399128303442035LL,
    -1658509282659454084LL, -8190332265239255687LL, -1492517620356299342LL,
    -5016179395587873849LL};

/* src/multirand.F90:390-553; returns 0 when every check passes */
int orc_multirand_selftest(orc_multirand *g, int al_int) {
  const int ntest = 10;
  const int64_t *head, *tail = NULL;
  int itail = 0, bad = 0;
  /* boundary checks :441-471 */
  int64_t k = INT64_MAX;
  if (int2real64(k) != 1.0) bad = 1;
  k = INT64_MIN;
  if (int2real64(k) != 0.0) bad = 1;
  orc_multirand_default_seeds(g, al_int);
  if (al_int == 2) {
    head = kat_mt_head;
    tail = kat_mt_tail;
    itail = 312 - ntest / 2;
  } else if (al_int == 3) {
    head = kat_sk_head;
    tail = kat_sk_tail;
    itail = 20632 - ntest / 2;
  } else {
    head = kat_kiss64;
  }
  for (int i = 0; i < ntest; i++)
    if (orc_multirand_int64(g) != head[i]) return 1;
  if (tail) {
    for (int i = ntest + 1; i <= itail; i++) (void)orc_multirand_int64(g);
    for (int i = 0; i < ntest; i++)
      if (orc_multirand_int64(g) != tail[i]) return 1;
  }
  return bad;
}

static const int64_t primes1[100] = {
    15484219, 15484223, 15484243, 15484247, 15484279, 15484333, 15484363,
    15484387, 15484393, 15484409, 15484421, 15484453, 15484457, 15484459,
    15484471, 15484489, 15484517, 15484519, 15484549, 15484559, 15484591,
    15484627, 15484631, 15484643, 15484661, 15484697, 15484709, 15484723,
    15484769, 15484771, 15484783, 15484817, 15484823, 15484873, 15484877,
    15484879, 15484901, 15484919, 15484939, 15484951, 15484961, 15484999,
    15485039, 15485053, 15485059, 15485077, 15485083, 15485143, 15485161,
    15485179, 15485191, 15485221, 15485243, 15485251, 15485257, 15485273,
    15485287, 15485291, 15485293, 15485299, 15485311, 15485321, 15485339,
    15485341, 15485357, 15485363, 15485383, 15485389, 15485401, 15485411,
    15485429, 15485441, 15485447, 15485471, 15485473, 15485497, 15485537,
    15485539, 15485543, 15485549, 15485557, 15485567, 15485581, 15485609,
    15485611, 15485621, 15485651, 15485653, 15485669, 15485677, 15485689,
    15485711, 15485737, 15485747, 15485761, 15485773, 15485783, 15485801,
    15485807, 15485837};
static const int64_t primes2[100] = {
    7001, 7013, 7019, 7027, 7039, 7043, 7057, 7069, 7079, 7103, 7109, 7121,
    7127, 7129, 7151, 7159, 7177, 7187, 7193, 7207, 7211, 7213, 7219, 7229,
    7237, 7243, 7247, 7253, 7283, 7297, 7307, 7309, 7321, 7331, 7333, 7349,
    7351, 7369, 7393, 7411, 7417, 7433, 7451, 7457, 7459, 7477, 7481, 7487,
    7489, 7499, 7507, 7517, 7523, 7529, 7537, 7541, 7547, 7549, 7559, 7561,
    7573, 7577, 7583, 7589, 7591, 7603, 7607, 7621, 7639, 7643, 7649, 7669,
    7673, 7681, 7687, 7691, 7699, 7703, 7717, 7723, 7727, 7741, 7753, 7757,
    7759, 7789, 7793, 7817, 7823, 7829, 7841, 7853, 7867, 7873, 7877, 7879,
    7883, 7901, 7907, 7919};

static int64_t iabs64(int64_t a) { return a < 0 ? -a : a; }

/* src/multirand.F90:132-383 */
int orc_multirand_init(orc_multirand *g, int al_int, int seed_type, int mype,
                       int warmup, int selftest) {
  int rc = 0;
  int64_t nseed;
  if (al_int != 2 && al_int != 3) al_int = 1;
  nseed = al_int == 2 ? 312 : (al_int == 3 ? 20635 : 4);
  g->al_int = al_int;
  if (selftest) rc = orc_multirand_selftest(g, al_int);
  g->al_int = al_int;

  if (seed_type == 3) {
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f) {
      seed_type = 2;
    } else {
      if (fread(g->seeds, 8, (size_t)nseed, f) != (size_t)nseed) rc = 3;
      if (al_int == 1) {
        while (g->seeds[1] == 0)
          if (fread(&g->seeds[1], 8, 1, f) != 1) break;
        while (g->seeds[0] == 0 && g->seeds[3] == 0) {
          if (fread(&g->seeds[0], 8, 1, f) != 1) break;
          if (fread(&g->seeds[3], 8, 1, f) != 1) break;
        }
      } else if (al_int == 3) {
        while (g->seeds[20634] == 0)
          if (fread(&g->seeds[20634], 8, 1, f) != 1) break;
      }
      fclose(f);
    }
  }
  if (seed_type != 3) {
    int64_t clock, add;
    uint64_t *tmp;
    if (al_int == 3 && g->seeds[20634] == 0)
      return 2; /* the reference loops forever at :346-348 */
    if (seed_type == 2) {
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      clock = (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
    } else {
      clock = primes1[1];
    }
    /* :307-320 -- KISS seeds from clock and mype */
    add = primes1[iabs64(clock + primes2[iabs64(clock) % 100] * mype) % 100] *
          mype;
    for (int i = 0; i < 4; i++) g->seeds[i] = (uint64_t)(clock + add);
    for (int64_t i = 0; i < 4; i++) {
      int64_t s = (int64_t)g->seeds[i];
      s += primes2[iabs64(s + primes1[iabs64(clock) % 100] * i) % 100] * i;
      g->seeds[i] = (uint64_t)s;
    }
    /* :322-350 -- KISS randomises the seed array */
    tmp = (uint64_t *)malloc(sizeof(uint64_t) * ORC_MULTIRAND_NSEED);
    memcpy(tmp, g->seeds, sizeof(uint64_t) * ORC_MULTIRAND_NSEED);
    for (int i = 0; i < 20; i++) tmp[0] = kiss64(g);
    for (int64_t i = 1; i < nseed; i++) tmp[i] = kiss64(g);
    if (al_int == 1) {
      while (tmp[1] == 0) tmp[1] = kiss64(g);
      while (tmp[0] == 0 && tmp[3] == 0) {
        tmp[0] = kiss64(g);
        tmp[3] = kiss64(g);
      }
    }
    memcpy(g->seeds, tmp, sizeof(uint64_t) * (size_t)nseed);
    free(tmp);
  }
  if (al_int == 2)
    g->iseed = 312;
  else if (al_int == 3)
    g->iseed = 20632;
  for (int64_t i = 0; i < (int64_t)warmup * nseed; i++)
    (void)orc_multirand_int64(g);
  return rc;
}

/* ======================================================================
 * ownership -- PETSC_DECIDE split: n/size + (rank < n%size)
 * (PetscSplitOwnership; used at src/pic1dp_particle.F90:89-94,129)
 * ====================================================================== */
int64_t orc_local_size(int64_t n, int rank, int size) {
  return n / size + ((n % size) > rank ? 1 : 0);
}

/* src/pic1dp_particle.F90:240-248 */
int64_t orc_particle_np(const orc_input *in, int isp, int mype, int npe) {
  int64_t nlocal = orc_local_size(in->nparticle_max, mype, npe);
  int64_t spare = in->nparticle_max - in->species_nparticle_init[isp];
  int64_t unload = spare / npe;
  if (mype == 0) unload += spare % npe;
  return nlocal - unload;
}

/* ======================================================================
 * particle_load -- src/pic1dp_particle.F90:162-265 (one species)
 * ====================================================================== */
void orc_particle_load_species(const orc_input *in, int isp, orc_multirand *g,
                               int64_t n, double *x, double *v, double *p,
                               double *w) {
  const double T = in->species_temperature[isp];
  const double T2 = in->species_temperature2[isp];
  const double m = in->species_mass[isp];
  const double den = in->species_density[isp];
  const double v0 = in->species_v0[isp];
  const double ninit = (double)in->species_nparticle_init[isp];
  const double lx = in->lx, vmax = in->v_max;

  if (in->imarker == 1) { /* :172-178 */
    const double sig = sqrt(T / m);
    const double pc = den * lx / ninit;
    orc_multirand_gaussian_array64(g, v, n);
    for (int64_t i = 0; i < n; i++) {
      v[i] = v[i] * sig + v0;
      p[i] = pc;
    }
  } else { /* :179-218 */
    orc_multirand_real_array64(g, v, n);
    for (int64_t i = 0; i < n; i++) v[i] = (v[i] - 0.5) * 2.0 * vmax;
    if (in->iptcldist == 1) { /* :183-186 */
      const double K = den * lx * 2.0 * vmax / ninit;
      const double s2pi = sqrt(2.0 * ORC_PI);
      for (int64_t i = 0; i < n; i++) {
        double v2 = v[i] * v[i];
        p[i] = K * v2 * exp(-v2 / 2.0) / s2pi;
      }
    } else if (in->iptcldist == 2) { /* :188-196 */
      const double K = den * lx * 2.0 * vmax / ninit;
      const double two_tm = 2.0 * T / m;
      const double nrm = sqrt(8.0 * ORC_PI * T / m);
      for (int64_t i = 0; i < n; i++) {
        double vp = v[i] + v0, vm = v[i] - v0;
        p[i] = K * (exp(-(vp * vp) / two_tm) + exp(-(vm * vm) / two_tm)) / nrm;
      }
    } else if (in->iptcldist == 3) { /* :198-209 */
      const double K = 1.0 * lx * 2.0 * vmax / ninit;
      const double two_tm = 2.0 * T / m, two_tm2 = 2.0 * T2 / m;
      const double nrm = sqrt(2.0 * ORC_PI * T / m);
      const double nrm2 = sqrt(2.0 * ORC_PI * T2 / m);
      const double beam = 1.0 - den;
      for (int64_t i = 0; i < n; i++) {
        double vm = v[i] - v0;
        p[i] = K * (den * exp(-(v[i] * v[i]) / two_tm) / nrm +
                    beam * exp(-(vm * vm) / two_tm2) / nrm2);
      }
    } else { /* :211-217 */
      const double K = den * lx * 2.0 * vmax / ninit;
      const double two_tm = 2.0 * T / m;
      const double nrm = sqrt(2.0 * ORC_PI * T / m);
      for (int64_t i = 0; i < n; i++) {
        double vm = v[i] - v0;
        p[i] = K * exp(-(vm * vm) / two_tm) / nrm;
      }
    }
  }
  /* :222-223 uniform in x */
  orc_multirand_real_array64(g, x, n);
  for (int64_t i = 0; i < n; i++) x[i] = x[i] * lx;
  /* :225-232 initial mode perturbation */
  for (int64_t i = 0; i < n; i++) w[i] = 0.0;
  for (int im = 0; im < in->init_nmode; im++) {
    const double k = 2.0 * ORC_PI / lx * (double)in->init_mode[im];
    const double ac = in->init_mode_cos[im], as = in->init_mode_sin[im];
    for (int64_t i = 0; i < n; i++) {
      /* GCC's middle end (hence gfortran -O3, the reference build) turns the
       * cos/sin pair of one argument into a single sincos call; glibc's sincos
       * and sin()/cos() differ in the last bit for ~0.05% of arguments, so the
       * call is made explicit here instead of being left to the optimiser */
      double sn, cs;
      sincos(k * x[i], &sn, &cs);
      w[i] = w[i] + ac * cs + as * sn;
    }
  }
  /* :234-237, input_pertb_shape == 1.0 (src/pic1dp_input.F90:271) */
  for (int64_t i = 0; i < n; i++) w[i] = w[i] * p[i] * 1.0;
  /* :260-264 nonlinear: p = f/g = f0/g + delta f/g */
  if (in->linear == 0)
    for (int64_t i = 0; i < n; i++) p[i] = p[i] + w[i];
}

/* ======================================================================
 * interaction_collect_charge, iptclshape == 4
 * ====================================================================== */

/* src/pic1dp_interaction.F90:96-114 */
void orc_deposit_species_idx(const orc_input *in, int64_t np, double *x,
                             const double *q, double *charge1, int32_t *ix_out,
                             int64_t *count) {
  const double lx = in->lx;
  const int32_t nx = in->nx;
  const double dnx = (double)nx;
  for (int64_t ip = 0; ip < np; ip++) {
    double px = fmod(x[ip], lx);  /* :102 */
    if (px < 0.0) px = px + lx;   /* :104 */
    x[ip] = px;
    double sx = px / lx * dnx;    /* :106 */
    int32_t ix = (int32_t)floor(sx);
    sx = 1.0 - (sx - (double)ix); /* :108 */
    /* memory safety only: px + lx can round to lx; the
     * reference would write out of bounds, fold to cell 0 instead */
    if (ix >= nx) ix = 0;
    if (ix_out) ix_out[ip] = ix;
    if (count) count[ix]++;
    charge1[ix] = charge1[ix] + sx * q[ip]; /* :110 */
    ix = ix + 1;
    if (ix > nx - 1) ix = 0;
    charge1[ix] = charge1[ix] + (1.0 - sx) * q[ip]; /* :113 */
  }
}

void orc_deposit_species(const orc_input *in, int64_t np, double *x,
                         const double *q, double *charge1) {
  orc_deposit_species_idx(in, np, x, q, charge1, NULL, NULL);
}

/* libm exp() element by element: the transcendental of the weight equation
 * (src/pic1dp_interaction.F90:278-321) as the reference's compiler links it; lets a
 * test bound the device's exp against it */
void orc_exp_array(const double *x, double *y, int64_t n) {
  for (int64_t i = 0; i < n; i++) y[i] = exp(x[i]);
}

/* src/pic1dp_interaction.F90:138-148 */
void orc_chargeden_from_charge(const orc_input *in, const double *charge1,
                               double *chargeden) {
  const double dnx = (double)in->nx;
  for (int ix = 0; ix < in->nx; ix++) {
    double c = charge1[ix] * dnx / in->lx;
    if (in->deltaf == 0)
      for (int s = 0; s < in->nspecies; s++)
        c = c - in->species_charge[s] * in->species_density[s];
    chargeden[ix] = c;
  }
}

/* ======================================================================
 * interaction_push_particle
 * ====================================================================== */

/* src/pic1dp_interaction.F90:178-189 (VecCopy over the whole local vector) */
void orc_push_backup(int64_t nalloc, const double *x, const double *v,
                     const double *w, double *xb, double *vb, double *wb,
                     int deltaf) {
  memcpy(xb, x, sizeof(double) * (size_t)nalloc);
  memcpy(vb, v, sizeof(double) * (size_t)nalloc);
  if (deltaf == 1) memcpy(wb, w, sizeof(double) * (size_t)nalloc);
}

/* -d f0/dv / f0, src/pic1dp_interaction.F90:274-326 */
static inline double dlnf0(const orc_input *in, int isp, double v) {
  const double T = in->species_temperature[isp];
  const double T2 = in->species_temperature2[isp];
  const double m = in->species_mass[isp];
  const double den = in->species_density[isp];
  const double v0 = in->species_v0[isp];
  if (in->iptcldist == 1) { /* :276 */
    return v - 2.0 / v;
  } else if (in->iptcldist == 2) { /* :278-292 */
    const double two_tm = 2.0 * T / m;
    double vp = v + v0, vm = v - v0;
    double ep = exp(-(vp * vp) / two_tm), em = exp(-(vm * vm) / two_tm);
    return (vp * ep + vm * em) / (ep + em) * m / T;
  } else if (in->iptcldist == 3) { /* :294-321 */
    const double tm = T / m, tm2 = T2 / m;
    const double two_tm = 2.0 * T / m, two_tm2 = 2.0 * T2 / m;
    const double stm = sqrt(tm), stm2 = sqrt(tm2);
    const double beam = 1.0 - den;
    double vm = v - v0;
    double e1 = exp(-(v * v) / two_tm), e2 = exp(-(vm * vm) / two_tm2);
    double num = den * v / tm * e1 / stm + beam * vm / tm2 * e2 / stm2;
    double dnm = den * e1 / stm + beam * e2 / stm2;
    return num / dnm;
  }
  return (v - v0) / (T / m); /* :323-325 */
}

/* src/pic1dp_interaction.F90:238-339 */
void orc_push_species(const orc_input *in, int isp, int irk, const double *E,
                      int64_t np, double *x, double *v, const double *p,
                      double *w, const double *xb, const double *vb,
                      const double *wb) {
  const double dt = irk == 1 ? 0.5 * in->dt : in->dt; /* :179,192 */
  const double lx = in->lx;
  const int32_t nx = in->nx;
  const double dnx = (double)nx;
  const double Z = in->species_charge[isp], m = in->species_mass[isp];
  for (int64_t ip = 0; ip < np; ip++) {
    double sx = x[ip] / lx * dnx; /* :250 */
    int32_t ix = (int32_t)floor(sx);
    sx = 1.0 - (sx - (double)ix); /* :252 */
    if (ix >= nx) ix = 0;         /* memory safety, see deposit */
    double e = E[ix] * sx;        /* :254 */
    ix = ix + 1;
    if (ix > nx - 1) ix = 0;
    e = e + E[ix] * (1.0 - sx); /* :257 */
    double pv = v[ip];
    x[ip] = xb[ip] + dt * pv; /* :261 */
    if (in->deltaf == 1) {
      double tmp1 = in->linear == 1 ? p[ip] * e : (p[ip] - w[ip]) * e;
      double tmp2 = dlnf0(in, isp, pv);
      w[ip] = wb[ip] + dt * tmp1 * tmp2 * Z / m; /* :329 */
    }
    if (in->linear == 0) v[ip] = vb[ip] + dt * e * Z / m; /* :336 */
  }
}

/* ======================================================================
 * field -- src/pic1dp_field.F90
 * ====================================================================== */

/* cos / sin that the optimiser cannot pair into sincos */
static double (*volatile orc_cos_ptr)(double) = cos;
static double (*volatile orc_sin_ptr)(double) = sin;
static double orc_cos_alone(double x) { return orc_cos_ptr(x); }
static double orc_sin_alone(double x) { return orc_sin_ptr(x); }

/* operators of field_init, :158-210 */
orc_field *orc_field_new(const orc_input *in) {
  orc_field *f = (orc_field *)calloc(1, sizeof(orc_field));
  const int nx = in->nx, nm = in->nmode;
  f->nx = nx;
  f->nmode = nm;
  f->fourier_re = (double *)malloc(sizeof(double) * (size_t)nx * nm);
  f->fourier_im = (double *)malloc(sizeof(double) * (size_t)nx * nm);
  f->grad_inv = (double *)malloc(sizeof(double) * (size_t)nm);
  for (int im = 0; im < nm; im++) /* :165-166 */
    f->grad_inv[im] = 1.0 / (2.0 * ORC_PI / in->lx * (double)in->modes[im]);
  /* the reference fills the two matrices in two separate loops (:186-189 and
   * :194-197): plain cos() and plain sin(), never merged into sincos */
  for (int ix = 0; ix < nx; ix++)
    for (int im = 0; im < nm; im++) {
      double th = 2.0 * ORC_PI / (double)nx * (double)in->modes[im] * (double)ix;
      f->fourier_re[(size_t)ix * nm + im] = orc_cos_alone(th);
    }
  for (int ix = 0; ix < nx; ix++)
    for (int im = 0; im < nm; im++) {
      double th = 2.0 * ORC_PI / (double)nx * (double)in->modes[im] * (double)ix;
      f->fourier_im[(size_t)ix * nm + im] = -orc_sin_alone(th);
    }
  return f;
}

void orc_field_free(orc_field *f) {
  if (!f) return;
  free(f->fourier_re);
  free(f->fourier_im);
  free(f->grad_inv);
  free(f);
}

/* field_solve_electric, :231-257, in PETSc's summation order.
 * One rank (SeqAIJ): MatMultTranspose accumulates y[col] += a(row,col)*x[row] over
 * ascending rows; MatMult / MatMultAdd sum a row's entries in ascending column
 * order; VecScale multiplies by the scalar (here a pre-formed reciprocal).
 * npe ranks (MPI-AIJ; the operators are created with PETSC_DECIDE row blocks,
 * src/pic1dp_global.F90:96-133, and the reference is run as mpiexec -n 4,
 * run/Makefile:41): MatMultTranspose_MPIAIJ forms every rank's contribution from
 * its own rows -- ascending, from zero, exactly the loop above restricted to the
 * rank's block of n/npe + (rank < n%npe) rows -- and a reverse VecScatter with
 * ADD_VALUES adds the contributions into the owner's entry, which already holds
 * the owner's own.  The owner of mode entry m follows from the PETSC_DECIDE split
 * of the nmode entries (src/pic1dp_field.F90:86-88); the order in which the
 * received contributions are added is the scatter's (message completion): stated
 * here as the canonical one -- the owner's own block first, then the other ranks
 * in rank order, which for mode entry 0 (the one kept mode of the default input,
 * owned by rank 0) is plain rank order.  The inverse (MatMult, then MatMultAdd onto its
 * result) follows MPI-AIJ's two products per row: the columns of the row's own rank
 * (its diagonal block: the PETSC_DECIDE block of the nmode entries that rank holds)
 * ascending, then every other column ascending; for one rank, and for nmode <= 2, that
 * is the one-rank sum. */
void orc_field_solve_ranks(const orc_input *in, const orc_field *f, int npe,
                           const double *rho, double *E, double *mode_re,
                           double *mode_im) {
  const int nx = f->nx, nm = f->nmode;
  const double dnx = (double)in->nx;
  const double sc_im = -1.0 / dnx, sc_re = 1.0 / dnx;
  if (npe < 1) npe = 1;
  for (int im = 0; im < nm; im++) {
    int owner = 0; /* PETSC_DECIDE split of the nmode entries (src/pic1dp_field.F90:86-88): whose block holds im */
    for (int64_t first = 0; owner < npe - 1; owner++) {
      first += orc_local_size(nm, owner, npe);
      if (im < first) break;
    }
    double tot_im = 0.0, tot_re = 0.0;
    for (int k = 0; k < npe; k++) {
      /* the owner's own rows first, then ranks 0, 1, ... (skipping the owner) */
      const int r = k == 0 ? owner : (k <= owner ? k - 1 : k);
      int64_t lo = 0;
      for (int q = 0; q < r; q++) lo += orc_local_size(nx, q, npe);
      const int64_t hi = lo + orc_local_size(nx, r, npe);
      double p_im = 0.0, p_re = 0.0;
      for (int64_t ix = lo; ix < hi; ix++) {
        p_im += f->fourier_re[(size_t)ix * nm + im] * rho[ix]; /* :231 */
        p_re += f->fourier_im[(size_t)ix * nm + im] * rho[ix]; /* :236 */
      }
      if (k == 0) {
        tot_im = p_im;
        tot_re = p_re;
      } else {
        tot_im += p_im;
        tot_re += p_re;
      }
    }
    mode_im[im] = tot_im;
    mode_re[im] = tot_re;
  }
  for (int im = 0; im < nm; im++) {
    mode_im[im] = mode_im[im] * sc_im;          /* :234 */
    mode_re[im] = mode_re[im] * sc_re;          /* :239 */
    mode_re[im] = mode_re[im] * f->grad_inv[im]; /* :243 */
    mode_im[im] = mode_im[im] * f->grad_inv[im]; /* :246 */
  }
  for (int r = 0, row = 0; r < npe; r++) {
    /* rank r's rows [row, row_end) and the mode entries it owns [own, own_end) */
    const int row_end = row + (int)orc_local_size(nx, r, npe);
    int own = 0;
    for (int q = 0; q < r; q++) own += (int)orc_local_size(nm, q, npe);
    const int own_end = own + (int)orc_local_size(nm, r, npe);
    for (int ix = row; ix < row_end; ix++) {
      double s = 0.0;
      for (int im = own; im < own_end; im++) /* :251, diagonal block */
        s += f->fourier_re[(size_t)ix * nm + im] * mode_re[im];
      for (int im = 0; im < nm; im++)        /* :251, the other ranks' columns */
        if (im < own || im >= own_end) s += f->fourier_re[(size_t)ix * nm + im] * mode_re[im];
      for (int im = own; im < own_end; im++) /* :253 */
        s += f->fourier_im[(size_t)ix * nm + im] * mode_im[im];
      for (int im = 0; im < nm; im++)
        if (im < own || im >= own_end) s += f->fourier_im[(size_t)ix * nm + im] * mode_im[im];
      E[ix] = s * 2.0; /* :256 */
    }
    row = row_end;
  }
}

/* the one-rank run (SeqAIJ): one block, ascending rows */
void orc_field_solve(const orc_input *in, const orc_field *f,
                     const double *rho, double *E, double *mode_re,
                     double *mode_im) {
  orc_field_solve_ranks(in, f, 1, rho, E, mode_re, mode_im);
}

/* CPU statement of the engine's OPT-IN finite-difference solver (NOT part of
 * the reference; checks pic1dp_hip_set_field_solver(1) only):
 * (phi[i-1] - 2 phi[i] + phi[i+1])/h^2 = -(rho[i] - <rho>), phi[0] = 0, periodic;
 * E[i] = -(phi[i+1] - phi[i-1])/(2h).  Thomas algorithm on the nx-1 unknowns. */
void orc_field_solve_fd(const orc_input *in, const double *rho, double *E) {
  const int nx = in->nx, n = nx - 1;
  const double h = in->lx / (double)nx;
  double mean = 0.0;
  for (int i = 0; i < nx; i++) mean += rho[i];
  mean /= (double)nx;
  double *cp = (double *)malloc(sizeof(double) * (size_t)nx);
  double *dp = (double *)malloc(sizeof(double) * (size_t)nx);
  double *phi = (double *)calloc((size_t)nx + 1, sizeof(double));
  /* rows j = 0..n-1 for phi[j+1]: -phi[j] + 2 phi[j+1] - phi[j+2] = h^2 (rho[j+1]-mean) */
  for (int j = 0; j < n; j++) {
    double a = j == 0 ? 0.0 : -1.0, b = 2.0, c = j == n - 1 ? 0.0 : -1.0;
    double d = h * h * (rho[j + 1] - mean);
    double den = j == 0 ? b : b - a * cp[j - 1];
    cp[j] = c / den;
    dp[j] = j == 0 ? d / den : (d - a * dp[j - 1]) / den;
  }
  for (int j = n - 1; j >= 0; j--)
    phi[j + 1] = j == n - 1 ? dp[j] : dp[j] - cp[j] * phi[j + 2];
  phi[0] = 0.0;
  for (int i = 0; i < nx; i++) {
    int ip = i + 1 == nx ? 0 : i + 1, im = i == 0 ? nx - 1 : i - 1;
    E[i] = -(phi[ip] - phi[im]) / (2.0 * h);
  }
  free(cp);
  free(dp);
  free(phi);
}

/* src/pic1dp_output.F90:120-124 (VecNorm NORM_2, then squared) */
double orc_field_energy(const orc_input *in, const double *E) {
  double s = 0.0;
  for (int ix = 0; ix < in->nx; ix++) s += E[ix] * E[ix];
  double nrm = sqrt(s);
  return nrm * nrm * in->lx / (double)in->nx;
}

/* src/pic1dp_output.F90:126-172 for deltaf==1 (sums run over the whole
 * local vector, not just particle_np) */
void orc_energy_sums(int64_t nalloc, const double *v, const double *p,
                     const double *w, int deltaf, double out[3]) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t i = 0; i < nalloc; i++) {
    double v2 = v[i] * v[i];
    s0 += v2;
    s1 += v2 * p[i];
    if (deltaf == 1) s2 += v2 * w[i];
  }
  out[0] = s0;
  out[1] = s1;
  out[2] = deltaf == 1 ? s2 : s1;
}

/* src/pic1dp_output.F90:239-315: raw (unscaled, unreduced) histograms */
void orc_ptcldist(const orc_input *in, int64_t np, const double *x,
                  const double *v, const double *p, const double *w,
                  double *markr_xv, double *total_xv, double *pertb_xv,
                  double *markr_v, double *total_v, double *pertb_v) {
  const int nxo = in->nx_opd, nvo = in->nv_opd;
  const double lx = in->lx, vmax = in->v_max;
  for (int64_t ip = 0; ip < np; ip++) {
    if (fabs(v[ip]) >= vmax) continue; /* :241 */
    double sx = x[ip] / lx * (double)nxo;
    int ix = (int)floor(sx);
    sx = 1.0 - (sx - (double)ix);
    double sv = (v[ip] + vmax) / (vmax * 2.0) * (double)(nvo - 1);
    int iv = (int)floor(sv);
    sv = 1.0 - (sv - (double)iv);
    if (ix < 0 || ix >= nxo || iv < 0 || iv + 1 >= nvo) continue; /* safety */
    double pp = p[ip], pw = w ? w[ip] : 0.0;
    for (int pass = 0; pass < 2; pass++) {
      int a = iv * nxo + ix, b = (iv + 1) * nxo + ix;
      markr_xv[a] += sx * sv;
      total_xv[a] += sx * sv * pp;
      if (in->deltaf == 1) pertb_xv[a] += sx * sv * pw;
      markr_xv[b] += sx * (1.0 - sv);
      total_xv[b] += sx * (1.0 - sv) * pp;
      if (in->deltaf == 1) pertb_xv[b] += sx * (1.0 - sv) * pw;
      ix = ix + 1; /* :274-276 */
      if (ix > nxo - 1) ix = 0;
      sx = 1.0 - sx;
    }
    markr_v[iv] += sv;
    total_v[iv] += sv * pp;
    if (in->deltaf == 1) pertb_v[iv] += sv * pw;
    markr_v[iv + 1] += (1.0 - sv);
    total_v[iv + 1] += (1.0 - sv) * pp;
    if (in->deltaf == 1) pertb_v[iv + 1] += (1.0 - sv) * pw;
  }
}

/* ======================================================================
 * marker optimisation -- src/pic1dp_particle.F90:356-813 (one rank)
 * Indices are kept 1-based like the reference (arrays are passed 0-based, so
 * element ip lives at [ip-1]); stale bin references after a swap-with-last are
 * reproduced, not repaired.
 * ====================================================================== */

/* :372-388 */
void orc_dist_pertb_abs_v(const orc_input *in, int64_t np, const double *v,
                          const double *w, double *hist) {
  const double vmax = in->v_max;
  const int nv = in->nv;
  for (int64_t ip = 0; ip < np; ip++) {
    if (fabs(v[ip]) >= vmax) continue; /* :375 */
    double sv = (v[ip] + vmax) / (vmax * 2.0) * (double)(nv - 1);
    int iv = (int)floor(sv);
    sv = 1.0 - (sv - (double)iv);
    hist[iv] = hist[iv] + sv * fabs(w[ip]);
    hist[iv + 1] = hist[iv + 1] + (1.0 - sv) * fabs(w[ip]);
  }
}

static double hist_max(const orc_input *in, const double *hist) {
  double m = hist[0];
  for (int i = 1; i < in->nv; i++)
    if (hist[i] > m) m = hist[i];
  return m;
}

/* |delta f| interpolated at v, and the v bin: :449-463 (same in remove, split) */
static double df_at(const orc_input *in, const double *hist, double v, int *iv_out) {
  const int nv = in->nv;
  double sv = (v + in->v_max) / (in->v_max * 2.0) * (double)(nv - 1);
  int iv = (int)floor(sv);
  double df;
  if (iv < 0) {
    iv = 0;
    df = hist[iv];
  } else if (iv >= nv - 1) {
    iv = nv - 1;
    df = hist[iv];
  } else {
    sv = 1.0 - (sv - (double)iv);
    df = hist[iv] * sv + hist[iv + 1] * (1.0 - sv);
  }
  *iv_out = iv;
  return df;
}

/* particle_merge, :411-519 */
void orc_particle_merge(const orc_input *in, double thsh, const double *hist,
                        int64_t *np_io, double *x, double *v, double *p, double *w) {
  const int nx = in->nx, nv = in->nv;
  const double lx = in->lx;
  int64_t np = *np_io;
  /* ipbin(ix, iv, iw, 1) and ipbin_top(ix, iv, iw): one slot per bin */
  int64_t *bin = (int64_t *)calloc((size_t)nx * nv * 2, sizeof(int64_t));
  char *full = (char *)calloc((size_t)nx * nv * 2, 1);
  const double df_thsh = hist_max(in, hist) * thsh;
  int64_t ip = 0;
  for (;;) {
    ip = ip + 1;
    if (ip > np) break;
    int iv;
    double df = df_at(in, hist, v[ip - 1], &iv);
    if (df >= df_thsh) continue; /* :466 */
    double px = fmod(x[ip - 1], lx); /* :469-471 */
    if (px < 0.0) px = px + lx;
    x[ip - 1] = px;
    double sx = px / lx * (double)nx;
    int ix = (int)floor(sx);
    if (ix >= nx) ix = nx - 1; /* memory safety only (px == lx) */
    const int iw = w[ip - 1] > 0.0 ? 1 : 0;
    const size_t b = ((size_t)ix * nv + iv) * 2 + iw;
    if (!full[b]) { /* :480-482 */
      bin[b] = ip;
      full[b] = 1;
    } else { /* :483-505 */
      const int64_t ip1 = bin[b];
      x[ip1 - 1] = (w[ip1 - 1] * x[ip1 - 1] + w[ip - 1] * x[ip - 1]) / (w[ip1 - 1] + w[ip - 1]);
      v[ip1 - 1] = (w[ip1 - 1] * v[ip1 - 1] + w[ip - 1] * v[ip - 1]) / (w[ip1 - 1] + w[ip - 1]);
      p[ip1 - 1] = p[ip1 - 1] + p[ip - 1];
      w[ip1 - 1] = w[ip1 - 1] + w[ip - 1];
      if (ip < np) {
        x[ip - 1] = x[np - 1];
        v[ip - 1] = v[np - 1];
        p[ip - 1] = p[np - 1];
        w[ip - 1] = w[np - 1];
        ip = ip - 1;
      }
      np = np - 1;
      full[b] = 0;
    }
  }
  free(bin);
  free(full);
  *np_io = np;
}

/* particle_remove, :531-602 */
void orc_particle_remove(const orc_input *in, double thsh, const double *hist,
                         orc_multirand *g, int64_t *np_io, double *x, double *v,
                         double *p, double *w) {
  int64_t np = *np_io;
  const double hmax = hist_max(in, hist);
  const double df_thsh = hmax * thsh;
  int64_t ip = 0;
  for (;;) {
    ip = ip + 1;
    if (ip > np) break;
    int iv;
    double df = df_at(in, hist, v[ip - 1], &iv);
    if (in->typeremove == 1 && df >= df_thsh) continue; /* :567-570 */
    df = df / hmax;                                      /* :571 */
    const double dice = orc_multirand_real64(g);
    if ((in->typeremove == 1 && dice < in->remove_frac) ||
        (in->typeremove == 2 && dice > df)) { /* :577-588 */
      if (ip < np) {
        x[ip - 1] = x[np - 1];
        v[ip - 1] = v[np - 1];
        p[ip - 1] = p[np - 1];
        w[ip - 1] = w[np - 1];
        ip = ip - 1;
      }
      np = np - 1;
    } else if (in->typeremove == 1) { /* :590-593 */
      p[ip - 1] = p[ip - 1] / (1.0 - in->remove_frac);
      w[ip - 1] = w[ip - 1] / (1.0 - in->remove_frac);
    } else { /* :594-596 */
      p[ip - 1] = p[ip - 1] / df;
      w[ip - 1] = w[ip - 1] / df;
    }
  }
  *np_io = np;
}

/* particle_split, :610-715 */
void orc_particle_split(const orc_input *in, double thsh, const double *hist,
                        orc_multirand *g, int64_t nalloc, int64_t *np_io, double *x,
                        double *v, double *p, double *w) {
  const int ng = in->split_ngroup;
  const int64_t np = *np_io;
  if (nalloc - np < 2 * ng - 1) return; /* :638-639 */
  int64_t np_inc = 0;
  const double df_thsh = hist_max(in, hist) * thsh;
  double *grand = (double *)malloc(sizeof(double) * (size_t)ng);
  const double share = (double)ng * 2.0;
  for (int64_t ip = 1; ip <= np; ip++) {
    if (nalloc - (np + np_inc) < 2 * ng - 1) break; /* :655-656 */
    int iv;
    double df = df_at(in, hist, v[ip - 1], &iv);
    if (df <= df_thsh) continue; /* :676 */
    orc_multirand_gaussian_array64(g, grand, ng);
    for (int k = 0; k < ng; k++) /* :680-681 */
      grand[k] = grand[k] * 2.0 * in->v_max / (double)in->nv * in->split_dv_sig_frac;
    for (int ig = 1; ig <= ng; ig++) { /* :686-707 */
      int64_t ip1 = np + np_inc + ig * 2 - 1;
      x[ip1 - 1] = x[ip - 1];
      v[ip1 - 1] = v[ip - 1] + grand[ig - 1];
      p[ip1 - 1] = p[ip - 1] / share;
      if (in->deltaf == 1) w[ip1 - 1] = w[ip - 1] / share;
      ip1 = ig == ng ? ip : np + np_inc + ig * 2;
      x[ip1 - 1] = x[ip - 1];
      v[ip1 - 1] = v[ip - 1] - grand[ig - 1];
      p[ip1 - 1] = p[ip - 1] / share;
      if (in->deltaf == 1) w[ip1 - 1] = w[ip - 1] / share;
    }
    np_inc = np_inc + (2 * ng - 1);
  }
  free(grand);
  *np_io = np + np_inc;
}

/* equilibrium f0(v) of the full-f branch, src/pic1dp_output.F90:375-451
 * (the reference normalises by T/m where sqrt(T/m) would be expected) */
static double output_f0(const orc_input *in, int isp, double sv) {
  const double T = in->species_temperature[isp], T2 = in->species_temperature2[isp];
  const double m = in->species_mass[isp], den = in->species_density[isp];
  const double v0 = in->species_v0[isp];
  if (in->iptcldist == 1) /* :380-381 */
    return den * (sv * sv) * exp(-(sv * sv) / 2.0) / sqrt(2.0 * ORC_PI);
  if (in->iptcldist == 2) /* :390-396 */
    return den * (exp(-((sv + v0) * (sv + v0)) / (2.0 * T / m)) +
                  exp(-((sv - v0) * (sv - v0)) / (2.0 * T / m))) /
           (sqrt(8.0 * ORC_PI) * T / m);
  if (in->iptcldist == 3) /* :410-419 */
    return den * exp(-(sv * sv) / (2.0 * T / m)) / (sqrt(2.0 * ORC_PI) * T / m) +
           (1.0 - den) * exp(-((sv - v0) * (sv - v0)) / (2.0 * T2 / m)) /
               (sqrt(2.0 * ORC_PI) * T2 / m);
  return den * exp(-((sv - v0) * (sv - v0)) / (2.0 * T / m)) /
         (sqrt(2.0 * ORC_PI) * T / m); /* :438-443 */
}

/* root-rank post-processing of output_ptcldist, src/pic1dp_output.F90:328-454,
 * applied to the rank-summed histograms */
void orc_ptcldist_finish(const orc_input *in, int isp, double *markr_xv,
                         double *total_xv, double *pertb_xv, double *markr_v,
                         double *total_v, double *pertb_v) {
  const int nxo = in->nx_opd, nvo = in->nv_opd;
  const int nxv = nxo * nvo;
  const double delv_inv = (double)(nvo - 1) / (2.0 * in->v_max); /* :203-205 */
  const double delx_inv = (double)nxo / in->lx;
  if (in->linear == 1) { /* :328-331 */
    for (int i = 0; i < nxv; i++) total_xv[i] = total_xv[i] + pertb_xv[i];
    for (int i = 0; i < nvo; i++) total_v[i] = total_v[i] + pertb_v[i];
  }
  for (int i = 0; i < nxv; i++) { /* :362-363 */
    markr_xv[i] = markr_xv[i] * delx_inv * delv_inv;
    total_xv[i] = total_xv[i] * delx_inv * delv_inv;
  }
  for (int i = 0; i < nvo; i++) { /* :364-365 */
    markr_v[i] = markr_v[i] * delv_inv;
    total_v[i] = total_v[i] * delv_inv;
  }
  if (in->deltaf == 1) { /* :366-369 */
    for (int i = 0; i < nxv; i++) pertb_xv[i] = pertb_xv[i] * delx_inv * delv_inv;
    for (int i = 0; i < nvo; i++) pertb_v[i] = pertb_v[i] * delv_inv;
  } else { /* :371-452 */
    for (int iv = 0; iv < nvo; iv++) {
      double sv = ((double)iv / (double)(nvo - 1) * 2.0 - 1.0) * in->v_max;
      double f0 = output_f0(in, isp, sv);
      for (int ix = 0; ix < nxo; ix++)
        pertb_xv[iv * nxo + ix] = total_xv[iv * nxo + ix] - f0;
      pertb_v[iv] = total_v[iv] - in->lx * f0;
    }
  }
}

/* ======================================================================
 * driver -- src/pic1dp.F90:64-109 with npe virtual reference ranks
 * ====================================================================== */
struct orc_sim {
  orc_input in;
  int npe, nthreads;
  orc_field *fld;
  int64_t *nalloc;              /* [npe] */
  int64_t *np;                  /* [npe][nspecies] */
  double **arr;                 /* [npe][nspecies][7] */
  double *charge_rank;          /* [npe][nx] per-rank charge2 */
  double *charge1, *chargeden, *E, *mode_re, *mode_im;
  int32_t itime;
  double time;
  orc_multirand **rng;          /* [npe] each rank's generator, kept after the load */
  int imerge, iremove, isplit;  /* particle_imerge / _iremove / _isplit */
};

#define SIM_ARR(s, r, isp, k) ((s)->arr[((size_t)(r) * (s)->in.nspecies + (isp)) * 7 + (k)])

orc_sim *orc_sim_new(const orc_input *in, int npe) {
  orc_sim *s = (orc_sim *)calloc(1, sizeof(orc_sim));
  s->in = *in;
  s->npe = npe;
  s->nthreads = 1;
  s->fld = orc_field_new(in);
  s->nalloc = (int64_t *)calloc((size_t)npe, sizeof(int64_t));
  s->np = (int64_t *)calloc((size_t)npe * in->nspecies, sizeof(int64_t));
  s->arr = (double **)calloc((size_t)npe * in->nspecies * 7, sizeof(double *));
  for (int r = 0; r < npe; r++) {
    s->nalloc[r] = orc_local_size(in->nparticle_max, r, npe);
    for (int isp = 0; isp < in->nspecies; isp++) {
      s->np[(size_t)r * in->nspecies + isp] = orc_particle_np(in, isp, r, npe);
      for (int k = 0; k < 7; k++)
        SIM_ARR(s, r, isp, k) =
            (double *)calloc((size_t)s->nalloc[r] + 1, sizeof(double));
    }
  }
  s->charge_rank = (double *)calloc((size_t)npe * in->nx, sizeof(double));
  s->charge1 = (double *)calloc((size_t)in->nx, sizeof(double));
  s->chargeden = (double *)calloc((size_t)in->nx, sizeof(double));
  s->E = (double *)calloc((size_t)in->nx, sizeof(double));
  s->mode_re = (double *)calloc((size_t)in->nmode, sizeof(double));
  s->mode_im = (double *)calloc((size_t)in->nmode, sizeof(double));
  s->rng = (orc_multirand **)calloc((size_t)npe, sizeof(orc_multirand *));
  /* particle_init, src/pic1dp_particle.F90:73-87 */
  s->imerge = in->nmerge > 0 ? 1 : 0;
  s->iremove = in->nremove > 0 ? 1 : 0;
  s->isplit = in->nsplit > 0 ? 1 : 0;
  return s;
}

void orc_sim_free(orc_sim *s) {
  if (!s) return;
  for (size_t i = 0; i < (size_t)s->npe * s->in.nspecies * 7; i++) free(s->arr[i]);
  free(s->arr);
  for (int r = 0; r < s->npe; r++) orc_multirand_free(s->rng[r]);
  free(s->rng);
  free(s->nalloc);
  free(s->np);
  free(s->charge_rank);
  free(s->charge1);
  free(s->chargeden);
  free(s->E);
  free(s->mode_re);
  free(s->mode_im);
  orc_field_free(s->fld);
  free(s);
}

void orc_sim_set_threads(orc_sim *s, int nthreads) {
  s->nthreads = nthreads < 1 ? 1 : nthreads;
}

/* particle_load on every virtual rank, src/pic1dp_particle.F90:159-265 */
int orc_sim_load(orc_sim *s) {
  int rc_all = 0;
#pragma omp parallel for num_threads(s->nthreads) schedule(static, 1)
  for (int r = 0; r < s->npe; r++) {
    orc_multirand *g = orc_multirand_new();
    int rc = orc_multirand_init(g, s->in.multirand_al_int,
                                s->in.multirand_seed_type, r,
                                s->in.multirand_warmup, s->in.multirand_selftest);
    if (rc) {
#pragma omp critical
      rc_all = rc;
    }
    if (rc != 2)
      for (int isp = 0; isp < s->in.nspecies; isp++)
        orc_particle_load_species(&s->in, isp, g, s->nalloc[r],
                                  SIM_ARR(s, r, isp, 0), SIM_ARR(s, r, isp, 1),
                                  SIM_ARR(s, r, isp, 2), SIM_ARR(s, r, isp, 3));
    orc_multirand_free(s->rng[r]);
    s->rng[r] = g;
  }
  s->itime = 0;
  s->time = 0.0;
  return rc_all;
}

/* interaction_collect_charge, src/pic1dp_interaction.F90:79-151 */
void orc_sim_collect_charge(orc_sim *s) {
  const orc_input *in = &s->in;
  const int nx = in->nx;
#pragma omp parallel for num_threads(s->nthreads) schedule(static, 1)
  for (int r = 0; r < s->npe; r++) {
    double *c2 = s->charge_rank + (size_t)r * nx;
    double *c1 = (double *)malloc(sizeof(double) * (size_t)nx);
    for (int ix = 0; ix < nx; ix++) c2[ix] = 0.0; /* :81 */
    for (int isp = 0; isp < in->nspecies; isp++) {
      for (int ix = 0; ix < nx; ix++) c1[ix] = 0.0; /* :83 */
      const double *q = in->deltaf == 1 ? SIM_ARR(s, r, isp, 3) : SIM_ARR(s, r, isp, 2);
      orc_deposit_species(in, s->np[(size_t)r * in->nspecies + isp],
                          SIM_ARR(s, r, isp, 0), q, c1);
      for (int ix = 0; ix < nx; ix++) /* :126-127 */
        c2[ix] = c2[ix] + c1[ix] * in->species_charge[isp];
    }
    free(c1);
  }
  /* MPI_Allreduce(SUM) :132 -- summed here in rank order */
  for (int ix = 0; ix < nx; ix++) {
    double t = s->charge_rank[ix];
    for (int r = 1; r < s->npe; r++) t = t + s->charge_rank[(size_t)r * nx + ix];
    s->charge1[ix] = t;
  }
  orc_chargeden_from_charge(in, s->charge1, s->chargeden);
}

void orc_sim_solve_field(orc_sim *s) {
  orc_field_solve_ranks(&s->in, s->fld, s->npe, s->chargeden, s->E, s->mode_re, s->mode_im);
}

/* interaction_push_particle, src/pic1dp_interaction.F90:161-370 */
void orc_sim_push(orc_sim *s, int irk) {
  const orc_input *in = &s->in;
#pragma omp parallel for num_threads(s->nthreads) schedule(static, 1)
  for (int r = 0; r < s->npe; r++)
    for (int isp = 0; isp < in->nspecies; isp++) {
      double *x = SIM_ARR(s, r, isp, 0), *v = SIM_ARR(s, r, isp, 1);
      double *p = SIM_ARR(s, r, isp, 2), *w = SIM_ARR(s, r, isp, 3);
      double *xb = SIM_ARR(s, r, isp, 4), *vb = SIM_ARR(s, r, isp, 5);
      double *wb = SIM_ARR(s, r, isp, 6);
      if (irk == 1) orc_push_backup(s->nalloc[r], x, v, w, xb, vb, wb, in->deltaf);
      orc_push_species(in, isp, irk, s->E, s->np[(size_t)r * in->nspecies + isp],
                       x, v, p, w, xb, vb, wb);
    }
}

/* time loop body, src/pic1dp.F90:79-93 */
void orc_sim_step(orc_sim *s, int nsteps) {
  for (int it = 0; it < nsteps; it++) {
    for (int irk = 1; irk <= 2; irk++) {
      orc_sim_push(s, irk);
      orc_sim_optimize(s, irk); /* src/pic1dp.F90:82 */
      orc_sim_collect_charge(s);
      orc_sim_solve_field(s);
    }
    s->itime += 1;
    s->time = s->time + s->in.dt;
  }
}

/* particle_compute_dist_pertb_abs_v over all ranks (:356-403, MPI_Allreduce in
 * rank order) for one species */
static void sim_dist_pertb_abs_v(orc_sim *s, int isp, double *hist) {
  const int nv = s->in.nv;
  double *loc = (double *)malloc(sizeof(double) * (size_t)nv);
  for (int r = 0; r < s->npe; r++) {
    for (int i = 0; i < nv; i++) loc[i] = 0.0;
    orc_dist_pertb_abs_v(&s->in, s->np[(size_t)r * s->in.nspecies + isp], SIM_ARR(s, r, isp, 1),
                         SIM_ARR(s, r, isp, 3), loc);
    for (int i = 0; i < nv; i++) hist[i] = r == 0 ? loc[i] : hist[i] + loc[i];
  }
  free(loc);
}

/* particle_optimize, src/pic1dp_particle.F90:724-783 */
int orc_sim_optimize(orc_sim *s, int irk) {
  const orc_input *in = &s->in;
  int done = 0;
  if (in->deltaf == 0) return 0;
  double *hist = (double *)malloc(sizeof(double) * (size_t)in->nv * in->nspecies);
  for (int kind = 0; kind < 3; kind++) {
    int *idx = kind == 0 ? &s->imerge : (kind == 1 ? &s->iremove : &s->isplit);
    const int n = kind == 0 ? in->nmerge : (kind == 1 ? in->nremove : in->nsplit);
    const double *tt = kind == 0 ? in->tmerge : (kind == 1 ? in->tremove : in->tsplit);
    const double *th = kind == 0 ? in->thshmerge : (kind == 1 ? in->thshremove : in->thshsplit);
    if (!(*idx > 0 && *idx <= n)) continue;
    if (!(s->time + in->dt >= tt[*idx - 1] && irk == 2)) continue;
    /* the histogram of every species first (it is global), then rank by rank */
    for (int isp = 0; isp < in->nspecies; isp++) sim_dist_pertb_abs_v(s, isp, hist + (size_t)isp * in->nv);
    for (int r = 0; r < s->npe; r++)
      for (int isp = 0; isp < in->nspecies; isp++) {
        int64_t *np = &s->np[(size_t)r * in->nspecies + isp];
        double *x = SIM_ARR(s, r, isp, 0), *v = SIM_ARR(s, r, isp, 1);
        double *p = SIM_ARR(s, r, isp, 2), *w = SIM_ARR(s, r, isp, 3);
        const double *h = hist + (size_t)isp * in->nv;
        if (kind == 0)
          orc_particle_merge(in, th[*idx - 1], h, np, x, v, p, w);
        else if (kind == 1)
          orc_particle_remove(in, th[*idx - 1], h, s->rng[r], np, x, v, p, w);
        else
          orc_particle_split(in, th[*idx - 1], h, s->rng[r], s->nalloc[r], np, x, v, p, w);
      }
    *idx += 1;
    done = 1;
  }
  free(hist);
  return done;
}

orc_multirand *orc_sim_rank_rng(orc_sim *s, int rank) { return s->rng[rank]; }
void orc_sim_set_rank_np(orc_sim *s, int rank, int isp, int64_t np) {
  s->np[(size_t)rank * s->in.nspecies + isp] = np;
}

int32_t orc_sim_itime(const orc_sim *s) { return s->itime; }
double orc_sim_time(const orc_sim *s) { return s->time; }
double orc_sim_field_energy(const orc_sim *s) { return orc_field_energy(&s->in, s->E); }

void orc_sim_get_field(const orc_sim *s, double *E, double *rho,
                       double *mode_re, double *mode_im) {
  if (E) memcpy(E, s->E, sizeof(double) * (size_t)s->in.nx);
  if (rho) memcpy(rho, s->chargeden, sizeof(double) * (size_t)s->in.nx);
  if (mode_re) memcpy(mode_re, s->mode_re, sizeof(double) * (size_t)s->in.nmode);
  if (mode_im) memcpy(mode_im, s->mode_im, sizeof(double) * (size_t)s->in.nmode);
}

void orc_sim_set_field(orc_sim *s, const double *E) {
  memcpy(s->E, E, sizeof(double) * (size_t)s->in.nx);
}

int64_t orc_sim_rank_np(const orc_sim *s, int rank, int isp) {
  return s->np[(size_t)rank * s->in.nspecies + isp];
}
int64_t orc_sim_rank_nalloc(const orc_sim *s, int rank) { return s->nalloc[rank]; }
double *orc_sim_array(orc_sim *s, int rank, int isp, int which) {
  return SIM_ARR(s, rank, isp, which);
}

/* VecSum on each rank then the scalar all-reduce, rank order */
void orc_sim_energy_sums(const orc_sim *s, int isp, double out[3]) {
  out[0] = out[1] = out[2] = 0.0;
  for (int r = 0; r < s->npe; r++) {
    double t[3];
    orc_energy_sums(s->nalloc[r], SIM_ARR(s, r, isp, 1), SIM_ARR(s, r, isp, 2),
                    SIM_ARR(s, r, isp, 3), s->in.deltaf, t);
    for (int k = 0; k < 3; k++) out[k] = out[k] + t[k];
  }
}

/* output_ptcldist over all virtual ranks: per-rank histograms, MPI_Reduce in
 * rank order (src/pic1dp_output.F90:333-358), then the root's finish */
void orc_sim_ptcldist(const orc_sim *s, int isp, int finish, double *markr_xv,
                      double *total_xv, double *pertb_xv, double *markr_v,
                      double *total_v, double *pertb_v) {
  const orc_input *in = &s->in;
  const int nxv = in->nx_opd * in->nv_opd, nvo = in->nv_opd;
  const size_t ntot = (size_t)3 * nxv + 3 * nvo;
  double *acc = (double *)calloc(ntot, sizeof(double));
  double *loc = (double *)malloc(ntot * sizeof(double));
  for (int r = 0; r < s->npe; r++) {
    memset(loc, 0, ntot * sizeof(double));
    orc_ptcldist(in, s->np[(size_t)r * in->nspecies + isp], SIM_ARR(s, r, isp, 0),
                 SIM_ARR(s, r, isp, 1), SIM_ARR(s, r, isp, 2), SIM_ARR(s, r, isp, 3),
                 loc, loc + nxv, loc + 2 * nxv, loc + 3 * nxv, loc + 3 * nxv + nvo,
                 loc + 3 * nxv + 2 * nvo);
    for (size_t i = 0; i < ntot; i++) acc[i] = r == 0 ? loc[i] : acc[i] + loc[i];
  }
  if (finish)
    orc_ptcldist_finish(in, isp, acc, acc + nxv, acc + 2 * nxv, acc + 3 * nxv,
                        acc + 3 * nxv + nvo, acc + 3 * nxv + 2 * nvo);
  memcpy(markr_xv, acc, sizeof(double) * nxv);
  memcpy(total_xv, acc + nxv, sizeof(double) * nxv);
  memcpy(pertb_xv, acc + 2 * nxv, sizeof(double) * nxv);
  memcpy(markr_v, acc + 3 * nxv, sizeof(double) * nvo);
  memcpy(total_v, acc + 3 * nxv + nvo, sizeof(double) * nvo);
  memcpy(pertb_v, acc + 3 * nxv + 2 * nvo, sizeof(double) * nvo);
  free(acc);
  free(loc);
}

/* realbuf of output_field, src/pic1dp_output.F90:117-172 */
void orc_sim_output_scalars(const orc_sim *s, double *out) {
  const orc_input *in = &s->in;
  out[0] = s->time;
  out[1] = orc_field_energy(in, s->E);
  for (int isp = 0; isp < in->nspecies; isp++) {
    double t[3];
    orc_sim_energy_sums(s, isp, t);
    double total = t[1], pert;
    if (in->deltaf == 1) {
      pert = t[2];
      if (in->linear == 1) total = total + pert; /* :152-155 */
    } else {                                      /* :156-170 */
      pert = total;
      if (in->iptcldist == 1)
        pert = pert - 3.0 * in->species_density[isp] * in->lx;
      else if (in->iptcldist == 0)
        pert = pert - in->species_temperature[isp] / in->species_mass[isp] *
                          in->species_density[isp] * in->lx;
    }
    out[2 + 3 * isp] = t[0];
    out[3 + 3 * isp] = total;
    out[4 + 3 * isp] = pert;
  }
}

/* src/pic1dp.F90:133-148 */
int orc_check_termination(const orc_input *in, int32_t itime, double time) {
  return (itime >= in->ntime_max || time + ORC_SQRT_EPS >= in->time_max) ? 1 : 0;
}

/* src/pic1dp.F90:98-106 */
int orc_output_due(const orc_input *in, double time, int itermination) {
  double a = fmod(time + ORC_SQRT_EPS, in->output_interval);
  double b = fmod(time + ORC_SQRT_EPS - in->dt, in->output_interval);
  return (a < b || itermination == 1) ? 1 : 0;
}
