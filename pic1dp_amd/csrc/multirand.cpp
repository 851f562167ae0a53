// multirand.cpp -- see multirand.hpp.  Host code of the product loader.
#include "multirand.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>

namespace pic1dp {

namespace {

// seeding tables of multirand_init, src/multirand.F90:163-195 (data)
constexpr int64_t kPrimesA[100] = {
    15484219, 15484223, 15484243, 15484247, 15484279, 15484333, 15484363, 15484387, 15484393,
    15484409, 15484421, 15484453, 15484457, 15484459, 15484471, 15484489, 15484517, 15484519,
    15484549, 15484559, 15484591, 15484627, 15484631, 15484643, 15484661, 15484697, 15484709,
    15484723, 15484769, 15484771, 15484783, 15484817, 15484823, 15484873, 15484877, 15484879,
    15484901, 15484919, 15484939, 15484951, 15484961, 15484999, 15485039, 15485053, 15485059,
    15485077, 15485083, 15485143, 15485161, 15485179, 15485191, 15485221, 15485243, 15485251,
    15485257, 15485273, 15485287, 15485291, 15485293, 15485299, 15485311, 15485321, 15485339,
    15485341, 15485357, 15485363, 15485383, 15485389, 15485401, 15485411, 15485429, 15485441,
    15485447, 15485471, 15485473, 15485497, 15485537, 15485539, 15485543, 15485549, 15485557,
    15485567, 15485581, 15485609, 15485611, 15485621, 15485651, 15485653, 15485669, 15485677,
    15485689, 15485711, 15485737, 15485747, 15485761, 15485773, 15485783, 15485801, 15485807,
    15485837};
constexpr int64_t kPrimesB[100] = {
    7001, 7013, 7019, 7027, 7039, 7043, 7057, 7069, 7079, 7103, 7109, 7121, 7127, 7129, 7151,
    7159, 7177, 7187, 7193, 7207, 7211, 7213, 7219, 7229, 7237, 7243, 7247, 7253, 7283, 7297,
    7307, 7309, 7321, 7331, 7333, 7349, 7351, 7369, 7393, 7411, 7417, 7433, 7451, 7457, 7459,
    7477, 7481, 7487, 7489, 7499, 7507, 7517, 7523, 7529, 7537, 7541, 7547, 7549, 7559, 7561,
    7573, 7577, 7583, 7589, 7591, 7603, 7607, 7621, 7639, 7643, 7649, 7669, 7673, 7681, 7687,
    7691, 7699, 7703, 7717, 7723, 7727, 7741, 7753, 7757, 7759, 7789, 7793, 7817, 7823, 7829,
    7841, 7853, 7867, 7873, 7877, 7879, 7883, 7901, 7907, 7919};

// known answers of multirand_selftest, src/multirand.F90:396-425 (data)
constexpr int64_t kKatKiss[10] = {
    8932985056925012148LL,  5710300428094272059LL, -104233206776033023LL, -4143107803135683366LL,
    542381058189297533LL,   -4244931820854714191LL, 6853720724624422285LL, -767542866500872268LL,
    -257204313086867125LL,  8128797625455304420LL};
constexpr int64_t kKatMtHead[10] = {
    -3932459287431434586LL, 4620546740167642908LL, -5337173792191653896LL, -983805426561117294LL,
    355488278567739596LL,   7469126240319926998LL, 4635995468481642529LL,  418970542659199878LL,
    -8842573084457035060LL, 6358044926049913402LL};
constexpr int64_t kKatMtTail[10] = {
    -7948593974297132281LL, 1921007855220546564LL, 7643484074408755248LL, -7128315020423208677LL,
    1370093900783164344LL,  6776537281339823025LL, 3450492372588984223LL, -9045729527952115285LL,
    7896519943553875907LL,  -4143300141377237606LL};
constexpr int64_t kKatSkHead[10] = {
    6140839658375754198LL, -95225469143006167LL,  -9148462456964506707LL, 3912874252778582253LL,
    6801212277726928591LL, -809575511391043410LL, -397286769868273005LL,  4963780769400405858LL,
    2406624640673457322LL, 1246843699883922102LL};
constexpr int64_t kKatSkTail[10] = {
    -1387224431860786161LL, -8846516422183390713LL, 8111357788999165247LL, 444070776306226770LL,
    -7730678117654887867LL, -296399128303442035LL,  -1658509282659454084LL, -8190332265239255687LL,
    -1492517620356299342LL, -5016179395587873849LL};

inline uint64_t xs_step(uint64_t s) {  // the 13/17/43 xorshift shared by KISS and SuperKISS
  s ^= s << 13;
  s ^= s >> 17;
  s ^= s << 43;
  return s;
}

inline int64_t mag(int64_t a) { return a < 0 ? -a : a; }

}  // namespace

Multirand::Multirand() : q_(kStateWords, 0) {}

uint64_t Multirand::kiss() {
  // x: multiply-with-carry pair (q0, carry q3); q1 xorshift; q2 congruential
  const uint64_t x = q_[0];
  const uint64_t t = (x << 58) + q_[3];
  const uint64_t sx = x >> 63, st = t >> 63;
  q_[3] = (sx == st) ? (x >> 6) + sx : (x >> 6) + 1 - ((x + t) >> 63);
  q_[0] = x + t;
  q_[1] = xs_step(q_[1]);
  q_[2] = 6906969069ULL * q_[2] + 1234567ULL;
  return q_[0] + q_[1] + q_[2];
}

void Multirand::refill_mt() {
  constexpr int N = 312, M = 156;
  constexpr uint64_t HI = 0xFFFFFFFF80000000ULL, LO = 0x7FFFFFFFULL, A = 0xB5026F5AA96619E9ULL;
  auto twist = [&](uint64_t u, uint64_t l) {
    const uint64_t y = (u & HI) | (l & LO);
    return (y >> 1) ^ ((y & 1ULL) ? A : 0ULL);
  };
  for (int i = 0; i < N - M; ++i) q_[i] = q_[i + M] ^ twist(q_[i], q_[i + 1]);
  for (int i = N - M; i < N - 1; ++i) q_[i] = q_[i + M - N] ^ twist(q_[i], q_[i + 1]);
  q_[N - 1] = q_[M - 1] ^ twist(q_[N - 1], q_[0]);
  pos_ = 0;
}

uint64_t Multirand::mt() {
  if (pos_ >= 312) refill_mt();
  uint64_t y = q_[pos_++];
  y ^= (y >> 29) & 0x5555555555555555ULL;
  y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
  y ^= (y << 37) & 0xFFF7EEE000000000ULL;
  y ^= y >> 43;
  return y;
}

void Multirand::refill_superkiss() {
  // complementary multiply-with-carry over the lag table, carry in q[kLag]
  uint64_t c = q_[kLag];
  for (int i = 0; i < kLag; ++i) {
    const uint64_t v = q_[i];
    const uint64_t z = ((v << 41) >> 1) + ((v << 39) >> 1) + (c >> 1);
    const uint64_t low = c & 1ULL;
    c = (v >> 23) + (v >> 25) + (z >> 63);
    q_[i] = ~((z << 1) + low);
  }
  q_[kLag] = c;
  pos_ = 0;
}

uint64_t Multirand::superkiss() {
  if (pos_ >= kLag) refill_superkiss();
  q_[kLag + 1] = q_[kLag + 1] * 6906969069ULL + 123ULL;
  q_[kLag + 2] = xs_step(q_[kLag + 2]);
  return q_[pos_++] + q_[kLag + 1] + q_[kLag + 2];
}

uint64_t Multirand::next() {
  switch (engine_) {
    case MT19937_64: return mt();
    case SUPERKISS64: return superkiss();
    default: return kiss();
  }
}

void Multirand::fill_real(double *a, int64_t n) {
  for (int64_t i = 0; i < n; ++i) a[i] = to_real(next());
}

void Multirand::fill_gaussian(double *a, int64_t n) {
  constexpr double kMax = 9223372036854775807.0;  // 2**63-1 -> 2**63 as double
  int64_t i = 0;
  if (gauss_held_ && n > 0) {
    a[i++] = gauss_val_;
    gauss_held_ = false;
  }
  for (; i < n; i += 2) {
    double x, y, r2;
    do {
      x = static_cast<double>(static_cast<int64_t>(next())) / kMax;
      y = static_cast<double>(static_cast<int64_t>(next())) / kMax;
      r2 = x * x + y * y;
    } while (!(r2 > 0.0 && r2 < 1.0));
    const double f = std::sqrt((-2.0 * std::log(r2)) / r2);
    a[i] = x * f;
    if (i + 1 < n) {
      a[i + 1] = y * f;
    } else {
      gauss_val_ = y * f;
      gauss_held_ = true;
    }
  }
}

void Multirand::default_seeds(int al_int) {
  engine_ = al_int;
  if (al_int == MT19937_64) {
    q_[0] = 5489ULL;
    for (int i = 1; i < 312; ++i)
      q_[i] = 6364136223846793005ULL * (q_[i - 1] ^ (q_[i - 1] >> 62)) + static_cast<uint64_t>(i);
    pos_ = 312;
  } else if (al_int == SUPERKISS64) {
    uint64_t cng = 12367890123456ULL, xs = 521288629546311ULL;
    q_[kLag] = 36243678541ULL;
    for (int i = 0; i < kLag; ++i) {
      cng = cng * 6906969069ULL + 123ULL;
      xs = xs_step(xs);
      q_[i] = cng + xs;
    }
    q_[kLag + 1] = cng;
    q_[kLag + 2] = xs;
    pos_ = kLag;
  } else {
    q_[0] = 1234567890987654321ULL;
    q_[1] = 362436362436362436ULL;
    q_[2] = 1066149217761810ULL;
    q_[3] = 123456123456123456ULL;
  }
}

bool Multirand::selftest(int al_int) {
  bool ok = to_real(static_cast<uint64_t>(INT64_MAX)) == 1.0 &&
            to_real(static_cast<uint64_t>(INT64_MIN)) == 0.0;
  default_seeds(al_int);
  const int64_t *head = kKatKiss, *tail = nullptr;
  int tail_at = 0;
  if (al_int == MT19937_64) {
    head = kKatMtHead;
    tail = kKatMtTail;
    tail_at = 312 - 5;
  } else if (al_int == SUPERKISS64) {
    head = kKatSkHead;
    tail = kKatSkTail;
    tail_at = kLag - 5;
  }
  for (int i = 0; i < 10; ++i)
    if (static_cast<int64_t>(next()) != head[i]) return false;
  if (tail) {
    for (int i = 10; i < tail_at; ++i) next();
    for (int i = 0; i < 10; ++i)
      if (static_cast<int64_t>(next()) != tail[i]) return false;
  }
  return ok;
}

Multirand::Status Multirand::init(int al_int, int seed_type, int mype, int warmup, bool do_selftest) {
  if (al_int != MT19937_64 && al_int != SUPERKISS64) al_int = KISS64;
  const int64_t nseed = al_int == MT19937_64 ? 312 : (al_int == SUPERKISS64 ? kStateWords : 4);
  Status st = OK;
  engine_ = al_int;
  if (do_selftest && !selftest(al_int)) st = SELFTEST_FAILED;
  engine_ = al_int;

  if (seed_type == 3) {
    std::FILE *f = std::fopen("/dev/urandom", "rb");
    if (!f) {
      seed_type = 2;  // the reference falls back to the clock
    } else {
      auto word = [&](uint64_t &w) { return std::fread(&w, 8, 1, f) == 1; };
      for (int64_t i = 0; i < nseed; ++i)
        if (!word(q_[i])) st = IO_ERROR;
      if (al_int == KISS64) {
        while (q_[1] == 0 && word(q_[1])) {}
        while (q_[0] == 0 && q_[3] == 0 && word(q_[0]) && word(q_[3])) {}
      } else if (al_int == SUPERKISS64) {
        while (q_[kLag + 2] == 0 && word(q_[kLag + 2])) {}
      }
      std::fclose(f);
    }
  }
  if (seed_type != 3) {
    // With SuperKISS the reference spins forever unless an earlier self-test
    // left a non-zero xorshift word (src/multirand.F90:346-348): refuse instead.
    if (al_int == SUPERKISS64 && q_[kLag + 2] == 0) return WOULD_HANG;
    int64_t clock;
    if (seed_type == 2) {
      clock = std::chrono::duration_cast<std::chrono::nanoseconds>(
                  std::chrono::steady_clock::now().time_since_epoch()).count();
    } else {
      clock = kPrimesA[1];
    }
    const int64_t cmod = mag(clock) % 100;
    const int64_t rank_term = kPrimesA[mag(clock + kPrimesB[cmod] * mype) % 100] * mype;
    for (int64_t i = 0; i < 4; ++i) {
      int64_t s = clock + rank_term;
      s += kPrimesB[mag(s + kPrimesA[cmod] * i) % 100] * i;
      q_[i] = static_cast<uint64_t>(s);
    }
    // a KISS stream, warmed by 20 draws, fills the engine's whole seed array;
    // slot 0 keeps the last warm-up draw
    std::vector<uint64_t> fresh(static_cast<size_t>(nseed));
    for (int i = 0; i < 20; ++i) fresh[0] = kiss();
    for (int64_t i = 1; i < nseed; ++i) fresh[i] = kiss();
    if (al_int == KISS64) {
      while (fresh[1] == 0) fresh[1] = kiss();
      while (fresh[0] == 0 && fresh[3] == 0) {
        fresh[0] = kiss();
        fresh[3] = kiss();
      }
    }
    for (int64_t i = 0; i < nseed; ++i) q_[i] = fresh[i];
  }
  if (al_int == MT19937_64) pos_ = 312;
  if (al_int == SUPERKISS64) pos_ = kLag;
  for (int64_t i = 0, n = static_cast<int64_t>(warmup) * nseed; i < n; ++i) next();
  return st;
}

}  // namespace pic1dp
