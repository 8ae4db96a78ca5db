// mt19937_cpython.cc — host-only Mersenne Twister that reproduces CPython's `random` module
// bit for bit, so HER future-index picks (reference src/buffer.py:153, random.randint) and
// batch draws (src/buffer.py:124, random.sample) match the reference under a fixed seed.
//
// Restated from the published algorithms:
//   MT19937 (Matsumoto & Nishimura 1998, init_genrand / init_by_array / tempering),
//   CPython 3.10 Modules/_randommodule.c (seed(int) -> init_by_array over the 32-bit
//   little-endian words of |n|; getrandbits(k<=32) = genrand_uint32() >> (32-k);
//   random() = (a*2^26 + b) / 2^53 with a = u32>>5, b = u32>>6),
//   CPython 3.10 Lib/random.py (_randbelow_with_getrandbits, randrange/randint, sample).
#include "common.h"

#include <vector>

namespace {
constexpr int kN = 624;
constexpr int kM = 397;
}  // namespace

struct gcrl_mt {
  uint32_t w[kN];
  int idx;  // next word to temper; kN = regenerate first

  void init_scalar(uint32_t s) {
    w[0] = s;
    for (int i = 1; i < kN; ++i) w[i] = 1812433253u * (w[i - 1] ^ (w[i - 1] >> 30)) + (uint32_t)i;
    idx = kN;
  }

  void init_key(const uint32_t* key, int len) {
    init_scalar(19650218u);
    int i = 1, j = 0;
    for (int k = (kN > len ? kN : len); k > 0; --k) {
      w[i] = (w[i] ^ ((w[i - 1] ^ (w[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= kN) { w[0] = w[kN - 1]; i = 1; }
      if (++j >= len) j = 0;
    }
    for (int k = kN - 1; k > 0; --k) {
      w[i] = (w[i] ^ ((w[i - 1] ^ (w[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= kN) { w[0] = w[kN - 1]; i = 1; }
    }
    w[0] = 0x80000000u;
  }

  void twist() {
    auto mix = [](uint32_t hi, uint32_t lo) -> uint32_t {
      uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
      return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    int k = 0;
    for (; k < kN - kM; ++k) w[k] = w[k + kM] ^ mix(w[k], w[k + 1]);
    for (; k < kN - 1; ++k) w[k] = w[k + kM - kN] ^ mix(w[k], w[k + 1]);
    w[kN - 1] = w[kM - 1] ^ mix(w[kN - 1], w[0]);
    idx = 0;
  }

  inline uint32_t next_u32() {
    if (idx >= kN) twist();
    uint32_t y = w[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }

  inline uint32_t bits(int k) { return next_u32() >> (32 - k); }

  // Lib/random.py _randbelow_with_getrandbits: k = n.bit_length(); redraw while r >= n
  inline uint32_t below(uint32_t n) {
    int k = 32 - __builtin_clz(n);
    uint32_t r = bits(k);
    while (r >= n) r = bits(k);
    return r;
  }
};

extern "C" {

gcrl_mt* gcrl_mt_create(void) {
  gcrl_mt* mt = new gcrl_mt;
  uint32_t key0 = 0;
  mt->init_key(&key0, 1);
  return mt;
}

void gcrl_mt_destroy(gcrl_mt* mt) { delete mt; }

int gcrl_mt_seed(gcrl_mt* mt, uint64_t seed) {
  GCRL_CHECK_ARG(mt != nullptr, "gcrl_mt_seed: null handle");
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  mt->init_key(key, key[1] ? 2 : 1);
  return GCRL_OK;
}

int gcrl_mt_get_state(const gcrl_mt* mt, uint32_t* state625) {
  GCRL_CHECK_ARG(mt && state625, "gcrl_mt_get_state: null argument");
  std::memcpy(state625, mt->w, sizeof(uint32_t) * kN);
  state625[kN] = (uint32_t)mt->idx;
  return GCRL_OK;
}

int gcrl_mt_set_state(gcrl_mt* mt, const uint32_t* state625) {
  GCRL_CHECK_ARG(mt && state625, "gcrl_mt_set_state: null argument");
  GCRL_CHECK_ARG(state625[kN] <= (uint32_t)kN, "gcrl_mt_set_state: index %u > 624", state625[kN]);
  std::memcpy(mt->w, state625, sizeof(uint32_t) * kN);
  mt->idx = (int)state625[kN];
  return GCRL_OK;
}

uint32_t gcrl_mt_getrandbits(gcrl_mt* mt, int k) {
  if (!mt || k < 1 || k > 32) { gcrl::fail(GCRL_ERR_ARG, "gcrl_mt_getrandbits: k must be 1..32"); return 0; }
  return mt->bits(k);
}

uint32_t gcrl_mt_randbelow(gcrl_mt* mt, uint32_t n) {
  if (!mt || n == 0) { gcrl::fail(GCRL_ERR_ARG, "gcrl_mt_randbelow: n must be >= 1"); return 0; }
  return mt->below(n);
}

int64_t gcrl_mt_randint(gcrl_mt* mt, int64_t a, int64_t b) {
  if (!mt || b < a || (b - a) >= 0xffffffffLL) {
    gcrl::fail(GCRL_ERR_ARG, "gcrl_mt_randint: empty or too wide range");
    return a;
  }
  return a + (int64_t)mt->below((uint32_t)(b - a + 1));
}

double gcrl_mt_random(gcrl_mt* mt) {
  if (!mt) { gcrl::fail(GCRL_ERR_ARG, "gcrl_mt_random: null handle"); return 0.0; }
  uint32_t a = mt->next_u32() >> 5, b = mt->next_u32() >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

// random.sample(population, k) index stream (Lib/random.py, 3.10):
//   setsize = 21 (+ 4**ceil(log(3k,4)) when k > 5)
//   n <= setsize : pool path   j = randbelow(n-i); res = pool[j]; pool[j] = pool[n-i-1]
//   else         : set path    j = randbelow(n), redraw while j already selected
int gcrl_mt_sample_indices(gcrl_mt* mt, uint32_t n, uint32_t k, uint32_t* out) {
  GCRL_CHECK_ARG(mt && (out || k == 0), "gcrl_mt_sample_indices: null argument");
  if (k > n) return gcrl::fail(GCRL_ERR_NOT_ENOUGH, "Sample larger than population (k=%u > n=%u)", k, n);
  if (k == 0) return GCRL_OK;
  uint64_t setsize = 21;
  if (k > 5) {
    uint64_t p = 1;  // smallest power of 4 that is >= 3k  (3k is never a power of 4)
    while (p < 3ull * k) p *= 4;
    setsize += p;
  }
  if (n <= setsize) {
    std::vector<uint32_t> pool(n);
    for (uint32_t i = 0; i < n; ++i) pool[i] = i;
    for (uint32_t i = 0; i < k; ++i) {
      uint32_t j = mt->below(n - i);
      out[i] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  } else {
    // open-addressing membership table, load <= 1/4
    uint32_t cap = 16;
    while (cap < 4 * k) cap <<= 1;
    std::vector<uint32_t> tab(cap, 0xffffffffu);
    const uint32_t mask = cap - 1;
    for (uint32_t i = 0; i < k; ++i) {
      for (;;) {
        uint32_t j = mt->below(n);
        uint32_t h = (j * 2654435761u) & mask;
        bool seen = false;
        while (tab[h] != 0xffffffffu) {
          if (tab[h] == j) { seen = true; break; }
          h = (h + 1) & mask;
        }
        if (seen) continue;
        tab[h] = j;
        out[i] = j;
        break;
      }
    }
  }
  return GCRL_OK;
}

int gcrl_mt_future_indices(gcrl_mt* mt, int T, int k_future, uint8_t* out) {
  GCRL_CHECK_ARG(mt && T >= 1 && T <= 255 && k_future >= 0, "gcrl_mt_future_indices: bad T=%d k=%d", T, k_future);
  GCRL_CHECK_ARG(out || k_future * (T - 1) == 0, "gcrl_mt_future_indices: null output");
  int o = 0;
  for (int i = 0; i + 1 < T; ++i)
    for (int r = 0; r < k_future; ++r)  // randint(i+1, T-1) = i+1 + randbelow(T-1-i)
      out[o++] = (uint8_t)(i + 1 + (int)mt->below((uint32_t)(T - 1 - i)));
  return GCRL_OK;
}

}  // extern "C"
