// her_ring.h — internal view of the HER replay ring (shared with the update engine).
//
// HBM layout.  One transition = one packed record of RS floats, 64-byte aligned:
//     [ s (S) | a (A) | pad ]  [ ns (S) | pad ]  [ r | d ]  zero pad
//       <-- SA4 = roundup(S+A,4) -->  <-- S4 = roundup(S,4) -->      RS = roundup(SA4+S4+2, 16)
// Records are contiguous in arrival order: ring[cap][RS].  A uniformly random row gather then
// touches ceil(4*RS/128) whole 128-B lines per row (PickAndPlace: 2 lines for 208 useful
// bytes) where five separate field arrays would touch 6-7.  Every field group starts on a
// 16-byte boundary, so the update engine's batch matrices ([s|a], [ns|..], [s|..], row stride
// SA4) are plain float4 copies of record slices, and [s|a] is already the critic's input row.  Logical index j (0 = oldest, what random.sample indexes in the reference's
// deque, src/buffer.py:124) lives at physical row (head + j) mod cap; head/len are host state.
//
// Staging: per env, up to flush_len records of RG = roundup(RW+G, 16) floats: the ring record's
// RW = SA4+S4+2 leading floats, then ag (G)          (the dg column is the goal slot of s itself)
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "common.h"

namespace gcrl {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// uniform integer in [0, n) from a 64-bit counter hash — the device-RNG ("fast") mode of the
// future-index pick and of the batch draw.  Restated bit-for-bit in oracle/her_oracle.py.
__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint32_t hash_below(uint64_t seed, uint64_t stream, uint64_t ctr,
                                                uint32_t n) {
  uint64_t h = mix64(mix64(seed ^ (stream * 0xd1342543de82ef95ull)) + ctr);
  return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);  // multiply-high range reduction
}

// device-RNG batch draw: batch element t of draw d is pi_{seed,d}(t), pi a keyed permutation of [0, n)
// (a 4-round balanced Feistel network over the next even power of two, cycle-walked back into range).
// A permutation gives the batch its without-replacement property by construction — no rejection, no
// communication between the elements: every gather lane computes its own row index, so the device mode
// needs neither a host RNG nor an index upload.  Restated in oracle/her_oracle.py (HashRng.sample).
struct IdxGen {
  uint64_t seed, draw0;   // draw number of the launch's first batch
  uint32_t n;             // population (ring length)
  int B, half_bits;       // batch size; the Feistel halves are half_bits wide
};

__host__ __device__ inline int feistel_half_bits(uint32_t n) {
  int bits = 2;
  while (bits < 32 && (1ull << bits) < (uint64_t)n) ++bits;
  return (bits + 1) >> 1;
}

__host__ __device__ inline uint32_t feistel_index(uint64_t seed, uint64_t draw, uint32_t n, int hb, uint32_t t) {
  const uint32_t mask = hb >= 32 ? 0xffffffffu : ((1u << hb) - 1u);
  const uint64_t key = mix64(seed ^ (draw * 0xd1342543de82ef95ull) ^ 0x5bd1e995ull);
  uint64_t x = t;
  do {
    uint32_t L = (uint32_t)(x >> hb) & mask, R = (uint32_t)x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t f = (uint32_t)(mix64(key + (uint64_t)r * 0x9e3779b97f4a7c15ull + R) >> 32) & mask;
      const uint32_t nl = R;
      R = L ^ f;
      L = nl;
    }
    x = ((uint64_t)L << hb) | R;
  } while (x >= (uint64_t)n);
  return (uint32_t)x;
}

__host__ __device__ inline uint32_t idxgen_at(const IdxGen& g, long long row) {
  const long long m = row / g.B;
  return feistel_index(g.seed, g.draw0 + (uint64_t)m, g.n, g.half_bits, (uint32_t)(row - m * g.B));
}

}  // namespace gcrl

struct gcrl_her {
  gcrl_her_config cfg;
  int S, A, G, SA4, S4, RW, RS, RG;   // o_ns = SA4, o_r = SA4+S4, o_d = o_r+1, o_ag = RW
  gcrl_mt* rng = nullptr;
  bool own_rng = false;
  hipStream_t stream = nullptr;

  float* ring = nullptr;   // [cap][RS]
  float* stage = nullptr;  // [nenvs][flush_len][RG]
  int64_t head = 0, len = 0;
  std::vector<int> staged;
  uint64_t episodes_flushed = 0;  // stream id of the device RNG
  uint64_t draws_done = 0;        // batch-draw counter of the device RNG
  gcrl::IdxGen last_gen{};        // device-RNG mode: what the next gather launch computes its indices from
  bool idx_on_device = true;      // false: no index array was uploaded (device-RNG mode)
  uint64_t mutation_epoch = 0;    // bumped by every flush

  // index upload: pinned host slots -> idx_dev
  static constexpr int kSlots = 8;
  uint32_t* idx_pinned[kSlots] = {};
  hipEvent_t slot_ev[kSlots] = {};
  size_t slot_rows = 0;
  int next_slot = 0;
  uint32_t* idx_dev = nullptr;
  size_t idx_dev_rows = 0;
  // pinned slots for whole-episode uploads
  float* epi_pinned[kSlots] = {};
  hipEvent_t epi_ev[kSlots] = {};
  int next_epi_slot = 0;
  // future indices of a flush launch when they exceed the inline kernel-argument space (k_future >= 42)
  uint8_t* fut_dev = nullptr;
  uint8_t* fut_pinned[kSlots] = {};
  hipEvent_t fut_ev[kSlots] = {};
  int next_fut_slot = 0;
  // GCRL_REWARD_HOST: the caller's compute_reward evaluates every relabel slot of a flush on the host (gcrl_her_set_reward_callback)
  gcrl_reward_fn reward_cb = nullptr;
  void* reward_cb_user = nullptr;
  std::vector<float> ag_mirror;          // [nenvs][flush_len][G]: host copy of the staged achieved goals
  std::vector<float> cb_ag, cb_goal;     // pair lists handed to the callback
  float* rew_dev = nullptr;
  float* rew_pinned[kSlots] = {};
  hipEvent_t rew_ev[kSlots] = {};
  int next_rew_slot = 0;
  // raw observation rows of a vector-env step and their normalised [obs | goal] state matrices (gcrl_her_process_step)
  float* ps_dev = nullptr;
  float* ps_pinned[kSlots] = {};
  size_t ps_floats = 0;
  // per-env payload of a vector-env step (gcrl_her_push_batch)
  float* pay_dev = nullptr;
  float* pay_pinned[kSlots] = {};

  // gather-launch timing (gcrl_her_profile_*)
  bool prof = false;
  std::vector<hipEvent_t> prof_a, prof_b;
  std::vector<int64_t> prof_pair_rows;   // rows of each pending bracketed launch
  int64_t prof_class_rows = 0;           // statistics are kept for the LARGEST launch size seen (a cycle's main gather)
  size_t prof_used = 0;
  int64_t prof_launches = 0, prof_rows = 0;
  double prof_ms = 0.0;

  // NULL -> the handle's own stream; GCRL_STREAM_LEGACY -> HIP's legacy default stream
  hipStream_t pick(void* s) const {
    if (!s) return stream;
    if (s == GCRL_STREAM_LEGACY) return (hipStream_t) nullptr;
    return (hipStream_t)s;
  }
};

struct gcrl_normalizer;

namespace gcrl {

// device RunningNormalizer (normalizer.hip): statistics update from device rows; out[:, col0:col0+D] = normalize(x)
// (z == null: plain copy)
void normalizer_view(const gcrl_normalizer* z, const double** mean, const double** var, double** count, double* clip, int* mode = nullptr);   // mode: norm_math.h bits
void normalizer_updated(gcrl_normalizer* z);   // call after enqueuing a launch that updates z
int normalizer_update_dev(gcrl_normalizer* z, const float* x_dev, int n, int ld, hipStream_t st);
int normalizer_apply_dev(const gcrl_normalizer* z, const float* x_dev, int n, int ld, int D, float* out_dev, int ld_out, int col0,
                         hipStream_t st);

// Draw M*B logical indices (random.sample semantics per batch) into a pinned slot and upload
// them to h->idx_dev on `st`.  idx_host != NULL: use those instead of drawing.  Returns the
// pinned host copy through *host_copy (valid until kSlots more uploads).
int her_upload_indices(gcrl_her* h, int B, int M, const uint32_t* idx_host, hipStream_t st,
                       const uint32_t** host_copy);

// Gather for the update engine: rows idx_dev[0..n) -> three GEMM-ready matrices with row
// stride ldx = roundup(S+A,4): sa = [s|a], nsa = [ns|0..], spa = [s|0..] (may be null: not written); r[n], d[n].
// `idx_dev` may point into pinned (device-mapped) host memory: the update engine's head launch of a call reads its 256
// indices straight from the upload block.  cp_*: the same launch also copies cp_bytes (a multiple of 16) from cp_src
// (pinned host) to cp_dst (device) — the call's control block travels with its first gather instead of as copies before it.
int her_gather_update(gcrl_her* h, const uint32_t* idx_dev, int64_t n, float* sa, float* nsa,
                      float* spa, int ldx, float* r, float* d, hipStream_t st, const void* cp_src = nullptr,
                      void* cp_dst = nullptr, size_t cp_bytes = 0);

}  // namespace gcrl
