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

// device-RNG batch draw (host side): the first B DISTINCT values of hash_below(seed, draw, ctr = 0, 1, ...)
// in draw order (what oracle/her_oracle.py HashRng.sample restates); duplicates found through a small
// open-addressing table instead of a scan of the batch so far (B = 2048: 4 M compares per batch)
inline void hash_draw_batch(uint64_t seed, uint64_t draw, uint32_t n, int B, uint32_t* out, std::vector<uint32_t>& table) {
  size_t cap = 64;
  while (cap < (size_t)B * 2) cap <<= 1;
  table.assign(cap, 0xffffffffu);
  uint64_t ctr = 0;
  for (int i = 0; i < B;) {
    const uint32_t j = hash_below(seed ^ 0x5bd1e995u, draw, ctr++, n);
    size_t slot = (j * 2654435761u) & (cap - 1);
    bool dup = false;
    while (table[slot] != 0xffffffffu) {
      if (table[slot] == j) { dup = true; break; }
      slot = (slot + 1) & (cap - 1);
    }
    if (dup) continue;
    table[slot] = j;
    out[i++] = j;
  }
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
  // per-env payload of a vector-env step (gcrl_her_push_batch)
  float* pay_dev = nullptr;
  float* pay_pinned[kSlots] = {};

  // gather-launch timing (gcrl_her_profile_*)
  bool prof = false;
  std::vector<hipEvent_t> prof_a, prof_b;
  size_t prof_used = 0;
  int64_t prof_launches = 0, prof_rows = 0;
  double prof_ms = 0.0, prof_clk_ticks = 0.0;
  unsigned long long* prof_clk = nullptr;   // [pairs][2] device words
  int prof_clk_khz = 100000;

  // NULL -> the handle's own stream; GCRL_STREAM_LEGACY -> HIP's legacy default stream
  hipStream_t pick(void* s) const {
    if (!s) return stream;
    if (s == GCRL_STREAM_LEGACY) return (hipStream_t) nullptr;
    return (hipStream_t)s;
  }
};

namespace gcrl {

// Draw M*B logical indices (random.sample semantics per batch) into a pinned slot and upload
// them to h->idx_dev on `st`.  idx_host != NULL: use those instead of drawing.  Returns the
// pinned host copy through *host_copy (valid until kSlots more uploads).
int her_upload_indices(gcrl_her* h, int B, int M, const uint32_t* idx_host, hipStream_t st,
                       const uint32_t** host_copy);

// Gather for the update engine: rows idx_dev[0..n) -> three GEMM-ready matrices with row
// stride ldx = roundup(S+A,4): sa = [s|a], nsa = [ns|0..], spa = [s|0..]; r[n], d[n].
int her_gather_update(gcrl_her* h, const uint32_t* idx_dev, int64_t n, float* sa, float* nsa,
                      float* spa, int ldx, float* r, float* d, hipStream_t st);

}  // namespace gcrl
