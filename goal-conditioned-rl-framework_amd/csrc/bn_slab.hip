// bn_slab.hip — [Linear -> BatchNorm1d(train) -> ReLU] of the SAC / TQC actor (src/model.py:100-123) as ONE launch per
// layer and direction, for batches of up to 512 rows.
//
// BatchNorm's statistics are per COLUMN over all rows of the batch, so the natural owner of a statistic is a workgroup
// that holds a column slab over every row: a workgroup here owns 16 output columns x all B rows.  It computes its slab of
// z = X W^T + b on the matrix cores (8 waves x 64 rows, K walked in 32-wide chunks straight from L2: the operand is 512 KB
// at most and every workgroup streams all of it), then — the accumulators still in registers — the column means, the
// centred second moments (the two-pass form torch uses), normalises, applies the affine map and ReLU, and writes h, xhat
// and 1/std.  The launch-per-op form was GEMM -> row-block partial statistics -> merge + apply: three launches of
// 4.7-7.6 us each per layer for 0.13 GFLOP (profiles/r03_kernel_stats_sac_slide_b512.csv), on ONE dependency chain.
//
// The backward slab owns 16 columns of a layer's OUTPUT gradient: dh = sum over the consuming layers of G_up W_up (the
// dX GEMM of the layer above — or of the two heads), ReLU mask from xhat, the two column sums BatchNorm's backward needs,
// dz in place of xhat, dgamma / dbeta and their sum of squares for the global-norm clip: one launch instead of a GEMM and
// two BatchNorm launches.
//
// Running statistics: a two-input launch (actor.sample(next_state) and actor.sample(states) co-scheduled, agent.hip) has
// the two inputs' statistics in different workgroups, and the update order matters (next_state's batch first): the batch
// statistics go to `bstat` and the step's tanh-Gaussian launch applies them in order (bn_running_update, ops_sac.hip).
#include "ops.h"

#include "gemm_mfma.h"
#include "meet.h"
#include "sac_select.h"
#include <cstdlib>
#include <cstring>
#ifdef GCRL_SLAB_STAMPS
#include <cstdio>
#include <vector>
#endif

namespace gcrl {

namespace {

constexpr float kEps = 1e-5f;          // nn.BatchNorm1d defaults (as in ops_sac.hip)
constexpr int kWaves = 8;              // waves of a workgroup that holds ALL rows of its slab (512 threads); the row-split forms: 8 or 4 (template parameter WV)
// 16-row tiles per wave (template parameter NT): 4 = a workgroup holds all 512 rows of its 16 columns; 1 = a workgroup holds 128
// rows and the row groups of a slab exchange their column partials through memory (see slab_exchange)
#ifndef GCRL_SLAB_NS
#define GCRL_SLAB_NS 4
#endif
constexpr int kNSDefault = GCRL_SLAB_NS;   // chunks of global loads in flight per wave
// A CU keeps ~64 vector-memory instructions in flight (rowchain.h measured the same ceiling): 8 waves x 4 stages x (A + W) = 64.  With more stages
// the issue itself stalls (round 5, 8 waves: 8 stages +1.7 us per launch, 16 stages +16 us); the 4-wave row-split form has room for 8.
template <int NT, int WV> struct SlabStages { static constexpr int value = (NT == 1 && WV == 4) ? 2 * kNSDefault : kNSDefault; };
constexpr int kCK = 16;                // k per chunk (the k-permutation of gemm_mfma.h: a lane's 4 consecutive k feed 4 MFMAs)

// 4 consecutive floats at byte offset `off` (one 16-byte load, or element loads when the operand is not 16-byte
// addressable); offsets past the descriptor's extent return 0
// (`soff`: a wave-uniform byte offset — the chunk's — added by the load instruction itself: no vector arithmetic per chunk)
template <bool VEC>
__device__ inline v4f ld4b(__amdgpu_buffer_rsrc_t rs, int off, int soff) {
  v4f r;
  if (VEC) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = __uint_as_float(a[q]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + 4 * q, soff, 0));
  }
  return r;
}

// sum over the rows of the batch for this lane's column: lanes (i, g) hold partial sums of column i; result in every lane
template <int WV>
__device__ inline float col_sum(float v, float (*red)[16], int wave, int li, int lg) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  __syncthreads();
  if (lg == 0) red[wave][li] = v;
  __syncthreads();
  float s = red[0][li];
#pragma unroll
  for (int w = 1; w < WV; ++w) s += red[w][li];
  return s;
}

struct Operand { const float* p; long long ld; int K; };

// acc[t] += A[rows of tile t][k] * Bm(k, col) over one operand pair.
//   A: row-major [B][lda], k contiguous.      BROW: Bm(k, n) = W[n*ldb + k] (forward: weight rows, k contiguous)
//                                             else: Bm(k, n) = W[k*ldb + n] (backward: the consuming layer's weight, n contiguous)
// The A operand goes through a WAVE-PRIVATE LDS image: the MFMA wants lane i to hold row i, and loading it that way makes
// every 16-lane group of a load instruction touch 16 different cache lines — the vector cache looks up one line per cycle,
// so a K = 256 layer took 24 us for 7 us of MFMAs, whatever the prefetch depth (rocprofv3; first version of this file).
// Here a load instruction covers 16 rows x 64 contiguous bytes (lane = 4 * row + quad), the quads land in LDS at
// slot quad ^ ((row >> 1) & 3) of their 64-byte row (conflict-free for these stores and for the fragment reads: 8 lanes on 8
// consecutive rows at one k-quad), and lane (i, g) reads row 16 t + i, quad g back.  Wave-private: LDS instructions of one
// wave execute in order, no barrier.  Two chunks of global loads stay in flight in registers.
// WTILE (backward form, round 5): the chunk's 16 k x 16 column tile of W arrives as ONE 16-byte load per lane (lane = 4 * k + column quad: four
// 64-byte row segments per instruction) and reaches the fragment layout through a second wave-private LDS image ([16][20] floats: the fragment
// reads — lane (i, g) takes W[4 g + q][i] — hit 32 different banks per half wave).  Before: four 4-byte loads per lane and chunk, i.e. five
// vector-memory instructions per chunk against the ~64 a CU keeps in flight (4 waves x 8 stages x 5 = 160): the backward slab launch did not gain
// from the 64-row workgroups (10.5 us where the forward launch went from 10.4 to 8.6).  Needs 16-byte addressable W rows (the launcher checks).
template <bool VEC, bool BROW, int kNT, int kNS, bool WTILE = false>
__device__ inline void slab_gemm(v4f (&acc)[kNT], float* lds, const float* A, long long lda, const float* W, long long ldb, int K, int B,
                                 int ncols, int row0, int col0, int lane) {
  constexpr int kPast = 0x7ffffff0;
  const int li = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsa = wave_uniform_rsrc_n(A, (long long)(B - 1) * lda + K);
  const __amdgpu_buffer_rsrc_t rsw = wave_uniform_rsrc_n(W, BROW ? (long long)(ncols - 1) * ldb + K : (long long)(K - 1) * ldb + ncols);
  const int srow = lane >> 2, sq = lane & 3;      // staging role: row srow (+16 t) of the wave's 64, k-quad sq
  int aoff[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) aoff[t] = (int)(((long long)min(row0 + 16 * t + srow, B - 1) * lda + 4 * sq) * 4);
  const int st_off = srow * 16 + ((sq ^ ((srow >> 1) & 3)) << 2);         // (+ 256 t floats)
  const int rd_off = li * 16 + ((lg ^ ((li >> 1) & 3)) << 2);
  const int col = min(col0 + li, ncols - 1);
  const int woff = BROW ? (int)(((long long)col * ldb + 4 * lg) * 4) : (int)(((long long)(4 * lg) * ldb + col) * 4);
  const int ldb4 = (int)(ldb * 4);                // (operands are < 2 GiB)
  const int wtoff = (int)(((long long)srow * ldb + col0 + 4 * sq) * 4);   // WTILE staging role: k row srow of the chunk, column quad sq
  float* ldsw = lds + 256 * kNT;
  const int nfull = K / kCK;                      // whole chunks: the pipelined loop; a partial last chunk follows on its own
  // kNS register stages of global loads in flight (the loops over them are fully unrolled: compile-time stage indices —
  // a stage picked at run time would be a dynamically indexed array, i.e. scratch memory).  The operand was written by the
  // previous launch and comes from the memory side of this XCD's L2: ~2 us a round trip; with two chunks in flight a K = 256
  // layer took 17.8 us (one round trip per two chunks), measured.
  v4f a[kNS][kNT], w[kNS];
  // unconditional: chunks >= nc get offsets past the extents (zeros, no traffic).  A WHOLE chunk's offset rides in the load
  // instruction's scalar offset (no vector arithmetic per chunk: a 64-bit multiply per W load was 2.3 us of the backward launch's
  // 15); the scalar offset is not part of the descriptor's range check, so the partial last chunk — whose reads do run past
  // rows and past the operand — keeps everything in the checked vector offset.
  auto load = [&](int c, int nc, v4f (&a)[kNT], v4f& w, bool tail = false) {
    const bool in = c < nc;
    const int k0 = c * kCK;
    const int so = (in && !tail) ? k0 * 4 : 0, vo = tail ? k0 * 4 : 0;
#pragma unroll
    for (int t = 0; t < kNT; ++t) a[t] = ld4b<VEC>(rsa, in ? aoff[t] + vo : kPast, so);
    if (BROW) {
      w = ld4b<VEC>(rsw, in ? woff + vo : kPast, so);
    } else if (WTILE) {
      w = ld4b<true>(rsw, (in && k0 + srow < K) ? wtoff + (tail ? k0 * ldb4 : 0) : kPast, (in && !tail) ? k0 * ldb4 : 0);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsw, (in && k0 + 4 * lg + q < K) ? woff + (tail ? (k0 + q) * ldb4 : 0) : kPast,
                                                                    (in && !tail) ? (k0 + q) * ldb4 : 0, 0));
    }
  };
  auto mac = [&](const v4f (&a)[kNT], const v4f& w) {
#pragma unroll
    for (int t = 0; t < kNT; ++t) *reinterpret_cast<v4f*>(lds + 256 * t + st_off) = a[t];
    if (WTILE) *reinterpret_cast<v4f*>(ldsw + srow * 20 + 4 * sq) = w;
    // (other lanes' stores feed this lane's reads: the compiler must keep the order — the hardware does anyway)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v4f f[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) f[t] = *reinterpret_cast<const v4f*>(lds + 256 * t + rd_off);
    v4f wf = w;
    if (WTILE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[q] = ldsw[(4 * lg + q) * 20 + li];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < kNT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t][q], wf[q], acc[t], 0, 0, 0);
  };
  // NOTHING in the loop is conditional, and sched_barrier keeps every stage's consumer where it is written: with an
  // `if (c >= nchunk) return` in mac, or with the k-tail masks inside (the scheduler hoisted all four stages' selects to
  // the top of the body), the compiler drained every load in flight — s_waitcnt vmcnt(0) — once per round of stages and
  // the launch ran at one memory round trip per round whatever kNS.  Chunks past the end are loaded as zeros.
  // the partial last chunk is requested FIRST, into registers of its own (after the loop it would be one more exposed round
  // trip: the whole launch, for a first layer with K = 22 or the heads' K = 3)
  v4f at[kNT], wt;
  load(nfull, (K % kCK) ? nfull + 1 : 0, at, wt, /*tail=*/true);
  // (the prologue's loads in stage order too: the loop header's wait count is the minimum over both ways into the loop)
#pragma unroll
  for (int s = 0; s < kNS; ++s) { __builtin_amdgcn_sched_barrier(0); load(s, nfull, a[s], w[s]); }
  for (int c = 0; c < nfull; c += kNS) {
#pragma unroll
    for (int s = 0; s < kNS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      mac(a[s], w[s]);
      load(c + s + kNS, nfull, a[s], w[s]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (K % kCK) {   // (uniform) the partial chunk: the row's bytes past K belong to other columns or rows — mask both operands
    const int ks = nfull * kCK + 4 * sq, kb = nfull * kCK + 4 * lg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int t = 0; t < kNT; ++t) at[t][q] = ks + q < K ? at[t][q] : 0.f;
      if (!WTILE) wt[q] = kb + q < K ? wt[q] : 0.f;   // (the tile form's rows k >= K were never loaded)
    }
    mac(at, wt);
  }
}

// ---- row groups of a slab exchanging their column partials (NT = 1: a workgroup holds 128 rows) -------------------------------
// With all 512 rows in one workgroup a K = 256 layer is ~7 us of MFMAs on 16-32 CUs.  Split over RS row groups the GEMM is
// 1/RS of that on RS times the CUs, and the groups need each other's column partials ONCE per launch: every group publishes two
// 16-float vectors with agent-scope stores and meets the others at the slab's counter (meet.h: stores drained before the
// arrival, one monotonic 64-bit counter per slab, bounded wait, host-visible status bit on a timeout).  All RS x slabs x inputs
// workgroups are resident at once (the launcher checks against the kernel's occupancy on a device the process has to itself), so a
// waiting group never keeps an awaited one from being scheduled.  Every group then merges the RS partials in index order: the
// same result in all of them.
constexpr int kSc1 = 16;                 // agent-scope cache policy of the raw buffer builtins (gfx94x / gfx950)
struct Xchg { float* buf; unsigned int* bar; unsigned int* status; int df, nslots; };     // buf [slot][RS][32] floats; bar [slot][32] words: a 64-bit counter per 128-byte line

// a, b: this group's two values for column li (valid in wave 0, lanes lg == 0).  Returns false on a timed-out wait.
__device__ inline bool slab_exchange(const Xchg& x, int slot, int RS, int r, float a, float b, int wave, int li, int lg, float (&oa)[8],
                                     float (&ob)[8], unsigned int* s_flag) {
  const __amdgpu_buffer_rsrc_t rs = wave_uniform_rsrc(x.buf + (long long)slot * RS * 32);
  if (wave == 0 && lg == 0) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), rs, (r * 32 + li) * 4, 0, kSc1);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b), rs, (r * 32 + 16 + li) * 4, 0, kSc1);
  }
  const bool ok = meet(reinterpret_cast<unsigned long long*>(x.bar + (long long)slot * 32), (unsigned)RS, true, s_flag, x.status, MEET_ERR_BN_SLAB);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    oa[j] = j < RS ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (j * 32 + li) * 4, 0, kSc1)) : 0.f;
    ob[j] = j < RS ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (j * 32 + 16 + li) * 4, 0, kSc1)) : 0.f;
  }
  return ok;
}

// The same exchange with the DATA AS ITS OWN FLAG (round 5; the form above costs four dependent memory round trips — drain the stores, arrive, see
// the last arrival, fetch the partials — 1.8-2.0 us of a 6 us launch by the kernel's clock stamps).  A group's two values for a column travel as ONE
// 8-byte word whose all-ones pattern means "not written yet" (no finite or hardware-generated NaN pair has it); the readers poll the words themselves:
// one round trip after the last writer's store.  The words are double-buffered by the parity of the slab's LAUNCH COUNT (word 0 of its `bar` line:
// device-resident, so graph replays count too): a launch publishes into and polls parity p = count & 1, every group resets its own words of parity
// p ^ 1 (read last by the previous launch, complete by now; written next by the following one) and group 0 stores count + 1 once it has seen every
// word — by then every group of the slab has published, hence read the count (`seq` is in every wave's registers before the workgroup barrier that
// precedes the publication: the callers consume it in front of their first col_sum).
//   buf as 8-byte words: [parity][slot][8 groups][16 columns]
// (the high half of the 8-byte word: the test hook's fault word — gcrl_agent_debug_meet_fault — nonzero: group 1 withholds its words ONCE)
// (an ordinary L1-bypassing load, not an atomic one: the compiler waits for an atomic load where it is issued — 1 us at the top of the kernel)
__device__ inline unsigned long long slab_seq_load(const Xchg& x, int slot) {
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rs = wave_uniform_rsrc(reinterpret_cast<const float*>(x.bar + (long long)slot * 32));
  // (a zero the compiler cannot see through: with a visibly uniform address it moves the result to scalar registers — v_readfirstlane, and the wait
  // for the load with it — right behind the load)
  int zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  const v2u w = __builtin_amdgcn_raw_buffer_load_b64(rs, zero, 0, kSc1);
  return ((unsigned long long)w[1] << 32) | (unsigned long long)w[0];
}
__device__ inline bool slab_exchange_df(const Xchg& x, int slot, int RS, int r, unsigned long long seq_fault, float a, float b, int wave, int li, int lg,
                                        float (&oa)[8], float (&ob)[8]) {
  const unsigned int seq = (unsigned int)seq_fault;
  const bool withhold = r == 1 && (seq_fault >> 32) != 0ull;
  unsigned long long* words = reinterpret_cast<unsigned long long*>(x.buf);
  const unsigned int par = seq & 1u;
  unsigned long long* mine = words + (((long long)par * x.nslots + slot) * 8) * 16;
  unsigned long long* other = words + (((long long)(par ^ 1u) * x.nslots + slot) * 8) * 16;
  if (wave == 0 && lg == 0) {
    const unsigned long long w = ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
    if (!withhold) __hip_atomic_store(mine + r * 16 + li, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (li == 0) x.bar[(long long)slot * 32 + 1] = 0u;
    __hip_atomic_store(other + r * 16 + li, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned long long v[8];
  bool ok = true;
  for (int spins = 0;; ++spins) {
    bool miss = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = j < RS ? __hip_atomic_load(mine + j * 16 + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      miss = miss || v[j] == ~0ull;
    }
    if (__builtin_amdgcn_ballot_w64(miss) == 0ull) break;          // (wave-uniform: every lane of the wave leaves together)
    if (spins >= kMeetSpinMax) { ok = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    oa[j] = __uint_as_float((unsigned int)v[j]);
    ob[j] = __uint_as_float((unsigned int)(v[j] >> 32));
  }
  if (wave == 0 && (threadIdx.x & 63) == 0) {
    if (!ok && x.status) __hip_atomic_fetch_or(x.status, (unsigned int)MEET_ERR_BN_SLAB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (r == 0) __hip_atomic_store(x.bar + (long long)slot * 32, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return ok;
}

struct FwdProb { const float* X; long long x_slot; float* h; float* xhat; float* invstd; float* bstat; };
struct FwdArgs {
  FwdProb p[2];
  const int* slot;
  const float *W, *bias, *gamma, *beta;
  long long ldx;
  int B, H, K, RS;
  Xchg x;
#ifdef GCRL_SLAB_STAMPS
  unsigned long long* stamps;   // development build (tools/slab_stamps.sh): [workgroup][8] constant-rate clock (100 MHz) stamps
#endif
};
#ifdef GCRL_SLAB_STAMPS
#define SLAB_STAMP(k) do { if (threadIdx.x == 0) g.stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define SLAB_STAMP(k) do { } while (0)
#endif

template <bool VEC, int NT, int WV>
__global__ __launch_bounds__(64 * WV) void bn_linear_fwd_slab_kernel(FwdArgs g) {
  __shared__ float red[WV][16];
  __shared__ unsigned int s_flag;
  __shared__ __attribute__((aligned(16))) float stage[WV][16 * NT * kCK];   // wave-private images of the A operand
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int prob = NT == 1 ? (int)blockIdx.z : (int)blockIdx.y, rgrp = NT == 1 ? (int)blockIdx.y : 0;
  const FwdProb me = g.p[prob];
  const int B = g.B, H = g.H;
  constexpr int kRowsWg = 16 * NT * WV;
  const int col0 = blockIdx.x * 16, col = col0 + li, row0 = rgrp * kRowsWg + wave * 16 * NT;
  const int nl = min(kRowsWg, B - rgrp * kRowsWg);              // rows of this workgroup (>= 1: launcher)
  const long long sl = (g.slot && me.x_slot) ? (long long)*g.slot : 0;
  v4f acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
  // epilogue operands first: their latency hides behind the GEMM
  const float bias = col < H ? g.bias[col] : 0.f, gm = col < H ? g.gamma[col] : 0.f, bt = col < H ? g.beta[col] : 0.f;
  const int xslot = prob * (H / 16) + (int)blockIdx.x;
  unsigned long long seq = 0;
  if (NT == 1 && g.RS > 1 && g.x.df) seq = slab_seq_load(g.x, xslot);   // (its round trip hides behind the GEMM)
  SLAB_STAMP(0);
  slab_gemm<VEC, true, NT, SlabStages<NT, WV>::value>(acc, stage[wave], me.X + sl * me.x_slot, g.ldx, g.W, g.K, g.K, B, H, row0, col0, lane);
  SLAB_STAMP(1);
  // acc[t][r] = z[row0 + 16 t + 4 lg + r][col] - bias
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[t][r] += bias;
      if (row0 + 16 * t + 4 * lg + r < B) s += acc[t][r];
    }
  asm volatile("" ::"v"(seq));                                      // the launch count has arrived in every wave before the barriers below
  float mean = col_sum<WV>(s, red, wave, li, lg) / (float)nl;      // of this workgroup's rows
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + 16 * t + 4 * lg + r < B) { const float d = acc[t][r] - mean; q += d * d; }
  float m2 = col_sum<WV>(q, red, wave, li, lg);
  SLAB_STAMP(2);
  if (NT == 1 && g.RS > 1) {
    // merge of the row groups' (n_j, mean_j, M2_j) in index order: mean = sum n_j mean_j / B, M2 = sum (M2_j + n_j (mean_j - mean)^2)
    float pm[8], pq[8];
    const bool ok = g.x.df ? slab_exchange_df(g.x, xslot, g.RS, rgrp, seq, mean, m2, wave, li, lg, pm, pq)
                           : slab_exchange(g.x, xslot, g.RS, rgrp, mean, m2, wave, li, lg, pm, pq, &s_flag);
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < g.RS) sm += pm[j] * (float)min(kRowsWg, B - j * kRowsWg);
    mean = sm / (float)B;
    m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < g.RS) { const float dm = pm[j] - mean; m2 += pq[j] + dm * dm * (float)min(kRowsWg, B - j * kRowsWg); }
    if (!ok) mean = __builtin_nanf("");                        // a timed-out exchange must not pass for a result
  }
  SLAB_STAMP(3);
  const float var = m2 / (float)B;                              // biased: what normalises the batch
  const float invstd = 1.0f / sqrtf(var + kEps);
  if (col < H) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 16 * t + 4 * lg + r;
        if (row >= B) continue;
        const float xh = (acc[t][r] - mean) * invstd;
        const float y = xh * gm + bt;
        const long long idx = (long long)row * H + col;
        me.h[idx] = y > 0.f ? y : 0.f;
        if (me.xhat) me.xhat[idx] = xh;
      }
    if (rgrp == 0 && wave == 0 && lg == 0) {
      if (me.invstd) me.invstd[col] = invstd;
      me.bstat[col] = mean;
      me.bstat[H + col] = var;
    }
  }
#ifdef GCRL_SLAB_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SLAB_STAMP(4);
#endif
}

struct BwdArgs {
  Operand up[2]; const float* Wup[2]; long long ldw[2]; int nup;   // dh = sum_u G_u[B][K_u] . W_u[K_u][H]
  float* xhat_dz;          // [B][H]: xhat in, dz out (same elements, same thread)
  const float *invstd, *gamma, *beta;
  float *dgamma, *dbeta, *sumsq_out;   // sumsq_out[slab]: sum of squares of this slab's dgamma | dbeta (may be null)
  int B, H, RS;
  int wtile;               // every W_u has 16-byte addressable rows: slab_gemm's WTILE form (row-split launches)
  Xchg x;
};
// FOLD (round 5, the top layer of the SAC / TQC actor in the row-split form): the launch takes the sampling backward (ops_sac.hip
// tanh_gauss_bwd_kernel: 5 us of launch for B x A elements) as its prologue — every workgroup forms the two heads' output gradients of ITS rows
// from the critics' action gradients (the workgroups of slab 0 also write them out: the heads' dW problems read them), keeps them in LDS and
// contracts them with the heads' weights (K = 2 x action_dim: plain multiply-adds, mean head first) in place of the MFMA pass over memory — and the
// single-workgroup actor-loss selection + log-alpha gradient block of that launch as ONE MORE workgroup (blockIdx.x == H / 16, row group 0).
struct BwdFoldArgs {
  BwdArgs g;
  TanhGaussBwdArgs tg;
  ActorSelArgs sel; AlphaArgs al; int rider;
};

template <int NT, int WV, bool FOLD>
__device__ __forceinline__ void bn_linear_bwd_slab_body(const BwdArgs& g, const TanhGaussBwdArgs* tg) {
  __shared__ float red[WV][16];
  __shared__ unsigned int s_flag;
  __shared__ __attribute__((aligned(16))) float stage[WV][16 * NT * kCK + 320];   // wave-private images of the A operand and of W's tile
  __shared__ float s_gh[FOLD ? 16 * NT * WV : 1][FOLD ? 33 : 1];   // [local row][mean head 0..15 | log_std head 16..31]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int B = g.B, H = g.H;
  constexpr int kRowsWg = 16 * NT * WV;
  const int rgrp = NT == 1 ? (int)blockIdx.y : 0;
  const int col0 = blockIdx.x * 16, col = col0 + li, row0 = rgrp * kRowsWg + wave * 16 * NT;
  const bool okc = col < H;
  v4f acc[NT], xh[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {     // xhat first: its round trip hides behind the GEMM
    acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 16 * t + 4 * lg + r;
      xh[t][r] = (okc && row < B) ? g.xhat_dz[(long long)row * H + col] : 0.f;
    }
  }
  const float gm = okc ? g.gamma[col] : 0.f, bt = okc ? g.beta[col] : 0.f, is = okc ? g.invstd[col] : 0.f;
  unsigned long long seq = 0;
  if (NT == 1 && g.RS > 1 && g.x.df) seq = slab_seq_load(g.x, (int)blockIdx.x);
  if (FOLD) {
    const TanhGaussBwdArgs& t = *tg;
    const int A = t.A;
    // the heads' weight columns of this lane first (their round trip overlaps the element pass)
    float wm[16], wl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      wm[j] = (j < A && okc) ? g.Wup[0][(long long)j * g.ldw[0] + col] : 0.f;
      wl[j] = (j < A && okc) ? g.Wup[1][(long long)j * g.ldw[1] + col] : 0.f;
    }
    const long long slot_off = t.act_slot_stride ? (long long)t.cur->batch_slot * t.act_slot_stride : 0;
    const float alpha = t.alpha_dev ? *t.alpha_dev : t.alpha_const;
    for (int e = threadIdx.x; e < kRowsWg * A; e += 64 * WV) {
      const int rl = e / A, j = e - rl * A, b = rgrp * kRowsWg + rl;
      float gmu = 0.f, gls = 0.f;
      if (b < B) {
        tanh_gauss_bwd_elem(t, slot_off, alpha, b, j, gmu, gls);
        if (blockIdx.x == 0) { t.gmu[(long long)b * t.ld_g + j] = gmu; t.gls[(long long)b * t.ld_g + j] = gls; }
      }
      s_gh[rl][j] = gmu; s_gh[rl][16 + j] = gls;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* gh = s_gh[wave * 16 * NT + 16 * tt + 4 * lg + r];
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < A) d += gh[j] * wm[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < A) d += gh[16 + j] * wl[j];
        acc[tt][r] = d;
      }
  } else {
    if (NT == 1 && g.wtile) {   // (uniform)
      for (int u = 0; u < g.nup; ++u)
        slab_gemm<true, false, NT, SlabStages<NT, WV>::value, NT == 1>(acc, stage[wave], g.up[u].p, g.up[u].ld, g.Wup[u], g.ldw[u], g.up[u].K, B, H, row0, col0, lane);
    } else {
      for (int u = 0; u < g.nup; ++u)
        slab_gemm<true, false, NT, SlabStages<NT, WV>::value>(acc, stage[wave], g.up[u].p, g.up[u].ld, g.Wup[u], g.ldw[u], g.up[u].K, B, H, row0, col0, lane);
    }
  }
  // dy = dh where the forward's output was positive (mask recomputed from xhat exactly as the forward computed y)
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool in = row0 + 16 * t + 4 * lg + r < B;
      const float y = xh[t][r] * gm + bt;
      const float dy = (in && y > 0.f) ? acc[t][r] : 0.f;
      acc[t][r] = dy;
      s1 += dy;
      s2 += dy * xh[t][r];
    }
  asm volatile("" ::"v"(seq));
  float sum_dy = col_sum<WV>(s1, red, wave, li, lg);
  float sum_dyx = col_sum<WV>(s2, red, wave, li, lg);
  if (NT == 1 && g.RS > 1) {
    float pa[8], pb[8];
    const bool ok = g.x.df ? slab_exchange_df(g.x, (int)blockIdx.x, g.RS, rgrp, seq, sum_dy, sum_dyx, wave, li, lg, pa, pb)
                           : slab_exchange(g.x, (int)blockIdx.x, g.RS, rgrp, sum_dy, sum_dyx, wave, li, lg, pa, pb, &s_flag);
    sum_dy = 0.f; sum_dyx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < g.RS) { sum_dy += pa[j]; sum_dyx += pb[j]; }
    if (!ok) sum_dy = __builtin_nanf("");
  }
  const float m1 = sum_dy / (float)B, m2 = sum_dyx / (float)B, k = gm * is;
  if (okc) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 16 * t + 4 * lg + r;
        if (row < B) g.xhat_dz[(long long)row * H + col] = (acc[t][r] - m1 - xh[t][r] * m2) * k;
      }
    if (rgrp == 0 && wave == 0 && lg == 0) { g.dgamma[col] = sum_dyx; g.dbeta[col] = sum_dy; }
  }
  if (g.sumsq_out && rgrp == 0 && wave == 0) {
    float q = (okc && lg == 0) ? sum_dyx * sum_dyx + sum_dy * sum_dy : 0.f;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);   // over the slab's 16 columns
    if (lane == 0) g.sumsq_out[blockIdx.x] = q;
  }
}

template <int NT, int WV>
__global__ __launch_bounds__(64 * WV) void bn_linear_bwd_slab_kernel(BwdArgs g) { bn_linear_bwd_slab_body<NT, WV, false>(g, nullptr); }

template <int WV>
__global__ __launch_bounds__(64 * WV) void bn_linear_bwd_slab_fold_kernel(BwdFoldArgs f) {
  if ((int)blockIdx.x == f.g.H / 16) {   // the selection + log-alpha workgroup (its row groups > 0: nothing to do)
    if (blockIdx.y != 0 || !f.rider) return;
    __shared__ float scratch[16];
    actor_select_body(f.sel, scratch);
    __syncthreads();
    alpha_body(f.al, scratch);
    return;
  }
  bn_linear_bwd_slab_body<1, WV, true>(f.g, &f.tg);
}

bool aligned16(const void* p) { return ((unsigned long long)p & 15ull) == 0; }

}  // namespace

bool bn_slab_ok(int B, int H) { return B >= 1 && B <= 16 * 4 * kWaves && H >= 16 && H % 16 == 0; }
long long bn_slab_xchg_floats(int H) { return 2LL * 2 * (H / 16) * 8 * 32; }   // [parity][input][slab][<= 8 row groups][16] 8-byte words
// "not written yet" in every word of the exchange scratch, launch counts / meeting counters at zero: creation, after a timed-out wait, and
// before every launch on scratch that launches of other shapes share (abi_misc.hip)
int bn_slab_scratch_reset(float* xchg, unsigned int* bar, int H, hipStream_t st) {
  GCRL_HIP(hipMemsetAsync(xchg, 0xFF, (size_t)bn_slab_xchg_floats(H) * sizeof(float), st));
  GCRL_HIP(hipMemsetAsync(bar, 0, (size_t)bn_slab_bar_words(H) * sizeof(unsigned int), st));
  return GCRL_OK;
}
// the exchange of the row-split forms: the data as its own flag (default) or round 4's counter meeting (GCRL_SLAB_MEET=1)
static int split_df() {
  static const int df = std::getenv("GCRL_SLAB_MEET") ? 0 : 1;
  return df;
}
bool bn_slab_data_flag() { return split_df() != 0; }
long long bn_slab_bar_words(int H) { return 2LL * (H / 16) * 32; }

// rows split: 1 (a workgroup holds all rows) or ceil(B / 128) row groups that exchange their partials (scratch required).  The
// split form waits inside the launch: admitted only when all `slabs_x_inputs` x groups workgroups are resident at once by the
// kernel's own occupancy, on a device this process has to itself (meet.h).  A function of the shapes, the device and the
// process-wide sharing switch only: a launch's summation order never changes from step to step.
// waves per workgroup of the row-split forms: 4 (64 rows per workgroup, up to 8 row groups; the default since round 5) or 8 (128 rows,
// up to 4 groups: round 3's form, GCRL_SLAB_WAVES=8).  A function of the environment only: a launch's summation order never changes.
static int split_waves() {
  static const int wv = (std::getenv("GCRL_SLAB_WAVES") && std::atoi(std::getenv("GCRL_SLAB_WAVES")) == 8) ? 8 : 4;
  return wv;
}
static const void* fwd_split_kernel(bool vec) {
  if (split_waves() == 4) return vec ? (const void*)bn_linear_fwd_slab_kernel<true, 1, 4> : (const void*)bn_linear_fwd_slab_kernel<false, 1, 4>;
  return vec ? (const void*)bn_linear_fwd_slab_kernel<true, 1, 8> : (const void*)bn_linear_fwd_slab_kernel<false, 1, 8>;
}
static const void* bwd_split_kernel() {
  return split_waves() == 4 ? (const void*)bn_linear_bwd_slab_kernel<1, 4> : (const void*)bn_linear_bwd_slab_kernel<1, 8>;
}
static const void* bwd_fold_kernel() {
  return split_waves() == 4 ? (const void*)bn_linear_bwd_slab_fold_kernel<4> : (const void*)bn_linear_bwd_slab_fold_kernel<8>;
}
static int row_split(int want, int B, const float* xchg, const unsigned int* bar, const void* kernel, long long slabs_x_inputs) {
  const int rows = 16 * split_waves();
  if (want <= 1 || B <= rows || !xchg || !bar) return 1;
  const int rs = (B + rows - 1) / rows;          // <= 8 (bn_slab_ok: B <= 512)
  return slabs_x_inputs * rs <= meet_capacity(kernel, 64 * split_waves(), 0) ? rs : 1;
}
int bn_slab_row_split(int B, int H, int n_inputs) {   // what the launchers will choose for a layer that asks for the split (agent.hip: build)
  static float dummy_x; static unsigned int dummy_b;
  const int f = row_split(4, B, &dummy_x, &dummy_b, fwd_split_kernel(true), (long long)(H / 16) * n_inputs);
  const int b = row_split(4, B, &dummy_x, &dummy_b, bwd_split_kernel(), (long long)(H / 16));
  return (f > 1 && b > 1) ? f : 1;
}

// the top layer's backward launch can take the sampling backward and the selection block along (BnSlabBwd::fold_tg): the row-split form only
bool bn_slab_bwd_can_fold(int B, int H, int A) {
  static float dummy_x; static unsigned int dummy_b;
  return A >= 1 && A <= 16 && row_split(4, B, &dummy_x, &dummy_b, bwd_fold_kernel(), (long long)(H / 16)) > 1;
}

int launch_bn_linear_fwd_slab(hipStream_t st, const BnSlabFwd& f) {
  GCRL_CHECK_ARG(bn_slab_ok(f.B, f.H) && (f.n == 1 || f.n == 2) && f.K >= 1, "bn_linear_fwd_slab: B=%d H=%d K=%d n=%d", f.B, f.H, f.K, f.n);
  FwdArgs g;
  for (int i = 0; i < 2; ++i) {
    const BnSlabFwdProb& p = f.p[i < f.n ? i : 0];
    g.p[i] = FwdProb{p.X, p.x_slot, p.h, p.xhat, p.invstd, p.bstat};
  }
  g.slot = f.slot; g.W = f.W; g.bias = f.bias; g.gamma = f.gamma; g.beta = f.beta; g.ldx = f.ldx;
  g.B = f.B; g.H = f.H; g.K = f.K;
  bool vec = f.ldx % 4 == 0 && f.K % 4 == 0 && aligned16(f.W);
  for (int i = 0; i < f.n; ++i) vec = vec && aligned16(f.p[i].X) && f.p[i].x_slot % 4 == 0;
  g.RS = row_split(f.rsplit, f.B, f.xchg, f.bar, fwd_split_kernel(vec), (long long)(f.H / 16) * f.n);
  g.x = Xchg{f.xchg, f.bar, f.status, split_df(), 2 * (f.H / 16)};
#ifdef GCRL_SLAB_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  static long long launches = 0;
  if (!stamps_dev) { GCRL_HIP(hipMalloc((void**)&stamps_dev, 1024 * 8 * 8)); }
  GCRL_HIP(hipMemsetAsync(stamps_dev, 0, 1024 * 8 * 8, st));
  g.stamps = stamps_dev;
#endif
  if (g.RS > 1) {
    const dim3 grid(f.H / 16, g.RS, f.n);
    if (split_waves() == 4) {
      if (vec) hipLaunchKernelGGL((bn_linear_fwd_slab_kernel<true, 1, 4>), grid, dim3(64 * 4), 0, st, g);
      else hipLaunchKernelGGL((bn_linear_fwd_slab_kernel<false, 1, 4>), grid, dim3(64 * 4), 0, st, g);
    } else {
      if (vec) hipLaunchKernelGGL((bn_linear_fwd_slab_kernel<true, 1, 8>), grid, dim3(64 * 8), 0, st, g);
      else hipLaunchKernelGGL((bn_linear_fwd_slab_kernel<false, 1, 8>), grid, dim3(64 * 8), 0, st, g);
    }
  } else {
    const dim3 grid(f.H / 16, f.n);
    if (vec) hipLaunchKernelGGL((bn_linear_fwd_slab_kernel<true, 4, kWaves>), grid, dim3(64 * kWaves), 0, st, g);
    else hipLaunchKernelGGL((bn_linear_fwd_slab_kernel<false, 4, kWaves>), grid, dim3(64 * kWaves), 0, st, g);
  }
  GCRL_HIP(hipGetLastError());
#ifdef GCRL_SLAB_STAMPS
  {   // launches number GCRL_SLAB_STAMPS_AT .. +8 leave their per-workgroup stamps in the file GCRL_SLAB_STAMPS (never under a graph capture)
    static const long long at = std::getenv("GCRL_SLAB_STAMPS_AT") ? std::atoll(std::getenv("GCRL_SLAB_STAMPS_AT")) : 3000;
    const char* path = std::getenv("GCRL_SLAB_STAMPS");
    if (path && launches >= at && launches < at + 9) {
      std::vector<unsigned long long> h(1024 * 8);
      GCRL_HIP(hipStreamSynchronize(st));
      GCRL_HIP(hipMemcpy(h.data(), stamps_dev, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE* fp = std::fopen(path, launches == at ? "wb" : "ab")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
    }
    ++launches;
  }
#endif
  return GCRL_OK;
}

int launch_bn_linear_bwd_slab(hipStream_t st, const BnSlabBwd& b) {
  GCRL_CHECK_ARG(bn_slab_ok(b.B, b.H) && (b.nup == 1 || b.nup == 2), "bn_linear_bwd_slab: B=%d H=%d nup=%d", b.B, b.H, b.nup);
  BwdArgs g;
  g.nup = b.nup;
  for (int u = 0; u < 2; ++u) {
    const int v = u < b.nup ? u : 0;
    GCRL_CHECK_ARG(b.G[v] && b.W[v] && b.K[v] >= 1 && b.ldg[v] % 4 == 0 && aligned16(b.G[v]) && b.ldw[v] >= b.H,
                   "bn_linear_bwd_slab: upstream operand %d (16-byte rows of G, W rows of at least H floats)", v);
    g.up[u] = Operand{b.G[v], b.ldg[v], b.K[v]};
    g.Wup[u] = b.W[v]; g.ldw[u] = b.ldw[v];
  }
  g.xhat_dz = b.xhat_dz; g.invstd = b.invstd; g.gamma = b.gamma; g.beta = b.beta;
  g.dgamma = b.dgamma; g.dbeta = b.dbeta; g.sumsq_out = b.sumsq_out;
  g.B = b.B; g.H = b.H;
  g.wtile = std::getenv("GCRL_NO_SLAB_WTILE") ? 0 : 1;   // (A/B knob)
  for (int u = 0; u < b.nup; ++u) g.wtile = g.wtile && aligned16(b.W[u]) && b.ldw[u] % 4 == 0;
  g.RS = row_split(b.rsplit, b.B, b.xchg, b.bar, b.fold_tg ? bwd_fold_kernel() : bwd_split_kernel(), (long long)(b.H / 16));
  g.x = Xchg{b.xchg, b.bar, b.status, split_df(), 2 * (b.H / 16)};
  if (b.fold_tg) {
    const TanhGaussBwdArgs& t = *b.fold_tg;
    GCRL_CHECK_ARG(g.RS > 1 && b.nup == 2 && t.A >= 1 && t.A <= 16 && t.B == b.B && b.K[0] == t.A && b.K[1] == t.A && t.gmu == b.G[0] && t.gls == b.G[1] &&
                       t.ld_g == b.ldg[0] && (!b.fold_sel) == (!b.fold_al),
                   "bn_linear_bwd_slab: the folded sampling backward needs the row-split form and the two heads as consumers (RS=%d, nup=%d, A=%d)", g.RS, b.nup, t.A);
    BwdFoldArgs f;
    std::memset(&f, 0, sizeof(f));
    f.g = g; f.tg = t;
    if (b.fold_sel) { f.sel = *b.fold_sel; f.al = *b.fold_al; f.rider = 1; }
    const dim3 grid(b.H / 16 + 1, g.RS);
    if (split_waves() == 4) hipLaunchKernelGGL((bn_linear_bwd_slab_fold_kernel<4>), grid, dim3(64 * 4), 0, st, f);
    else hipLaunchKernelGGL((bn_linear_bwd_slab_fold_kernel<8>), grid, dim3(64 * 8), 0, st, f);
    GCRL_HIP(hipGetLastError());
    return GCRL_OK;
  }
  if (g.RS > 1 && split_waves() == 4) hipLaunchKernelGGL((bn_linear_bwd_slab_kernel<1, 4>), dim3(b.H / 16, g.RS), dim3(64 * 4), 0, st, g);
  else if (g.RS > 1) hipLaunchKernelGGL((bn_linear_bwd_slab_kernel<1, 8>), dim3(b.H / 16, g.RS), dim3(64 * 8), 0, st, g);
  else hipLaunchKernelGGL((bn_linear_bwd_slab_kernel<4, kWaves>), dim3(b.H / 16), dim3(64 * kWaves), 0, st, g);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
