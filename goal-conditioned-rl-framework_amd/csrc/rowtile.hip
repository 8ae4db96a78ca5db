// rowtile.hip — the DDPG critic phase (K) and actor phase (P) of rowchain.hip in WEIGHT-SLICE form (round 4; opt-in: GCRL_ROWTILE=1).
//
// rowchain_ddpg_kernel gives every 4 batch rows a workgroup that streams each layer's WHOLE weight matrix (256 KB at
// H = 256) through one CU: a layer pass is bound by that CU's vector-memory path (2.85 us measured, 1.7 us at the L1's
// 64 B/clk), ten dependent passes per phase, half the chip idle at B = 256.  Here a workgroup owns the 16 x 16 output
// tile (row block rb, column block cb) of EVERY layer of its role's chain: per layer it reads its 16 columns of the weight
// matrix (16 KB, independent of computed data) and its row block's 16 x H activations (16 KB) and issues 16 MFMAs per wave
// (v_mfma_f32_16x16x4_f32, k split over the four waves, partial sums through LDS).  The price is a hand-off per layer between
// the H / 16 workgroups of a row block.  tools/microbench_rowtile.hip (profiles/r04_rowtile_microbench.txt): 1.8-2.2 us per
// layer of a bare chain against 2.85; inside the engine (heads, two roles, saves) 2.7-2.9 us per hand-off cycle: the step is
// 55.4 us against 56.3 with the row-chain launch — measured, and left opt-in.
//
// Roles (src/agent.py:1288-1317; the arithmetic per row is rowchain.hip's, the summation order inside a dot product is not):
//   P   actor phase: actor forward (saved) -> a = tanh(head) -> critic forward on [s | a] -> partials of Q(s, pi(s)), constant
//       upstream -1/B -> critic input-gradient chain -> action gradient, tanh' -> actor gradient chain (saved)
//   K   critic phase: the target chain (target actor -> a' -> target critic -> partials of Q') and the online critic's forward
//       on [s | a] (saved) are independent — their layers share the hand-off cycle, two tiles per cycle — then y, dq and the
//       online critic's gradient chain (saved)
// Small heads (N <= 16 outputs) are computed REDUNDANTLY by every workgroup of a row block from the row block's full
// activations (one more group of MFMAs on the fragments it has loaded anyway) — no extra hand-off.
//
// Hand-off: THE DATA IS THE FLAG.  Every word of a hand-off buffer is 0xFFFFFFFF ("not written yet": a NaN pattern no fp32
// arithmetic produces) between launches.  The tile leaves wave 0 as ONE 16-byte store per lane — no drain, no counter —
// and a consumer wave simply loads its fragments (sc1: past the CU's L1) until none of their words is that pattern.
// Once a workgroup holds ALL tiles of a stage, every workgroup of the row block has stored its tile of that stage, which it
// did after reading the stage before: so the workgroup puts its own tile of the PREVIOUS stage back to "not written yet",
// ready for the next launch (hipGraph replays included).  The last stage of a chain and the scalar-head partials are put back
// at the start of the next launch: their readers come several hand-offs later.  A poll is bounded; a timed-out poll sets
// MEET_ERR_ROWCHAIN in the host-visible status word: the next synchronising call returns GCRL_ERR_STATE and re-initialises
// every hand-off word (agent.hip: rowtile_reset).
// Stores: agent-scope write-through (sc1) unless the workgroups of the row block found each other on ONE XCD (wave 1
// publishes the workgroup's XCC_ID behind a drained store with its arrival at the row block's monotonic counter, whose
// returned value names the round; all read the same H / 16 words, so all decide alike): then plain stores — the XCD's L2 is
// the point of coherence for its own CUs and serves the sc1 loads — 1.8 instead of 2.2 us per layer of a bare chain.  The
// launch order puts a row block's workgroups on one XCD (workgroup w runs on XCD w % 8, tools/xcc_census.hip) and keeps a
// tile's workgroup index the same in every launch; a different placement costs speed, never correctness.  Nothing crosses
// roles.  All workgroups of the launch must be resident at once (rowtile_ok).
#include "rowchain.h"
#include "meet.h"

#include <algorithm>
#include <cstdlib>

namespace gcrl {
namespace {

constexpr int kTileThreads = 256;
constexpr int kSc1 = 16;
enum { ROLE_P = 0, ROLE_K = 1 };

__device__ inline unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // HW_REG_XCC_ID[3:0]

// development: -DGCRL_RT_STAMPS leaves device-clock stamps of workgroup (row block 0, column block 0) of each role (tools/rt_stamps.py)
#ifdef GCRL_RT_STAMPS
#ifndef GCRL_RT_STAMPS_MASK
#define GCRL_RT_STAMPS_MASK 3   // which launches: 3 = both roles in one launch, 1 = actor phase alone, 2 = critic phase alone
#endif
__device__ unsigned long long g_rt_stamps[3][64];   // [role][stamp] (third row unused)
#define RT_STAMP(i) do { if (threadIdx.x == 0 && c.rb == 0 && c.cb == GCRL_RT_STAMPS && t.role_mask == GCRL_RT_STAMPS_MASK) g_rt_stamps[role][i] = wall_clock64(); } while (0)
#else
#define RT_STAMP(i) do { } while (0)
#endif
#ifdef GCRL_RT_STAMPS
#define RT_STAMPW() do { RT_STAMP(sidx < 62 ? sidx : 62); ++sidx; } while (0)
#else
#define RT_STAMPW() do { } while (0)
#endif
#define RT_STAMPR() RT_STAMPW()
#define RT_STAMPA() RT_STAMPW()

// "not written yet": the hand-off buffers hold this word until their tile arrives (a NaN pattern no arithmetic produces:
// hardware NaNs are 0x7FC00000 / 0xFFC00000)
constexpr unsigned int kSent = 0xFFFFFFFFu;

struct TileCtx {
  int rb, cb, wave, lane, li, lg, ncb;
  bool plain;                  // wave 0: plain stores (row block on one XCD)
  unsigned long long target;   // counters' value once every workgroup of the row block has arrived this launch
  unsigned int* status;
  bool failed;
};

// a timed-out wait: the host-visible status word (behind this wave's stores: tools/check_release_isa.py reads every atomic that
// follows write-through stores as an arrival)
__device__ inline void report_timeout(const TileCtx& c) {
  drain_stores();
  if (c.lane == 0 && c.status) __hip_atomic_fetch_or(c.status, MEET_ERR_ROWCHAIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a wave waits for the stage's arrivals (every lane polls the same word: one request)
__device__ inline void tile_wait(TileCtx& c, const unsigned long long* ctr) {
  if (c.failed) return;
  int spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.target) {
    if (++spins >= kMeetSpinMax) {
      report_timeout(c);
      c.failed = true;
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

__device__ inline void tile_store(const TileCtx& c, float* buf, long long nfloats, int off_floats, v4f v) {
  v4u o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  if (c.failed && o[0] != kSent) o = (v4u){0x7FC00000u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u};   // after a timed-out wait: NaN, not a plausible number
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(buf, nfloats);
  if (c.plain) __builtin_amdgcn_raw_buffer_store_b128(o, rs, off_floats * 4, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(o, rs, off_floats * 4, 0, kSc1);
}
__device__ inline void tile_store1(const TileCtx& c, float* buf, long long nfloats, int off_floats, float v, bool through) {
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(buf, nfloats);
  if (c.plain && !through) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off_floats * 4, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off_floats * 4, 0, kSc1);
}

// B operand of a tile: columns cb*16 .. +15 of M (row-major [J][ldm]), rows of this wave's k share, k-permuted like the A
// fragments (MFMA (t, s) consumes k = kb + 16 t + 4 lg + s).  Rows >= J read 0 (descriptor extent).
__device__ inline void load_b(float (&bw)[4][4], const TileCtx& c, const float* M, int ldm, int J, int kper) {
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(M, (long long)J * ldm);
  const int kb = c.wave * kper, nt = kper >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      bw[t][s] = (t < nt && kb + 16 * t < J) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ((kb + 16 * t + 4 * c.lg + s) * ldm + c.cb * 16 + c.li) * 4, 0, 0)) : 0.f;
}
// B operand of a small head: out[row][o] = sum_k x[row][k] * W[o * ldw + k], o < n_out <= 16 (k contiguous: 16-byte loads)
__device__ inline void load_bh(float (&bw)[4][4], const TileCtx& c, const float* W, int ldw, int n_out, int J, int kper) {
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(W, (long long)n_out * ldw);
  const int kb = c.wave * kper, nt = kper >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    v4u v = {0u, 0u, 0u, 0u};
    if (t < nt && kb + 16 * t < J && c.li < n_out) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (c.li * ldw + kb + 16 * t + 4 * c.lg) * 4, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) bw[t][s] = __uint_as_float(v[s]);
  }
}
// A operand from a handed-off [B][H] buffer: rows rb*16 + li, k = kb + 16 t + 4 lg .. + 3 (sc1: past the CU's L1)
__device__ inline bool load_a(v4f (&af)[4], const TileCtx& c, const float* X, long long nfloats, int H, int kper) {
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(X, nfloats);
  const int kb = c.wave * kper, nt = kper >> 4;
  bool missing = false;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    v4u v = {0u, 0u, 0u, 0u};
    if (t < nt) v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((c.rb * 16 + c.li) * H + kb + 16 * t + 4 * c.lg) * 4, 0, kSc1);
    missing = missing || v[0] == kSent || v[1] == kSent || v[2] == kSent || v[3] == kSent;
    af[t] = (v4f){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  }
  return missing;
}
// A operand from the workgroup's staged input rows (LDS [16][ldsx], zero padded to a multiple of 16 columns)
__device__ inline void load_a_lds(v4f (&af)[4], const TileCtx& c, const float* xs, int ldsx, int J16, int kper) {
  const int kb = c.wave * kper, nt = kper >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    af[t] = (t < nt && kb + 16 * t < J16) ? *(const v4f*)(xs + c.li * ldsx + kb + 16 * t + 4 * c.lg) : (v4f){0.f, 0.f, 0.f, 0.f};
}
// transposed product: lane li = output row, acc[r] = column 4 lg + r
__device__ inline v4f tile_mma(const float (&bw)[4][4], const v4f (&af)[4], int kper) {
  const int nt = kper >> 4;
  v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nt) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][0], af[t][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][1], af[t][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][2], af[t][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][3], af[t][3], acc1, 0, 0, 0);
    }
  return acc0 + acc1;
}
// the four waves' partial tiles meet in LDS; wave 0 gets the sum: lane -> row lane >> 2, columns 4 (lane & 3) .. + 3
__device__ inline v4f tile_reduce(const TileCtx& c, float* part, int& pb, v4f acc) {
  float* p = part + pb * 1024;
  *(v4f*)(p + c.wave * 256 + c.li * 16 + 4 * c.lg) = acc;
  __syncthreads();
  v4f v = {0.f, 0.f, 0.f, 0.f};
  if (c.wave == 0) {
    v = *(const v4f*)(p + 4 * c.lane);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *(const v4f*)(p + w * 256 + 4 * c.lane);
  }
  pb ^= 1;
  return v;
}

// sum over the row block's H / 16 column-block partials of a scalar head, [cb][16 rows] floats, in a fixed order (the same in
// every workgroup of the row block): lane -> row lane >> 2; its four lanes take four column blocks each
__device__ inline float sum_partials(TileCtx& c, const float* qp, long long nfloats, int base) {
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(qp, nfloats);
  const int row = c.lane >> 2, j = c.lane & 3;
  float v[4];   // (H <= 256: at most 16 column blocks; every load requested before the first is used)
  for (int spins = 0;; ++spins) {
    bool missing = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cbi = 4 * i + j;
      const unsigned int w = cbi < c.ncb ? __builtin_amdgcn_raw_buffer_load_b32(rs, (base + cbi * 16 + row) * 4, 0, kSc1) : 0u;
      missing = missing || w == kSent;
      v[i] = __uint_as_float(w);
    }
    if (c.failed || !__any(missing)) break;
    if (spins >= kMeetSpinMax) {
      report_timeout(c);
      c.failed = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = v[i] + __shfl_xor(v[i], 1, 64);
    t = t + __shfl_xor(t, 2, 64);
    s += t;
  }
  return s;
}

__global__ __launch_bounds__(kTileThreads) void rowtile_ddpg_kernel(RowTileArgs t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const RowChainArgs& a = t.rc;
  const int H = a.critic[0].H, L = a.critic[0].L, S = a.S, A = a.A, B = a.B;
  const int ncb = H >> 4, nrb = B >> 4;
  const int tid = threadIdx.x;
  TileCtx c;
  c.lane = tid & 63; c.li = c.lane & 15; c.lg = c.lane >> 4; c.ncb = ncb;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.plain = false; c.failed = false; c.status = t.status; c.target = 0;
  // role and tile.  A row block's workgroups sit on one XCD when the dispatcher deals workgroups round-robin over the 8 XCDs
  // (always the two-role order, whichever roles run: a tile's workgroup index — and with it its XCD — is then the same in
  // every launch of the handle; workgroups of a role that does not run exit at once)
  const int units = 2 * nrb;
  int u;
  if ((units & 7) == 0 && !t.force_linear) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    u = (slot / ncb) * 8 + xcd;
    c.cb = slot % ncb;
  } else {
    u = blockIdx.x / ncb;
    c.cb = blockIdx.x % ncb;
  }
  const int role = u & 1;
  c.rb = u >> 1;
  if (blockIdx.x == 0 && tid == 0) { a.cb->cur_b = a.cb->cur; a.cb->prev_b = a.cb->prev; }
  if (!((t.role_mask >> role) & 1)) return;
  const int row0 = c.rb * 16;
  const long long BH = (long long)B * H;
  // LDS
  const int ldsx = t.ldsx;
  float* part = lds;                       // [2 buffers][2 chains][4 waves][256]
  float* xs = part + 4096;                 // [16][ldsx] staged input rows (+ the action columns): s (P) / s' (K)
  float* xs2 = xs + 16 * ldsx;             // [16][ldsx] K: the rows [s | a] of the online critic
  float* hd = xs2 + 16 * ldsx;             // [16][16] a head's result
  float* as_ = hd + 256;                   // [16][16] tanh(actor head) (P)
  float* own = as_ + 256;                  // [2L][256] own tiles (wave 0; lane-private 16-byte slots)
  __shared__ unsigned int s_plain, s_fail;
  int pb = 0;
  int sidx = 4;
  (void)sidx;
  if (a.clk && tid == 0 && c.rb == 0 && c.cb == 0) atomicMin(&a.clk[0], (unsigned long long)wall_clock64());
  RT_STAMP(0);

  const int kperx = t.kperx, kperh = H >> 2;
  const int orow = c.lane >> 2, ocol = c.cb * 16 + 4 * (c.lane & 3);   // wave 0: own tile element group
  const int ooff = (row0 + orow) * H + ocol;
  // hand-off buffers of this role: tile stages [4L][B][H]; partials [3 kinds][B/16][H/16][16]
  float* xb = t.xb + (long long)role * 4 * L * BH;
  auto X = [&](int stage) { return xb + (long long)stage * BH; };
  const long long qn = 3LL * nrb * ncb * 16;
  const int qbase_t = (0 * nrb + c.rb) * ncb * 16, qbase_o = (1 * nrb + c.rb) * ncb * 16, qbase_p = (2 * nrb + c.rb) * ncb * 16;
  const v4f sent4 = {__uint_as_float(kSent), __uint_as_float(kSent), __uint_as_float(kSent), __uint_as_float(kSent)};

  // first-layer weights: requested before anything that depends on the step's control block
  float bw[4][4], bw2[4][4];
  v4f af[4], af2[4];
  v4f bias = {0.f, 0.f, 0.f, 0.f}, bias2 = {0.f, 0.f, 0.f, 0.f};
  v4f acc, acc2 = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f}, v2 = {0.f, 0.f, 0.f, 0.f};
  const RowNet& cr = a.critic[0];
  {
    const RowNet& n0 = role == ROLE_P ? a.actor : a.tactor;
    load_b(bw, c, n0.Wt + n0.wt[0], H, n0.jpad0, kperx);
    if (c.wave == 0) bias = *(const v4f*)(n0.P + n0.b[0] + ocol);
    if (role == ROLE_K) {
      load_b(bw2, c, cr.Wt + cr.wt[0], H, cr.jpad0, kperx);
      if (c.wave == 0) bias2 = *(const v4f*)(cr.P + cr.b[0] + ocol);
    }
  }
  const StepCtrl sc = role == ROLE_P ? *a.cur_p : *a.cur_k;
  // Which store form?  Wave 1 publishes this workgroup's XCD (a drained write-through store) and arrives at the row block's counter
  // (monotonic: the value the add returns names the round); after layer 0 it waits for the row block's arrivals (bounded) and
  // reads the H / 16 words: all on one XCD -> plain stores.
  unsigned long long* ctr = t.ctr + (long long)(role * nrb + c.rb) * 16;
  unsigned int* xid = t.xid + (long long)(role * nrb + c.rb) * 32;
  if (tid == 64) {
    __hip_atomic_store(xid + c.cb, xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain_stores();
    const unsigned long long t0 = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.target = (t0 / (unsigned long long)ncb + 1ull) * (unsigned long long)ncb;
  }
  c.target = __shfl(c.target, 0, 64);   // (wave 1: lane 0's value; other waves: 0, unused)
  if (tid == 0) { s_plain = 0u; s_fail = 0u; }
  // last launch's leftovers -> "not written yet": the final tile stage of the chain (nobody could tell when its readers were
  // done) and the scalar-head partials.  Their readers of THIS launch come several hand-offs later, each of which needs a tile
  // of this workgroup that it stores after these resets have been acknowledged (the waits for its loads below).
  if (c.wave == 0) {
    tile_store(c, X(3 * L + 1), BH, ooff, sent4);   // (write-through: the store form is not decided yet)
    if ((c.lane & 3) == 0) {
      if (role == ROLE_P) tile_store1(c, t.qpart, qn, qbase_p + c.cb * 16 + orow, __uint_as_float(kSent), true);
      else {
        tile_store1(c, t.qpart, qn, qbase_t + c.cb * 16 + orow, __uint_as_float(kSent), true);
        tile_store1(c, t.qpart, qn, qbase_o + c.cb * 16 + orow, __uint_as_float(kSent), true);
      }
    }
  }
  // input rows -> LDS (zero padded): P: s;  K: s' and [s | a]
  {
    const int W16 = t.w16;
    const float* rows = (role == ROLE_P ? a.sa : a.nsa) + (long long)sc.batch_slot * a.slot_x + (long long)row0 * a.ldx;
    const float* rows2 = a.sa + (long long)sc.batch_slot * a.slot_x + (long long)row0 * a.ldx;
    for (int i = tid; i < 16 * W16; i += kTileThreads) {
      const int r = i / W16, cc = i - r * W16;
      xs[r * ldsx + cc] = cc < S ? rows[(long long)r * a.ldx + cc] : 0.f;
      if (role == ROLE_K) xs2[r * ldsx + cc] = cc < S + A ? rows2[(long long)r * a.ldx + cc] : 0.f;
    }
  }

  auto fwd_epi = [&](v4f x, v4f b) {
    x += b;
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = act_apply(x[q], EPI_LEAKY);
    return x;
  };
  auto bwd_epi = [&](v4f x, v4f h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] *= act_deriv(h[q], MUL_DLEAKY);
    return x;
  };
  auto own_at = [&](int i) { return (v4f*)(own + i * 256 + 4 * c.lane); };
  auto save = [&](float* buf, v4f x) {
    if (c.failed) x = (v4f){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    *(v4f*)(buf + ooff) = x;
  };   // for the dW launch: a plain store, visible at the kernel boundary
  // The hand-off: the DATA is the flag.  Every wave loads its fragments of the row block's tiles (sc1: past the CU's L1) until none
  // of its words is the "not written yet" pattern; the workgroup votes and repeats together.  Once a set of stages has arrived,
  // every workgroup of the row block has stored its tiles of that set — which it did AFTER reading the previous set — so the
  // previous set's own tiles go back to "not written yet" for the next launch.
  // (No workgroup vote per poll: a wave polls for itself — tools/microbench_rowtile.hip: 1.79 vs 2.13 us per layer — and the
  // reset of the previous set waits for the barrier of the partial-sum exchange, behind which every wave has its fragments.)
  float* prev1 = nullptr; float* prev2 = nullptr;      // the set consumed before the current one
  float* cur1 = nullptr; float* cur2 = nullptr;
  bool need1 = false, need2 = false;
  auto consume_begin = [&](float* X1, float* X2) {     // first poll: requested BEFORE the caller's weight loads
    for (int i = 0; i < t.sleep_first; ++i) __builtin_amdgcn_s_sleep(1);
    cur1 = X1; cur2 = X2;
    need1 = __any(load_a(af, c, X1, BH, H, kperh));
    need2 = X2 ? __any(load_a(af2, c, X2, BH, H, kperh)) : false;
  };
  auto consume_end = [&]() {
    for (int spins = 0; (need1 || need2) && !c.failed; ++spins) {
      if (spins >= kMeetSpinMax) {
        report_timeout(c);
        c.failed = true;
        if (c.lane == 0) s_fail = 1u;   // (wave 0 poisons what this workgroup hands on and saves: a timed-out step must not pass for a result)
        break;
      }
      for (int i = 0; i < t.sleep_poll; ++i) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      if (need1) need1 = __any(load_a(af, c, cur1, BH, H, kperh));
      if (need2) need2 = __any(load_a(af2, c, cur2, BH, H, kperh));
    }
    RT_STAMPW();
  };
  // the four waves' partial tiles of ONE or TWO chains meet in LDS (one barrier); wave 0 gets the sums
  auto reduce = [&](bool two) {
    float* p = part + pb * 2048;
    *(v4f*)(p + c.wave * 256 + c.li * 16 + 4 * c.lg) = acc;
    if (two) *(v4f*)(p + 1024 + c.wave * 256 + c.li * 16 + 4 * c.lg) = acc2;
    __syncthreads();
    if (c.wave == 0) {
      c.plain = s_plain != 0;   // (wave 1's decision, once it is made: the stores before it go write-through)
      if (s_fail) c.failed = true;
      if (cur1 != prev1 || cur2 != prev2) {   // a new set has been consumed by every wave: the one before it goes back to "not written yet"
        if (prev1) tile_store(c, prev1, BH, ooff, sent4);
        if (prev2) tile_store(c, prev2, BH, ooff, sent4);
      }
      v = *(const v4f*)(p + 4 * c.lane);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *(const v4f*)(p + w * 256 + 4 * c.lane);
      if (two) {
        v2 = *(const v4f*)(p + 1024 + 4 * c.lane);
#pragma unroll
        for (int w = 1; w < 4; ++w) v2 += *(const v4f*)(p + 1024 + w * 256 + 4 * c.lane);
      }
    }
    prev1 = cur1; prev2 = cur2;
    pb ^= 1;
    RT_STAMPR();
  };
  // wave 1, after layer 0: the row block's first arrivals are in (bounded wait) -> one XCD?  Wave 0 reads the answer behind the
  // next barrier.
  auto decide_plain = [&]() {
    if (c.wave != 1) return;
    tile_wait(c, ctr);
    const unsigned int mine = c.lane < ncb ? __hip_atomic_load(xid + c.lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const unsigned int first = __shfl(mine, 0, 64);
    const bool same = c.lane >= ncb || (mine == first && mine == xcc_id() + 1u);
    const bool plain = !t.force_sc1 && !c.failed && __all(same);
    if (c.lane == 0) s_plain = plain ? 1u : 0u;
  };
  // hidden layers l_beg .. l_end-1 of `net`: layer l reads stage st0 + l - 1 and hands its tile on as stage st0 + l; own tiles kept
  // at own[own0 + l]; sv: where the dW launch finds the layer's output ([L][B][H]) or null
  auto hidden_rest = [&](const RowNet& net, int st0, float* sv, int own0, int l_beg, int l_end) {
    for (int l = l_beg; l < l_end; ++l) {
      consume_begin(X(st0 + l - 1), nullptr);
      load_b(bw, c, net.Wt + net.wt[l], H, H, kperh);
      if (c.wave == 0) bias = *(const v4f*)(net.P + net.b[l] + ocol);
      consume_end();
      acc = tile_mma(bw, af, kperh);
      reduce(false);
      if (c.wave == 0) {
        v = fwd_epi(v, bias);
        *own_at(own0 + l) = v;
        tile_store(c, X(st0 + l), BH, ooff, v);
        if (sv) save(sv + (long long)l * BH, v);
        RT_STAMPA();
      }
    }
  };
  // input-gradient chain: g[l-1] = (g[l] . W[l]) * act'(h[l-1]) for l = L-1 .. 1 (own tiles of h at own[own0 + l - 1]); stage
  // st0 + l holds g[l]
  auto grad_rest = [&](const RowNet& net, int st0, float* sv, int own0) {
    for (int l = L - 1; l >= 1; --l) {
      consume_begin(X(st0 + l), nullptr);
      load_b(bw, c, net.P + net.w[l], H, H, kperh);
      consume_end();
      acc = tile_mma(bw, af, kperh);
      reduce(false);
      if (c.wave == 0) {
        v = bwd_epi(v, *own_at(own0 + l - 1));
        if (l > 1) tile_store(c, X(st0 + l - 1), BH, ooff, v);
        if (sv) save(sv + (long long)(l - 1) * BH, v);
        RT_STAMPA();
      }
    }
  };
  // a head on a handed-off stage: out[row][o] -> wave 0's v (lane: row, outputs 4 (lane & 3) .. + 3)
  auto head_on = [&](const float* W, int n_out, int stage) {
    consume_begin(X(stage), nullptr);
    load_bh(bw, c, W, H, n_out, H, kperh);
    consume_end();
    acc = tile_mma(bw, af, kperh);
    reduce(false);
  };
  // partial of a scalar head over this tile's 16 columns -> qp[cb][row] (read by workgroups of the SAME role and row block)
  auto head_partial = [&](v4f h, const float* Wh, int base) {
    const v4f w = *(const v4f*)(Wh + ocol);
    float s = ((h[0] * w[0] + h[1] * w[1]) + h[2] * w[2]) + h[3] * w[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((c.lane & 3) == 0) tile_store1(c, t.qpart, qn, base + c.cb * 16 + orow, s, false);
  };

  if (role == ROLE_P) {
    // tile stages: l: hA[l];  L + l: critic activations;  2L + l: critic gradients;  3L + l: gA[l]
    __syncthreads();
    load_a_lds(af, c, xs, ldsx, t.w16, kperx);
    acc = tile_mma(bw, af, kperx);
    RT_STAMP(1);
    reduce(false);
    if (c.wave == 0) {
      v = fwd_epi(v, bias);
      *own_at(0) = v;
      tile_store(c, X(0), BH, ooff, v);   // (write-through: the store form is not decided yet)
      save(a.hA, v);
      RT_STAMPA();
    }
    decide_plain();
    // layer 1
    {
      consume_begin(X(0), nullptr);
      load_b(bw, c, a.actor.Wt + a.actor.wt[1], H, H, kperh);
      if (c.wave == 0) bias = *(const v4f*)(a.actor.P + a.actor.b[1] + ocol);
      consume_end();
      acc = tile_mma(bw, af, kperh);
      reduce(false);
      if (c.wave == 0) {
        v = fwd_epi(v, bias);
        *own_at(1) = v;
        tile_store(c, X(1), BH, ooff, v);
        save(a.hA + BH, v);
        RT_STAMPA();
      }
    }
    hidden_rest(a.actor, 0, a.hA, 0, 2, L);
    // actor head -> a = tanh(.) -> critic layer 0 on [s | a]
    load_b(bw2, c, cr.Wt + cr.wt[0], H, cr.jpad0, kperx);
    const v4f bias_c0 = *(const v4f*)(cr.P + cr.b[0] + ocol);
    head_on(a.actor.P + a.actor.w[L], A, L - 1);
    if (c.wave == 0) {
      const int o0 = 4 * (c.lane & 3);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (o0 + q < A) {
          const float act = tanhf(v[q] + a.actor.P[a.actor.b[L] + o0 + q]);
          as_[orow * 16 + o0 + q] = act;
          xs[orow * ldsx + S + o0 + q] = act;
        }
    }
    __syncthreads();
    load_a_lds(af, c, xs, ldsx, t.w16, kperx);
    acc = tile_mma(bw2, af, kperx);
    reduce(false);
    if (c.wave == 0) {
      v = fwd_epi(v, bias_c0);
      *own_at(L) = v;
      tile_store(c, X(L), BH, ooff, v);
      RT_STAMPA();
    }
    hidden_rest(cr, L, nullptr, L, 1, L - 1);
    // last critic layer: its tile stays here — Q(s, pi(s)) partial (a metric), then the head's backward: upstream -1/B
    consume_begin(X(L + L - 2), nullptr);
    load_b(bw, c, cr.Wt + cr.wt[L - 1], H, H, kperh);
    if (c.wave == 0) bias = *(const v4f*)(cr.P + cr.b[L - 1] + ocol);
    consume_end();
    acc = tile_mma(bw, af, kperh);
    reduce(false);
    if (c.wave == 0) {
      v = fwd_epi(v, bias);
      const float* Wh = cr.P + cr.w[L];
      head_partial(v, Wh, qbase_p);
      const v4f w = *(const v4f*)(Wh + ocol);
      const float gb = -1.0f / (float)B;
      v4f g;
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = (0.f + gb * w[q]) * act_deriv(v[q], MUL_DLEAKY);
      tile_store(c, X(2 * L + L - 1), BH, ooff, g);
      RT_STAMPA();
    }
    // critic input-gradient chain; column block 0 also finishes Q(s, pi(s))
    for (int l = L - 1; l >= 1; --l) {
      consume_begin(X(2 * L + l), nullptr);
      load_b(bw, c, cr.P + cr.w[l], H, H, kperh);
      consume_end();
      acc = tile_mma(bw, af, kperh);
      reduce(false);
      if (c.wave == 0) {
        v = bwd_epi(v, *own_at(L + l - 1));
        tile_store(c, X(2 * L + l - 1), BH, ooff, v);
        RT_STAMPA();
        if (l == L - 1 && c.cb == 0) {
          const float q2v = sum_partials(c, t.qpart, qn, qbase_p);
          if ((c.lane & 3) == 0) a.q2[row0 + orow] = q2v + cr.P[cr.b[L]];
        }
      }
    }
    // action gradient da = g0 . W0[:, S + o] (rows S.. of the [in][out] copy), through the tanh, then the actor head's backward
    head_on(cr.Wt + cr.wt[0] + (long long)S * H, A, 2 * L);
    if (c.wave == 0) {
      const int o0 = 4 * (c.lane & 3);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (o0 + q < A) {
          const float g = v[q] * act_deriv(as_[orow * 16 + o0 + q], MUL_DTANH);
          hd[orow * 16 + o0 + q] = g;
          if (c.cb == 0) a.dz[(long long)(row0 + orow) * a.Apad + o0 + q] = g;
        }
    }
    __syncthreads();
    if (c.wave == 0) {
      const float* Wh = a.actor.P + a.actor.w[L];
      v4f s = {0.f, 0.f, 0.f, 0.f};
      for (int o = 0; o < A; ++o) s += hd[orow * 16 + o] * *(const v4f*)(Wh + (long long)o * H + ocol);
      v = bwd_epi(s, *own_at(L - 1));
      tile_store(c, X(3 * L + L - 1), BH, ooff, v);
      save(a.gA + (long long)(L - 1) * BH, v);
      RT_STAMPA();
    }
    grad_rest(a.actor, 3 * L, a.gA, 0);
  } else {
    // K: the target chain (target actor -> a' -> target critic -> partials of Q') and the online critic's forward are independent:
    // their layers go through the hand-off cycle TOGETHER (two tiles per cycle) while both have layers left.
    // tile stages: l: target actor, L + l: target critic (l < L - 1);  2L + l: hC[l] (l < L - 1);  3L + l: gC[l]
    const RowNet& tc = a.tcritic[0];
    float rr = 0.f, dd = 0.f;
    if (c.wave == 0) {
      rr = a.rbuf[(long long)sc.batch_slot * a.slot_rd + row0 + orow];
      dd = a.dbuf[(long long)sc.batch_slot * a.slot_rd + row0 + orow];
    }
    __syncthreads();
    load_a_lds(af, c, xs, ldsx, t.w16, kperx);
    load_a_lds(af2, c, xs2, ldsx, t.w16, kperx);
    acc = tile_mma(bw, af, kperx);
    acc2 = tile_mma(bw2, af2, kperx);
    RT_STAMP(1);
    reduce(true);
    const float* Whc = cr.P + cr.w[L];
    if (c.wave == 0) {
      v = fwd_epi(v, bias);
      v2 = fwd_epi(v2, bias2);
      *own_at(0) = v2;
      tile_store(c, X(0), BH, ooff, v);   // (write-through: the store form is not decided yet)
      tile_store(c, X(2 * L), BH, ooff, v2);
      save(a.hC, v2);
      RT_STAMPA();
    }
    decide_plain();
    for (int l = 1; l < L; ++l) {
      consume_begin(X(l - 1), X(2 * L + l - 1));
      load_b(bw, c, a.tactor.Wt + a.tactor.wt[l], H, H, kperh);
      load_b(bw2, c, cr.Wt + cr.wt[l], H, H, kperh);
      if (c.wave == 0) { bias = *(const v4f*)(a.tactor.P + a.tactor.b[l] + ocol); bias2 = *(const v4f*)(cr.P + cr.b[l] + ocol); }
      consume_end();
      acc = tile_mma(bw, af, kperh);
      acc2 = tile_mma(bw2, af2, kperh);
      reduce(true);
      if (c.wave == 0) {
        v = fwd_epi(v, bias);
        v2 = fwd_epi(v2, bias2);
        *own_at(l) = v2;
        tile_store(c, X(l), BH, ooff, v);
        if (l < L - 1) tile_store(c, X(2 * L + l), BH, ooff, v2);
        else head_partial(v2, Whc, qbase_o);
        save(a.hC + (long long)l * BH, v2);
        RT_STAMPA();
      }
    }
    // (wave 0 keeps the online critic's last tile in v2)
    const v4f h3 = v2;
    // target actor head -> a' = tanh(.) -> target critic layer 0 on [s' | a']
    load_b(bw2, c, tc.Wt + tc.wt[0], H, tc.jpad0, kperx);
    const v4f bias_c0 = *(const v4f*)(tc.P + tc.b[0] + ocol);
    head_on(a.tactor.P + a.tactor.w[L], A, L - 1);
    if (c.wave == 0) {
      const int o0 = 4 * (c.lane & 3);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (o0 + q < A) xs[orow * ldsx + S + o0 + q] = tanhf(v[q] + a.tactor.P[a.tactor.b[L] + o0 + q]);
    }
    __syncthreads();
    load_a_lds(af, c, xs, ldsx, t.w16, kperx);
    acc = tile_mma(bw2, af, kperx);
    reduce(false);
    if (c.wave == 0) {
      v = fwd_epi(v, bias_c0);
      tile_store(c, X(L), BH, ooff, v);
      RT_STAMPA();
    }
    hidden_rest(tc, L, nullptr, L, 1, L - 1);
    consume_begin(X(L + L - 2), nullptr);
    load_b(bw, c, tc.Wt + tc.wt[L - 1], H, H, kperh);
    if (c.wave == 0) bias = *(const v4f*)(tc.P + tc.b[L - 1] + ocol);
    consume_end();
    acc = tile_mma(bw, af, kperh);
    reduce(false);
    if (c.wave == 0) {
      v = fwd_epi(v, bias);
      head_partial(v, tc.P + tc.w[L], qbase_t);
      RT_STAMPA();
      // q of the rows, Q' of the rows, TD target, loss gradient (src/agent.py:1311-1317)
      const float qv = sum_partials(c, t.qpart, qn, qbase_o) + cr.P[cr.b[L]];
      const float tq = sum_partials(c, t.qpart, qn, qbase_t) + tc.P[tc.b[L]];
      float y = __fadd_rn(rr, __fmul_rn(__fmul_rn(a.gamma, __fsub_rn(1.0f, dd)), tq));
      y = fminf(fmaxf(y, a.clamp_lo), 0.0f);
      const float diff = __fsub_rn(qv, y);
      const float g = (2.0f / (float)B) * diff;
      if (c.cb == 0 && (c.lane & 3) == 0) { a.q[row0 + orow] = qv; a.y[row0 + orow] = y; a.dq[row0 + orow] = g; }
      const v4f w = *(const v4f*)(Whc + ocol);
      v4f gt;
#pragma unroll
      for (int q = 0; q < 4; ++q) gt[q] = (0.f + g * w[q]) * act_deriv(h3[q], MUL_DLEAKY);
      tile_store(c, X(3 * L + L - 1), BH, ooff, gt);
      save(a.gC + (long long)(L - 1) * BH, gt);
      RT_STAMPA();
    }
    grad_rest(cr, 3 * L, a.gC, 0);
  }
  if (a.clk && tid == 0) atomicMax(&a.clk[1], (unsigned long long)wall_clock64());
}

#ifdef GCRL_RT_STAMPS
extern "C" int gcrl_debug_rt_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rt_stamps), sizeof(unsigned long long) * 3 * 64) == hipSuccess ? 0 : -1;
}
#endif
size_t rowtile_lds_bytes(int H, int L, int w16) { return (size_t)(4096 + 2 * 16 * (w16 + 4) + 512 + 2 * L * 256) * sizeof(float); }

}  // namespace

bool rowtile_shape_ok(int B, int H, int L, int S, int A, int C) {
  return C == 1 && H >= 64 && H <= 256 && H % 64 == 0 && B >= 16 && B % 16 == 0 && L >= 2 && L <= kRowMaxLayers && A >= 1 && A <= 16 &&
         ((S + A + 15) & ~15) <= 256;
}

// every workgroup of the launch waits for others: all of them must be resident at once on a device this process has to itself
bool rowtile_ok(int B, int H, int L, int S, int A, int C) {
  if (!rowtile_shape_ok(B, H, L, S, A, C)) return false;
  const int w16 = (S + A + 15) & ~15;
  const size_t lds = rowtile_lds_bytes(H, L, w16);
  return 2LL * (B / 16) * (H / 16) <= meet_capacity((const void*)rowtile_ddpg_kernel, kTileThreads, lds);
}

long long rowtile_ctr_words(int B, int L) { return 2LL * (B / 16) * 16; }   // 64-bit words: one 128-byte line per (role, row block)
long long rowtile_xb_floats(int B, int H, int L) { return 2LL * 4 * L * B * H; }
long long rowtile_part_floats(int B, int H) { return 3LL * (B / 16) * (H / 16) * 16; }

int launch_rowtile_ddpg(hipStream_t st, RowTileArgs t) {
  const RowChainArgs& a = t.rc;
  const int H = a.critic[0].H, L = a.critic[0].L;
  GCRL_CHECK_ARG(rowtile_shape_ok(a.B, H, L, a.S, a.A, a.C) && a.actor.L == L && a.actor.H == H && a.target_kind == TGT_DDPG &&
                     a.loss_kind == LOSS_MSE && !a.given_next && !a.p_critic_only && t.xb && t.qpart && t.ctr && t.xid,
                 "rowtile: unsupported configuration (B=%d H=%d L=%d A=%d C=%d)", a.B, H, L, a.A, a.C);
  t.nroles = 0; t.role_mask = 0;
  static const int dbg_roles = std::getenv("GCRL_RT_ROLES") ? std::atoi(std::getenv("GCRL_RT_ROLES")) : 3;   // timing experiments only (wrong results): 1 = P, 2 = K
  if (a.nblk_p && (dbg_roles & 1)) { t.roles[t.nroles++] = ROLE_P; t.role_mask |= 1 << ROLE_P; }
  if (a.nblk_k && (dbg_roles & 2)) { t.roles[t.nroles++] = ROLE_K; t.role_mask |= 1 << ROLE_K; }
  if (t.nroles == 0) return GCRL_OK;
  static const bool lin = std::getenv("GCRL_RT_LINEAR") != nullptr;   // experiment: never the XCD-aligned workgroup order
  t.force_linear = lin ? 1 : 0;
  static const int sl_first = std::getenv("GCRL_RT_SLEEP_FIRST") ? std::atoi(std::getenv("GCRL_RT_SLEEP_FIRST")) : 0;   // experiment knobs (64-clock units)
  static const int sl_poll = std::getenv("GCRL_RT_SLEEP_POLL") ? std::atoi(std::getenv("GCRL_RT_SLEEP_POLL")) : 1;
  t.sleep_first = sl_first; t.sleep_poll = sl_poll;
  t.nstage = 2 + 4 * L;
  t.w16 = (a.S + a.A + 15) & ~15;
  t.ldsx = t.w16 + 4;
  t.kperx = ((t.w16 / 4 + 15) / 16) * 16;
  const size_t lds = rowtile_lds_bytes(H, L, t.w16);
  const int grid = 2 * (a.B / 16) * (H / 16);
  hipLaunchKernelGGL(rowtile_ddpg_kernel, dim3(grid), dim3(kTileThreads), lds, st, t);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
