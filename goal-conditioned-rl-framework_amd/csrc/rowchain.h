// rowchain.h — row-block MLP primitives: a workgroup keeps kRows batch rows resident in LDS and
// walks them through consecutive Linear layers, so a whole forward+backward chain of the small
// actor / critic networks (src/model.py:18,57,63) is ONE launch instead of one launch per layer.
//
// The step is latency-bound (256-row batches, 256-wide layers: 33 MFLOP per layer), so what
// matters is the number of dependent launches, not MFMA occupancy.  Rows of a batch are
// independent through every forward and dX pass; only dW = G^T X contracts over the batch and
// stays a separate (full-chip) GEMM launch.
//
//   out[r][c] = sum_j x[r][j] * M[j][c]      r < kRows,  M row-major [J][ldm] in global memory
//
// serves both directions: forward uses M = W^T (the [in][out] copy kept next to every H x H
// weight), dX uses M = W (torch's [out][in] layout read as [j = out][c = in]).  Either way lane l
// owns columns 4l..4l+3 of a 256-column chunk and streams ROWS of M with 16-byte loads (a wave
// reads one contiguous 1 KB row per k: perfectly coalesced, every WG reads the same rows -> L2
// hits), the four waves split j four ways and the partial sums meet in LDS.
//
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction.  Block b = lane>>2
// takes A_b[i] from lane 4b+i and B_b[j] from lane 4b+j and accumulates D_b[i][j] in VGPR i of
// lane 4b+j.  Feeding every block the same four activations x[4g+i][k] (lane l supplies row l&3)
// and B = component q of the lane's float4 of M gives, in VGPR i of lane l,
// out[4g+i][4l+q] — 64 FLOP/clk/SIMD, the fp32 MFMA peak, with only FOUR rows per row group
// (the 16x16x4 form needs sixteen rows per tile, i.e. 4x fewer workgroups at B = 256).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_mfma.h"
#include "ops.h"

namespace gcrl {

constexpr int kRowThreads = 256;   // 4 waves: j is split four ways
constexpr int kRowChunk = 256;     // columns per pass (64 lanes x 4)

__device__ inline __amdgpu_buffer_rsrc_t bounded_rsrc(const float* p, long long nfloats) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  const int bytes = __builtin_amdgcn_readfirstlane((int)(nfloats * 4));
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, bytes, 0x00020000);
}

// One wave's k-share of one 256-column chunk.  RG row groups of 4 rows.
//   xs   LDS activations [4*RG][ldx] (16-byte aligned rows)
//   rs   buffer resource over M (reads past the end return 0)
//   jb, je   this wave's j range (multiples of 4)
template <int RG>
__device__ __forceinline__ void rows_matmul_wave(const float* xs, int ldx, __amdgpu_buffer_rsrc_t rs, int ldm, int c0,
                                        int jb, int je, v4f (&acc)[RG][4]) {
  const int lane = threadIdx.x & 63;
  // k per stage; two stages in flight.  A sharp optimum: 16 loads per wave = 64 per CU in flight.  Measured end
  // to end (headline bench): U = 4 -> 8.5k steps/s, 8 -> 17.0k, 12 -> 11.1k, 16 -> 10.6k — past 64 outstanding
  // vector-memory instructions per CU the issue itself stalls, and the in-order wave cannot reach its MFMAs.
  constexpr int U = 8;
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[g][q] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int col_b = (c0 + 4 * lane) * 4;
  auto load = [&](v4u (&w)[U], int j) {
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, col_b + (j + u) * ldm * 4, 0, 0);
  };
  auto compute = [&](const v4u (&w)[U], int j) {
#pragma unroll
    for (int u4 = 0; u4 < U; u4 += 4) {
      if (j + u4 < je) {
        v4f xa[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) xa[g] = *(const v4f*)(xs + (4 * g + (lane & 3)) * ldx + j + u4);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < RG; ++g)
              acc[g][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[g][u], __uint_as_float(w[u4 + u][q]), acc[g][q], 0, 0, 0);
      }
    }
  };
  v4u wa[U], wb[U];
  load(wa, jb);
  for (int j = jb; j < je; j += 2 * U) {
    load(wb, j + U);
    compute(wa, j);
    load(wa, j + 2 * U);
    compute(wb, j + U);
  }
}

// y = act(x . M + bias) [* act'(hprev)] for the block's rows; all kRowThreads threads call it.
//   part   LDS scratch 2 x [4 waves][4*RG][kRowChunk] (two buffers, `pbuf` picks one)
//   ys     LDS output [4*RG][ldy]   (must not alias xs)
//   save   optional global copy of y: row r of the block at save[r*ld_save + c]
//   mulH   dX form: y *= act'(mulH[r*ld_mul + c]) (saved post-activation of the layer below, global)
//   chained (N <= kRowChunk only): wave w finishes exactly the columns that are ITS j-share of the
//          next pass (whose J is this N), so the next pass may start without a second barrier — a
//          wave reads back only what it wrote itself.  `part` alternates between two buffers (a fast
//          wave's next exchange must not overwrite what a slow wave is still summing); whoever reads
//          ys across waves afterwards (heads, element-wise steps) must __syncthreads() first.
template <int RG>
__device__ __forceinline__ void rows_linear(const float* xs, int ldx, int J, const float* M, int ldm, int N, const float* bias,
                                   int epi, float* part, float* ys, int ldy, float* save, long long ld_save,
                                   int rows_valid, const float* mulH = nullptr, long long ld_mul = 0, int mul = MUL_NONE,
                                   bool chained = false, int pbuf = 0) {
  constexpr int R = 4 * RG;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(M, (long long)J * ldm);
  const int per = ((J + 3) / 4 + 3) & ~3;
  const int jb = wave * per, je = min(J, jb + per);
  if (chained) {
    part += pbuf * (4 * R * kRowChunk);
    // this wave's outputs: all rows x its share [cb, ce) of the next pass
    const int pern = ((N + 3) / 4 + 3) & ~3, q4 = pern >> 2;   // float4 items per row
    const int cb = wave * pern;
    v4f eb[RG], eh[RG];
#pragma unroll
    for (int it = 0; it < RG; ++it) {
      const int idx = lane + 64 * it, r = idx / q4, cc = cb + 4 * (idx - r * q4);
      const bool ok = idx < R * q4 && cc < N;
      eb[it] = (bias && ok) ? *(const v4f*)(bias + cc) : (v4f){0.f, 0.f, 0.f, 0.f};
      eh[it] = (mulH && ok && r < rows_valid) ? *(const v4f*)(mulH + (long long)r * ld_mul + cc) : (v4f){0.f, 0.f, 0.f, 0.f};
    }
    v4f acc[RG][4];
    rows_matmul_wave<RG>(xs, ldx, rs, ldm, 0, jb, je, acc);
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(v4f*)(part + ((wave * R + 4 * g + i) * kRowChunk) + 4 * lane) =
            (v4f){acc[g][0][i], acc[g][1][i], acc[g][2][i], acc[g][3][i]};
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RG; ++it) {
      const int idx = lane + 64 * it, r = idx / q4, cc = cb + 4 * (idx - r * q4);
      if (idx < R * q4 && cc < N) {
        v4f v = *(const v4f*)(part + r * kRowChunk + cc);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *(const v4f*)(part + (w * R + r) * kRowChunk + cc);
        v += eb[it];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = act_apply(v[q], epi);
          if (mul != MUL_NONE) v[q] *= act_deriv(eh[it][q], mul);
        }
        *(v4f*)(ys + r * ldy + cc) = v;
        if (save && r < rows_valid) *(v4f*)(save + (long long)r * ld_save + cc) = v;
      }
    }
    return;
  }
  for (int c0 = 0; c0 < N; c0 += kRowChunk) {
    const int c = c0 + 4 * lane;
    // epilogue operands first: their latency hides behind the weight stream
    v4f b = (v4f){0.f, 0.f, 0.f, 0.f}, hm[RG];
    if (bias && c < N) b = *(const v4f*)(bias + c);
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int r = wave + 4 * rr;
      hm[rr] = (mulH && c < N && r < rows_valid) ? *(const v4f*)(mulH + (long long)r * ld_mul + c) : (v4f){0.f, 0.f, 0.f, 0.f};
    }
    v4f acc[RG][4];
    rows_matmul_wave<RG>(xs, ldx, rs, ldm, c0, jb, je, acc);
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(v4f*)(part + ((wave * R + 4 * g + i) * kRowChunk) + 4 * lane) =
            (v4f){acc[g][0][i], acc[g][1][i], acc[g][2][i], acc[g][3][i]};
    __syncthreads();
    if (c < N) {
#pragma unroll
      for (int rr = 0; rr < RG; ++rr) {
        const int r = wave + 4 * rr;
        v4f v = *(const v4f*)(part + (0 * R + r) * kRowChunk + 4 * lane);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *(const v4f*)(part + (w * R + r) * kRowChunk + 4 * lane);
        v += b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = act_apply(v[q], epi);
          if (mul != MUL_NONE) v[q] *= act_deriv(hm[rr][q], mul);
        }
        *(v4f*)(ys + r * ldy + c) = v;
        if (save && r < rows_valid) *(v4f*)(save + (long long)r * ld_save + c) = v;
      }
    }
    __syncthreads();
  }
}

// small heads: out[r][o] = epi(sum_k x[r][k] * W[o*ldw + k] + bias[o]) for o < n_out <= 16, into
// LDS sm[r*16 + o].  A wave owns the (r, o) pairs wave, wave+4, ... and works on all of them AT ONCE:
// G = 64 / pairs lanes per pair (a power of two), lane-strided partial sums, an xor-shuffle reduce
// inside each group, bias + activation by the group leaders (pair after pair, with the actor head's
// tanh in lane 0 each time, a 4-output head cost 3.6 us against 0.6 us for a 1-output one).
template <int RG>
__device__ inline void rows_head(const float* xs, int ldx, int K, const float* W, long long ldw, const float* bias,
                                 int n_out, int epi, float* sm) {
  constexpr int R = 4 * RG;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int P = (R * n_out + 3) >> 2;   // pairs per wave (<= 64)
  int G = 64;
  while (G * P > 64) G >>= 1;
  const int j = lane / G, sub = lane - j * G;
  const int p = wave + 4 * j;
  const bool mine = j < P && p < R * n_out;
  const int r = mine ? p / n_out : 0, o = mine ? p - r * n_out : 0;
  float s = 0.f;
  if (mine) {
    if (((K | (int)ldw) & 3) == 0) {   // 16-byte LDS reads (round 4: the one-float-per-lane loop was 16 dependent rounds of LDS latency)
      for (int k = 4 * sub; k < K; k += 4 * G) {
        const v4f x = *(const v4f*)(xs + r * ldx + k), w = *(const v4f*)(W + (long long)o * ldw + k);
        s += ((x[0] * w[0] + x[1] * w[1]) + x[2] * w[2]) + x[3] * w[3];
      }
    } else {
      for (int k = sub; k < K; k += G) s += xs[r * ldx + k] * W[(long long)o * ldw + k];
    }
  }
  for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (mine && sub == 0) {
    const float v = s + (bias ? bias[o] : 0.f);
    // fp32 tanh here (the GEMM epilogue's EPI_TANH rounds an fp64 tanh: ~1 us for these few lanes)
    sm[r * 16 + o] = epi == EPI_TANH ? tanhf(v) : act_apply(v, epi);
  }
}

// ---------------------------------------------------------------- DDPG step as row-block chains
constexpr int kRowMaxLayers = 8;

struct RowNet {
  const float* P;    // parameters, torch layout (Linear weight [out][in])
  const float* Wt;   // [in][out] copies of the hidden-layer weights (layer 0: zero rows up to jpad0)
  long long w[kRowMaxLayers + 1], b[kRowMaxLayers + 1];  // offsets into P; entry L = the head
  long long wt[kRowMaxLayers];                           // offsets into Wt
  int in, jpad0, H, L, out;
};

// Blocks [0, nblk_k) run the critic phase K of the step described by *cur_k, blocks
// [nblk_k, nblk_k + nblk_p) the actor phase P of the step described by *cur_p (the previous step in
// the software-pipelined schedule, the same one otherwise).  Everything a dW GEMM or the optimiser
// needs afterwards is written to global memory.
struct RowChainArgs {
  const StepCtrl* cur_k; const StepCtrl* cur_p;
  CtrlBlock* cb;   // block 0 refreshes cb->cur_b / prev_b (the optimiser launch reads those)
  RowNet actor, tactor, critic[2], tcritic[2];
  int C;                      // critics in the K role (1 DDPG, 2 TD3); the P role uses critic[0]
  int target_kind, loss_kind; // TGT_DDPG (clamped) / TGT_MIN; LOSS_MSE / LOSS_SMOOTH_L1
  // TD3 target-policy smoothing (src/agent.py:174-179): injected N(0,1) [B][A] or null (device RNG)
  const float* noise; float policy_noise, noise_clamp; unsigned long long seed;
  // SAC (the BatchNorm actor runs outside, src/agent.py:557-570, :516-521): the K role takes the next
  // action from the action columns of nsa (given_next) and subtracts alpha*logp_next in the target
  // (TGT_MIN_ENT); the P role is critic-only (p_critic_only): both critics on the rows
  // [s (from sa) | pi(s) (from pi)], d(-min(q1,q2))/dq, each critic's input gradient w.r.t. the action into dz[c]
  int given_next, p_critic_only;
  const float* logp_next; float alpha;
  const float* pi;   // pi(s) [B][Apad] (P role, critic-only form)
  const float* sa; const float* nsa; const float* rbuf; const float* dbuf;   // + batch_slot * slot_*
  long long slot_x, slot_rd;
  int ldx, ldl, B, S, A, Apad;
  int nblk_k, nblk_p;
  int linear_roles;           // 1 (default): roles in launch order (K's workgroups first); 0: by XCD (GCRL_ROW_XCD_ROLES=1: A/B knob, agent_rowchain.inc)
  float *hC, *gC, *q, *y, *dq;      // K: activations / pre-activation gradients [C][L][B][H], q / dq [C][B], y [B]
  float *hA, *gA, *hC2, *q2, *dz;   // P: ..., critic activations (scratch) [C][L][B][H], Q(s, pi(s)) [C][B], d(pre-tanh) [C][B][Apad]
  float gamma, clamp_lo;
  unsigned long long* clk;   // profiling: {first block start, last block end} in wall_clock64 ticks, or null
  // split form (launch_rowchain_split): the roles of a phase run in DIFFERENT workgroups, two launches per phase
  float* qt;       // [C][B] target-critic outputs (written by the forward launch, read by the backward launch)
  // part 3 (both parts in ONE launch): meeting counters of the row blocks, [2 phases][nblk][32 words] (a 64-bit counter at the
  // head of each 128-byte line, meet.h), zero-initialised; `status`: host-visible word that takes MEET_ERR_ROWCHAIN on a timed-out wait
  unsigned int* bar;
  unsigned int* status;
  // phase 0 of part 3 as producers / consumers (meet.h): the target roles only arrive, the online-critic roles only wait for
  // them (word 4 + 2 * critic of the row block's line: that consumer's own launch count) — admissible WITHOUT all workgroups
  // being resident at once (TD3 at batch 2048: 1 024 workgroups of 8 rows)
  int producers_first;
  // DDPG (rowchain_ddpg_kernel): the critic phase as the two roles above inside the fused launch (needs bar, qt, producers_first)
  int k_split;
};

// Twin-critic phases as role-parallel launches (SAC; TD3 at small batches).  In the fused kernel a workgroup
// walks its rows through BOTH critics and BOTH target critics one after the other (16 dependent layer passes
// at L = 3); the chains of different networks are independent until the TD target / the min-selection, so
//   part 1 (forward)    K: roles [target critic k | online critic k], P: roles [critic k on [s|pi(s)]]
//   part 2 (backward)   K: y, dq, input-gradient chain of critic k;   P: min-selection, chain, action gradient
// run each role in its own workgroups: L passes + (L-1) passes on the critical path, the whole chip busy.
// Same per-row arithmetic in the same order as the fused kernel (bitwise the same results).
//   phase 0 = K (critic phase), 1 = P (actor phase, critic-only form); part 1 | 2
// part 3 = parts 1 and 2 in ONE launch (round 3): what part 2 needs from other workgroups is a handful of scalars per ROW BLOCK
// (the target critics' outputs for the TD target; the other critic's Q for the min-selection), so the role workgroups of a row
// block meet at a counter — published with agent-scope stores, a bounded wait on a generation word (the last arriver resets the
// counter and bumps it: ready for the next launch, graph replays included) — and the online-critic workgroups go straight on to
// their backward chains.  All role x block workgroups of a launch are resident at once (<= 2 per CU at SAC's 512 rows: the
// launcher checks), so a waiting workgroup never keeps an awaited one off the chip; a wait that times out poisons the gradient
// with NaN instead of hanging.  Same per-row arithmetic: bitwise the results of the two-launch form.
// SAC (round 5): the BatchNorm actor's two heads (src/model.py:114-115) and its sampling (:125-141) INSIDE the chain launches — before, one launch
// of 11 us between the actor's last slab launch and the critic phase (sac_heads.h).  A role workgroup that needs actions takes the last hidden
// activation of ITS rows, forms both heads' outputs (rows_head) and samples: the target-critic roles of phase 0 the next action a' (no_grad pass
// on next_state; role 0 also publishes logp_next for the online roles' TD target), the critic roles of phase 1 pi(s) (role 0 also writes what the
// actor's backward and the actor loss read: pi, logp, eps, std, the heads' outputs).  One more workgroup of the phase-0 launch applies the
// BatchNorm running-statistics update the sampling launch used to carry.
struct HeadsFold {
  int on;
  // 1: pi(s) and everything phase 1 reads are formed by one more role of phase 0's launch (a workgroup per row block, dispatched as the critic roles
  // finish: beside the online roles' backward passes) — phase 1 then runs without this record, on the stored pi
  int cur_in_k;
  const float* P;                          // the actor's parameter block
  long long w_mean, b_mean, w_ls, b_ls;    // offsets of the two heads ([A][H] weights, [A] biases)
  const float* h_next; const float* h_cur; // [B][H] last hidden activation of actor(next_state) / actor(state)
  const float* eps_next; const float* eps_cur;   // injected N(0,1) [B][A] or null (counter hash: streams 1 / 2 of the seed, as the sampling launch)
  float* logp_next;                        // [B]
  float* pi; float* logp; float* save_eps; float* save_std; float* head;   // phase 1: [B][Apad], [B], [B][A], [B][A], [B][2*Apad] (mu | ls_raw)
  BnRunning run;                           // layers > 0: the running-statistics rider (phase 0)
};
int launch_rowchain_split(hipStream_t st, const RowChainArgs& a, int rg, int phase, int part, const HeadsFold* hf = nullptr);
bool rowchain_merge_ok(int rg, int ldl, int A, int H, int C, int B);   // part 3 admissible: all its workgroups resident at once on an unshared device

// Batched actor inference (select_action, src/agent.py:1345-1366) as one row-block launch: 4 observation
// rows per workgroup through the actor's hidden layers and its tanh head.
struct RowActArgs {
  RowNet actor;
  const float* obs; int ld_obs;   // [n, >= S]
  float* out; int ld_out;         // [n, >= A]
  int n, S, A, ldl;
  // fused acting entry (gcrl_agent_observe_act): `obs` then holds RAW rows [n][S] = [observation (D) | goal (S - D)]; the
  // first D columns go through (x - mean) / (sqrt(var) + 1e-8), clipped to +-clip (float64, rounded to float32), when
  // nz_mean is set, the goal columns when nzg_mean is; post != 0: out64[n][A] = mode 1: clip(tanh(y) + noise, -1, 1),
  // mode 2 (post == 2): clip(tanh(y), -1, 1), mode 3: y   (select_action's arithmetic, src/agent.py:1345-1366, :253-270)
  const double* nz_mean; const double* nz_var; double nz_clip; int D;
  const double* nzg_mean; const double* nzg_var; double nzg_clip;
  int nz_mode, nzg_mode;   // norm_math.h bits (float32 statistics of loaded normalisers, float64 rows)
  int post; const double* noise; double* out64;
};
int launch_rowchain_act(hipStream_t st, const RowActArgs& a);
// The same with the rows (and the exploration noise) travelling INSIDE the kernel arguments and the float64 actions written
// straight to host-visible (pinned, mapped) memory, followed by one 8-byte flag per workgroup (round 4: one vector-env step of the
// acting side was two staged copies up, a launch, a copy down and a stream synchronisation — 30 us for a 3 us kernel).  The host
// waits for the flags instead of synchronising the stream.  n * S <= kActInlineFloats, n * A <= kActInlineNoise.
constexpr int kActInlineFloats = 640, kActInlineNoise = 96;
struct RowActInline {
  RowActArgs base;                       // obs / noise / out64 are ignored: the inline arrays and out_host are used
  double* out_host;                      // device pointer of the pinned block: [n][A] doubles
  unsigned long long* flag_host;         // ... and of its flags: flag_host[workgroup] = seq once that workgroup's rows are out
  unsigned long long seq;
  int with_noise;
  float obs_inl[kActInlineFloats];
  double noise_inl[kActInlineNoise];
};
int launch_rowchain_act_inline(hipStream_t st, const RowActInline& a);

// Weight-slice form of the DDPG launch (rowtile.hip): a workgroup owns the 16 x 16 tile (row block, column block) of every
// layer of its role's chain; the H / 16 workgroups of a row block hand the layer outputs to each other inside the launch.
struct RowTileArgs {
  RowChainArgs rc;            // networks, batch arrays, outputs (nblk_k / nblk_p != 0: which phases run)
  float* xb;                  // hand-off tile stages [2 roles][4L][B][H], every word 0xFFFFFFFF ("not written yet") between launches
  float* qpart;               // scalar-head partials [3][B/16][H/16][16]: Q' (target chain), q (online critic), Q(s, pi(s)); same convention
  unsigned long long* ctr;    // first-arrival counters [2 roles][B/16][16 words]: one 128-byte line each, zero-initialised, monotonic
  unsigned int* xid;          // [2 roles][B/16][32]: the XCD of each workgroup of a row block (+1)
  unsigned int* status;       // host-visible word: MEET_ERR_ROWCHAIN on a timed-out wait
  int force_sc1;              // never the plain-store form (GCRL_ROWTILE_SC1=1)
  // filled by the launcher
  int roles[3], nroles, nstage, w16, ldsx, kperx, force_linear, role_mask, sleep_first, sleep_poll;
};
bool rowtile_shape_ok(int B, int H, int L, int S, int A, int C);
bool rowtile_ok(int B, int H, int L, int S, int A, int C);   // ... and all 3 * (B/16) * (H/16) workgroups resident at once, device not shared
long long rowtile_ctr_words(int B, int L);                   // 64-bit words
long long rowtile_xb_floats(int B, int H, int L);
long long rowtile_part_floats(int B, int H);
int launch_rowtile_ddpg(hipStream_t st, RowTileArgs t);

// rows per workgroup = 4*rg, rg in {1, 2, 4}
int launch_rowchain_ddpg(hipStream_t st, const RowChainArgs& a, int rg);
size_t rowchain_lds_bytes(int rg, int ldl, int A, int H, int C);

// (re)build the [in][out] copies of one net's hidden-layer weights from its parameter block
int launch_wt_rebuild(hipStream_t st, const RowNet& net, float* Wt);

}  // namespace gcrl
