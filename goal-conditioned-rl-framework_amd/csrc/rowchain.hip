// rowchain.hip — the DDPG critic phase (K) and actor phase (P) as row-block chains: one launch
// walks each block of 4/8/16 batch rows through every forward and input-gradient pass of the
// phase (rowchain.h), leaving only the batch-contracting dW GEMMs and the optimiser to follow.
//
//   K (src/agent.py:1311-1317, critic_update):  target actor -> target critic -> TD target y;
//       online critic on [s|a] (activations saved) -> q; dq = 2(q-y)/B; critic input-gradient
//       chain, every layer's pre-activation gradient saved for dW.
//   P (src/agent.py:1288-1300, actor_update):  actor on s (saved) -> a = tanh(.); critic on
//       [s|a] -> q2 (= -loss*B); critic input-gradient chain down to the action columns;
//       d(pre-tanh) = da*(1-a^2); actor gradient chain, saved for dW.
#include "rowchain.h"
#include "norm_math.h"
#include "meet.h"
#include "sac_select.h"

#include <algorithm>

namespace gcrl {
namespace {

template <int RG>
__device__ inline void load_rows(float* X, int ldl, const float* src, long long ld_src, int ncols, int jpad, int rv) {
  constexpr int R = 4 * RG;
  for (int i = threadIdx.x; i < R * jpad; i += kRowThreads) {
    const int r = i / jpad, c = i - r * jpad;
    X[r * ldl + c] = (r < rv && c < ncols) ? src[(long long)r * ld_src + c] : 0.f;
  }
}

// hidden layers 0..L-1 of `net` on the rows in X0 (jpad0 columns, zero padded); returns the LDS
// buffer holding the last hidden activation.  save: [L][B][H] global, row r0 of the block.
template <int RG>
__device__ inline float* mlp_hidden(const RowNet& net, const float* X0, float* X1, float* X2, int ldl, float* part,
                                    float* save, long long BH, long long row0, int rv, float* last_out = nullptr) {
  const float* in = X0;
  float* out = X1;
  const bool chained = RG == 1 && net.H <= kRowChunk;   // one barrier per layer (rows_linear); measured slower at 8 rows
  for (int l = 0; l < net.L; ++l) {
    if (last_out && l == net.L - 1) out = last_out;   // e.g. kept aside while the ping-pong buffers are reused
    rows_linear<RG>(in, ldl, l == 0 ? net.jpad0 : net.H, net.Wt + net.wt[l], net.H, net.H, net.P + net.b[l], EPI_LEAKY,
                    part, out, ldl, save ? save + l * BH + row0 * net.H : nullptr, net.H, rv, nullptr, 0, MUL_NONE,
                    chained, l & 1);
    in = out;
    out = (out == X1) ? X2 : X1;
  }
  if (chained) __syncthreads();   // the callers read the result across waves
  return (float*)in;
}

// G <- (upstream[r][o] . Whead[o][k]) * act'(h[r][k]) in place over h (LDS), n_up <= 16 head outputs.
// A thread owns four consecutive k (16-byte LDS accesses; H % 4 == 0): one pass for 4 rows x 256 columns (round 4: the
// element-per-thread loop was four dependent rounds of LDS reads, 0.8-1.9 us per call by the kernel's own clock stamps).
// q_out (optional, n_up == 1): ALSO the head's forward value q[r] = sum_k h[r][k] * Whead[k] + q_bias -> q_out[r * 16], formed
// from the same reads — for callers whose upstream gradient does not depend on it (DDPG's actor phase: d(-mean Q)/dq = -1/B),
// which saves the separate head pass.  Needs the H / 4 threads of a row inside one wavefront (H = 64, 128, 256).
template <int RG>
__device__ inline void head_backward(float* h, int ldl, int H, const float* Wh, int n_up, const float* up /* LDS [R][16] */,
                                     float* save, int rv, float* q_out = nullptr, float q_bias = 0.f) {
  constexpr int R = 4 * RG;
  const int H4 = H >> 2;
  for (int i = threadIdx.x; i < R * H4; i += kRowThreads) {
    const int r = i / H4, k = (i - r * H4) << 2;
    const v4f hv = *(const v4f*)(h + r * ldl + k);
    v4f s = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int o = 0; o < n_up; ++o) s += up[r * 16 + o] * *(const v4f*)(Wh + (long long)o * H + k);
    v4f g;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = s[q] * act_deriv(hv[q], MUL_DLEAKY);
    if (q_out) {   // (uniform) the rows of a wave: 64 / H4 of them, H4 lanes each
      const v4f w = *(const v4f*)(Wh + k);
      float qs = ((hv[0] * w[0] + hv[1] * w[1]) + hv[2] * w[2]) + hv[3] * w[3];
      for (int off = H4 >> 1; off > 0; off >>= 1) qs += __shfl_xor(qs, off, 64);
      if ((i - r * H4) == 0) q_out[r * 16] = qs + q_bias;
    }
    *(v4f*)(h + r * ldl + k) = g;
    if (save && r < rv) *(v4f*)(save + (long long)r * H + k) = g;
  }
}
__device__ inline bool head_fusable(int H) { return H == 64 || H == 128 || H == 256; }

// pre-activation gradients of hidden layers L-2..0 from the one of layer L-1 (in G, LDS)
template <int RG>
__device__ inline float* grad_chain(const RowNet& net, float* G, float* X1, float* X2, int ldl, float* part,
                                    const float* hsaved, float* gsave, long long BH, long long row0, int rv) {
  float* in = G;
  const bool chained = RG == 1 && net.H <= kRowChunk;
  for (int l = net.L - 1; l >= 1; --l) {
    float* out = (in == X1) ? X2 : X1;
    rows_linear<RG>(in, ldl, net.H, net.P + net.w[l], net.H, net.H, nullptr, EPI_NONE, part, out, ldl,
                    gsave ? gsave + (l - 1) * BH + row0 * net.H : nullptr, net.H, rv,
                    hsaved + (l - 1) * BH + row0 * net.H, net.H, MUL_DLEAKY, chained, l & 1);
    in = out;
  }
  if (chained) __syncthreads();
  return in;
}

// stage `n` floats of global memory into LDS (all threads)
// Prologue staging in two phases — every global load of the block's inputs is requested (into registers)
// before the first LDS store waits for any of them: the phases used to alternate per region and the
// prologue was seven memory round trips (4.9 us) long.
template <int N>
struct Staged { float v[N]; };

template <int N>
__device__ inline void seg_load(Staged<N>& s, const float* src, int n) {
#pragma unroll
  for (int u = 0; u < N; ++u) { const int i = threadIdx.x + u * kRowThreads; s.v[u] = i < n ? src[i] : 0.f; }
}
template <int N>
__device__ inline void seg_store(const Staged<N>& s, float* dst, const float* src, int n) {
#pragma unroll
  for (int u = 0; u < N; ++u) { const int i = threadIdx.x + u * kRowThreads; if (i < n) dst[i] = s.v[u]; }
  for (int i = threadIdx.x + N * kRowThreads; i < n; i += kRowThreads) dst[i] = src[i];   // oversize regions: the rest directly
}
template <int RG, int N>
__device__ inline void rows_load(Staged<N>& s, const float* src, long long ld_src, int ncols, int jpad, int rv) {
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int i = threadIdx.x + u * kRowThreads, r = i / jpad, c = i - r * jpad;
    s.v[u] = (i < 4 * RG * jpad && r < rv && c < ncols) ? src[(long long)r * ld_src + c] : 0.f;
  }
}
template <int RG, int N>
__device__ inline void rows_store(const Staged<N>& s, float* X, int ldl, const float* src, long long ld_src, int ncols, int jpad, int rv) {
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int i = threadIdx.x + u * kRowThreads, r = i / jpad, c = i - r * jpad;
    if (i < 4 * RG * jpad) X[r * ldl + c] = s.v[u];
  }
  for (int i = threadIdx.x + N * kRowThreads; i < 4 * RG * jpad; i += kRowThreads) {
    const int r = i / jpad, c = i - r * jpad;
    X[r * ldl + c] = (r < rv && c < ncols) ? src[(long long)r * ld_src + c] : 0.f;
  }
}

// (eight loads in flight per thread before the first LDS store: one memory round trip per 2048 floats)
__device__ inline void stage(float* dst, const float* src, int n) {
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * kRowThreads) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * kRowThreads; v[u] = i < n ? src[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * kRowThreads; if (i < n) dst[i] = v[u]; }
  }
}

template <int RG, bool HF = false>
__device__ __forceinline__ void rowchain_split_body(const RowChainArgs& a, int phase, int part, int bid, const HeadsFold& hfr = HeadsFold{});

template <int RG>
__global__ __launch_bounds__(kRowThreads) void rowchain_ddpg_kernel(RowChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  kernarg_warm<sizeof(RowChainArgs)>();
  // k_split (round 4, DDPG): the critic phase K as TWO roles in this launch — its target chain (target actor -> target critic -> Q')
  // and its online critic's forward are independent, so K's critical path shrinks from 11 layer passes to 6 + 2 (the split kernel's
  // phase 0, part 3, producers / consumers form: workgroups [0, nblk_k) the target role, [nblk_k, 2 nblk_k) the online critic, which
  // waits for its rows' Q' only) and the launch is bounded by the actor phase's 10 passes instead.  Same per-row arithmetic.
  if (a.k_split && a.nblk_k) {
    if ((int)blockIdx.x < 2 * a.nblk_k) {
      if (a.clk && threadIdx.x == 0 && blockIdx.x == 0) atomicMin(&a.clk[0], (unsigned long long)wall_clock64());
      rowchain_split_body<RG>(a, 0, 3, (int)blockIdx.x);
      if (a.clk && threadIdx.x == 0) atomicMax(&a.clk[1], (unsigned long long)wall_clock64());
      return;
    }
  }
  const int kblocks = (a.k_split && a.nblk_k) ? 2 * a.nblk_k : a.nblk_k;
  constexpr int R = 4 * RG;
  const int ldl = a.ldl, H = a.critic[0].H, S = a.S, A = a.A, B = a.B;
  float* X0 = lds;
  float* X1 = X0 + R * ldl;
  float* X2 = X1 + R * ldl;
  float* XS = X2 + R * ldl;               // second input rows (K) / last actor activation (P)
  float* part = XS + R * ldl;
  float* sm = part + R * 16 + (RG == 1 ? 2 : 1) * 4 * R * kRowChunk;   // [R][16] head outputs (part[0..R*16): smoothing noise; then 2 exchange buffers)
  float* sm2 = sm + R * 16;               // [R][16] second small array
  float* sm3 = sm2 + R * 16;              // [R][16] reward / done
  float* hw = sm3 + R * 16;               // head weights of the role: [A*H | H | max(A*H, H)], then head biases [16 | 16]
  float* hb = hw + max(max(2 * A + 1, A + 2 * a.C), a.C * (A + 1)) * H;
  // Which role, which row block.  Workgroup w runs on XCD w % 8 and every XCD has its own L2: with the roles in launch order every
  // XCD streamed BOTH roles' weights (PMC: 32 MB HBM-side per launch for 3.3 MB of distinct weights).  When both roles have the
  // same number of row blocks (a multiple of 4) the critic phase K takes XCDs 0-3 and the actor phase P XCDs 4-7: an XCD fetches
  // one role's networks only.  (A wrong guess about the placement costs speed, never correctness.)
  bool role_k = (int)blockIdx.x < kblocks;
  int blk = role_k ? (int)blockIdx.x : (int)blockIdx.x - kblocks;
  if (a.nblk_k == a.nblk_p && (a.nblk_k & 3) == 0 && !a.linear_roles && !a.k_split) {
    const int xcd = (int)blockIdx.x & 7;
    role_k = xcd < 4;
    blk = ((int)blockIdx.x >> 3) * 4 + (xcd & 3);
  }
  const long long row0 = (long long)blk * R;
  const int rv = min(R, B - (int)row0);
  const long long BH = (long long)B * H;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) { a.cb->cur_b = a.cb->cur; a.cb->prev_b = a.cb->prev; }
  if (a.clk && tid == 0 && blk == 0) atomicMin(&a.clk[0], (unsigned long long)wall_clock64());

  // Everything that does not depend on computed data is requested NOW (inputs, rewards, head
  // weights and biases): each of these was a separate exposed memory round trip (~1 us) in the
  // middle of the chain.
  if (role_k) {
    const StepCtrl c = *a.cur_k;
    const int C = a.C;
    const float* ns_rows = a.nsa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    const float* sa_rows = a.sa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    const float* rr = a.rbuf + (long long)c.batch_slot * a.slot_rd + row0;
    const float* dd = a.dbuf + (long long)c.batch_slot * a.slot_rd + row0;
    constexpr int NX = 2 * RG;   // covers row widths up to 128 floats
    const int jp_ns = max(a.tactor.jpad0, a.tcritic[0].jpad0), nc_ns = a.given_next ? S + A : S;
    float* hw_ta = hw; float* hw_tc = hw + A * H; float* hw_c = hw_tc + C * H;
    const float* src_ta = a.tactor.P + a.tactor.w[a.tactor.L];
    // phase 1: request everything
    Staged<NX> s_ns, s_sa;
    rows_load<RG>(s_ns, ns_rows, a.ldx, nc_ns, jp_ns, rv);
    rows_load<RG>(s_sa, sa_rows, a.ldx, S + A, a.critic[0].jpad0, rv);
    float v_r = 0.f, v_d = 0.f, v_lp = 0.f, v_hb = 0.f;
    if (tid < R && tid < rv) { v_r = rr[tid]; v_d = dd[tid]; if (a.logp_next) v_lp = a.logp_next[row0 + tid]; }
    if (tid < A && !a.given_next) v_hb = a.tactor.P[a.tactor.b[a.tactor.L] + tid];
    if (tid >= 32 && tid < 32 + C) v_hb = a.tcritic[tid - 32].P[a.tcritic[tid - 32].b[a.tcritic[tid - 32].L]];
    if (tid >= 64 && tid < 64 + C) v_hb = a.critic[tid - 64].P[a.critic[tid - 64].b[a.critic[tid - 64].L]];
    Staged<8> s_ta, s_tc[2], s_c[2];
    if (!a.given_next) seg_load(s_ta, src_ta, A * H);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < C) {
        seg_load(s_tc[k], a.tcritic[k].P + a.tcritic[k].w[a.tcritic[k].L], H);
        seg_load(s_c[k], a.critic[k].P + a.critic[k].w[a.critic[k].L], H);
      }
    // phase 2: into LDS
    rows_store<RG>(s_ns, X0, ldl, ns_rows, a.ldx, nc_ns, jp_ns, rv);
    rows_store<RG>(s_sa, XS, ldl, sa_rows, a.ldx, S + A, a.critic[0].jpad0, rv);
    if (tid < R) { sm3[tid * 16] = v_r; sm3[tid * 16 + 1] = v_d; sm3[tid * 16 + 2] = v_lp; }
    if (tid < A && !a.given_next) hb[tid] = v_hb;
    if (tid >= 32 && tid < 32 + C) hb[16 + (tid - 32)] = v_hb;
    if (tid >= 64 && tid < 64 + C) hb[18 + (tid - 64)] = v_hb;
    if (!a.given_next) seg_store(s_ta, hw_ta, src_ta, A * H);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < C) {
        seg_store(s_tc[k], hw_tc + k * H, a.tcritic[k].P + a.tcritic[k].w[a.tcritic[k].L], H);
        seg_store(s_c[k], hw_c + k * H, a.critic[k].P + a.critic[k].w[a.critic[k].L], H);
      }
    if (a.target_kind == TGT_MIN && !a.given_next && tid < R * A) {
      // smoothing noise of this block's rows, same draw as td3_smooth_kernel (element i = row*A + j)
      const int r = tid / A, o = tid - r * A;
      const long long i = (row0 + r) * A + o;
      float e = 0.f;
      if (r < rv) e = a.noise ? a.noise[i] : hash_normal(a.seed, (((unsigned long long)c.rng_hi << 32) | c.rng_lo) + (unsigned long long)i);
      part[tid] = fminf(fmaxf(__fmul_rn(e, a.policy_noise), -a.noise_clamp), a.noise_clamp);
    }
    __syncthreads();
    float* h;
    if (!a.given_next) {
      // target actor on ns (+ clipped smoothing noise, TD3)
      h = mlp_hidden<RG>(a.tactor, X0, X1, X2, ldl, part + R * 16, nullptr, BH, row0, rv);
      rows_head<RG>(h, ldl, H, hw_ta, H, hb, A, EPI_TANH, sm);
      __syncthreads();
      if (tid < R * A) {
        const int r = tid / A, o = tid - r * A;
        float act = sm[r * 16 + o];
        if (a.target_kind == TGT_MIN) act = fminf(fmaxf(__fadd_rn(act, part[tid]), -1.0f), 1.0f);
        X0[r * ldl + S + o] = act;
      }
      __syncthreads();
    }
    // target critic(s) on [ns | a']
    for (int k = 0; k < C; ++k) {
      h = mlp_hidden<RG>(a.tcritic[k], X0, X1, X2, ldl, part + R * 16, nullptr, BH, row0, rv);
      rows_head<RG>(h, ldl, H, hw_tc + k * H, H, hb + 16 + k, 1, EPI_NONE, sm + 4 + k);   // sm[r*16 + 4 + k]
      __syncthreads();
    }
    if (tid < R) {
      // y = r + gamma*(1-d)*tq, tq = Q' (DDPG, y clamped to [-1/(1-gamma), 0]) or min(Q1', Q2') (TD3):
      // same roundings as td_loss_kernel
      const int r = tid;
      float tq = a.target_kind == TGT_DDPG ? sm[r * 16 + 4] : fminf(sm[r * 16 + 4], sm[r * 16 + 5]);
      if (a.target_kind == TGT_MIN_ENT) tq = __fsub_rn(tq, __fmul_rn(a.alpha, sm3[r * 16 + 2]));
      float y = __fadd_rn(sm3[r * 16], __fmul_rn(__fmul_rn(a.gamma, __fsub_rn(1.0f, sm3[r * 16 + 1])), tq));
      if (a.target_kind == TGT_DDPG) y = fminf(fmaxf(y, a.clamp_lo), 0.0f);
      sm2[r * 16 + 1] = y;
      if (r < rv) a.y[row0 + r] = y;
    }
    // online critic(s) on [s | a]: forward (activations saved), loss gradient, input-gradient chain
    for (int k = 0; k < C; ++k) {
      h = mlp_hidden<RG>(a.critic[k], XS, X1, X2, ldl, part + R * 16, a.hC + (long long)k * a.critic[k].L * BH, BH, row0, rv);
      rows_head<RG>(h, ldl, H, hw_c + k * H, H, hb + 18 + k, 1, EPI_NONE, sm);
      __syncthreads();
      if (tid < R) {
        const int r = tid;
        const float q = sm[r * 16], y = sm2[r * 16 + 1];
        const float diff = __fsub_rn(q, y);
        float g;
        if (a.loss_kind == LOSS_MSE) g = (2.0f / (float)B) * diff;            // d mse_loss / dq
        else { const float n1 = 1.0f / (float)B; g = (diff < -1.0f) ? -n1 : (diff > 1.0f ? n1 : n1 * diff); }   // smooth-L1
        if (r >= rv) g = 0.f;
        sm2[r * 16] = g;
        if (r < rv) { a.q[(long long)k * B + row0 + r] = q; a.dq[(long long)k * B + row0 + r] = g; }
      }
      __syncthreads();
      float* gsave = a.gC + (long long)k * a.critic[k].L * BH;
      head_backward<RG>(h, ldl, H, hw_c + k * H, 1, sm2, gsave + (a.critic[k].L - 1) * BH + row0 * H, rv);
      __syncthreads();
      grad_chain<RG>(a.critic[k], h, X1, X2, ldl, part + R * 16, a.hC + (long long)k * a.critic[k].L * BH, gsave, BH, row0, rv);
    }
  } else if (a.p_critic_only) {
    const StepCtrl c = *a.cur_p;
    const int C = a.C;
    const float* s_rows = a.sa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    {
      const int jpad = a.critic[0].jpad0;
      for (int i = tid; i < R * jpad; i += kRowThreads) {
        const int r = i / jpad, cc = i - r * jpad;
        float v = 0.f;
        if (r < rv) {
          if (cc < S) v = s_rows[(long long)r * a.ldx + cc];
          else if (cc < S + A) v = a.pi[(row0 + r) * a.Apad + (cc - S)];
        }
        X0[r * ldl + cc] = v;
      }
    }
    float* hw_c = hw; float* hw_da = hw + C * H;   // heads [C][H], then rows S..S+A-1 of each W0^T [C][A][H]
    for (int k = 0; k < C; ++k) {
      stage(hw_c + k * H, a.critic[k].P + a.critic[k].w[a.critic[k].L], H);
      stage(hw_da + k * A * H, a.critic[k].Wt + a.critic[k].wt[0] + (long long)S * H, A * H);
    }
    if (tid < C) hb[tid] = a.critic[tid].P[a.critic[tid].b[a.critic[tid].L]];
    __syncthreads();
    for (int k = 0; k < C; ++k) {
      float* h = mlp_hidden<RG>(a.critic[k], X0, X1, X2, ldl, part + R * 16, a.hC2 + (long long)k * a.critic[k].L * BH, BH, row0, rv);
      rows_head<RG>(h, ldl, H, hw_c + k * H, H, hb + k, 1, EPI_NONE, sm + k);   // sm[r*16 + k]
      __syncthreads();
    }
    if (tid < R) {
      // d(-mean min(q1, q2))/dq: to the smaller, split on ties (actor_select_kernel)
      const int r = tid;
      const float q0 = sm[r * 16], q1 = C > 1 ? sm[r * 16 + 1] : INFINITY;
      const float gb = r < rv ? -1.0f / (float)B : 0.f;
      const float w0 = q0 < q1 ? 1.f : (q0 == q1 ? 0.5f : 0.f);
      sm2[r * 16] = gb * w0;
      sm2[r * 16 + 1] = gb * (1.f - w0);
      if (r < rv) { a.q2[row0 + r] = q0; if (C > 1) a.q2[(long long)B + row0 + r] = q1; }
    }
    __syncthreads();
    for (int k = 0; k < C; ++k) {
      const float* hs = a.hC2 + (long long)k * a.critic[k].L * BH;
      const float* hsrc = hs + (a.critic[k].L - 1) * BH + row0 * H;
      for (int i = tid; i < R * H; i += kRowThreads) {
        const int r = i / H, kk = i - r * H;
        XS[r * ldl + kk] = r < rv ? hsrc[(long long)r * H + kk] : 0.f;
      }
      __syncthreads();
      head_backward<RG>(XS, ldl, H, hw_c + k * H, 1, sm2 + k, nullptr, rv);
      __syncthreads();
      float* g0 = grad_chain<RG>(a.critic[k], XS, X1, X2, ldl, part + R * 16, hs, nullptr, BH, row0, rv);
      rows_head<RG>(g0, ldl, H, hw_da + k * A * H, H, nullptr, A, EPI_NONE, sm);
      __syncthreads();
      if (tid < R * A) {
        const int r = tid / A, o = tid - r * A;
        if (r < rv) a.dz[((long long)k * B + row0 + r) * a.Apad + o] = sm[r * 16 + o];
      }
      __syncthreads();
    }
  } else {
    const StepCtrl c = *a.cur_p;
    const float* sa_rows = a.sa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    constexpr int NX = 2 * RG;
    const int jp_s = max(a.actor.jpad0, a.critic[0].jpad0);
    float* hw_a = hw; float* hw_c = hw + A * H; float* hw_da = hw_c + H;
    const float* src_a = a.actor.P + a.actor.w[a.actor.L];
    const float* src_c = a.critic[0].P + a.critic[0].w[a.critic[0].L];
    const float* src_da = a.critic[0].Wt + a.critic[0].wt[0] + (long long)S * H;   // rows S..S+A-1 of W0^T
    Staged<NX> s_s;
    rows_load<RG>(s_s, sa_rows, a.ldx, S, jp_s, rv);
    Staged<8> s_a, s_hc, s_da;
    seg_load(s_a, src_a, A * H);
    seg_load(s_hc, src_c, H);
    seg_load(s_da, src_da, A * H);
    float v_hb = 0.f;
    if (tid < A) v_hb = a.actor.P[a.actor.b[a.actor.L] + tid];
    if (tid == 32) v_hb = a.critic[0].P[a.critic[0].b[a.critic[0].L]];
    rows_store<RG>(s_s, X0, ldl, sa_rows, a.ldx, S, jp_s, rv);
    seg_store(s_a, hw_a, src_a, A * H);
    seg_store(s_hc, hw_c, src_c, H);
    seg_store(s_da, hw_da, src_da, A * H);
    if (tid < A) hb[tid] = v_hb;
    if (tid == 32) hb[16] = v_hb;
    const bool fuse_q = head_fusable(H);   // Q(s, pi(s)) feeds a metric only: its head pass rides in the head's backward pass (below)
    if (fuse_q && tid < R) sm2[tid * 16] = (tid < rv) ? -1.0f / (float)B : 0.f;   // d(-mean Q)/dq, constant
    __syncthreads();
    // the last actor activation lands in XS and stays there: the critic chain reuses X1 / X2
    float* h = mlp_hidden<RG>(a.actor, X0, X1, X2, ldl, part + R * 16, a.hA, BH, row0, rv, XS);
    rows_head<RG>(h, ldl, H, hw_a, H, hb, A, EPI_TANH, sm);
    __syncthreads();
    if (tid < R * A) { const int r = tid / A, o = tid - r * A; X0[r * ldl + S + o] = sm[r * 16 + o]; }
    __syncthreads();
    // critic on [s | pi(s)]
    h = mlp_hidden<RG>(a.critic[0], X0, X1, X2, ldl, part + R * 16, a.hC2, BH, row0, rv);
    if (fuse_q) {
      head_backward<RG>(h, ldl, H, hw_c, 1, sm2, nullptr, rv, sm3, hb[16]);
      __syncthreads();
      if (tid < rv) a.q2[row0 + tid] = sm3[tid * 16];
    } else {
      rows_head<RG>(h, ldl, H, hw_c, H, hb + 16, 1, EPI_NONE, sm2);
      __syncthreads();
      if (tid < R) {
        if (tid < rv) a.q2[row0 + tid] = sm2[tid * 16];
        sm2[tid * 16] = (tid < rv) ? -1.0f / (float)B : 0.f;   // d(-mean Q)/dq
      }
      __syncthreads();
      head_backward<RG>(h, ldl, H, hw_c, 1, sm2, nullptr, rv);
      __syncthreads();
    }
    float* g0 = grad_chain<RG>(a.critic[0], h, X1, X2, ldl, part + R * 16, a.hC2, nullptr, BH, row0, rv);
    // da[r][j] = g0[r][:] . W0[:, S+j]  (row S+j of the [in][out] copy), then through the tanh
    rows_head<RG>(g0, ldl, H, hw_da, H, nullptr, A, EPI_NONE, sm2);
    __syncthreads();
    if (tid < R * A) {
      const int r = tid / A, o = tid - r * A;
      const float act = sm[r * 16 + o];
      const float g = sm2[r * 16 + o] * act_deriv(act, MUL_DTANH);
      sm2[r * 16 + o] = g;
      if (r < rv) a.dz[(row0 + r) * a.Apad + o] = g;
    }
    __syncthreads();
    head_backward<RG>(XS, ldl, H, hw_a, A, sm2, a.gA + (a.actor.L - 1) * BH + row0 * H, rv);
    __syncthreads();
    grad_chain<RG>(a.actor, XS, X1, X2, ldl, part + R * 16, a.hA, a.gA, BH, row0, rv);
  }
  if (a.clk && tid == 0) atomicMax(&a.clk[1], (unsigned long long)wall_clock64());
}

// ---- role-parallel form (rowchain.h: launch_rowchain_split) -------------------------------------------------
// part 3: the role workgroups of a row block meet (meet.h: every wave drains its agent-scope stores before the arrival; one
// monotonic 64-bit counter per row block and phase; those that need the others' values wait, bounded).  Returns false on a timed-out wait.
__device__ inline bool rc_meet(unsigned int* words, int arrivals, bool wait, unsigned int* s_flag, unsigned int* status) {
  return meet(reinterpret_cast<unsigned long long*>(words), (unsigned)arrivals, wait, s_flag, status, MEET_ERR_ROWCHAIN);
}
__device__ inline void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the two heads of the BatchNorm actor on the rows in `hrows` (global [.][H], this block's rows: contiguous) and the sampling (rowchain.h HeadsFold):
// action -> X0's action columns, log-prob terms -> sm3; mu / ls_raw stay in sm / sm2; eps / std -> es[r * 16 + o] / es[R * 16 + r * 16 + o].
// hw_m: LDS room for 2 * A * H floats; es: 2 * R * 16 floats.  Two steps, like the kernels' other prologue staging: heads_request puts every
// global load in flight (into registers) BEFORE the caller's own staging waits for anything — requested after it, the three regions were three more
// dependent round trips in front of the first layer pass (measured: the fold gained 2 us of the launch's 11).
struct HeadsStaged { Staged<4> wm, wl, h; float bm, bl; };
template <int RG>
__device__ __forceinline__ void heads_request(HeadsStaged& g, const HeadsFold& f, const float* hrows, int H, int A, int rv) {
  seg_load(g.wm, f.P + f.w_mean, A * H);
  seg_load(g.wl, f.P + f.w_ls, A * H);
  seg_load(g.h, hrows, rv * H);
  const int tid = threadIdx.x;
  g.bm = tid < A ? f.P[f.b_mean + tid] : 0.f;
  g.bl = tid < A ? f.P[f.b_ls + tid] : 0.f;
}
template <int RG>
__device__ __forceinline__ void heads_sample_rows(const HeadsStaged& g, const HeadsFold& f, const float* hrows, const float* eps, int rng_stream, unsigned long long seed,
                                         const StepCtrl& c, float* X0, float* X1, int ldl, int H, int S, int A, long long row0, int rv,
                                         float* hw_m, float* hb, float* sm, float* sm2, float* sm3, float* es) {
  constexpr int R = 4 * RG;
  const int tid = threadIdx.x;
  seg_store(g.wm, hw_m, f.P + f.w_mean, A * H);
  seg_store(g.wl, hw_m + A * H, f.P + f.w_ls, A * H);
  const bool one_pass = 2 * A <= 16;   // both heads as ONE 2A-output pass (weights and biases are contiguous in LDS: mean rows, then log_std rows)
  if (tid < A) { hb[32 + tid] = g.bm; hb[(one_pass ? 32 + A : 48) + tid] = g.bl; }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid + u * kRowThreads, r = i / H, kk = i - r * H;
    if (i < R * H) X1[r * ldl + kk] = g.h.v[u];             // (rows >= rv: zeros, seg_load's bound)
  }
  for (int i = tid + 4 * kRowThreads; i < R * H; i += kRowThreads) {
    const int r = i / H, kk = i - r * H;
    X1[r * ldl + kk] = r < rv ? hrows[(long long)r * H + kk] : 0.f;
  }
  __syncthreads();
  if (one_pass) {
    rows_head<RG>(X1, ldl, H, hw_m, H, hb + 32, 2 * A, EPI_NONE, sm);
    __syncthreads();
    if (tid < R * A) { const int r = tid / A, o = tid - r * A; sm2[r * 16 + o] = sm[r * 16 + A + o]; }   // (the callers read log_std's outputs from sm2)
  } else {
    rows_head<RG>(X1, ldl, H, hw_m, H, hb + 32, A, EPI_NONE, sm);
    rows_head<RG>(X1, ldl, H, hw_m + A * H, H, hb + 48, A, EPI_NONE, sm2);
  }
  __syncthreads();
  if (tid < R * A) {
    const int r = tid / A, o = tid - r * A;
    float t = 0.f;
    if (r < rv) {
      const TgElem e = tanh_gauss_elem_raw(eps, seed, rng_stream, c, sm[r * 16 + o], sm2[r * 16 + o], (row0 + r) * A + o);
      t = e.t;
      sm3[r * 16 + o] = e.term;
      es[r * 16 + o] = e.e; es[R * 16 + r * 16 + o] = e.sd;
    }
    X0[r * ldl + S + o] = t;
  }
  __syncthreads();
}
// what the actor's backward and the actor loss read of actor.sample(states), from heads_sample_rows' LDS results (one role per row block calls it)
template <int RG>
__device__ __forceinline__ void heads_write_cur(const HeadsFold& f, const float* X0, int ldl, int S, int A, int Apad, long long row0, int rv,
                                       const float* sm, const float* sm2, const float* sm3, const float* es);
__device__ inline float logp_row(const float* sm3, int r, int A) {
  float lp = 0.f;
  for (int j = 0; j < A; ++j) lp = __fadd_rn(lp, sm3[r * 16 + j]);   // in action order, as the sampling launch adds
  return lp;
}

template <int RG>
__device__ __forceinline__ void heads_write_cur(const HeadsFold& f, const float* X0, int ldl, int S, int A, int Apad, long long row0, int rv,
                                       const float* sm, const float* sm2, const float* sm3, const float* es) {
  constexpr int R = 4 * RG;
  const int tid = threadIdx.x;
  if (tid < R * A) {
    const int r = tid / A, o = tid - r * A;
    if (r < rv) {
      const long long i = (row0 + r) * A + o;
      f.pi[(row0 + r) * Apad + o] = X0[r * ldl + S + o];
      f.save_eps[i] = es[r * 16 + o]; f.save_std[i] = es[R * 16 + r * 16 + o];
      f.head[(row0 + r) * 2 * Apad + o] = sm[r * 16 + o];
      f.head[(row0 + r) * 2 * Apad + Apad + o] = sm2[r * 16 + o];
    }
  }
  if (tid < rv) f.logp[row0 + tid] = logp_row(sm3, tid, A);
}

// HF: the instantiation of rowchain_split_heads_kernel (the record is a kernel argument: read in place, never through a pointer that could be null —
// with `const HeadsFold*` and run-time null checks hipcc copied the 192-byte record to per-thread scratch once a third use site appeared)
template <int RG, bool HF>
__device__ __forceinline__ void rowchain_split_body(const RowChainArgs& a, int phase, int part, int bid, const HeadsFold& hfr) {
  const HeadsFold* const hf = &hfr;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int R = 4 * RG;
  const int ldl = a.ldl, H = a.critic[0].H, S = a.S, A = a.A, B = a.B, C = a.C;
  float* X0 = lds;
  float* X1 = X0 + R * ldl;
  float* X2 = X1 + R * ldl;
  float* XS = X2 + R * ldl;
  float* part_ = XS + R * ldl;
  float* sm = part_ + R * 16 + (RG == 1 ? 2 : 1) * 4 * R * kRowChunk;
  float* sm2 = sm + R * 16;
  float* sm3 = sm2 + R * 16;
  float* hw = sm3 + R * 16;
  float* hb = hw + max(max(2 * A + 1, A + 2 * C), C * (A + 1)) * H;
  const int nblk = (B + R - 1) / R;
  const int role = bid / nblk;
  const int blk = bid - role * nblk;
  const bool fold = HF && hfr.on != 0;

  const long long row0 = (long long)blk * R;
  const int rv = min(R, B - (int)row0);
  const long long BH = (long long)B * H;
  const int tid = threadIdx.x;
  const StepCtrl c = phase == 0 ? *a.cur_k : *a.cur_p;
  if (fold && phase == 0 && part != 2 && role == 2 * C) {
    // the launch's extra workgroups (highest indices, waiting for nobody: they start as role workgroups finish and run beside the online roles'
    // backward passes): with cur_in_k one per row block forms pi(s) and what phase 1 and the actor's backward read of it — in the online role's
    // own workgroup the same work sat on the launch's critical path (34.4 us against 32.2) —; block 0 also applies BatchNorm's running statistics
    // (input 0's batch, then input 1's)
    if (hf->cur_in_k) {
      HeadsStaged hst;
      heads_request<RG>(hst, *hf, hf->h_cur + row0 * H, H, A, rv);
      heads_sample_rows<RG>(hst, *hf, hf->h_cur + row0 * H, hf->eps_cur, 2, a.seed, c, X0, X1, ldl, H, S, A, row0, rv, hw, hb, sm, sm2, sm3, part_);
      heads_write_cur<RG>(*hf, X0, ldl, S, A, a.Apad, row0, rv, sm, sm2, sm3, part_);
    }
    if (blk == 0 && hf->run.layers) bn_running_update(hf->run);
    return;
  }
  const bool merged = part == 3;                 // parts 1 and 2 in this launch (rowchain.h)
  __shared__ unsigned int s_flag;
  unsigned int* meet = merged ? a.bar + ((long long)phase * nblk + blk) * 32 : nullptr;
  bool met = true;
  if (phase == 0 && part != 2 && bid == 0 && tid == 0) { a.cb->cur_b = a.cb->cur; a.cb->prev_b = a.cb->prev; }

  if (phase == 0 && part != 2 && role < C) {
    // ---- target critic `role` on [ns | a'] (a' given: SAC; else the target actor runs first: TD3)
    const int k = role;
    const float* ns_rows = a.nsa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    const int jp_ns = max(a.tactor.jpad0, a.tcritic[0].jpad0), nc_ns = (a.given_next && !fold) ? S + A : S;
    float* hw_ta = hw; float* hw_tc = hw + (fold ? 2 * A : A) * H;
    HeadsStaged hst;
    if (fold) heads_request<RG>(hst, *hf, hf->h_next + row0 * H, H, A, rv);
    load_rows<RG>(X0, ldl, ns_rows, a.ldx, nc_ns, jp_ns, rv);
    stage(hw_tc, a.tcritic[k].P + a.tcritic[k].w[a.tcritic[k].L], H);
    if (tid == 0) hb[16] = a.tcritic[k].P[a.tcritic[k].b[a.tcritic[k].L]];
    if (!a.given_next) {
      stage(hw_ta, a.tactor.P + a.tactor.w[a.tactor.L], A * H);
      if (tid < A) hb[tid] = a.tactor.P[a.tactor.b[a.tactor.L] + tid];
      if (a.target_kind == TGT_MIN && tid < R * A) {
        const int r = tid / A, o = tid - r * A;
        const long long i = (row0 + r) * A + o;
        float e = 0.f;
        if (r < rv) e = a.noise ? a.noise[i] : hash_normal(a.seed, (((unsigned long long)c.rng_hi << 32) | c.rng_lo) + (unsigned long long)i);
        part_[tid] = fminf(fmaxf(__fmul_rn(e, a.policy_noise), -a.noise_clamp), a.noise_clamp);
      }
    }
    if (fold) {   // a' = actor.sample(next_state) for these rows; role 0 publishes logp_next (the online roles read it after the meeting)
      heads_sample_rows<RG>(hst, *hf, hf->h_next + row0 * H, hf->eps_next, 1, a.seed, c, X0, X1, ldl, H, S, A, row0, rv, hw, hb, sm, sm2, sm3, part_);
      if (k == 0 && tid < rv) st_agent(hf->logp_next + row0 + tid, logp_row(sm3, tid, A));
    }
    __syncthreads();
    float* h;
    if (!a.given_next) {
      h = mlp_hidden<RG>(a.tactor, X0, X1, X2, ldl, part_ + R * 16, nullptr, BH, row0, rv);
      rows_head<RG>(h, ldl, H, hw_ta, H, hb, A, EPI_TANH, sm);
      __syncthreads();
      if (tid < R * A) {
        const int r = tid / A, o = tid - r * A;
        float act = sm[r * 16 + o];
        if (a.target_kind == TGT_MIN) act = fminf(fmaxf(__fadd_rn(act, part_[tid]), -1.0f), 1.0f);
        X0[r * ldl + S + o] = act;
      }
      __syncthreads();
    }
    h = mlp_hidden<RG>(a.tcritic[k], X0, X1, X2, ldl, part_ + R * 16, nullptr, BH, row0, rv);
    rows_head<RG>(h, ldl, H, hw_tc, H, hb + 16, 1, EPI_NONE, sm);
    __syncthreads();
    if (tid < rv) st_agent(a.qt + (long long)k * B + row0 + tid, sm[tid * 16]);
    if (merged) {                                                    // (the target roles only report in)
      if (a.producers_first) meet_produce(reinterpret_cast<unsigned long long*>(meet));
      else rc_meet(meet, 2 * C, false, &s_flag, a.status);
    }
    return;
  }
  int k2 = role;                                 // the critic whose backward part runs in this workgroup
  if (phase == 0 && part != 2) {
    // ---- online critic `role - C` on [s | a]: forward, activations saved
    const int k = role - C;
    const float* sa_rows = a.sa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    load_rows<RG>(XS, ldl, sa_rows, a.ldx, S + A, a.critic[0].jpad0, rv);
    stage(hw, a.critic[k].P + a.critic[k].w[a.critic[k].L], H);
    if (tid == 0) hb[18] = a.critic[k].P[a.critic[k].b[a.critic[k].L]];
    __syncthreads();
    float* h = mlp_hidden<RG>(a.critic[k], XS, X1, X2, ldl, part_ + R * 16, a.hC + (long long)k * a.critic[k].L * BH, BH, row0, rv);
    rows_head<RG>(h, ldl, H, hw, H, hb + 18, 1, EPI_NONE, sm);
    __syncthreads();
    if (tid < rv) a.q[(long long)k * B + row0 + tid] = sm[tid * 16];
    if (!merged) return;
    // both target critics' outputs of these rows are out
    if (a.producers_first)
      met = meet_consume(reinterpret_cast<unsigned long long*>(meet), reinterpret_cast<unsigned long long*>(meet) + 2 + k, (unsigned)C, &s_flag, a.status, MEET_ERR_ROWCHAIN);
    else met = rc_meet(meet, 2 * C, true, &s_flag, a.status);
    k2 = k;
  }
  if (phase == 0) {
    // ---- part 2: TD target, loss gradient and input-gradient chain of critic `k2`
    const int k = k2;
    const int L = a.critic[k].L;
    const float* hs = a.hC + (long long)k * L * BH;
    const float* hsrc = hs + (L - 1) * BH + row0 * H;
    stage(hw, a.critic[k].P + a.critic[k].w[L], H);
    for (int i = tid; i < R * H; i += kRowThreads) {
      const int r = i / H, kk = i - r * H;
      XS[r * ldl + kk] = r < rv ? hsrc[(long long)r * H + kk] : 0.f;
    }
    if (tid < R) {
      const int r = tid;
      float g = 0.f;
      if (r < rv) {
        const float rew = a.rbuf[(long long)c.batch_slot * a.slot_rd + row0 + r], dn = a.dbuf[(long long)c.batch_slot * a.slot_rd + row0 + r];
        const float t0 = ld_agent(a.qt + row0 + r), t1 = C > 1 ? ld_agent(a.qt + (long long)B + row0 + r) : t0;
        float tq = a.target_kind == TGT_DDPG ? t0 : fminf(t0, t1);
        if (a.target_kind == TGT_MIN_ENT) tq = __fsub_rn(tq, __fmul_rn(a.alpha, fold ? ld_agent(a.logp_next + row0 + r) : a.logp_next[row0 + r]));
        float y = __fadd_rn(rew, __fmul_rn(__fmul_rn(a.gamma, __fsub_rn(1.0f, dn)), tq));
        if (a.target_kind == TGT_DDPG) y = fminf(fmaxf(y, a.clamp_lo), 0.0f);
        const float q = a.q[(long long)k * B + row0 + r];
        const float diff = __fsub_rn(q, y);
        if (a.loss_kind == LOSS_MSE) g = (2.0f / (float)B) * diff;
        else { const float n1 = 1.0f / (float)B; g = (diff < -1.0f) ? -n1 : (diff > 1.0f ? n1 : n1 * diff); }
        if (!met) g = __builtin_nanf("");          // a timed-out meeting must not pass for a result
        if (k == 0) a.y[row0 + r] = y;
        a.dq[(long long)k * B + row0 + r] = g;
      }
      sm2[r * 16] = g;
    }
    __syncthreads();
    float* gsave = a.gC + (long long)k * L * BH;
    head_backward<RG>(XS, ldl, H, hw, 1, sm2, gsave + (L - 1) * BH + row0 * H, rv);
    __syncthreads();
    grad_chain<RG>(a.critic[k], XS, X1, X2, ldl, part_ + R * 16, hs, gsave, BH, row0, rv);
    return;
  }
  if (part != 2) {
    // ---- P, part 1: critic `role` forward on [s | pi(s)]
    const int k = role;
    const float* s_rows = a.sa + (long long)c.batch_slot * a.slot_x + row0 * a.ldx;
    const int jpad = a.critic[0].jpad0;
    HeadsStaged hst;
    if (fold) heads_request<RG>(hst, *hf, hf->h_cur + row0 * H, H, A, rv);
    for (int i = tid; i < R * jpad; i += kRowThreads) {
      const int r = i / jpad, cc = i - r * jpad;
      float v = 0.f;
      if (r < rv) {
        if (cc < S) v = s_rows[(long long)r * a.ldx + cc];
        else if (cc < S + A && !fold) v = a.pi[(row0 + r) * a.Apad + (cc - S)];
      }
      if (!fold || cc < S || cc >= S + A) X0[r * ldl + cc] = v;
    }
    stage(hw, a.critic[k].P + a.critic[k].w[a.critic[k].L], H);
    if (tid == 0) hb[0] = a.critic[k].P[a.critic[k].b[a.critic[k].L]];
    if (fold) {   // pi(s) = actor.sample(states) for these rows; role 0 writes what the actor's backward and the actor loss read
      float* hw_m = hw + H;
      heads_sample_rows<RG>(hst, *hf, hf->h_cur + row0 * H, hf->eps_cur, 2, a.seed, c, X0, X1, ldl, H, S, A, row0, rv, hw_m, hb, sm, sm2, sm3, part_);
      if (k == 0) heads_write_cur<RG>(*hf, X0, ldl, S, A, a.Apad, row0, rv, sm, sm2, sm3, part_);
    }
    __syncthreads();
    float* h = mlp_hidden<RG>(a.critic[k], X0, X1, X2, ldl, part_ + R * 16, a.hC2 + (long long)k * a.critic[k].L * BH, BH, row0, rv);
    rows_head<RG>(h, ldl, H, hw, H, hb, 1, EPI_NONE, sm);
    __syncthreads();
    if (tid < rv) st_agent(a.q2 + (long long)k * B + row0 + tid, sm[tid * 16]);
    if (!merged) return;
    met = rc_meet(meet, C, true, &s_flag, a.status);                   // the other critic's Q of these rows is out
  }
  {
    // ---- P, part 2: d(-mean min(q1, q2))/dq_k, input-gradient chain of critic `role` down to the action columns
    const int k = role;
    const int L = a.critic[k].L;
    const float* hs = a.hC2 + (long long)k * L * BH;
    const float* hsrc = hs + (L - 1) * BH + row0 * H;
    float* hw_da = hw + H;
    stage(hw, a.critic[k].P + a.critic[k].w[L], H);
    stage(hw_da, a.critic[k].Wt + a.critic[k].wt[0] + (long long)S * H, A * H);
    for (int i = tid; i < R * H; i += kRowThreads) {
      const int r = i / H, kk = i - r * H;
      XS[r * ldl + kk] = r < rv ? hsrc[(long long)r * H + kk] : 0.f;
    }
    if (tid < R) {
      const int r = tid;
      float g = 0.f;
      if (r < rv) {
        const float q0 = ld_agent(a.q2 + row0 + r), q1 = C > 1 ? ld_agent(a.q2 + (long long)B + row0 + r) : INFINITY;
        const float gb = -1.0f / (float)B;
        const float w0 = q0 < q1 ? 1.f : (q0 == q1 ? 0.5f : 0.f);
        g = k == 0 ? gb * w0 : gb * (1.f - w0);
        if (!met) g = __builtin_nanf("");
      }
      sm2[r * 16] = g;
    }
    __syncthreads();
    head_backward<RG>(XS, ldl, H, hw, 1, sm2, nullptr, rv);
    __syncthreads();
    float* g0 = grad_chain<RG>(a.critic[k], XS, X1, X2, ldl, part_ + R * 16, hs, nullptr, BH, row0, rv);
    rows_head<RG>(g0, ldl, H, hw_da, H, nullptr, A, EPI_NONE, sm);
    __syncthreads();
    if (tid < R * A) {
      const int r = tid / A, o = tid - r * A;
      if (r < rv) a.dz[((long long)k * B + row0 + r) * a.Apad + o] = sm[r * 16 + o];
    }
  }
}

template <int RG>
__global__ __launch_bounds__(kRowThreads) void rowchain_split_kernel(RowChainArgs a, int phase, int part) {
  kernarg_warm<sizeof(RowChainArgs) + 8>();
  rowchain_split_body<RG>(a, phase, part, (int)blockIdx.x);
}
template <int RG>
__global__ __launch_bounds__(kRowThreads) void rowchain_split_heads_kernel(RowChainArgs a, int phase, int part, HeadsFold hf) {
  kernarg_warm<sizeof(RowChainArgs) + 8 + sizeof(HeadsFold)>();
  rowchain_split_body<RG, true>(a, phase, part, (int)blockIdx.x, hf);
}

// `row_at(i)`: element i of the launch's input rows ([n][ld_obs] flattened); `noise_at(t)`: exploration noise of action element t —
// functors, so that the inline form reads the kernel-argument segment by plain indexed loads (a POINTER into a by-value argument
// struct made hipcc copy the whole struct into every thread's scratch: 36 us for this kernel, round 4).  SYS: the float64 actions
// go out as system-scope (write-through) stores, for a host that polls a flag behind them.
template <bool SYS, typename RowAt, typename NoiseAt>
__device__ inline void rowchain_act_body(const RowActArgs& a, RowAt row_at, NoiseAt noise_at, bool has_noise, double* out64, float* lds) {
  constexpr int R = 4;
  const int ldl = a.ldl, H = a.actor.H, A = a.A, tid = threadIdx.x;
  float* X0 = lds;
  float* X1 = X0 + R * ldl;
  float* X2 = X1 + R * ldl;
  float* part = X2 + R * ldl;                 // two exchange buffers (chained layers)
  float* sm = part + 2 * 4 * R * kRowChunk;
  float* hw = sm + R * 16;
  float* hb = hw + A * H;
  const long long row0 = (long long)blockIdx.x * R;
  const int rv = min(R, a.n - (int)row0);
  const float* src_h = a.actor.P + a.actor.w[a.actor.L];
  const int jp = a.actor.jpad0;
  constexpr int NX = 2;                        // staged elements per thread: rows of up to 128 floats; wider rows take the loop below
  float s_x[NX];
  Staged<8> s_h;
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int i = tid + u * kRowThreads, r = i / jp, c = i - r * jp;
    s_x[u] = (i < R * jp && r < rv && c < a.S) ? row_at((row0 + r) * a.ld_obs + c) : 0.f;
  }
  seg_load(s_h, src_h, A * H);
  const float v_hb = tid < A ? a.actor.P[a.actor.b[a.actor.L] + tid] : 0.f;
  // An acting launch is a few workgroups long after the last optimiser launch: every layer's weights come from HBM, one
  // dependent ~2 us round trip per layer pass.  Touch one float of every 128-byte line of the LATER layers' [in][out] copies now
  // (fire and forget: the values are only consumed by a never-taken branch at the end), so that their passes find them in L2.
  float pf = 0.f;
  {
    const float* wl = a.actor.Wt + a.actor.wt[a.actor.L > 1 ? 1 : 0];
    const long long nfl = a.actor.L > 1 ? (long long)(a.actor.L - 1) * H * H : 0;
    if (nfl <= 64 * 1024)
      for (long long i = (long long)tid * 32; i < nfl; i += (long long)kRowThreads * 32) pf += wl[i];
  }
  // fused acting entry: the rows are raw; normalise this thread's elements on their way into LDS
  if (a.nz_mean || a.nzg_mean) {
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int i = tid + u * kRowThreads, r = i / jp, c = i - r * jp;
      if (i < R * jp && r < rv && c < a.S) {
        const bool ob = c < a.D;
        const double* m = ob ? a.nz_mean : a.nzg_mean;
        if (m) {
          const int j = ob ? c : c - a.D;
          const double* v = ob ? a.nz_var : a.nzg_var;
          const double clip = ob ? a.nz_clip : a.nzg_clip;
          const int md = ob ? a.nz_mode : a.nzg_mode;
          s_x[u] = norm_apply(s_x[u], m[j], norm_den(v[j], (md & NORM_F32) != 0), clip, norm_apply_f32(md));
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int i = tid + u * kRowThreads, r = i / jp, c = i - r * jp;
    if (i < R * jp) X0[r * ldl + c] = s_x[u];
  }
  for (int i = tid + NX * kRowThreads; i < R * jp; i += kRowThreads) {   // (rows wider than 128 floats: never with fused normalisation)
    const int r = i / jp, c = i - r * jp;
    X0[r * ldl + c] = (r < rv && c < a.S) ? row_at((row0 + r) * a.ld_obs + c) : 0.f;
  }
  seg_store(s_h, hw, src_h, A * H);
  if (tid < A) hb[tid] = v_hb;
  __syncthreads();
  float* h = mlp_hidden<1>(a.actor, X0, X1, X2, ldl, part, nullptr, 0, row0, rv);
  rows_head<1>(h, ldl, H, hw, H, hb, A, EPI_TANH, sm);
  __syncthreads();
  if (tid < R * A) {
    const int r = tid / A, o = tid - r * A;
    if (r < rv) {
      const float y = sm[r * 16 + o];
      if (!a.post) a.out[(row0 + r) * a.ld_out + o] = y;
      else {
        const long long t = (row0 + r) * A + o;
        double v = (double)y;
        if (a.post != 3) {
          v = (double)tanhf(y);
          if (a.post == 1 && has_noise) v += noise_at(t);
          v = fmin(fmax(v, -1.0), 1.0);
        }
        if (SYS) __hip_atomic_store(out64 + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else out64[t] = v;
      }
    }
  }
  if (pf == 1.2345678e-30f && a.post) out64[0] = (double)pf;   // (keeps the prefetch loads alive; never true in practice, harmless if it were)
}

__global__ __launch_bounds__(kRowThreads) void rowchain_act_kernel(RowActArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rowchain_act_body<false>(a, [&](long long i) { return a.obs[i]; }, [&](long long t) { return a.noise[t]; }, a.noise != nullptr, a.out64, lds);
}

// rows and noise inside the kernel arguments, actions and a completion flag per workgroup to host-visible memory (rowchain.h)
__global__ __launch_bounds__(kRowThreads) void rowchain_act_inline_kernel(RowActInline a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rowchain_act_body<true>(a.base, [&](long long i) { return a.obs_inl[i]; }, [&](long long t) { return a.noise_inl[t]; }, a.with_noise != 0, a.out_host, lds);
  // this workgroup's rows are out as write-through stores: drain them, then raise its flag (system scope: the host polls it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(a.flag_host + blockIdx.x, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Wt[wt[l] + k*H + o] = P[w[l] + o*in_l + k]; layer 0 rows in..jpad0-1 are zero
__global__ void wt_rebuild_kernel(RowNet net, float* Wt) {
  const int l = blockIdx.y;
  const int in = l == 0 ? net.in : net.H, J = l == 0 ? net.jpad0 : net.H, H = net.H;
  const long long n = (long long)J * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i / H), o = (int)(i - (long long)k * H);
    Wt[net.wt[l] + i] = k < in ? net.P[net.w[l] + (long long)o * in + k] : 0.f;
  }
}

}  // namespace

size_t rowchain_lds_bytes(int rg, int ldl, int A, int H, int C) {
  const int R = 4 * rg;
  return (size_t)(4 * R * ldl + (rg == 1 ? 2 : 1) * 4 * R * kRowChunk + 4 * R * 16 + std::max(std::max(2 * A + 1, A + 2 * C), C * (A + 1)) * H + 96) * sizeof(float);   // (the trailing floats: head biases — hb[0 .. 64))
}

// the merged (part 3) launches wait inside the kernel: all 2C x nblk workgroups of the larger one must be resident at once
bool rowchain_merge_ok(int rg, int ldl, int A, int H, int C, int B) {
  const size_t lds = rowchain_lds_bytes(rg, ldl, A, H, C);
  if (lds > 160 * 1024) return false;
  const void* k = rg == 1 ? (const void*)rowchain_split_kernel<1> : (rg == 2 ? (const void*)rowchain_split_kernel<2> : (const void*)rowchain_split_kernel<4>);
  if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
  const long long nblk = (B + 4 * rg - 1) / (4 * rg);
  // (the form with the actor's heads inside — SAC, rowchain.h HeadsFold — is another kernel with its own register count: both must fit)
  const void* kh = rg == 1 ? (const void*)rowchain_split_heads_kernel<1> : (rg == 2 ? (const void*)rowchain_split_heads_kernel<2> : (const void*)rowchain_split_heads_kernel<4>);
  if (lds > 64 * 1024 && hipFuncSetAttribute(kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
  // (its extra workgroup has the launch's LAST index and waits for nobody: it may start once a role workgroup has finished)
  return 2LL * C * nblk <= meet_capacity(k, kRowThreads, lds) && 2LL * C * nblk <= meet_capacity(kh, kRowThreads, lds);
}

int launch_rowchain_ddpg(hipStream_t st, const RowChainArgs& a, int rg) {
  GCRL_CHECK_ARG(rg == 1 || rg == 2 || rg == 4, "rowchain: rows per block must be 4, 8 or 16");
  GCRL_CHECK_ARG(a.critic[0].H % 4 == 0 && a.ldl % 4 == 0 && a.A <= 16 && a.C >= 1 && a.C <= 2,
                 "rowchain: unsupported shape (H=%d, A=%d, C=%d)", a.critic[0].H, a.A, a.C);
  const size_t lds = rowchain_lds_bytes(rg, a.ldl, a.A, a.critic[0].H, a.C);
  GCRL_CHECK_ARG(lds <= 160 * 1024, "rowchain: %zu bytes of LDS needed", lds);
  GCRL_CHECK_ARG(!a.k_split || (a.bar && a.qt && a.producers_first && a.C == 1), "rowchain: the two-role critic phase needs its meeting counters");
  const int grid = (a.k_split ? 2 : 1) * a.nblk_k + a.nblk_p;
  if (grid < 1) return GCRL_OK;
  auto go = [&](auto kern) -> int {
    static thread_local size_t raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
      GCRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRowThreads), lds, st, a);
    GCRL_HIP(hipGetLastError());
    return GCRL_OK;
  };
  if (rg == 1) return go(rowchain_ddpg_kernel<1>);
  if (rg == 2) return go(rowchain_ddpg_kernel<2>);
  return go(rowchain_ddpg_kernel<4>);
}

int launch_rowchain_split(hipStream_t st, const RowChainArgs& a, int rg, int phase, int part, const HeadsFold* hf) {
  GCRL_CHECK_ARG(rg == 1 || rg == 2 || rg == 4, "rowchain: rows per block must be 4, 8 or 16");
  GCRL_CHECK_ARG(a.critic[0].H % 4 == 0 && a.ldl % 4 == 0 && a.A <= 16 && a.C >= 1 && a.C <= 2 && (phase == 0 || phase == 1) &&
                     (part >= 1 && part <= 3) && (part != 3 || a.bar) && (phase == 0 || a.p_critic_only) && a.qt,
                 "rowchain split: unsupported shape (H=%d, A=%d, C=%d, phase %d part %d)", a.critic[0].H, a.A, a.C, phase, part);
  const size_t lds = rowchain_lds_bytes(rg, a.ldl, a.A, a.critic[0].H, a.C);
  GCRL_CHECK_ARG(lds <= 160 * 1024, "rowchain: %zu bytes of LDS needed", lds);
  const int nblk = (a.B + 4 * rg - 1) / (4 * rg);
  const int roles = (phase == 0 && part != 2) ? 2 * a.C : a.C;
  // part 3: every workgroup of the launch must be resident at once (rowchain_merge_ok: the kernel's occupancy at this LDS size
  // on a device the process has to itself)
  GCRL_CHECK_ARG(part != 3 || (phase == 0 && a.producers_first && !meet_device_shared()) || rowchain_merge_ok(rg, a.ldl, a.A, a.critic[0].H, a.C, a.B),
                 "rowchain split: %d workgroups of %zu bytes of LDS cannot all be resident (or the device is shared)", roles * nblk, lds);
  auto go = [&](auto kern) -> int {
    static thread_local size_t raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
      GCRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(roles * nblk), dim3(kRowThreads), lds, st, a, phase, part);
    GCRL_HIP(hipGetLastError());
    return GCRL_OK;
  };
  if (hf && hf->on) {   // the BatchNorm actor's heads and sampling inside the launch (rowchain.h HeadsFold); phase 0 carries the running-statistics workgroup
    GCRL_CHECK_ARG(a.given_next && a.p_critic_only && part != 2 && a.A <= 16, "rowchain split: the folded heads need the SAC form (given_next, critic-only actor phase)");
    const int grid = roles * nblk + (phase == 0 ? (hf->cur_in_k ? nblk : (hf->run.layers ? 1 : 0)) : 0);
    auto goh = [&](auto kern) -> int {
      static thread_local size_t raised = 0;
      if (lds > 64 * 1024 && lds > raised) {
        GCRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = lds;
      }
      hipLaunchKernelGGL(kern, dim3(grid), dim3(kRowThreads), lds, st, a, phase, part, *hf);
      GCRL_HIP(hipGetLastError());
      return GCRL_OK;
    };
    if (rg == 1) return goh(rowchain_split_heads_kernel<1>);
    if (rg == 2) return goh(rowchain_split_heads_kernel<2>);
    return goh(rowchain_split_heads_kernel<4>);
  }
  if (rg == 1) return go(rowchain_split_kernel<1>);
  if (rg == 2) return go(rowchain_split_kernel<2>);
  return go(rowchain_split_kernel<4>);
}

int launch_rowchain_act(hipStream_t st, const RowActArgs& a) {
  GCRL_CHECK_ARG(a.actor.H % 4 == 0 && a.ldl % 4 == 0 && a.A <= 16 && a.n >= 1, "rowchain act: unsupported shape");
  GCRL_CHECK_ARG(!(a.nz_mean || a.nzg_mean) || 4 * a.actor.jpad0 <= 2 * kRowThreads, "rowchain act: fused normalisation supports state_dim <= 128");
  const size_t lds = (size_t)(3 * 4 * a.ldl + 2 * 4 * 4 * kRowChunk + 4 * 16 + a.A * a.actor.H + 32) * sizeof(float);
  GCRL_CHECK_ARG(lds <= 160 * 1024, "rowchain act: %zu bytes of LDS needed", lds);
  static thread_local size_t raised = 0;
  if (lds > 64 * 1024 && lds > raised) {
    GCRL_HIP(hipFuncSetAttribute((const void*)rowchain_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised = lds;
  }
  hipLaunchKernelGGL(rowchain_act_kernel, dim3((a.n + 3) / 4), dim3(kRowThreads), lds, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_rowchain_act_inline(hipStream_t st, const RowActInline& a) {
  const RowActArgs& b = a.base;
  GCRL_CHECK_ARG(b.actor.H % 4 == 0 && b.ldl % 4 == 0 && b.A <= 16 && b.n >= 1 && b.post != 0 && a.out_host && a.flag_host, "rowchain act (inline): unsupported shape");
  GCRL_CHECK_ARG(b.n * b.ld_obs <= kActInlineFloats && b.n * b.A <= kActInlineNoise, "rowchain act (inline): %d rows do not fit the kernel arguments", b.n);
  GCRL_CHECK_ARG(!(b.nz_mean || b.nzg_mean) || 4 * b.actor.jpad0 <= 2 * kRowThreads, "rowchain act: fused normalisation supports state_dim <= 128");
  const size_t lds = (size_t)(3 * 4 * b.ldl + 2 * 4 * 4 * kRowChunk + 4 * 16 + b.A * b.actor.H + 32) * sizeof(float);
  GCRL_CHECK_ARG(lds <= 160 * 1024, "rowchain act: %zu bytes of LDS needed", lds);
  static thread_local size_t raised = 0;
  if (lds > 64 * 1024 && lds > raised) {
    GCRL_HIP(hipFuncSetAttribute((const void*)rowchain_act_inline_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised = lds;
  }
  hipLaunchKernelGGL(rowchain_act_inline_kernel, dim3((b.n + 3) / 4), dim3(kRowThreads), lds, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_wt_rebuild(hipStream_t st, const RowNet& net, float* Wt) {
  hipLaunchKernelGGL(wt_rebuild_kernel, dim3(64, net.L), dim3(256), 0, st, net, Wt);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
