// sac_select.h — device code of the SAC / TQC actor phase that more than one launch hosts (round 5): one (row, action) element of the sampling
// backward (src/model.py:125-141 differentiated), the actor-loss selection (src/agent.py:516-521, :916-925) and the log-alpha gradient /
// AdamW step (src/agent.py:532-546).  ops_sac.hip's launches use it; so does the top layer's backward slab launch (bn_slab.hip), which
// takes the sampling backward as its prologue and the selection + log-alpha block as one more workgroup.
#pragma once
#include <cmath>

#include "ops.h"

namespace gcrl {

__device__ inline float sel_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// sum over a block of up to 1024 threads (whole waves); result valid in every thread.  `scratch` holds 16 floats.
// Waves beyond the data contribute exact zeros, so the result does not depend on the block size chosen for small B.
__device__ inline float sel_block_sum(float v, float* scratch) {
  v = sel_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float s = scratch[0];
  for (int w = 1; w < nw; ++w) s += scratch[w];
  return s;
}

// one (row, action) element of SACActorModel.sample: action t, std sd, the log-prob term, the eps used
struct TgElem { float t, sd, term, e; };
// (eps: injected N(0,1) [B][A] or null: the counter hash, stream `rng_stream` of `seed`, at the step's counter + i)
__device__ inline TgElem tanh_gauss_elem_raw(const float* eps, unsigned long long seed, int rng_stream, const StepCtrl& c, float mu, float ls_raw, long long i) {
  const float ls = fminf(fmaxf(ls_raw, -20.0f), 2.0f);
  // exp / tanh / log go through fp64 and round once: log(1 - tanh^2 + 1e-8) amplifies a 1-ulp
  // tanh difference by 2|t|/(1-t^2), so the closer to correctly rounded, the closer to torch
  const float sd = (float)exp((double)ls);
  const float e = eps ? eps[i]
                        : hash_normal(seed + (unsigned long long)rng_stream,
                                      (((unsigned long long)c.rng_hi << 32) | c.rng_lo) + (unsigned long long)i);
  const float x = __fadd_rn(mu, __fmul_rn(e, sd));  // rsample: loc + eps*scale
  const float t = (float)tanh((double)x);
  // Normal(mu, sd).log_prob(x) - log(1 - tanh(x)^2 + 1e-8), every op rounded to fp32 as torch does
  const float df = __fsub_rn(x, mu);
  const float var = __fmul_rn(sd, sd);
  float term = __fsub_rn(__fsub_rn(__fdiv_rn(-__fmul_rn(df, df), __fmul_rn(2.0f, var)), (float)log((double)sd)),
                         0.91893853320467274f);
  const float om = __fadd_rn(__fsub_rn(1.0f, __fmul_rn(t, t)), 1e-8f);
  term = __fsub_rn(term, (float)log((double)om));
  return {t, sd, term, e};
}
__device__ inline TgElem tanh_gauss_elem(const TanhGaussArgs& a, const StepCtrl& c, float mu, float ls_raw, long long i) {
  return tanh_gauss_elem_raw(a.eps, a.seed, a.rng_stream, c, mu, ls_raw, i);
}

constexpr float kBnMomentum = 0.1f;   // nn.BatchNorm1d's default momentum
// running statistics from the batch statistics of the slab launches (bn_slab.hip), momentum 0.1, unbiased variance
__device__ inline void bn_running_update(const BnRunning& r) {
  const float ub = r.B > 1 ? (float)r.B / (float)(r.B - 1) : 1.0f;
  for (int e = threadIdx.x; e < r.layers * r.H; e += blockDim.x) {
    const int l = e / r.H, c = e - l * r.H;
    float rm = r.rmean[e], rv = r.rvar[e];
    for (int i = 0; i < r.n; ++i) {
      rm = (1.0f - kBnMomentum) * rm + kBnMomentum * r.bstat[i][(2 * l) * r.H + c];
      rv = (1.0f - kBnMomentum) * rv + kBnMomentum * (r.bstat[i][(2 * l + 1) * r.H + c] * ub);
    }
    r.rmean[e] = rm; r.rvar[e] = rv;
  }
}


// backward of SACActorModel.sample for element (b, j): gradients of the two heads' outputs
// (slot_off = cur->batch_slot * act_slot_stride; the callers skip the control-block load — a dependent memory round trip — when the stride is 0)
__device__ inline void tanh_gauss_bwd_elem(const TanhGaussBwdArgs& a, long long slot_off, float alpha, int b, int j, float& gmu, float& gls) {
  const long long i = (long long)b * a.A + j;
  float da = 0.f;
  for (int k = 0; k < a.C; ++k) da += a.dact[(long long)k * a.dact_stride + (long long)b * a.ld_dact + j];
  const float t = a.act[slot_off + (long long)b * a.ld_act + j];
  const float om = 1.0f - t * t;
  const float wlp = alpha / (float)a.B;  // d loss / d logp_b
  // d/dx [ -log(1 - tanh(x)^2 + 1e-8) ] = 2 t (1-t^2) / (1 - t^2 + 1e-8); the Normal terms of
  // the log-prob cancel through the reparameterisation (x - mu = eps*sd)
  const float dx = da * om + wlp * (2.0f * t * om / (om + 1e-8f));
  const float lsr = a.ls_raw[(long long)b * a.ld_head + j];
  const bool in_range = lsr >= -20.0f && lsr <= 2.0f;  // clamp backward
  gmu = dx;
  gls = in_range ? (dx * a.eps[i] * a.std[i] - wlp) : 0.f;
}

// (round 4: every operand of a thread's (up to) two rows is requested before any store — the compiler must assume dq aliases
// q / logp, and each row's later loads used to wait behind its first stores; ops.hip td_loss_kernel has the numbers)
struct SelRow { float q[kMaxCritics], lp; };
// the thread's share of sum_b (alpha * logp_b - sel_b); mb: row blockIdx.x * 256 + threadIdx.x only (multi-workgroup form)
__device__ inline float actor_select_acc(const ActorSelArgs& a, bool mb) {
  const float alpha = a.alpha_dev ? *a.alpha_dev : a.alpha_const;
  const int B = a.B, C = a.C, keep = a.C - a.drop;
  const float gb = -1.0f / (float)B;
  float acc = 0.f;
  auto fetch = [&](SelRow& w, int b) {
    const bool in = b < B;
#pragma unroll
    for (int k = 0; k < kMaxCritics; ++k) w.q[k] = (in && k < C) ? a.q[(long long)k * B + b] : INFINITY;
    w.lp = in ? a.logp[b] : 0.f;
  };
  auto finish = [&](SelRow& w, int b) {
    float* q = w.q;
    float sel;
    if (C == 2 && a.drop == 0) {
      // torch.min(q1, q2): gradient to the smaller, split on ties
      sel = fminf(q[0], q[1]);
      const float w0 = q[0] < q[1] ? 1.f : (q[0] == q[1] ? 0.5f : 0.f);
      a.dq[b] = gb * w0;
      a.dq[(long long)B + b] = gb * (1.f - w0);
    } else {
      // rank of each critic in the ascending (stable) order; the lowest `keep` carry gradient
      const float gk = gb / (float)keep;
#pragma unroll
      for (int k = 0; k < kMaxCritics; ++k) {
        if (k < C) {
          int rank = 0;
#pragma unroll
          for (int j = 0; j < kMaxCritics; ++j)
            if (j < C && (q[j] < q[k] || (q[j] == q[k] && j < k))) ++rank;
          a.dq[(long long)k * B + b] = rank < keep ? gk : 0.f;
        }
      }
#pragma unroll
      for (int pass = 0; pass < kMaxCritics - 1; ++pass)
#pragma unroll
        for (int k = 0; k < kMaxCritics - 1 - pass; ++k) {
          const float lo = fminf(q[k], q[k + 1]), hi = fmaxf(q[k], q[k + 1]);
          q[k] = lo; q[k + 1] = hi;
        }
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCritics; ++k) if (k < keep) s = __fadd_rn(s, q[k]);
      sel = s / (float)keep;
    }
    acc += __fsub_rn(__fmul_rn(alpha, w.lp), sel);
  };
  if (mb) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    SelRow w;
    fetch(w, b);
    if (b < B) finish(w, b);
    return acc;
  }
  for (int b0 = threadIdx.x; b0 < B; b0 += 2 * blockDim.x) {
    const int b1 = b0 + blockDim.x;
    SelRow w0, w1;
    fetch(w0, b0);
    fetch(w1, b1);
    finish(w0, b0);
    if (b1 < B) finish(w1, b1);
  }
  return acc;
}
__device__ inline void actor_select_body(const ActorSelArgs& a, float* scratch) {
  const StepCtrl c = *a.cur;
  const float acc = sel_block_sum(actor_select_acc(a, false), scratch);
  if (threadIdx.x == 0) a.metrics[(long long)c.metrics_slot * kMetricFloats + MET_ACTOR_LOSS] = acc / (float)a.B;
}

__device__ inline void alpha_body(const AlphaArgs& a, float* scratch) {
  const StepCtrl c = *a.cur;
  float* met = a.metrics + (long long)c.metrics_slot * kMetricFloats;
  if (!c.do_alpha) {  // `gradient_step <= alpha_min_steps: return 0.0`
    if (threadIdx.x == 0 && a.phase != 1) { met[MET_ALPHA_LOSS] = 0.f; met[MET_ALPHA] = *a.alpha; }
    return;
  }
  if (a.phase != 1) {
    float s = 0.f;
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) s += a.logp[b] + a.target_entropy;
    s = sel_block_sum(s, scratch);
    if (threadIdx.x == 0) {
      const float mean_x = s / (float)a.B;
      met[MET_ALPHA_LOSS] = -(*a.log_alpha * mean_x);
      *a.grad_out = -mean_x;
    }
  }
  if (a.phase != 0 && threadIdx.x == 0) {
    const AlphaStep st{a.log_alpha, a.m, a.v, a.alpha, a.grad_out, a.beta2, a.w1, a.w2, a.eps, a.metrics};
    alpha_step(st, c);
  }
}


}  // namespace gcrl
