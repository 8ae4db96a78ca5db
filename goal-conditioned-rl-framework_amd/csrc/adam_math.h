// adam_math.h — the arithmetic the optimiser launches share (ops.hip: adam_kernel / adam_pair_kernel; dw_adam.hip: the launch
// that forms the weight gradients AND steps them).  One definition each, so an element's new parameter / moments / target, the
// clip norm and the riding metrics are the same bits whichever launch computed them.
// Reference: torch.optim.Adam / AdamW single-tensor path (lerp, addcmul, addcdiv order), torch.nn.utils.clip_grad_norm_,
// Polyak src/agent.py:1260-1271, metrics src/agent.py:1296-1300, :1334-1343.
#pragma once
#include <hip/hip_runtime.h>

#include "ops.h"

namespace gcrl {

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// sum over a 256-thread block; result valid in every thread.  `scratch` holds 4 floats.
__device__ inline float block_sum_256(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// the StepCtrl triple of an optimiser (0 actor, 1 critic, 2 alpha)
struct AdamStepScalars { float step_size, bc2s, decay; };
__device__ inline AdamStepScalars adam_scalars(const StepCtrl& c, int which) {
  if (which == 0) return {c.step_size_actor, c.bc2s_actor, c.decay_actor};
  if (which == 1) return {c.step_size_critic, c.bc2s_critic, c.decay_critic};
  return {c.step_size_alpha, c.bc2s_alpha, c.decay_alpha};
}

// one element: torch's single-tensor Adam(W) op order.  g_raw: the raw gradient; gmul = grad_scale * clip coefficient.
struct AdamElem { float p, m, v; };
__device__ inline AdamElem adam_elem(float g_raw, float pi, float mi, float v_old, float gmul, const AdamStepScalars& s, float beta2, float w1,
                                     float w2, float eps) {
  const float gi = __fmul_rn(g_raw, gmul);
  if (s.decay != 1.0f) pi = __fmul_rn(pi, s.decay);
  mi = __fadd_rn(mi, __fmul_rn(w1, __fsub_rn(gi, mi)));
  const float vi = __fadd_rn(__fmul_rn(v_old, beta2), __fmul_rn(__fmul_rn(w2, gi), gi));
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), s.bc2s), eps);
  pi = __fadd_rn(pi, __fdiv_rn(__fmul_rn(-s.step_size, mi), denom));
  return {pi, mi, vi};
}
__device__ inline float polyak_elem(float tau, float p_new, float one_m_tau, float t_old) {
  return __fadd_rn(__fmul_rn(tau, p_new), __fmul_rn(one_m_tau, t_old));
}

// clip_grad_norm_'s coefficient from the net's sum of squares (fp64) — and the post-clip norm the reference logs
__device__ inline float clip_coef(double sumsq, float gscale, float clip, float* norm_out) {
  const float norm = gscale * (float)sqrt(sumsq);
  float coef = 1.0f;
  if (clip >= 0.f) coef = fminf(clip / (norm + 1e-6f), 1.0f);
  *norm_out = norm * coef;
  return coef;
}

// riders of an optimiser launch's first workgroup (256 threads; every thread calls): a scalar mean ...
__device__ inline void rider_mean_metric(const float* x, int n, float scale, float* dst) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {   // (eight loads in flight per thread: see rider_td_metrics)
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = i0 + u * 256 < n ? x[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (i0 + u * 256 < n) s += xv[u];
  }
  s = block_sum_256(s, scratch);
  if (threadIdx.x == 0) *dst = scale * (s / (float)n);
}
// ... and the TD metrics of a critic step whose loss was formed inside the row-chain launch: the same sums, in the same order,
// as td_loss_kernel forms (q: [C][n], y: [n])
__device__ inline void rider_td_metrics(const float* q, const float* y, int n, int C, int loss_kind, float* met) {
  __shared__ float scratch[4];
  float loss[2] = {0.f, 0.f}, td = 0.f, qs = 0.f;
  // eight rows' operands per thread are requested before any is used (round 5: as a plain loop the 2 048-row TD3 batch was eight dependent
  // memory round trips in ONE workgroup — the optimiser launch's long pole: 13.2 us where the other workgroups take 5); same sums, same order
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
    float yv[8], qv[2][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      yv[u] = i < n ? y[i] : 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) qv[k][u] = (k < C && i < n) ? q[(long long)k * n + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + u * 256 >= n) break;
      float tdmax = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k < C) {
          const float diff = __fsub_rn(qv[k][u], yv[u]);
          const float ad = fabsf(diff);
          if (loss_kind == LOSS_MSE) loss[k] += diff * diff;
          else loss[k] += (ad < 1.0f) ? 0.5f * diff * diff : ad - 0.5f;
          tdmax = fmaxf(tdmax, ad);
          qs += qv[k][u];
        }
      }
      td += tdmax;
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k < C) {
      const float s = block_sum_256(loss[k], scratch);
      if (threadIdx.x == 0) met[MET_CRITIC_LOSS + k] = s / (float)n;
    }
  }
  td = block_sum_256(td, scratch);
  qs = block_sum_256(qs, scratch);
  if (threadIdx.x == 0) {
    met[MET_TD] = td / (float)n;
    met[MET_Q] = qs / (float)(n * C);
  }
}

// prev <- cur, cur <- table[cursor++]: what begin_step(shift) would do for the next step (one thread)
__device__ inline void ctrl_advance(CtrlBlock* cb) {
  const int c = cb->cursor;
  cb->prev = cb->cur;
  cb->cur = cb->table[c];
  cb->cursor = c + 1;
}

}  // namespace gcrl
