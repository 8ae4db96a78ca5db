// ops.hip — TD target + critic loss, gradient norm, Adam/AdamW (+ global-norm clip + Polyak),
// and the small per-step helper kernels.  Reference arithmetic being restated:
//   TD targets   src/agent.py:1311-1317 (DDPG), :173-186 (TD3), :557-570 (SAC), :960-979 (TQC)
//   losses       F.mse_loss / F.smooth_l1_loss (mean) and their backward
//   clip         torch.nn.utils.clip_grad_norm_  (coef = max_norm/(norm+1e-6), clamped to 1)
//   optimiser    torch.optim.Adam / AdamW single-tensor path (lerp, addcmul, addcdiv order)
//   Polyak       tau*p + (1-tau)*p_target  (src/agent.py:1260-1271 etc.)
#include "ops.h"
#include "adam_math.h"
#include "meet.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace gcrl {
namespace {

// (wave_sum, wave_sum_d, block_sum_256: adam_math.h)
// sum over a block of up to 1024 threads (whole waves); result valid in every thread.  `scratch` holds 16 floats.
// Waves beyond the data contribute exact zeros, so the result does not depend on the block size chosen for small B.
__device__ inline float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float s = scratch[0];
  for (int w = 1; w < nw; ++w) s += scratch[w];
  return s;
}

__global__ void begin_step_kernel(CtrlBlock* cb, int shift) {
  if (threadIdx.x == 0) {
    const int c = cb->cursor;
    if (shift) cb->prev = cb->cur;
    cb->cur = cb->table[c];
    cb->cursor = c + 1;
  }
}

// ------------------------------------------------------------------ TD target + loss
// one block of 256..1024 threads (launch_td_loss): at B = 2048 a 256-thread block walked 8 dependent rounds of ~13 loads
// each (31 us in TQC's step).
// Round 4: EVERY operand of a thread's (up to) two rows is requested before anything is stored.  The output pointers may alias
// the inputs as far as the compiler knows, so a row's loads of q[k + 1] used to wait behind its store of dq[k]: five dependent
// round trips per row, two rows per thread at B = 2048 — 16.6 us for a kernel whose loads fit one round trip.  Same arithmetic
// in the same order as before (rows in increasing b per thread).
// One instance per (target kind, loss kind): a single workgroup's kernel runs from a cold instruction cache, and the generic
// body was 2 845 instructions (22 KB) of which a given agent executes a third.
constexpr int kTdPartStride = 16;   // floats per wave slot of the multi-workgroup form: kMaxCritics losses, td, q
static_assert(kMaxCritics + 2 <= kTdPartStride, "td_loss partial slot");
struct TdRow { float qt[kMaxCritics], q[kMaxCritics], r, d, lp, w; };
// MB (round 4, B >= 1024 with the agent's scratch): several workgroups of 256 threads, one row per thread.  A single workgroup
// is bound by its CU's issue rate — ~2 000 wave instructions x 4 cycles x 4 waves per SIMD: 13 of the kernel's 14.6 us at
// B = 2048 — not by memory.  Every wave leaves the partial sums of the (up to) seven logged quantities at
// part[(workgroup * 4 + wave) * 16 + q] with agent-scope stores and the workgroup takes a ticket (meet.h's release discipline);
// the last one adds the partials up in index order — the same result whichever workgroup is last — and writes the metrics.
// No workgroup waits for another: nothing to time out, safe on a shared device.
template <int TGT, int LOSS, bool MB>
__global__ __launch_bounds__(MB ? 256 : 1024) void td_loss_kernel(TdLossArgs a) {
  __shared__ float scratch[16];
  const StepCtrl c = *a.cur;
  if (a.refresh && blockIdx.x == 0 && threadIdx.x == 0) { a.refresh->cur_b = c; a.refresh->prev_b = a.refresh->prev; }
  const float* __restrict__ r = a.r + (long long)c.batch_slot * a.slot_stride;
  const float* __restrict__ d = a.d + (long long)c.batch_slot * a.slot_stride;
  const float alpha = a.alpha_dev ? *a.alpha_dev : a.alpha_const;
  const int B = a.B, C = a.C;
  const int keep = C - a.drop;
  const bool ent = TGT == TGT_MIN_ENT || TGT == TGT_TRUNC_ENT;
  float loss[kMaxCritics];
#pragma unroll
  for (int k = 0; k < kMaxCritics; ++k) loss[k] = 0.f;
  float td = 0.f, qsum = 0.f;
  const float mse_norm = 2.0f / (float)B, l1_norm = 1.0f / (float)B;

  auto fetch = [&](TdRow& w, int b) {
    const bool in = b < B;
#pragma unroll
    for (int k = 0; k < kMaxCritics; ++k) {
      w.qt[k] = (in && k < C) ? a.qt[(long long)k * B + b] : INFINITY;
      w.q[k] = (in && k < C) ? a.q[(long long)k * B + b] : 0.f;
    }
    w.r = in ? r[b] : 0.f;
    w.d = in ? d[b] : 0.f;
    w.lp = (in && ent) ? a.logp_next[b] : 0.f;
    w.w = (in && a.w) ? a.w[b] : 1.0f;
  };
  auto finish = [&](TdRow& w, int b) {
    float* qt = w.qt;
    float tq;
    if (TGT == TGT_DDPG) {
      tq = qt[0];
    } else if (TGT == TGT_MIN || TGT == TGT_MIN_ENT) {
      tq = fminf(qt[0], qt[1]);
    } else {
      // torch.sort over the critic axis, drop the largest `drop`, mean (src/agent.py:972-974)
#pragma unroll
      for (int pass = 0; pass < kMaxCritics - 1; ++pass)
#pragma unroll
        for (int k = 0; k < kMaxCritics - 1 - pass; ++k) {
          const float lo = fminf(qt[k], qt[k + 1]), hi = fmaxf(qt[k], qt[k + 1]);
          qt[k] = lo; qt[k + 1] = hi;
        }
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCritics; ++k) if (k < keep) s = __fadd_rn(s, qt[k]);
      tq = s / (float)keep;
    }
    if (ent) tq = __fsub_rn(tq, __fmul_rn(alpha, w.lp));
    // y = r + gamma * (1 - d) * tq   (left to right, one rounding per op)
    float y = __fadd_rn(w.r, __fmul_rn(__fmul_rn(a.gamma, __fsub_rn(1.0f, w.d)), tq));
    if (TGT == TGT_DDPG) y = fminf(fmaxf(y, a.clamp_lo), 0.0f);
    float tdmax = 0.f;
    const float wb = w.w;   // (weights * loss).mean(): every per-sample term scaled before the mean
#pragma unroll
    for (int k = 0; k < kMaxCritics; ++k) {
      if (k < C) {
        const float qc = w.q[k];
        const float diff = __fsub_rn(qc, y);
        const float ad = fabsf(diff);
        float g;
        if (LOSS == LOSS_MSE) {
          loss[k] += a.w ? wb * (diff * diff) : diff * diff;
          g = mse_norm * diff;
        } else {
          const float l = (ad < 1.0f) ? 0.5f * diff * diff : ad - 0.5f;
          loss[k] += a.w ? wb * l : l;
          g = (diff < -1.0f) ? -l1_norm : (diff > 1.0f ? l1_norm : l1_norm * diff);
        }
        if (a.w) g *= wb;
        a.dq[(long long)k * B + b] = g;
        tdmax = fmaxf(tdmax, ad);
        qsum += qc;
      }
    }
    td += tdmax;
    if (a.td_abs) a.td_abs[b] = tdmax;
  };
  // software-pipelined over the thread's rows: the next row's operands are in flight while this one is finished (one copy of
  // the arithmetic in the code)
  TdRow w;
  if (MB) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    fetch(w, b);
    if (b < B) finish(w, b);
  } else {
    fetch(w, threadIdx.x);
#pragma unroll 1
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      TdRow nx;
      fetch(nx, b + blockDim.x);
      finish(w, b);
      w = nx;
    }
  }
  float* met = a.metrics + (long long)c.metrics_slot * kMetricFloats;
  if (MB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = a.part + ((long long)blockIdx.x * 4 + wave) * kTdPartStride;
#pragma unroll
    for (int k = 0; k < kMaxCritics; ++k) {
      if (k < C) {
        const float v = wave_sum(loss[k]);
        if (lane == 0) __hip_atomic_store(mine + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const float tdw = wave_sum(td), qsw = wave_sum(qsum);
    if (lane == 0) {
      __hip_atomic_store(mine + kMaxCritics, tdw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + kMaxCritics + 1, qsw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    drain_stores();
    __syncthreads();
    unsigned int* s_ticket = reinterpret_cast<unsigned int*>(scratch);
    if (threadIdx.x == 0) *s_ticket = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_ticket != gridDim.x - 1) return;      // (uniform)
    if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    const int q = threadIdx.x, np = 4 * gridDim.x;
    if (q >= kMaxCritics + 2 || (q < kMaxCritics && q >= C)) return;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < np; ++i) s += __hip_atomic_load(a.part + (long long)i * kTdPartStride + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (q < kMaxCritics) met[MET_CRITIC_LOSS + q] = s / (float)B;
    else if (q == kMaxCritics) met[MET_TD] = s / (float)B;
    else met[MET_Q] = s / (float)(B * C);
    return;
  }
#pragma unroll
  for (int k = 0; k < kMaxCritics; ++k) {
    if (k < C) {
      const float s = block_sum(loss[k], scratch);
      if (threadIdx.x == 0) met[MET_CRITIC_LOSS + k] = s / (float)B;
    }
  }
  const float tds = block_sum(td, scratch);
  const float qs = block_sum(qsum, scratch);
  if (threadIdx.x == 0) {
    met[MET_TD] = tds / (float)B;
    met[MET_Q] = qs / (float)(B * C);
  }
}

__global__ __launch_bounds__(1024) void mean_metric_kernel(const StepCtrl* cur, const float* x, int n,
                                                           float scale, float* metrics, int idx) {
  __shared__ float scratch[16];
  float s = 0.f;
#pragma unroll 8
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) metrics[(long long)cur->metrics_slot * kMetricFloats + idx] = scale * (s / (float)n);
}

__global__ void fill_kernel(float* x, long long n, float v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

__global__ void td3_smooth_kernel(const StepCtrl* cur, float* act, long long slot_stride, int ld, int B,
                                  int A, const float* eps, float pn, float nc, unsigned long long seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * A) return;
  const StepCtrl c = *cur;
  const int b = i / A, j = i - b * A;
  float* p = act + (long long)c.batch_slot * slot_stride + (long long)b * ld + j;
  const float e = eps ? eps[i] : hash_normal(seed, (((unsigned long long)c.rng_hi << 32) | c.rng_lo) + (unsigned long long)i);
  const float noise = fminf(fmaxf(__fmul_rn(e, pn), -nc), nc);
  *p = fminf(fmaxf(__fadd_rn(*p, noise), -1.0f), 1.0f);
}

// ------------------------------------------------------------------ gradient norm
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long long n, long long net_stride,
                                                    float* partial) {
  __shared__ float scratch[4];
  const float* gp = g + (long long)blockIdx.y * net_stride;
  const long long chunk = (n + kNormBlocks - 1) / kNormBlocks;
  const long long beg = (long long)blockIdx.x * chunk;
  const long long end = beg + chunk < n ? beg + chunk : n;
  float s = 0.f;
  for (long long i = beg + threadIdx.x; i < end; i += 256) s += gp[i] * gp[i];
  s = block_sum_256(s, scratch);
  if (threadIdx.x == 0) partial[blockIdx.y * kNormBlocks + blockIdx.x] = s;
}

// two vectors of different length, one launch (data-parallel overlapped step: critic and actor norms)
__global__ __launch_bounds__(256) void sumsq2_kernel(const float* g0, long long n0, const float* g1, long long n1,
                                                     float* partial0, float* partial1) {
  __shared__ float scratch[4];
  const float* gp = blockIdx.y ? g1 : g0;
  const long long n = blockIdx.y ? n1 : n0;
  const long long chunk = (n + kNormBlocks - 1) / kNormBlocks;
  const long long beg = (long long)blockIdx.x * chunk;
  const long long end = beg + chunk < n ? beg + chunk : n;
  float s = 0.f;
  for (long long i = beg + threadIdx.x; i < end; i += 256) s += gp[i] * gp[i];
  s = block_sum_256(s, scratch);
  if (threadIdx.x == 0) (blockIdx.y ? partial1 : partial0)[blockIdx.x] = s;
}

// ------------------------------------------------------------------ Adam / AdamW
// riders_here: this launch's metric riders run in workgroup 0 of net 0 (the paired launch); else in a workgroup of their own (adam_kernel).
// nblocks: workgroups stepping elements (the flat form's stride)
__device__ inline void adam_riders(const AdamArgs& a, const StepCtrl& c) {
  if (a.mean_x) rider_mean_metric(a.mean_x, a.mean_n, a.mean_scale, a.metrics + (long long)c.metrics_slot * kMetricFloats + a.mean_index);
  if (a.td_q) rider_td_metrics(a.td_q, a.td_y, a.td_n, a.td_C, a.td_loss_kind, a.metrics + (long long)c.metrics_slot * kMetricFloats);
}
__device__ inline void adam_body(const AdamArgs& a, const int net, const bool riders_here = true, const unsigned nblocks = gridDim.x) {
  __shared__ float s_coef;
  const StepCtrl c = *a.cur;
  const AdamStepScalars sc = adam_scalars(c, a.which);
  const float gscale = c.grad_scale;
  const long long base = (long long)net * a.net_stride;
  float* __restrict__ p = a.p + base;
  const float* __restrict__ g = a.g + base;
  float* __restrict__ m = a.m + base;
  float* __restrict__ v = a.v + base;
  float* __restrict__ tp = a.target ? a.target + base : nullptr;
  // segmented launches: this thread's element is known before anything is loaded, so its operands are
  // requested NOW and arrive while the norm partials are being reduced (one memory round trip less on the
  // critical path of a kernel that is nothing but round trips)
  // (round 5: FOUR elements per thread — a block covers 1 024 flat elements or a 32 x 32 tile as four 16 x 16 sub-tiles.  With one element per
  // thread TD3's twin critics were 1 080 blocks of a few dependent round trips each, four waves of blocks on 256 CUs: 13.2 us for 7.7 MB.)
  AdamSeg sg;
  sg.tiled = 0; sg.nblk = 0;
  int lb = 0;
  long long my_i[kAdamPerThread];
#pragma unroll
  for (int u = 0; u < kAdamPerThread; ++u) my_i[u] = -1;
  if (a.n_seg > 0) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < kMaxAdamSeg; ++q)
      if (q < a.n_seg && (int)blockIdx.x >= a.seg[q].blk0) s = q;
    sg = a.seg[s];
    lb = (int)blockIdx.x - sg.blk0;
    if (lb < sg.nblk) {
      if (!sg.tiled) {
#pragma unroll
        for (int u = 0; u < kAdamPerThread; ++u) {
          const long long i = sg.beg + ((long long)lb * kAdamPerThread + u) * 256 + threadIdx.x;
          if (i < sg.beg + (long long)sg.rows * sg.cols) my_i[u] = i;
        }
      } else {
        const int tiles_k = (sg.cols + kAdamTile - 1) / kAdamTile;
#pragma unroll
        for (int u = 0; u < kAdamPerThread; ++u) {   // sub-tile u: rows + 16 * (u >> 1), columns + 16 * (u & 1)
          const int o = (lb / tiles_k) * kAdamTile + 16 * (u >> 1) + (threadIdx.x >> 4), k = (lb % tiles_k) * kAdamTile + 16 * (u & 1) + (threadIdx.x & 15);
          if (o < sg.rows && k < sg.cols) my_i[u] = sg.beg + (long long)o * sg.cols + k;
        }
      }
    }
  }
  float pre_g[kAdamPerThread], pre_p[kAdamPerThread], pre_m[kAdamPerThread], pre_v[kAdamPerThread], pre_t[kAdamPerThread];
#pragma unroll
  for (int u = 0; u < kAdamPerThread; ++u) {
    pre_g[u] = pre_p[u] = pre_m[u] = pre_v[u] = pre_t[u] = 0.f;
    if (my_i[u] >= 0) {
      pre_g[u] = g[my_i[u]]; pre_p[u] = p[my_i[u]]; pre_m[u] = m[my_i[u]]; pre_v[u] = v[my_i[u]];
      if (tp && a.polyak) pre_t[u] = tp[my_i[u]];
    }
  }
  {
    // ||g||: every block sums the same partials in the same order (deterministic), in fp64
    __shared__ double dred[4];
    const float* part = a.partial + (long long)net * a.part_stride;
    double s = 0.0;
    // independent 16-byte loads (a dependent scalar loop here cost ~4 us of L2 latency per launch)
    const int n4 = ((reinterpret_cast<uintptr_t>(part) & 15) == 0) ? a.nparts >> 2 : 0;
    const float4* part4 = reinterpret_cast<const float4*>(part);
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = part4[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
    for (int i = 4 * n4 + threadIdx.x; i < a.nparts; i += 256) s += (double)part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      s = dred[0] + dred[1] + dred[2] + dred[3];
      float post;
      s_coef = clip_coef(s, gscale, a.clip[net], &post);
      if (blockIdx.x == 0 && a.metrics)
        a.metrics[(long long)c.metrics_slot * kMetricFloats + a.metric_index + net] = post;
    }
  }
  __syncthreads();
  if (riders_here && blockIdx.x == 0 && net == 0) adam_riders(a, c);
  const float gmul = gscale * s_coef;
  const float w1 = a.w1, w2 = a.w2, one_m_tau = a.one_m_tau;
  const bool pk = tp && a.polyak;
  // one element: torch's single-tensor Adam(W) op order; returns the new parameter, *ti the new target
  auto step_vals = [&](long long i, float g_raw, float pi, float mi, float v_old, float t_old, float* ti) -> float {
    const AdamElem e = adam_elem(g_raw, pi, mi, v_old, gmul, sc, a.beta2, w1, w2, a.eps);
    p[i] = e.p; m[i] = e.m; v[i] = e.v;
    if (pk) { *ti = polyak_elem(a.tau, e.p, one_m_tau, t_old); tp[i] = *ti; }
    return e.p;
  };
  if (a.n_seg == 0) {
    float ti;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)nblocks * 256)
      step_vals(i, g[i], p[i], m[i], v[i], pk ? tp[i] : 0.f, &ti);
    return;
  }
  if (lb >= sg.nblk) return;   // paired launches are sized for the larger net
  float ti[kAdamPerThread], pi[kAdamPerThread];
#pragma unroll
  for (int u = 0; u < kAdamPerThread; ++u) {
    ti[u] = 0.f; pi[u] = 0.f;
    if (my_i[u] >= 0) pi[u] = step_vals(my_i[u], pre_g[u], pre_p[u], pre_m[u], pre_v[u], pre_t[u], &ti[u]);
  }
  if (!sg.tiled) return;
  __shared__ float tile_p[kAdamPerThread][16][17], tile_t[kAdamPerThread][16][17];
  const int tiles_k = (sg.cols + kAdamTile - 1) / kAdamTile;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
  for (int u = 0; u < kAdamPerThread; ++u) { tile_p[u][ty][tx] = pi[u]; tile_t[u][ty][tx] = ti[u]; }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kAdamPerThread; ++u) {
    const int o0 = (lb / tiles_k) * kAdamTile + 16 * (u >> 1), k0 = (lb % tiles_k) * kAdamTile + 16 * (u & 1);
    const int k = k0 + ty, o = o0 + tx;   // 16 consecutive o per copy row: 64-byte runs
    if (k < sg.cols && o < sg.rows) {
      const long long at = (long long)net * a.wt_net_stride + sg.dst + (long long)k * sg.rows + o;
      a.wt[at] = tile_p[u][tx][ty];
      if (pk && a.wt_target) a.wt_target[at] = tile_t[u][tx][ty];
    }
  }
}

__device__ inline void advance_ctrl(const AdamArgs& a) {
  if (a.advance && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    ctrl_advance(a.advance);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  // round 5: the metric riders (TD metrics of 2 048 rows x 2 critics at TD3's cfg 3: one memory round trip and four block reductions) have a
  // workgroup of their own — the launch's last — instead of extending workgroup 0's norm -> step chain
  const bool extra = a.mean_x || a.td_q;
  if (extra && blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y == 0) adam_riders(a, *a.cur);
    return;
  }
  adam_body(a, blockIdx.y, false, gridDim.x - (extra ? 1u : 0u));
  if (a.alpha.log_alpha && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) alpha_step(a.alpha, *a.cur);
  advance_ctrl(a);
}

// two independent single-net optimiser steps in one launch (software-pipelined DDPG: the critic of
// step i+1 and the actor of step i), blockIdx.y picks the argument set
__global__ __launch_bounds__(256) void adam_pair_kernel(AdamArgs a0, AdamArgs a1) {
  if (blockIdx.y == 0) adam_body(a0, 0);
  else adam_body(a1, 0);
  advance_ctrl(a0);
}

__global__ void polyak_kernel(const float* p, float* tp, long long n, float tau, float one_m_tau) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tp[i] = __fadd_rn(__fmul_rn(tau, p[i]), __fmul_rn(one_m_tau, tp[i]));
}

}  // namespace

// threads of a single-block reduction over n items: whole waves, 256..1024
static inline unsigned reduce_threads(long long n) { return (unsigned)std::min<long long>(1024, std::max<long long>(256, (n + 63) / 64 * 64)); }

int launch_begin_step(hipStream_t st, CtrlBlock* cb, int shift) {
  hipLaunchKernelGGL(begin_step_kernel, dim3(1), dim3(64), 0, st, cb, shift);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_td_loss(hipStream_t st, const TdLossArgs& a) {
  GCRL_CHECK_ARG(a.C >= 1 && a.C <= kMaxCritics && a.B >= 1, "td_loss: bad C=%d B=%d", a.C, a.B);
  const bool mb = a.part && a.ticket && a.B >= 1024;
  const dim3 th(mb ? 256 : reduce_threads(a.B)), gr(mb ? (a.B + 255) / 256 : 1);
#define GCRL_TD(T, L) if (a.target_kind == T && a.loss_kind == L) { if (mb) hipLaunchKernelGGL((td_loss_kernel<T, L, true>), gr, th, 0, st, a); \
                                                                      else hipLaunchKernelGGL((td_loss_kernel<T, L, false>), gr, th, 0, st, a); } else
  GCRL_TD(TGT_DDPG, LOSS_MSE) GCRL_TD(TGT_MIN, LOSS_SMOOTH_L1) GCRL_TD(TGT_MIN_ENT, LOSS_MSE) GCRL_TD(TGT_TRUNC_ENT, LOSS_MSE)
  GCRL_TD(TGT_DDPG, LOSS_SMOOTH_L1) GCRL_TD(TGT_MIN, LOSS_MSE) GCRL_TD(TGT_MIN_ENT, LOSS_SMOOTH_L1) GCRL_TD(TGT_TRUNC_ENT, LOSS_SMOOTH_L1)
  return fail(GCRL_ERR_ARG, "td_loss: bad target kind %d / loss kind %d", a.target_kind, a.loss_kind);
#undef GCRL_TD
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_mean_metric(hipStream_t st, const StepCtrl* cur, const float* x, int n, float scale,
                       float* metrics, int idx) {
  hipLaunchKernelGGL(mean_metric_kernel, dim3(1), dim3(reduce_threads(n)), 0, st, cur, x, n, scale, metrics, idx);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_fill(hipStream_t st, float* x, long long n, float v) {
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, v);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_td3_smooth(hipStream_t st, const StepCtrl* cur, float* act, long long slot_stride, int ld,
                      int B, int A, const float* eps, float pn, float nc, unsigned long long seed) {
  hipLaunchKernelGGL(td3_smooth_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, cur, act,
                     slot_stride, ld, B, A, eps, pn, nc, seed);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_sumsq(hipStream_t st, const float* g, long long n, long long net_stride, int nets,
                 float* partial) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks, nets), dim3(256), 0, st, g, n, net_stride, partial);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

// one element per thread up to 1024 blocks per net: the kernel is a chain of memory round trips
// (control record, norm partials, operands), so width beats per-thread work (4 elements per thread:
// 10.8 us for the 2 x 140k-parameter pair, 1: 8.6 us; 16: 21.8 us)
static unsigned adam_blocks(long long n) {
  long long blocks = (n + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  return (unsigned)blocks;
}

int launch_sumsq2(hipStream_t st, const float* g0, long long n0, float* partial0, const float* g1, long long n1,
                  float* partial1) {
  hipLaunchKernelGGL(sumsq2_kernel, dim3(kNormBlocks, 2), dim3(256), 0, st, g0, n0, g1, n1, partial0, partial1);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_adam(hipStream_t st, const AdamArgs& a) {
  GCRL_CHECK_ARG(a.nets >= 1 && a.nets <= kMaxCritics, "adam: bad net count %d", a.nets);
  const unsigned rider = (a.mean_x || a.td_q) ? 1u : 0u;   // (the riders' own workgroup: adam_kernel)
  hipLaunchKernelGGL(adam_kernel, dim3((a.n_seg ? (unsigned)a.seg_blocks : adam_blocks(a.n)) + rider, a.nets), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_adam_pair(hipStream_t st, const AdamArgs& a0, const AdamArgs& a1) {
  GCRL_CHECK_ARG(a0.nets == 1 && a1.nets == 1, "adam_pair: single-net argument sets only");
  const unsigned b0 = a0.n_seg ? (unsigned)a0.seg_blocks : adam_blocks(a0.n);
  const unsigned b1 = a1.n_seg ? (unsigned)a1.seg_blocks : adam_blocks(a1.n);
  hipLaunchKernelGGL(adam_pair_kernel, dim3(std::max(b0, b1), 2), dim3(256), 0, st, a0, a1);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_polyak(hipStream_t st, const float* p, float* target, long long n, double tau) {
  hipLaunchKernelGGL(polyak_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, target, n,
                     (float)tau, (float)(1.0 - tau));
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
