// ops_sac.hip — kernels only the stochastic-actor agents need:
//   BatchNorm1d(train)+ReLU fwd/bwd and eval        reference src/model.py:106-108
//   tanh-squashed Gaussian sample / log-prob        src/model.py:125-141
//   actor-loss critic selection (min / sort-trunc)  src/agent.py:516-521, :916-925
//   log-alpha AdamW step                            src/agent.py:532-546, :936-949
//   wavefront bitonic sort + truncated mean         generalisation of src/agent.py:919-921
#include "ops.h"
#include "meet.h"
#include "gemm_mfma.h"   // v4f
#include "sac_heads.h"
#include "sac_select.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace gcrl {
namespace {

constexpr float kBnEps = 1e-5f;       // nn.BatchNorm1d default eps
// (kBnMomentum, bn_running_update, tanh_gauss_elem: sac_select.h)

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
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
__device__ inline float block_sum_256(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
// (counter-hash normals: gcrl::hash_normal, ops.h — the one definition)

// BatchNorm statistics in two fully parallel, deterministic stages (a single column-owning block
// per 64 features left 252 of 256 CUs idle and serialised over the batch):
//   stage 1  grid (H/64, B/64): block (cb, rb) reduces its 64 rows x 64 columns to a per-column
//            partial — forward: (mean_b, M2_b) by a local two-pass; backward: (sum dy, sum dy*xhat)
//   stage 2  grid (H/64, B/16): every block merges the <= B/64 partials of its columns in the same
//            fixed order (Chan's parallel-variance merge / plain sums), then transforms its 16 rows.
// Thread layout: 16 column quads (one float4 = 4 columns per lane) x 16 row slots, so a wave-load moves four
// 256-B row segments.  These kernels move 0.5-2 MB and are nothing but memory round trips; what a launch costs
// is the number of vector-memory instructions a CU has to issue (64 in flight, ~1 us per batch of 64): with
// one column per lane the apply kernels issued 80 loads per thread — 16 rows + 64 row-block partials that all
// 64 blocks re-read lane by lane — and took 11.5 us (profiles/r02_kernel_stats_sac_slide_b512.csv).  Now a
// thread issues 4-6 float4 loads: its row(s), the affine pair, and ONE slice of the partials, which the block
// gathers cooperatively into LDS before every thread merges them.
// (Measured alternatives, both slower at B = 512 / H = 256: one launch with a block owning 16 columns and all
// rows — 16 blocks, 2 695 vs 2 773 SAC steps/s; one launch on this grid with every block re-reducing its
// columns over all rows — 128 dependent-latency loads per thread, 2 176 steps/s.)
constexpr int kBnRows = 64;      // rows per partial
constexpr int kBnSlots = 16;     // row slots of a block (256 threads / 16 column quads)
constexpr int kBnMaxPart = 32;   // row-block partials a block holds in registers (two per row slot: B <= 2048); more are read in the merge loops

__device__ inline v4f ld4(const float* p) { return *(const v4f*)p; }
__device__ inline v4f zero4() { return (v4f){0.f, 0.f, 0.f, 0.f}; }

// sum over the block's 16 row slots for every column quad (all threads get the result; same order everywhere)
__device__ inline v4f slot_sum(v4f v, v4f (*red)[16], int cq, int slot) {
  __syncthreads();
  red[slot][cq] = v;
  __syncthreads();
  v4f s = red[0][cq];
#pragma unroll
  for (int i = 1; i < kBnSlots; ++i) s += red[i][cq];
  return s;
}

// Merge of a problem's row-block partials, shared by the 16 row slots of a block (round 4).  Slot s owns partials s, s + 16, ...:
// the first two are requested straight into registers (with the caller's other loads, before anything waits), the rest —
// B > 2048, or SyncBN's world x nrb partials — are read in the merge loops.  Rounds 1-3 had every slot gather all partials
// through LDS and merge them redundantly: 64 dependent LDS reads per thread, 5.8 of the 11.1 us of a B = 2048, H = 512 launch
// (GCRL_BN_ABL ablation, tools/bn_layer_bench.py).
struct PartRegs { v4f a[2], b[2]; };
__device__ inline void part_request(PartRegs& r, const float* pa, const float* pb, int nrb, int H, int col, bool ok, int slot) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rb = slot + kBnSlots * u;
    const bool in = ok && rb < nrb;
    r.a[u] = in ? ld4(pa + (long long)rb * H + col) : zero4();
    r.b[u] = in ? ld4(pb + (long long)rb * H + col) : zero4();
  }
}

constexpr int kBnSlabMax = 4;
struct BnFwdPair { BnFwdProb p[2]; int n; int rpp; int slabs; int world, rank; };   // rpp: rows per partial (64: bn_stats_kernel, 16: GEMM epilogue); world / rank: BnSync

__global__ __launch_bounds__(256) void bn_stats_kernel(BnFwdPair pr, int B, int H) {
  __shared__ v4f red[kBnSlots][16];
  const float* __restrict__ z = pr.p[blockIdx.z].z;
  const int nrb_ = (B + kBnRows - 1) / kBnRows, nrbT = nrb_ * pr.world;
  float* __restrict__ part_mean = pr.p[blockIdx.z].scratch;
  float* __restrict__ part_m2 = part_mean + (long long)nrbT * H;
  const int cq = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cq;
  const int r0 = blockIdx.y * kBnRows, r1 = min(B, r0 + kBnRows);
  const bool ok = col < H;
  constexpr int NR = kBnRows / kBnSlots;
  v4f v[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int b = r0 + slot + kBnSlots * i;
    v[i] = (ok && b < r1) ? ld4(z + (long long)b * H + col) : zero4();
  }
  v4f s = zero4();
#pragma unroll
  for (int i = 0; i < NR; ++i) if (r0 + slot + kBnSlots * i < r1) s += v[i];
  const v4f mean = slot_sum(s, red, cq, slot) / (float)(r1 - r0);
  v4f q = zero4();
#pragma unroll
  for (int i = 0; i < NR; ++i) if (r0 + slot + kBnSlots * i < r1) { const v4f df = v[i] - mean; q += df * df; }
  const v4f m2 = slot_sum(q, red, cq, slot);
  if (ok && slot == 0) {
    *(v4f*)(part_mean + (long long)(pr.rank * nrb_ + blockIdx.y) * H + col) = mean;
    *(v4f*)(part_m2 + (long long)(pr.rank * nrb_ + blockIdx.y) * H + col) = m2;
  }
  if (ok && pr.world > 1 && slot > 0 && slot <= pr.world) {   // BnSync: the other ranks' slots of this row block := 0 (the exchange sums)
    const int g = slot - 1;
    if (g != pr.rank) {
      *(v4f*)(part_mean + (long long)(g * nrb_ + blockIdx.y) * H + col) = zero4();
      *(v4f*)(part_m2 + (long long)(g * nrb_ + blockIdx.y) * H + col) = zero4();
    }
  }
}

// Merge of a problem's row-block partials (mean_b, M2_b over n_b rows) for one column quad, the same in every block:
//   mean = sum n_b mean_b / B,   M2 = sum (M2_b + n_b (mean_b - mean)^2)      -> (mean, biased variance)
// — two passes without a division in the loop (Chan's sequential merge is a serial chain of divisions), each a sum over this
// slot's partials followed by the block's slot sum (fixed order: the result is the same in every block and on every rank).
// BnSync: nrb = world * nrb_local partials (rank-major), each over the rows of ITS rank's block; Bt = world * B rows in all.
__device__ inline void bn_merge_slots(const PartRegs& r, const float* gm, const float* gq, v4f (*red)[16], int cq, int slot, int H, int col,
                                      bool ok, int nrb, int B, int rpp, int world, v4f* mean_out, v4f* var_out) {
  const int nrb_l = nrb / world;
  const float Bt = (float)B * (float)world;
  auto rows = [&](int rb) { const int q = rb % nrb_l; return (float)(min(B, (q + 1) * rpp) - q * rpp); };
  v4f s = zero4();
#pragma unroll
  for (int u = 0; u < 2; ++u) if (slot + kBnSlots * u < nrb) s += r.a[u] * rows(slot + kBnSlots * u);
  for (int rb = slot + 2 * kBnSlots; rb < nrb; rb += kBnSlots) s += (ok ? ld4(gm + (long long)rb * H + col) : zero4()) * rows(rb);
  const v4f mean = slot_sum(s, red, cq, slot) / Bt;
  v4f m2 = zero4();
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (slot + kBnSlots * u < nrb) { const v4f dm = r.a[u] - mean; m2 += r.b[u] + dm * dm * rows(slot + kBnSlots * u); }
  for (int rb = slot + 2 * kBnSlots; rb < nrb; rb += kBnSlots) {
    const v4f dm = (ok ? ld4(gm + (long long)rb * H + col) : zero4()) - mean;
    m2 += (ok ? ld4(gq + (long long)rb * H + col) : zero4()) + dm * dm * rows(rb);
  }
  *mean_out = mean; *var_out = slot_sum(m2, red, cq, slot) / Bt;
}

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(BnFwdPair pr, int B, int H, const float* gamma, const float* beta,
                                                            float* rmean, float* rvar) {
  __shared__ v4f red[kBnSlots][16];
  const int cq = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cq;
  const bool ok = col < H;
  const int rpp = pr.rpp, nrb = ((B + rpp - 1) / rpp) * pr.world;      // (BnSync: every rank's partials)
  const BnFwdProb me = pr.p[blockIdx.z];
  const float* __restrict__ z = me.z;
  const float* __restrict__ part_mean = me.scratch;
  const float* __restrict__ part_m2 = me.scratch + (long long)nrb * H;
  // a block transforms `slabs` slabs of 16 rows (launcher: 1 up to B = 1024, then B/1024): the merge of the partials is paid
  // once per block
  const int slabs = pr.slabs;
  // request everything first: the partials' slice, the affine pair, this thread's rows
  // the block that updates the running statistics of a two-problem launch also needs problem 1's merged statistics
  const bool second = blockIdx.y == 0 && blockIdx.z == 0 && pr.n > 1;   // (uniform per block)
  PartRegs prg, prg1;
  part_request(prg, part_mean, part_m2, nrb, H, col, ok, slot);
  if (second) part_request(prg1, pr.p[1].scratch, pr.p[1].scratch + (long long)nrb * H, nrb, H, col, ok, slot);
  const v4f g = ok ? ld4(gamma + col) : zero4(), bt = ok ? ld4(beta + col) : zero4();
  v4f v[kBnSlabMax];
#pragma unroll
  for (int sl = 0; sl < kBnSlabMax; ++sl) {
    const int b = (blockIdx.y * slabs + sl) * kBnSlots + slot;
    v[sl] = (ok && sl < slabs && b < B) ? ld4(z + (long long)b * H + col) : zero4();
  }
  v4f rm0 = zero4(), rv0 = zero4();
  if (blockIdx.y == 0 && blockIdx.z == 0 && slot == 0 && ok) { rm0 = ld4(rmean + col); rv0 = ld4(rvar + col); }
  v4f mean, var;
  bn_merge_slots(prg, part_mean, part_m2, red, cq, slot, H, col, ok, nrb, B, rpp, pr.world, &mean, &var);   // biased variance: what normalises the batch
  v4f invstd;
#pragma unroll
  for (int q = 0; q < 4; ++q) invstd[q] = 1.0f / sqrtf(var[q] + kBnEps);
#pragma unroll
  for (int sl = 0; sl < kBnSlabMax; ++sl) {
    const int b = (blockIdx.y * slabs + sl) * kBnSlots + slot;
    if (ok && sl < slabs && b < B) {
      const long long idx = (long long)b * H + col;
      const v4f xh = (v[sl] - mean) * invstd;
      const v4f y = xh * g + bt;
      v4f hv;
#pragma unroll
      for (int q = 0; q < 4; ++q) hv[q] = y[q] > 0.f ? y[q] : 0.f;
      *(v4f*)(me.h + idx) = hv;
      if (me.xhat) *(v4f*)(me.xhat + idx) = xh;
    }
  }
  if (ok && blockIdx.y == 0 && slot == 0 && me.invstd) *(v4f*)(me.invstd + col) = invstd;
  // running statistics: problem 0's batch, then problem 1's (two forward calls, in that order)
  if (blockIdx.y == 0 && blockIdx.z == 0) {
    const int Bt = B * pr.world;
    const float ub = Bt > 1 ? (float)Bt / (float)(Bt - 1) : 1.0f;
    v4f rm = (1.0f - kBnMomentum) * rm0 + kBnMomentum * mean;
    v4f rv = (1.0f - kBnMomentum) * rv0 + kBnMomentum * (var * ub);
    if (second) {
      const float* pm1 = pr.p[1].scratch;
      const float* pq1 = pm1 + (long long)nrb * H;
      v4f m1, v1;
      bn_merge_slots(prg1, pm1, pq1, red, cq, slot, H, col, ok, nrb, B, rpp, pr.world, &m1, &v1);
      rm = (1.0f - kBnMomentum) * rm + kBnMomentum * m1;
      rv = (1.0f - kBnMomentum) * rv + kBnMomentum * (v1 * ub);
    }
    if (ok && slot == 0) {
      *(v4f*)(rmean + col) = rm;
      *(v4f*)(rvar + col) = rv;
    }
  }
}

__global__ void bn_relu_eval_kernel(const float* z, int B, int H, const float* gamma, const float* beta,
                                    const float* rmean, const float* rvar, float* h) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H) return;
  const int col = (int)(i % H);
  const float y = (z[i] - rmean[col]) * (1.0f / sqrtf(rvar[col] + kBnEps)) * gamma[col] + beta[col];
  h[i] = y > 0.f ? y : 0.f;
}

// backward: dy = dh (+ dh2) where the forward's output was positive.  The mask is recomputed from xhat exactly as the
// forward computed it (y = xhat*gamma + beta, the same two fp32 operations), so the saved h is not read again.
__device__ inline v4f bn_dy(v4f dh, v4f xh, v4f g, v4f bt) {
  const v4f y = xh * g + bt;
  v4f dy;
#pragma unroll
  for (int q = 0; q < 4; ++q) dy[q] = y[q] > 0.f ? dh[q] : 0.f;
  return dy;
}

__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dh, const float* __restrict__ dh2,
                                                           const float* __restrict__ xhat, const float* gamma, const float* beta,
                                                           int B, int H, float* __restrict__ part_dy,
                                                           float* __restrict__ part_dyx, int world, int rank) {
  __shared__ v4f red[kBnSlots][16];
  const int cq = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cq;
  const int r0 = blockIdx.y * kBnRows, r1 = min(B, r0 + kBnRows);
  const bool ok = col < H;
  constexpr int NR = kBnRows / kBnSlots;
  v4f vd[NR], vd2[NR], vx[NR];
  const v4f g = ok ? ld4(gamma + col) : zero4(), bt = ok ? ld4(beta + col) : zero4();
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int b = r0 + slot + kBnSlots * i;
    const bool in = ok && b < r1;
    const long long idx = (long long)b * H + col;
    vd[i] = in ? ld4(dh + idx) : zero4();
    vd2[i] = (in && dh2) ? ld4(dh2 + idx) : zero4();
    vx[i] = in ? ld4(xhat + idx) : zero4();
  }
  v4f s1 = zero4(), s2 = zero4();
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (ok && r0 + slot + kBnSlots * i < r1) {
      const v4f dy = bn_dy(dh2 ? vd[i] + vd2[i] : vd[i], vx[i], g, bt);
      s1 += dy;
      s2 += dy * vx[i];
    }
  }
  const v4f sum_dy = slot_sum(s1, red, cq, slot);
  const v4f sum_dyx = slot_sum(s2, red, cq, slot);
  const int nrb_ = gridDim.y;
  if (ok && slot == 0) {
    *(v4f*)(part_dy + (long long)(rank * nrb_ + blockIdx.y) * H + col) = sum_dy;
    *(v4f*)(part_dyx + (long long)(rank * nrb_ + blockIdx.y) * H + col) = sum_dyx;
  }
  if (ok && world > 1 && slot > 0 && slot <= world && slot - 1 != rank) {   // BnSync: the other ranks' slots := 0
    *(v4f*)(part_dy + (long long)((slot - 1) * nrb_ + blockIdx.y) * H + col) = zero4();
    *(v4f*)(part_dyx + (long long)((slot - 1) * nrb_ + blockIdx.y) * H + col) = zero4();
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float* __restrict__ dh, const float* __restrict__ dh2,
                                                                const float* __restrict__ xhat, const float* invstd,
                                                                const float* gamma, const float* beta,
                                                                const float* __restrict__ part_dy,
                                                                const float* __restrict__ part_dyx, int B, int H,
                                                                float* __restrict__ dz, float* dgamma, float* dbeta,
                                                                float* sumsq_out, int slabs, int world, int rank) {
  __shared__ v4f red[kBnSlots][16];
  const int cq = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cq;
  const bool ok = col < H;
  const int nrb_l = (B + kBnRows - 1) / kBnRows, nrb = nrb_l * world;   // (BnSync: every rank's partials, rank-major)
  PartRegs prg;
  part_request(prg, part_dy, part_dyx, nrb, H, col, ok, slot);
  const v4f g = ok ? ld4(gamma + col) : zero4(), bt = ok ? ld4(beta + col) : zero4(), is = ok ? ld4(invstd + col) : zero4();
  v4f vd[kBnSlabMax], vd2[kBnSlabMax], vx[kBnSlabMax];
#pragma unroll
  for (int sl = 0; sl < kBnSlabMax; ++sl) {
    const int b = (blockIdx.y * slabs + sl) * kBnSlots + slot;
    const bool in = ok && sl < slabs && b < B;
    const long long idx = (long long)b * H + col;
    vd[sl] = in ? ld4(dh + idx) : zero4();
    vd2[sl] = (in && dh2) ? ld4(dh2 + idx) : zero4();
    vx[sl] = in ? ld4(xhat + idx) : zero4();
  }
  // this slot's partials (s, s + 16, ...), then the block's slot sums (the forward's merge, above)
  v4f sum_dy = zero4(), sum_dyx = zero4();
  v4f own_dy = zero4(), own_dyx = zero4();      // this rank's share: what dgamma | dbeta hold (the gradient exchange sums the ranks')
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rb = slot + kBnSlots * u;
    sum_dy += prg.a[u]; sum_dyx += prg.b[u];     // (zeros past nrb)
    if (rb < nrb && rb / nrb_l == rank) { own_dy += prg.a[u]; own_dyx += prg.b[u]; }
  }
  for (int rb = slot + 2 * kBnSlots; rb < nrb; rb += kBnSlots) {
    const v4f a1 = ok ? ld4(part_dy + (long long)rb * H + col) : zero4(), a2 = ok ? ld4(part_dyx + (long long)rb * H + col) : zero4();
    sum_dy += a1; sum_dyx += a2;
    if (rb / nrb_l == rank) { own_dy += a1; own_dyx += a2; }
  }
  sum_dy = slot_sum(sum_dy, red, cq, slot);
  sum_dyx = slot_sum(sum_dyx, red, cq, slot);
  if (world > 1) { own_dy = slot_sum(own_dy, red, cq, slot); own_dyx = slot_sum(own_dyx, red, cq, slot); }
  const v4f k = g * is;
  const float Bt = (float)B * (float)world;
  const v4f m1 = sum_dy / Bt, m2 = sum_dyx / Bt;
  if (world > 1) { sum_dy = own_dy; sum_dyx = own_dyx; }
#pragma unroll
  for (int sl = 0; sl < kBnSlabMax; ++sl) {
    const int b = (blockIdx.y * slabs + sl) * kBnSlots + slot;
    if (ok && sl < slabs && b < B) {
      const v4f dy = bn_dy(dh2 ? vd[sl] + vd2[sl] : vd[sl], vx[sl], g, bt);
      *(v4f*)(dz + (long long)b * H + col) = (dy - m1 - vx[sl] * m2) * k;
    }
  }
  if (ok && blockIdx.y == 0 && slot == 0) { *(v4f*)(dgamma + col) = sum_dyx; *(v4f*)(dbeta + col) = sum_dy; }
  if (sumsq_out && blockIdx.y == 0) {   // sum of squares of this column block's dgamma | dbeta (global-norm clip, fixed slot)
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) q += sum_dyx[k] * sum_dyx[k] + sum_dy[k] * sum_dy[k];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);   // over the 16 column quads
    if (threadIdx.x == 0) sumsq_out[blockIdx.x] = q;
  }
}

__device__ inline void tanh_gauss_fwd_row(const TanhGaussArgs& a, int b);
__device__ inline void tanh_gauss_fwd_body(const TanhGaussArgs& a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  tanh_gauss_fwd_row(a, b);
}
__device__ inline void tanh_gauss_fwd_row(const TanhGaussArgs& a, int b) {
  const StepCtrl c = *a.cur;
  float* act = a.act + (long long)c.batch_slot * a.act_slot_stride + (long long)b * a.ld_act;
  float lp = 0.f;
  for (int j = 0; j < a.A; ++j) {
    const float mu = a.mu[(long long)b * a.ld_head + j];
    if (a.deterministic) { act[j] = (float)tanh((double)mu); continue; }
    const long long i = (long long)b * a.A + j;
    const TgElem r = tanh_gauss_elem(a, c, mu, a.ls_raw[(long long)b * a.ld_head + j], i);
    act[j] = r.t;
    lp = __fadd_rn(lp, r.term);
    if (a.save_eps) { a.save_eps[i] = r.e; a.save_std[i] = r.sd; }
  }
  if (!a.deterministic && a.logp) a.logp[b] = lp;
}

__global__ void tanh_gauss_fwd_kernel(TanhGaussArgs a) {
  if (a.run.layers && blockIdx.x == gridDim.x - 1) { bn_running_update(a.run); return; }   // (the launcher's extra workgroup)
  tanh_gauss_fwd_body(a);
}
__global__ void tanh_gauss_fwd2_kernel(TanhGaussArgs a0, TanhGaussArgs a1) {
  if (a0.run.layers && blockIdx.x == gridDim.x - 1) { if (blockIdx.y == 0) bn_running_update(a0.run); return; }
  if (blockIdx.y == 0) tanh_gauss_fwd_body(a0);
  else tanh_gauss_fwd_body(a1);
}

// The two heads and the sampling of one or two inputs as one launch (sac_heads.h): workgroup (x, y) owns rows [16 x, 16 x + 16) of
// input y.  (The last workgroup of row y = 0 is the running-statistics rider when tg[0].run.layers.)
__global__ __launch_bounds__(256) void heads_sample_kernel(HeadsSampleArgs h) {
  __shared__ float s_mu[16][17], s_ls[16][17], s_term[16][17];
  const int y = (int)blockIdx.y;
  const TanhGaussArgs& a = h.tg[y];
  if (h.tg[0].run.layers && blockIdx.x == gridDim.x - 1) { if (y == 0) bn_running_update(h.tg[0].run); return; }
  const int t = (int)blockIdx.x;
  if (t * 16 >= a.B) return;   // (uniform per workgroup)
  gemm_pin(h.mean[y]);
  const StepCtrl c = *a.cur;   // (requested here: its round trip — the record was rewritten by the previous step's last launch — hides behind the two tiles)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  float x_m, x_l, ss;
  gemm_batch_tile<1, 1, 4>(h.mean[y], t, x_m, ss);
  __syncthreads();   // (the tile body's partial-sum exchange buffer is about to be written again)
  gemm_batch_tile<1, 1, 4>(h.lstd[y], t, x_l, ss);
  s_mu[4 * lg + wave][li] = x_m;
  s_ls[4 * lg + wave][li] = x_l;
  __syncthreads();
  const int A = a.A;
  if ((int)threadIdx.x < 16 * A) {
    const int r = (int)threadIdx.x / A, j = (int)threadIdx.x - r * A;
    const int b = t * 16 + r;
    if (b < a.B) {
      const long long i = (long long)b * A + j;
      const TgElem e = tanh_gauss_elem(a, c, s_mu[r][j], s_ls[r][j], i);
      a.act[(long long)c.batch_slot * a.act_slot_stride + (long long)b * a.ld_act + j] = e.t;
      s_term[r][j] = e.term;
      if (a.save_eps) { a.save_eps[i] = e.e; a.save_std[i] = e.sd; }
    }
  }
  __syncthreads();
  if (threadIdx.x < 16 && a.logp) {
    const int b = t * 16 + (int)threadIdx.x;
    if (b < a.B) {
      float lp = 0.f;
      for (int j = 0; j < A; ++j) lp = __fadd_rn(lp, s_term[threadIdx.x][j]);   // in action order, as the row loop adds
      a.logp[b] = lp;
    }
  }
}

__global__ __launch_bounds__(1024) void actor_select_kernel(ActorSelArgs a) {
  __shared__ float scratch[16];
  actor_select_body(a, scratch);
}

__device__ inline void tanh_gauss_bwd_body(const TanhGaussBwdArgs& a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.A) return;
  const int b = i / a.A, j = i - b * a.A;
  const long long slot_off = a.act_slot_stride ? (long long)a.cur->batch_slot * a.act_slot_stride : 0;
  const float alpha = a.alpha_dev ? *a.alpha_dev : a.alpha_const;
  float gmu, gls;
  tanh_gauss_bwd_elem(a, slot_off, alpha, b, j, gmu, gls);
  a.gmu[(long long)b * a.ld_g + j] = gmu;
  a.gls[(long long)b * a.ld_g + j] = gls;
}
__global__ __launch_bounds__(256) void tanh_gauss_bwd_kernel(TanhGaussBwdArgs a) { tanh_gauss_bwd_body(a); }

__global__ __launch_bounds__(1024) void alpha_update_kernel(AlphaArgs a) {
  __shared__ float scratch[16];
  alpha_body(a, scratch);
}

// actor-loss selection and the log-alpha gradient (both single-block passes over logp) in one launch
__global__ __launch_bounds__(1024) void actor_select_alpha_kernel(ActorSelArgs s, AlphaArgs al) {
  __shared__ float scratch[16];
  if (blockIdx.x == 1) {   // the rider: ops.hip mean_metric_kernel's sums
    float m = 0.f;
#pragma unroll 8
    for (int i = threadIdx.x; i < s.mean_n; i += blockDim.x) m += s.mean_x[i];
    m = block_sum(m, scratch);
    if (threadIdx.x == 0) s.metrics[(long long)s.cur->metrics_slot * kMetricFloats + s.mean_index] = m / (float)s.mean_n;
    return;
  }
  actor_select_body(s, scratch);
  __syncthreads();
  alpha_body(al, scratch);
}

// Multi-workgroup form of the same launch (round 4, B >= 1024 with the agent's scratch; ops.hip td_loss_kernel<.., MB> has the
// reasoning: one workgroup is bound by its CU's issue rate).  Workgroups of 256 threads, one row per thread; every wave leaves
// its partial of the two sums (actor loss terms; logp + target entropy) at part[(workgroup * 4 + wave) * 2 + {0, 1}], the last
// workgroup to arrive adds them up in index order and finishes both results.  Workgroup gridDim.x - 1 is the q_value rider when
// there is one (it takes no ticket).
__global__ __launch_bounds__(256) void actor_select_alpha_mb_kernel(ActorSelArgs s, AlphaArgs al) {
  __shared__ float scratch[16];
  const int nb = (s.B + 255) / 256;
  if ((int)blockIdx.x >= nb) {   // the rider
    float m = 0.f;
#pragma unroll 8
    for (int i = threadIdx.x; i < s.mean_n; i += 256) m += s.mean_x[i];
    m = block_sum_256(m, scratch);
    if (threadIdx.x == 0) s.metrics[(long long)s.cur->metrics_slot * kMetricFloats + s.mean_index] = m / (float)s.mean_n;
    return;
  }
  const StepCtrl c = *s.cur;
  const int b = blockIdx.x * 256 + threadIdx.x;
  const float lp = (c.do_alpha && b < al.B) ? al.logp[b] + al.target_entropy : 0.f;
  const float acc = wave_sum(actor_select_acc(s, true));
  const float lps = wave_sum(lp);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    float* mine = s.part + ((long long)blockIdx.x * 4 + wave) * 2;
    __hip_atomic_store(mine, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + 1, lps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  drain_stores();
  __syncthreads();
  unsigned int* s_ticket = reinterpret_cast<unsigned int*>(scratch);
  if (threadIdx.x == 0) *s_ticket = __hip_atomic_fetch_add(s.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*s_ticket != (unsigned)(nb - 1)) return;   // (uniform)
  if (threadIdx.x == 0) __hip_atomic_store(s.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  if (threadIdx.x > 1) return;
  float t = 0.f;
#pragma unroll 8
  for (int i = 0; i < 4 * nb; ++i) t += __hip_atomic_load(s.part + (long long)i * 2 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* met = s.metrics + (long long)c.metrics_slot * kMetricFloats;
  if (threadIdx.x == 0) { met[MET_ACTOR_LOSS] = t / (float)s.B; return; }
  if (!c.do_alpha) { met[MET_ALPHA_LOSS] = 0.f; met[MET_ALPHA] = *al.alpha; return; }   // `gradient_step <= alpha_min_steps: return 0.0`
  const float mean_x = t / (float)al.B;
  met[MET_ALPHA_LOSS] = -(*al.log_alpha * mean_x);
  *al.grad_out = -mean_x;
}

// row-block SAC: the selection kernel's outputs are metrics and the log-alpha gradient only (the row-block launch forms
// its own min-selection), so its single block rides on the tanh-Gaussian backward launch as one extra block
__global__ __launch_bounds__(256) void tanh_gauss_bwd_select_kernel(TanhGaussBwdArgs a, ActorSelArgs s, AlphaArgs al) {
  if (blockIdx.x + 1 < gridDim.x) { tanh_gauss_bwd_body(a); return; }
  __shared__ float scratch[16];
  actor_select_body(s, scratch);
  __syncthreads();
  alpha_body(al, scratch);
}

// lane i holds element i of the row (+inf beyond `width`): 21 compare-exchange rounds with the
// partner lane i^j, direction from bit k of the lane id -> ascending order across the wave.
__global__ __launch_bounds__(256) void sort_trunc_kernel(const float* in, long long rows, int width, int drop,
                                                         float* sorted, float* mean) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v = lane < width ? in[row * width + lane] : INFINITY;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float o = __shfl_xor(v, j, 64);
      const bool up = (lane & k) == 0;          // ascending block?
      const bool lower = (lane & j) == 0;       // this lane is the lower index of the pair
      v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
    }
  }
  if (sorted && lane < width) sorted[row * width + lane] = v;
  const int keep = width - drop;
  float s = wave_sum(lane < keep ? v : 0.f);
  if (lane == 0 && mean) mean[row] = s / (float)keep;
}

// ---- distributional TQC (ops.h QuantileArgs) -----------------------------------------------------------------
__device__ inline float wave_bitonic_sort(float v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float o = __shfl_xor(v, j, 64);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
    }
  }
  return v;
}

// One wavefront per batch row.  (1) lane l < C*Q holds atom l of the pooled target atoms: bitonic sort across the wave,
// the lowest K = C*(Q - drop) kept, y_j = r + gamma*(1-d)*(z_(j) - alpha*logp').  (2) per critic c, lane i < Q owns atom
// z_i with tau_i = (2i+1)/(2Q): over the K targets u = y_j - z_i, rho = |tau_i - 1(u<0)| * huber(u), loss_c = mean over
// (B, Q, K), dz_i = -(1/(B*Q*K)) * sum_j |tau_i - 1(u<0)| * clamp(u, -1, 1).
__global__ __launch_bounds__(256) void quantile_td_kernel(QuantileArgs a) {
  __shared__ float ys[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + w;
  if (b >= a.B) return;
  const StepCtrl c = *a.cur;
  const int C = a.C, Q = a.Q, N = C * Q, K = C * (Q - a.drop);
  const float alpha = *a.alpha_dev;
  float v = INFINITY;
  if (lane < N) { const int cc = lane / Q, q = lane - cc * Q; v = a.zt[((long long)cc * a.B + b) * Q + q]; }
  v = wave_bitonic_sort(v, lane);
  const float rew = a.r[(long long)c.batch_slot * a.slot_stride + b], dn = a.d[(long long)c.batch_slot * a.slot_stride + b];
  const float k1 = __fmul_rn(a.gamma, __fsub_rn(1.0f, dn)), ent = __fmul_rn(alpha, a.logp_next[b]);
  const float y = lane < K ? __fadd_rn(rew, __fmul_rn(k1, __fsub_rn(v, ent))) : 0.f;
  ys[w][lane] = y;
  a.y[(long long)b * 64 + lane] = y;
  float ysum = lane < K ? y : 0.f;
  ysum = wave_sum(ysum);
  ysum = __shfl(ysum, 0, 64);
  const float ymean = ysum / (float)K;
  const float norm = 1.0f / ((float)a.B * (float)Q * (float)K);
  float tdmax = 0.f;
  for (int cc = 0; cc < C; ++cc) {
    float loss = 0.f, g = 0.f, z = 0.f;
    if (lane < Q) {
      z = a.z[((long long)cc * a.B + b) * Q + lane];
      const float tau = (2.0f * (float)lane + 1.0f) / (2.0f * (float)Q);
      for (int j = 0; j < K; ++j) {
        const float u = ys[w][j] - z;
        const float au = fabsf(u);
        const float hub = au <= 1.0f ? 0.5f * u * u : au - 0.5f;
        const float wgt = fabsf(tau - (u < 0.f ? 1.0f : 0.0f));
        loss += wgt * hub;
        g -= wgt * fminf(fmaxf(u, -1.0f), 1.0f);
      }
      a.dz[((long long)cc * a.B + b) * Q + lane] = g * norm;
    }
    loss = wave_sum(loss);
    float zs = wave_sum(lane < Q ? z : 0.f);
    if (lane == 0) {
      a.row_loss[(long long)cc * a.B + b] = loss;
      tdmax = fmaxf(tdmax, fabsf(ymean - zs / (float)Q));
    }
  }
  if (lane == 0) a.row_td[b] = tdmax;
}

// metrics of the step from the per-row sums (one block, fixed order): critic losses, td, and — nothing else: q_value
// is taken from the updated critics later, like the reference's TQC (src/agent.py:1016-1019)
__global__ __launch_bounds__(256) void quantile_metrics_kernel(QuantileArgs a) {
  __shared__ float scratch[4];
  const StepCtrl c = *a.cur;
  float* met = a.metrics + (long long)c.metrics_slot * kMetricFloats;
  const int K = a.C * (a.Q - a.drop);
  for (int cc = 0; cc < a.C; ++cc) {
    float s = 0.f;
    for (int b = threadIdx.x; b < a.B; b += 256) s += a.row_loss[(long long)cc * a.B + b];
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) met[MET_CRITIC_LOSS + cc] = s / ((float)a.B * (float)a.Q * (float)K);
  }
  float t = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) t += a.row_td[b];
  t = block_sum_256(t, scratch);
  if (threadIdx.x == 0) met[MET_TD] = t / (float)a.B;
}

__global__ __launch_bounds__(256) void quantile_actor_kernel(QuantileActorArgs a) {
  __shared__ float scratch[4];
  const StepCtrl c = *a.cur;
  const float alpha = *a.alpha_dev;
  const int N = a.C * a.Q;
  const float gz = -1.0f / ((float)a.B * (float)N);
  float acc = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    float s = 0.f;
    for (int cc = 0; cc < a.C; ++cc)
      for (int q = 0; q < a.Q; ++q) {
        const long long i = ((long long)cc * a.B + b) * a.Q + q;
        s += a.z[i];
        a.dz[i] = gz;
      }
    acc += __fsub_rn(__fmul_rn(alpha, a.logp[b]), s / (float)N);
  }
  acc = block_sum_256(acc, scratch);
  if (threadIdx.x == 0) a.metrics[(long long)c.metrics_slot * kMetricFloats + MET_ACTOR_LOSS] = acc / (float)a.B;
}

}  // namespace

int launch_quantile_td(hipStream_t st, const QuantileArgs& a) {
  GCRL_CHECK_ARG(a.C >= 1 && a.Q >= 1 && a.C * a.Q <= 64 && a.drop >= 0 && a.drop < a.Q, "quantile_td: need C*Q <= 64 and 0 <= drop < Q (C=%d, Q=%d, drop=%d)", a.C, a.Q, a.drop);
  hipLaunchKernelGGL(quantile_td_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  hipLaunchKernelGGL(quantile_metrics_kernel, dim3(1), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_quantile_actor(hipStream_t st, const QuantileActorArgs& a) {
  hipLaunchKernelGGL(quantile_actor_kernel, dim3(1), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

static inline unsigned reduce_threads(long long n) { return (unsigned)std::min<long long>(1024, std::max<long long>(256, (n + 63) / 64 * 64)); }
static inline int bn_slabs(int B) { static const char* e = getenv("GCRL_BN_SLABS"); if (e) return atoi(e); return std::min(kBnSlabMax, std::max(1, B / 1024)); }
static inline bool bn_aligned(const void* p) { return ((unsigned long long)p & 15ull) == 0; }

int launch_bn_relu_fwd(hipStream_t st, const float* z, int B, int H, const float* gamma,
                       const float* beta, float* h, float* xhat, float* invstd, float* rmean,
                       float* rvar, float* scratch) {
  const BnFwdProb one{z, h, xhat, invstd, scratch};
  return launch_bn_relu_fwd_multi(st, &one, 1, B, H, gamma, beta, rmean, rvar);
}

int launch_bn_relu_fwd_multi(hipStream_t st, const BnFwdProb* probs, int nprob, int B, int H, const float* gamma,
                             const float* beta, float* rmean, float* rvar, int rows_per_part, const BnSync* sync, bool stats_done) {
  GCRL_CHECK_ARG(nprob == 1 || nprob == 2, "bn_relu_fwd_multi: 1 or 2 problems");
  const int world = (sync && sync->world > 1) ? sync->world : 1;
  GCRL_CHECK_ARG(world == 1 || (rows_per_part == kBnRows && sync->exchange && world < kBnSlots && sync->rank >= 0 && sync->rank < world),
                 "bn_relu_fwd_multi: SyncBN needs the bn_stats form, an exchange function and world <= %d", kBnSlots - 1);
  GCRL_CHECK_ARG(rows_per_part == kBnRows || (rows_per_part == kBnFusedRows && (B + kBnFusedRows - 1) / kBnFusedRows <= kBnMaxPart),
                 "bn_relu_fwd_multi: %d rows per partial at B=%d", rows_per_part, B);
  GCRL_CHECK_ARG(H % 4 == 0 && bn_aligned(gamma) && bn_aligned(beta) && bn_aligned(rmean) && bn_aligned(rvar) &&
                     bn_aligned(probs[0].z) && bn_aligned(probs[0].h) && bn_aligned(probs[0].scratch),
                 "bn_relu_fwd: H must be a multiple of 4 and every operand 16-byte aligned (H=%d)", H);
  const int nrb = (B + kBnRows - 1) / kBnRows;
  BnFwdPair pr;
  pr.n = nprob;
  pr.p[0] = probs[0];
  pr.p[1] = nprob > 1 ? probs[1] : probs[0];
  pr.rpp = rows_per_part;
  pr.world = world; pr.rank = world > 1 ? sync->rank : 0;
  GCRL_CHECK_ARG(!stats_done || (rows_per_part == kBnRows && world == 1), "bn_relu_fwd_multi: partials from the tiled GEMM are 64-row ones of one rank");
  if (rows_per_part == kBnRows && !stats_done) {
    hipLaunchKernelGGL(bn_stats_kernel, dim3((H + 63) / 64, nrb, nprob), dim3(256), 0, st, pr, B, H);
    GCRL_HIP(hipGetLastError());
  }
  if (world > 1) {   // every rank's partials to every rank: one exchange when the problems' arrays are adjacent, else one each
    const long long n1 = 2LL * world * nrb * H;
    int rc = GCRL_OK;
    if (nprob == 2 && probs[1].scratch == probs[0].scratch + n1) rc = sync->exchange(probs[0].scratch, 2 * n1, st, sync->user);
    else if (nprob == 2 && probs[0].scratch == probs[1].scratch + n1) rc = sync->exchange(probs[1].scratch, 2 * n1, st, sync->user);
    else for (int i = 0; i < nprob && !rc; ++i) rc = sync->exchange(probs[i].scratch, n1, st, sync->user);
    if (rc) return rc;
    if (sync->result)   // (the apply launch only READS the partials)
      for (int i = 0; i < 2; ++i) pr.p[i].scratch = const_cast<float*>(sync->result(pr.p[i].scratch, sync->user));
  }
  pr.slabs = bn_slabs(B);
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((H + 63) / 64, (B + kBnSlots * pr.slabs - 1) / (kBnSlots * pr.slabs), nprob), dim3(256), 0,
                     st, pr, B, H, gamma, beta, rmean, rvar);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_bn_relu_eval(hipStream_t st, const float* z, int B, int H, const float* gamma,
                        const float* beta, const float* rmean, const float* rvar, float* h) {
  const long long n = (long long)B * H;
  hipLaunchKernelGGL(bn_relu_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, B, H,
                     gamma, beta, rmean, rvar, h);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_bn_relu_bwd(hipStream_t st, const float* dh, const float* dh2, const float* xhat, const float* invstd,
                       const float* gamma, const float* beta, int B, int H, float* dz, float* dgamma, float* dbeta,
                       float* scratch, float* sumsq_out, const BnSync* sync) {
  const int world = (sync && sync->world > 1) ? sync->world : 1, rank = world > 1 ? sync->rank : 0;
  GCRL_CHECK_ARG(world == 1 || (sync->exchange && world < kBnSlots && rank >= 0 && rank < world), "bn_relu_bwd: bad SyncBN arguments");
  GCRL_CHECK_ARG(H % 4 == 0 && bn_aligned(dh) && bn_aligned(dh2) && bn_aligned(xhat) && bn_aligned(invstd) && bn_aligned(gamma) &&
                     bn_aligned(beta) && bn_aligned(dz) && bn_aligned(dgamma) && bn_aligned(dbeta) && bn_aligned(scratch),
                 "bn_relu_bwd: H must be a multiple of 4 and every operand 16-byte aligned (H=%d)", H);
  const int nrb = (B + kBnRows - 1) / kBnRows;
  float* part_dy = scratch;
  float* part_dyx = scratch + (long long)world * nrb * H;
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((H + 63) / 64, nrb), dim3(256), 0, st, dh, dh2, xhat, gamma, beta, B, H, part_dy,
                     part_dyx, world, rank);
  GCRL_HIP(hipGetLastError());
  if (world > 1) {
    if (int rc = sync->exchange(scratch, 2LL * world * nrb * H, st, sync->user)) return rc;
    if (sync->result) {
      part_dy = const_cast<float*>(sync->result(scratch, sync->user));
      part_dyx = part_dy + (long long)world * nrb * H;
    }
  }
  const int slabs = bn_slabs(B);
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((H + 63) / 64, (B + kBnSlots * slabs - 1) / (kBnSlots * slabs)), dim3(256), 0, st, dh,
                     dh2, xhat, invstd, gamma, beta, part_dy, part_dyx, B, H, dz, dgamma, dbeta, sumsq_out, slabs, world, rank);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_heads_sample(hipStream_t st, HeadsSampleArgs& h) {
  GCRL_CHECK_ARG(h.n == 1 || h.n == 2, "heads_sample: %d inputs", h.n);
  int blocks = 0;
  for (int i = 0; i < h.n; ++i) {
    const TanhGaussArgs& a = h.tg[i];
    GCRL_CHECK_ARG(!a.deterministic && a.A >= 1 && a.A <= 16 && h.mean[i].M == a.B && h.lstd[i].M == a.B && h.mean[i].N == a.A && h.lstd[i].N == a.A &&
                       a.mu == h.mean[i].C && a.ls_raw == h.lstd[i].C && (long long)a.ld_head == h.mean[i].c_rs && (long long)a.ld_head == h.lstd[i].c_rs &&
                       !h.mean[i].slot == !h.mean[i].a_slot && h.mean[i].c_slot == 0 && h.lstd[i].c_slot == 0,
                   "heads_sample: input %d: the sampling must read what the two head problems write", i);
    GCRL_CHECK_ARG(gemm_shape_of(h.mean[i]) == 1 && gemm_shape_of(h.lstd[i]) == 1, "heads_sample: input %d: not the k-split 16x16 form", i);
    blocks = std::max(blocks, gemm_prepare_ksplit(h.mean[i]));
    (void)gemm_prepare_ksplit(h.lstd[i]);
  }
  hipLaunchKernelGGL(heads_sample_kernel, dim3((unsigned)blocks + (h.tg[0].run.layers ? 1 : 0), (unsigned)h.n), dim3(256), 0, st, h);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_tanh_gauss_fwd(hipStream_t st, const TanhGaussArgs& a) {
  hipLaunchKernelGGL(tanh_gauss_fwd_kernel, dim3((a.B + 255) / 256 + (a.run.layers ? 1 : 0)), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_tanh_gauss_fwd2(hipStream_t st, const TanhGaussArgs& a0, const TanhGaussArgs& a1) {
  hipLaunchKernelGGL(tanh_gauss_fwd2_kernel, dim3((std::max(a0.B, a1.B) + 255) / 256 + (a0.run.layers ? 1 : 0), 2), dim3(256), 0, st, a0, a1);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_actor_select(hipStream_t st, const ActorSelArgs& a) {
  GCRL_CHECK_ARG(a.C >= 1 && a.C <= kMaxCritics && a.drop >= 0 && a.drop < a.C, "actor_select: bad C=%d drop=%d", a.C, a.drop);
  hipLaunchKernelGGL(actor_select_kernel, dim3(1), dim3(reduce_threads(a.B)), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_actor_select_alpha(hipStream_t st, const ActorSelArgs& a, const AlphaArgs& al) {
  GCRL_CHECK_ARG(a.C >= 1 && a.C <= kMaxCritics && a.drop >= 0 && a.drop < a.C, "actor_select: bad C=%d drop=%d", a.C, a.drop);
  if (a.part && a.ticket && a.B >= 1024 && al.phase == 0 && al.B == a.B) {
    hipLaunchKernelGGL(actor_select_alpha_mb_kernel, dim3((a.B + 255) / 256 + (a.mean_x ? 1 : 0)), dim3(256), 0, st, a, al);
    GCRL_HIP(hipGetLastError());
    return GCRL_OK;
  }
  hipLaunchKernelGGL(actor_select_alpha_kernel, dim3(a.mean_x ? 2 : 1), dim3(reduce_threads(a.B)), 0, st, a, al);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_tanh_gauss_bwd(hipStream_t st, const TanhGaussBwdArgs& a) {
  hipLaunchKernelGGL(tanh_gauss_bwd_kernel, dim3((a.B * a.A + 255) / 256), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_tanh_gauss_bwd_select(hipStream_t st, const TanhGaussBwdArgs& a, const ActorSelArgs& s, const AlphaArgs& al) {
  GCRL_CHECK_ARG(s.C >= 1 && s.C <= kMaxCritics && s.drop >= 0 && s.drop < s.C, "actor_select: bad C=%d drop=%d", s.C, s.drop);
  hipLaunchKernelGGL(tanh_gauss_bwd_select_kernel, dim3((a.B * a.A + 255) / 256 + 1), dim3(256), 0, st, a, s, al);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_alpha_update(hipStream_t st, const AlphaArgs& a) {
  hipLaunchKernelGGL(alpha_update_kernel, dim3(1), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_sort_truncate_mean(hipStream_t st, const float* in, long long rows, int width, int drop,
                              float* sorted, float* mean) {
  GCRL_CHECK_ARG(in && rows >= 1 && width >= 1 && width <= 64 && drop >= 0 && drop < width,
                 "sort_truncate_mean: need 1 <= width <= 64 and 0 <= drop < width");
  hipLaunchKernelGGL(sort_trunc_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, in, rows, width,
                     drop, sorted, mean);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
