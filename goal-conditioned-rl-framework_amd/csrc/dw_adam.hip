// dw_adam.hip — the step's two all-row launches as ONE: every dW | db problem of a net (or of two nets: the overlapped DDPG
// step's critic and actor), the global-norm clip, Adam(W), Polyak, the [in][out] weight copies, the riding metrics and the
// control advance.  Protocol and reference lines: ops.h (DwAdamNetArgs).  The arithmetic is shared with the two-launch form
// (gemm_mfma.h gemm_batch_tile, adam_math.h), so the two forms give the same bits.
#include "dw_adam.h"

#include <algorithm>

#include "adam_math.h"
#include "meet.h"

namespace gcrl {
namespace {

constexpr unsigned long long kSlotEmpty = ~0ull;
constexpr int kLeaderMinTiles = 64;   // a negative quiet NaN with every payload bit set: never a sum of squares

// development build (-DGCRL_OF_STAMPS, tools/of_stamps.sh): thread 0 of every workgroup leaves the constant-rate clock (100 MHz) at
// its section boundaries in a.stamps[(net * 2048 + workgroup) * 8 + k]
#ifdef GCRL_OF_STAMPS
#define OF_STAMP(k) of_t[k] = wall_clock64()
#else
#define OF_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256, 5) void dw_adam_kernel(DwAdamArgs a) {
#ifdef GCRL_OF_STAMPS
  unsigned long long of_t[5];
#endif
  OF_STAMP(0);
  __shared__ float s_ss[4];
  __shared__ double dred[4];
  __shared__ float s_coef;
  __shared__ float tile_p[16][17], tile_t[16][17];
  const DwAdamNet& na = a.net[blockIdx.y];
  const DwAdamNetArgs& o = na.o;
  if ((int)blockIdx.x >= o.ntiles) return;   // (a paired launch is sized for the larger net; uniform per workgroup, before any barrier)
  // XCD-aware workgroup -> tile order (gemm_mfma.h xcd_tile_of, here for the 2-D grid: workgroup (x, y) runs on XCD
  // (x + y * gridDim.x) % 8): an XCD takes a contiguous range of the net's tiles — whole tile rows of a layer, i.e. that layer's
  // activations enter ONE L2 instead of eight.  A wrong guess about the placement costs speed, never correctness.
  const unsigned long long t_start = wall_clock64();
  int bid = (int)blockIdx.x;
  {
    const int per = o.ntiles >> 3;
    if (bid < (per << 3)) bid = ((bid + (int)(blockIdx.y * gridDim.x)) & 7) * per + (bid >> 3);
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kFusedMaxLayers; ++q)
    if (q < o.nl && bid >= na.d[q].tile0) pi = q;
  const GemmDesc& d = na.d[pi];
  const DwAdamLayer lay = o.lay[pi];
  const int t = bid - d.tile0;
  const int tn = t % d.tiles_n, tm = t / d.tiles_n;
  const int m0 = tm << 4, n0 = tn << 4;
  const int in = d.N - 1, out = d.M;   // the problem is [out][in | 1]: column `in` is the bias gradient

  // everything this workgroup will need that does not depend on the GEMM is requested first: the step's scalars, the launch
  // count, and this lane's parameter, moments and target
  const StepCtrl c = *o.cur;
  const unsigned int seq = o.seq[0], fault = o.seq[1];   // (fault: the test hook gcrl_agent_debug_meet_fault — workgroup 1's slot never arrives, once)
  const AdamStepScalars sc = adam_scalars(c, o.which);
  const int em = m0 + 4 * lg + wave, en = n0 + li;   // this lane's element of the tile (gemm_batch_tile's k-split layout)
  long long my_i = -1;
  if (em < out) {
    if (en < in) my_i = lay.pw + (long long)em * in + en;
    else if (en == in) my_i = lay.pb + em;
  }
  const bool pk = o.target && o.polyak;
  float pre_p = 0.f, pre_m = 0.f, pre_v = 0.f, pre_t = 0.f;
  if (my_i >= 0) {
    pre_p = o.p[my_i]; pre_m = o.m[my_i]; pre_v = o.v[my_i];
    if (pk) pre_t = o.target[my_i];
  }

  float x, ss;
  OF_STAMP(1);
  gemm_batch_tile<1, 1, 4>(d, t, x, ss);   // (t < d.ntiles by construction; the gradient element is also stored: get("grad:...") reads it)

  // the tile's sum of squares in the order adam_kernel adds up the batched launch's four per-wave partials of a tile
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if (lane == 0) s_ss[wave] = ss;
  __syncthreads();
  unsigned long long* mine = o.slots + (long long)(seq & 1u) * o.slot_stride;
  if (threadIdx.x == 0 && !(fault && bid == 1)) {
    const double dt = ((double)s_ss[0] + (double)s_ss[1]) + ((double)s_ss[2] + (double)s_ss[3]);
    const int slot = lay.slot0 + t;
    __hip_atomic_store(mine + slot, (unsigned long long)__double_as_longlong(dt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    o.slots[(long long)((seq & 1u) ^ 1u) * o.slot_stride + slot] = kSlotEmpty;   // nobody reads the other array in this launch; the kernel boundary publishes it
  }
  OF_STAMP(2);
  // riders of the net's first workgroup, while the other workgroups' slots arrive
  float* met = a.metrics + (long long)c.metrics_slot * kMetricFloats;
  if (bid == 0) {
    if (o.mean_x) rider_mean_metric(o.mean_x, o.mean_n, o.mean_scale, met + o.mean_index);
    if (o.td_q) rider_td_metrics(o.td_q, o.td_y, o.td_n, o.td_C, o.td_loss_kind, met);
  }
  // ||g||: the net's slots summed in ONE order (thread t: slots t, t + 256, ...; then lanes, then waves), in fp64.  Nets of
  // >= kLeaderMinTiles tiles: only the first eight workgroups of the net — one per XCD under round-robin dispatch — sweep the
  // slots; each leaves the sum in a result word (again its own flag), and every other workgroup polls the ONE word of the leader
  // that shares its XCD.  With every workgroup sweeping every slot the early finishers kept ~5 MB of slot loads per round in
  // flight in front of the operand loads of the workgroups still working (measured: the last tile done at 10-15 us instead of
  // 7.5); a leader's sweep is 37 lines.  Smaller nets: every workgroup sweeps (a few KB in all).
  {
    const bool lead_mode = a.leaders && o.ntiles >= kLeaderMinTiles;
    const bool sweeper = !lead_mode || blockIdx.x < 8;
    const int xc = ((int)blockIdx.x + (int)(blockIdx.y * gridDim.x)) & 7;
    unsigned long long* res = mine + (o.slot_stride - 8);
    bool ok = true;
    double s = 0.0;
    if (sweeper) {
      // a lane re-loads only the slots it has not seen yet
      unsigned long long w[kFusedMaxSlotsPerThread];
      int spins = 0;
      for (int i = 0; i < a.poll_first_sleep; ++i) __builtin_amdgcn_s_sleep(1);
      for (int i = 0; i < 4096 && (long long)(wall_clock64() - t_start) < (long long)a.poll_gate; ++i) __builtin_amdgcn_s_sleep(2);
#pragma unroll
      for (int u = 0; u < kFusedMaxSlotsPerThread; ++u) {
        const int i = (int)threadIdx.x + 256 * u;
        w[u] = i < o.ntiles ? __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
      for (;;) {
        bool all = true;
#pragma unroll
        for (int u = 0; u < kFusedMaxSlotsPerThread; ++u) all = all && w[u] != kSlotEmpty;
        if (all) break;
        if (++spins >= kMeetSpinMax) { ok = false; break; }
        for (int i = 0; i < a.poll_sleep; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int u = 0; u < kFusedMaxSlotsPerThread; ++u)
          if (w[u] == kSlotEmpty) w[u] = __hip_atomic_load(mine + (int)threadIdx.x + 256 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < kFusedMaxSlotsPerThread; ++u)
        if ((int)threadIdx.x + 256 * u < o.ntiles) s += __longlong_as_double((long long)w[u]);
      if (!ok) s = __longlong_as_double(0x7ff8000000000000ll);   // a slot never arrived: the step is poisoned (and reported below)
      s = wave_sum_d(s);
      if (lane == 0) dred[wave] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        s = dred[0] + dred[1] + dred[2] + dred[3];
        if (lead_mode) {
          __hip_atomic_store(res + xc, (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
          o.slots[(long long)((seq & 1u) ^ 1u) * o.slot_stride + (o.slot_stride - 8) + xc] = kSlotEmpty;
        }
      }
    } else if (threadIdx.x == 0) {
      unsigned long long w = __hip_atomic_load(res + xc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spins = 0; w == kSlotEmpty; ) {
        if (++spins >= kMeetSpinMax) { ok = false; w = 0x7ff8000000000000ull; break; }
        for (int i = 0; i < a.poll_sleep; ++i) __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(res + xc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s = __longlong_as_double((long long)w);
    }
    if (!ok && a.status) __hip_atomic_fetch_or(a.status, (unsigned int)MEET_ERR_DW_ADAM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the host learns it (meet.h)
    if (threadIdx.x == 0) {
      float post;
      s_coef = clip_coef(s, c.grad_scale, o.clip, &post);
      if (bid == 0 && a.metrics) met[o.metric_index] = post;
    }
  }
  __syncthreads();
  OF_STAMP(3);
  const float gmul = c.grad_scale * s_coef;
  float p_new = 0.f, t_new = 0.f;
  if (my_i >= 0) {
    const AdamElem e = adam_elem(x, pre_p, pre_m, pre_v, gmul, sc, a.beta2, a.w1, a.w2, a.eps);
    o.p[my_i] = e.p; o.m[my_i] = e.m; o.v[my_i] = e.v;
    p_new = e.p;
    if (pk) { t_new = polyak_elem(a.tau, e.p, a.one_m_tau, pre_t); o.target[my_i] = t_new; }
  }
  if (lay.wt_dst >= 0) {   // (uniform per workgroup) the [in][out] copy of a hidden layer's weight: 16 consecutive outputs per run
    tile_p[4 * lg + wave][li] = p_new;
    tile_t[4 * lg + wave][li] = t_new;
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int k = n0 + ty, oo = m0 + tx;
    if (k < in && oo < out) {
      const long long at = lay.wt_dst + (long long)k * out + oo;
      o.wt[at] = tile_p[tx][ty];
      if (pk && o.wt_target) o.wt_target[at] = tile_t[tx][ty];
    }
  }
  OF_STAMP(4);
#ifdef GCRL_OF_STAMPS
  if (a.stamps && threadIdx.x == 0)
    for (int k = 0; k < 5; ++k) a.stamps[((long long)blockIdx.y * 2048 + blockIdx.x) * 8 + k] = of_t[k];
#endif
  if (bid == 0 && threadIdx.x == 0) {
    o.seq[1] = 0u;
    o.seq[0] = seq + 1u;   // every workgroup of this launch read it before it published, and this workgroup has seen every slot
    if (blockIdx.y == 0 && a.advance) ctrl_advance(a.advance);
  }
}

}  // namespace

long long dw_adam_capacity() { return meet_capacity((const void*)dw_adam_kernel, 256, 0); }

int launch_dw_adam(hipStream_t st, DwAdamArgs& a) {
  GCRL_CHECK_ARG(a.nnets >= 1 && a.nnets <= 2, "dw_adam: %d nets (1 or 2)", a.nnets);
  int widest = 0;
  for (int i = 0; i < a.nnets; ++i) {
    DwAdamNet& n = a.net[i];
    GCRL_CHECK_ARG(n.o.nl >= 1 && n.o.nl <= kFusedMaxLayers && n.o.slots && n.o.seq && n.o.cur, "dw_adam: net %d: %d problems / missing slots", i, n.o.nl);
    int tiles = 0;
    for (int l = 0; l < n.o.nl; ++l) {
      GemmDesc& d = n.d[l];
      GCRL_CHECK_ARG(d.M >= 1 && d.N >= 2 && d.K >= 1 && d.A && d.B && d.C && d.ones_col && d.col_out && !d.bias && d.epi == EPI_NONE && d.mul == MUL_NONE &&
                         !d.bn_part && d.ksplit <= 1 && d.shape_hint == 0 && gemm_shape_of(d) == 1,
                     "dw_adam: net %d problem %d is not a dW | db problem of the k-split 16x16 form", i, l);
      d.sumsq_out = nullptr;
      d.a_vec = (d.a_cs == 1 && d.a_rs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
      d.b_vec = (d.b_rs == 1 && d.b_cs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
      d.a_rvec = (d.a_rs == 1 && d.a_cs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
      d.b_rvec = (d.b_cs == 1 && d.b_rs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
      d.tiles_n = (d.N + 15) / 16;
      d.ntiles = ((d.M + 15) / 16) * d.tiles_n;
      d.tile0 = tiles;
      tiles += d.ntiles;
    }
    n.o.ntiles = tiles;
    GCRL_CHECK_ARG(tiles <= 256 * kFusedMaxSlotsPerThread && tiles + 8 <= n.o.slot_stride, "dw_adam: net %d has %d tiles (slots: %d)", i, tiles, n.o.slot_stride);
    widest = std::max(widest, tiles);
  }
  // (the residency of widest * nnets workgroups is the caller's admission check — dw_adam_capacity — made once per agent)
  hipLaunchKernelGGL(dw_adam_kernel, dim3((unsigned)widest, (unsigned)a.nnets), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
