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
constexpr int kLeaderMinTiles = 64;
static_assert(kFusedMaxLayers == 5, "DwNetHead::tile0 holds the first tiles of problems 1..4");   // a negative quiet NaN with every payload bit set: never a sum of squares

// development build (-DGCRL_OF_STAMPS, tools/of_stamps.sh): thread 0 of every workgroup leaves the constant-rate clock (100 MHz) at
// its section boundaries in a.stamps[(net * 2048 + workgroup) * 8 + k]
#ifdef GCRL_OF_STAMPS
#define OF_STAMP(k) of_t[k] = wall_clock64()
#else
#define OF_STAMP(k) do { } while (0)
#endif

// a uniform, read-only-in-this-launch record through the scalar cache (constant address space): one s_load for the whole record
template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  typedef const unsigned int __attribute__((address_space(4))) cu32;
  cu32* q = reinterpret_cast<cu32*>(reinterpret_cast<uintptr_t>(p));
  union { T v; unsigned int w[sizeof(T) / 4]; } u;
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) u.w[i] = q[i];
  return u.v;
}

// a pointer that went through pin() is an integer as far as the compiler knows, and what it then dereferences is a FLAT access
// (slower, and counted on the LDS wait counter too): the accesses below go through pointers TYPED as global memory
typedef float __attribute__((address_space(1))) gfloat;
typedef unsigned long long __attribute__((address_space(1))) gu64;
typedef unsigned int __attribute__((address_space(1))) gu32;

// a POD record copied HERE, every dword of it in a scalar register: the loads of its fields cannot sink to their uses (left to
// itself the compiler fetched every field where it was first needed: a dozen dependent scalar round trips in front of the GEMM)
template <class T>
__device__ __forceinline__ void pin(T& v) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  unsigned int* w = reinterpret_cast<unsigned int*>(&v);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) asm volatile("" : "+s"(w[i]));
}

// What the kernel is handed: the host-side DwAdamArgs (whole GemmDesc records, 232 bytes each, read field by field where they are
// used) made the kernel start with a dozen DEPENDENT scalar loads — 2.2 us before its first operand request, another microsecond
// inside the tile body.  Here a net is a 48-dword header and 24-dword problems: the header arrives with the first round trip, the
// workgroup's problem, the step's scalars and the launch count with the second.
struct DwProb {
  const float* G; const float* X; float* dW; float* db;   // dW[out][in] = G^T X (G: [K][ldg], X: [K][ldx]), db = column sums of G
  long long pw, pb, wt_dst;                               // DwAdamLayer
  int ldg, ldx, out, in, K, tiles_n, slot0, x_slot;      // x_slot != 0: X lives in the batch array, + batch_slot * x_slot floats
};
struct DwNetHead {   // 16 dwords: the first round trip
  int ntiles, nl, which, slot_stride;
  int tile0[4];      // first tile of problems 1..4 (INT_MAX beyond the net's problems; problem 0 starts at 0)
  const StepCtrl* cur; unsigned int* seq; unsigned long long* slots;
  int polyak, grid_x;   // grid_x: the launch's gridDim.x (from the dispatch packet it would be one more dependent scalar load)
};
struct DwNetPtrs {   // 14 dwords
  float *p, *m, *v, *target, *wt, *wt_target;
  float clip; int pad;
};
struct DwNetK {
  DwNetHead h;
  DwNetPtrs ptr;
  int metric_index, mean_n;
  const float* mean_x; float mean_scale; int mean_index;
  const float* td_q; const float* td_y; int td_n, td_C, td_loss_kind, pad;
  DwProb prob[kFusedMaxLayers];
};
struct DwAdamK {
  float beta2, w1, w2, eps, tau, one_m_tau;
  int leaders, poll_gate, poll_first_sleep, poll_sleep;
  float* metrics; CtrlBlock* advance; unsigned int* status;
#ifdef GCRL_OF_STAMPS
  unsigned long long* stamps;
#endif
  DwNetK net[2];
};

__global__ __launch_bounds__(256, 5) void dw_adam_kernel(DwAdamK a) {
#ifdef GCRL_OF_STAMPS
  unsigned long long of_t[5];
#endif
  OF_STAMP(0);
  __shared__ float s_ss[4];
  __shared__ double dred[4];
  __shared__ float s_coef;
  __shared__ float tile_p[16][17], tile_t[16][17];
  const DwNetK& on = a.net[blockIdx.y];
  DwNetHead o = on.h;          // first round trip: 16 dwords
  pin(o);
  if ((int)blockIdx.x >= o.ntiles) return;   // (a paired launch is sized for the larger net; uniform per workgroup, before any barrier)
  // XCD-aware workgroup -> tile order (gemm_mfma.h xcd_tile_of, here for the 2-D grid: workgroup (x, y) runs on XCD
  // (x + y * gridDim.x) % 8): an XCD takes a contiguous range of the net's tiles — whole tile rows of a layer, i.e. that layer's
  // activations enter ONE L2 instead of eight.  A wrong guess about the placement costs speed, never correctness.
  const unsigned long long t_start = wall_clock64();
  int bid = (int)blockIdx.x;
  {
    const int per = o.ntiles >> 3;
    if (bid < (per << 3)) bid = ((bid + (int)blockIdx.y * o.grid_x) & 7) * per + (bid >> 3);
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int pi = (bid >= o.tile0[0]) + (bid >= o.tile0[1]) + (bid >= o.tile0[2]) + (bid >= o.tile0[3]);
  const int t0 = pi == 0 ? 0 : o.tile0[0] * (pi == 1) + o.tile0[1] * (pi == 2) + o.tile0[2] * (pi == 3) + o.tile0[3] * (pi == 4);
  // second round trip: the problem, the net's arrays, the step's scalars (the control block's copy, written by the launch in front)
  // and the launch count — the last two through the scalar cache (constant address space): uniform, and nothing in this launch
  // writes them before they are read
  DwProb pr = on.prob[pi];
  DwNetPtrs ptr = on.ptr;
  StepCtrl c = load_uniform(o.cur);
  unsigned long long sq = load_uniform(reinterpret_cast<const unsigned long long*>(o.seq));
  pin(pr); pin(ptr); pin(c); pin(sq);   // (all four requested, then waited for together)
  gfloat* const gp = (gfloat*)ptr.p; gfloat* const gm = (gfloat*)ptr.m; gfloat* const gv = (gfloat*)ptr.v; gfloat* const gt = (gfloat*)ptr.target;
  gfloat* const gwt = (gfloat*)ptr.wt; gfloat* const gwtt = (gfloat*)ptr.wt_target;
  gu64* const gslots = (gu64*)o.slots;
  const unsigned int seq = (unsigned int)sq, fault = (unsigned int)(sq >> 32);   // (fault: the test hook gcrl_agent_debug_meet_fault — workgroup 1's slot never arrives, once)
  const int t = bid - t0;
  const int tn = t % pr.tiles_n, tm = t / pr.tiles_n;
  const int m0 = tm << 4, n0 = tn << 4;
  const int in = pr.in, out = pr.out;   // the problem is [out][in | 1]: column `in` is the bias gradient
  struct { long long pw, pb, wt_dst; int slot0; } lay = {pr.pw, pr.pb, pr.wt_dst, pr.slot0};

  const AdamStepScalars sc = adam_scalars(c, o.which);
  const int em = m0 + 4 * lg + wave, en = n0 + li;   // this lane's element of the tile (gemm_batch_tile's k-split layout)
  long long my_i = -1;
  if (em < out) {
    if (en < in) my_i = lay.pw + (long long)em * in + en;
    else if (en == in) my_i = lay.pb + em;
  }
  const bool pk = ptr.target && o.polyak;

  // the problem as the batched launch's tile body wants it (agent.hip bwd_dw): everything else of the record is a literal here
  GemmDesc d;
  d.A = pr.G; d.a_rs = 1; d.a_cs = pr.ldg;
  d.B = pr.X + (pr.x_slot ? (long long)c.batch_slot * pr.x_slot : 0); d.b_rs = pr.ldx; d.b_cs = 1;
  // (ones_col through an opaque register: as a literal, the compiler turned the tile body's `ones column ? 1 : loaded value` selects
  // into a branch around the B loads of every chunk and drained the loads in flight — s_waitcnt vmcnt(0) — at each of them)
  int one = 1;
  asm volatile("" : "+s"(one));
  d.C = pr.dW; d.c_rs = pr.in; d.col_out = pr.db; d.ones_col = one;
  d.M = pr.out; d.N = pr.in + 1; d.K = pr.K;
  d.bias = nullptr; d.H = nullptr; d.h_rs = 0; d.epi = EPI_NONE; d.mul = MUL_NONE;
  d.slot = nullptr; d.a_slot = d.b_slot = d.c_slot = d.h_slot = 0;
  d.sumsq_out = nullptr; d.bn_part = nullptr;
  d.a_vec = d.b_vec = d.a_rvec = d.b_rvec = 0;   // (operands are batch-major: k runs along rows)
  d.tile0 = 0; d.tiles_n = pr.tiles_n; d.ntiles = 0x7fffffff;
  d.shape_hint = 0; d.ksplit = 0; d.kpart = nullptr; d.kticket = nullptr;
  float x, ss;
  OF_STAMP(1);
  gemm_batch_tile<1, 1, 4>(d, t, x, ss);   // (the gradient element is also stored: get("grad:...") reads it)

  // the tile's sum of squares in the order adam_kernel adds up the batched launch's four per-wave partials of a tile
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if (lane == 0) s_ss[wave] = ss;
  __syncthreads();
  gu64* mine = gslots + (long long)(seq & 1u) * o.slot_stride;
  if (threadIdx.x == 0 && !(fault && bid == 1)) {
    const double dt = ((double)s_ss[0] + (double)s_ss[1]) + ((double)s_ss[2] + (double)s_ss[3]);
    const int slot = lay.slot0 + t;
    __hip_atomic_store(mine + slot, (unsigned long long)__double_as_longlong(dt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    gslots[(long long)((seq & 1u) ^ 1u) * o.slot_stride + slot] = kSlotEmpty;   // nobody reads the other array in this launch; the kernel boundary publishes it
  }
  // this lane's parameter, moments and target: requested now, they arrive while the slots are awaited.  (In front of the tile body
  // they cost it a round trip: its k-loop's header waits for every load in flight — the registers its loads return in are reused
  // per iteration — and so the operand requests went out only after these had landed.)
  float pre_p = 0.f, pre_m = 0.f, pre_v = 0.f, pre_t = 0.f;
  if (my_i >= 0) {
    pre_p = gp[my_i]; pre_m = gm[my_i]; pre_v = gv[my_i];
    if (pk) pre_t = gt[my_i];
  }
  OF_STAMP(2);
  // riders of the net's first workgroup, while the other workgroups' slots arrive
  float* met = a.metrics + (long long)c.metrics_slot * kMetricFloats;
  if (bid == 0) {
    if (on.mean_x) rider_mean_metric(on.mean_x, on.mean_n, on.mean_scale, met + on.mean_index);
    if (on.td_q) rider_td_metrics(on.td_q, on.td_y, on.td_n, on.td_C, on.td_loss_kind, met);
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
    const int xc = ((int)blockIdx.x + (int)blockIdx.y * o.grid_x) & 7;
    gu64* res = mine + (o.slot_stride - 8);
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
          gslots[(long long)((seq & 1u) ^ 1u) * o.slot_stride + (o.slot_stride - 8) + xc] = kSlotEmpty;
        }
      }
    } else if (threadIdx.x == 0) {
      // (measured: four polls in flight, a quarter of a round trip apart, instead of one at a time — no gain, 52.5-52.7 vs
      // 52.2 us/step: the hop costs the store's way to the memory side plus one load round trip, not the sampling period)
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
      s_coef = clip_coef(s, c.grad_scale, ptr.clip, &post);
      if (bid == 0 && a.metrics) met[on.metric_index] = post;
    }
  }
  __syncthreads();
  OF_STAMP(3);
  const float gmul = c.grad_scale * s_coef;
  float p_new = 0.f, t_new = 0.f;
  if (my_i >= 0) {
    const AdamElem e = adam_elem(x, pre_p, pre_m, pre_v, gmul, sc, a.beta2, a.w1, a.w2, a.eps);
    gp[my_i] = e.p; gm[my_i] = e.m; gv[my_i] = e.v;
    p_new = e.p;
    if (pk) { t_new = polyak_elem(a.tau, e.p, a.one_m_tau, pre_t); gt[my_i] = t_new; }
  }
  if (lay.wt_dst >= 0) {   // (uniform per workgroup) the [in][out] copy of a hidden layer's weight: 16 consecutive outputs per run
    tile_p[4 * lg + wave][li] = p_new;
    tile_t[4 * lg + wave][li] = t_new;
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int k = n0 + ty, oo = m0 + tx;
    if (k < in && oo < out) {
      const long long at = lay.wt_dst + (long long)k * out + oo;
      gwt[at] = tile_p[tx][ty];
      if (pk && ptr.wt_target) gwtt[at] = tile_t[tx][ty];
    }
  }
  OF_STAMP(4);
#ifdef GCRL_OF_STAMPS
  if (a.stamps && threadIdx.x == 0)
    for (int k = 0; k < 5; ++k) a.stamps[((long long)blockIdx.y * 2048 + blockIdx.x) * 8 + k] = of_t[k];
#endif
  if (bid == 0 && threadIdx.x == 0) {
    ((gu32*)o.seq)[1] = 0u;
    ((gu32*)o.seq)[0] = seq + 1u;   // every workgroup of this launch read it before it published, and this workgroup has seen every slot
    if (blockIdx.y == 0 && a.advance) ctrl_advance(a.advance);
  }
}

}  // namespace

long long dw_adam_capacity() { return meet_capacity((const void*)dw_adam_kernel, 256, 0); }

int launch_dw_adam(hipStream_t st, DwAdamArgs& a) {
  GCRL_CHECK_ARG(a.nnets >= 1 && a.nnets <= 2, "dw_adam: %d nets (1 or 2)", a.nnets);
  DwAdamK k;
  std::memset(&k, 0, sizeof(k));
  k.beta2 = a.beta2; k.w1 = a.w1; k.w2 = a.w2; k.eps = a.eps; k.tau = a.tau; k.one_m_tau = a.one_m_tau;
  k.leaders = a.leaders; k.poll_gate = a.poll_gate; k.poll_first_sleep = a.poll_first_sleep; k.poll_sleep = a.poll_sleep;
  k.metrics = a.metrics; k.advance = a.advance; k.status = a.status;
#ifdef GCRL_OF_STAMPS
  k.stamps = a.stamps;
#endif
  int widest = 0;
  for (int i = 0; i < a.nnets; ++i) {
    DwAdamNet& n = a.net[i];
    DwNetK& kn = k.net[i];
    GCRL_CHECK_ARG(n.o.nl >= 1 && n.o.nl <= kFusedMaxLayers && n.o.slots && n.o.seq && n.o.cur, "dw_adam: net %d: %d problems / missing slots", i, n.o.nl);
    int tiles = 0;
    for (int l = 0; l < n.o.nl; ++l) {
      GemmDesc& d = n.d[l];
      GCRL_CHECK_ARG(d.M >= 1 && d.N >= 2 && d.K >= 1 && d.A && d.B && d.C && d.ones_col && d.col_out && !d.bias && d.epi == EPI_NONE && d.mul == MUL_NONE &&
                         !d.bn_part && d.ksplit <= 1 && d.shape_hint == 0 && gemm_shape_of(d) == 1,
                     "dw_adam: net %d problem %d is not a dW | db problem of the k-split 16x16 form", i, l);
      // ... in agent.hip bwd_dw's layout: A = G read k-major, B = X read k-major, C = dW [out][in]; a batch-slot lookup only on X,
      // through the control record this net's optimiser reads
      GCRL_CHECK_ARG(d.a_rs == 1 && d.b_cs == 1 && d.c_rs == d.N - 1 && d.a_cs < (1LL << 31) && d.b_rs < (1LL << 31) && d.a_slot == 0 && d.c_slot == 0 &&
                         (!d.slot || (d.slot == &n.o.cur->batch_slot && d.b_slot < (1LL << 31))),
                     "dw_adam: net %d problem %d is not laid out like a dW | db problem", i, l);
      DwProb& p = kn.prob[l];
      p.G = d.A; p.X = d.B; p.dW = d.C; p.db = d.col_out;
      p.pw = n.o.lay[l].pw; p.pb = n.o.lay[l].pb; p.wt_dst = n.o.lay[l].wt_dst;
      p.ldg = (int)d.a_cs; p.ldx = (int)d.b_rs; p.out = d.M; p.in = d.N - 1; p.K = d.K;
      p.tiles_n = (d.N + 15) / 16; p.slot0 = n.o.lay[l].slot0; p.x_slot = d.slot ? (int)d.b_slot : 0;
      if (l >= 1) kn.h.tile0[l - 1] = tiles;
      tiles += ((d.M + 15) / 16) * p.tiles_n;
    }
    n.o.ntiles = tiles;
    GCRL_CHECK_ARG(tiles <= 256 * kFusedMaxSlotsPerThread && tiles + 8 <= n.o.slot_stride, "dw_adam: net %d has %d tiles (slots: %d)", i, tiles, n.o.slot_stride);
    widest = std::max(widest, tiles);
    for (int l = n.o.nl; l < kFusedMaxLayers; ++l) kn.h.tile0[l - 1] = 0x7fffffff;
    kn.h.ntiles = tiles; kn.h.nl = n.o.nl; kn.h.which = n.o.which; kn.h.slot_stride = n.o.slot_stride;
    kn.h.cur = n.o.cur; kn.h.seq = n.o.seq; kn.h.slots = n.o.slots;
    kn.ptr.p = n.o.p; kn.ptr.m = n.o.m; kn.ptr.v = n.o.v; kn.ptr.target = n.o.target; kn.ptr.wt = n.o.wt; kn.ptr.wt_target = n.o.wt_target;
    kn.ptr.clip = n.o.clip; kn.h.polyak = n.o.polyak; kn.metric_index = n.o.metric_index;
    kn.mean_x = n.o.mean_x; kn.mean_n = n.o.mean_n; kn.mean_scale = n.o.mean_scale; kn.mean_index = n.o.mean_index;
    kn.td_q = n.o.td_q; kn.td_y = n.o.td_y; kn.td_n = n.o.td_n; kn.td_C = n.o.td_C; kn.td_loss_kind = n.o.td_loss_kind;
  }
  // (the residency of widest * nnets workgroups is the caller's admission check — dw_adam_capacity — made once per agent)
  for (int i = 0; i < a.nnets; ++i) k.net[i].h.grid_x = widest;
  hipLaunchKernelGGL(dw_adam_kernel, dim3((unsigned)widest, (unsigned)a.nnets), dim3(256), 0, st, k);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

}  // namespace gcrl
