// her_ring.hip — HER replay ring in HBM: staging, relabel+flush kernel, batch gather kernels.
//
// Replaces reference class HERBuffer (src/buffer.py:92-179):
//   push      src/buffer.py:110-119  -> gcrl_her_push / gcrl_her_push_episode (+ stage kernel)
//   apply_her src/buffer.py:143-179  -> her_flush_kernel (LDS-staged goal swap + reward)
//   sample    src/buffer.py:121-135  -> gcrl_her_sample (her_gather_kernel)
// Record layout and ring indexing: her_ring.h.
#include "her_ring.h"
#include "norm_math.h"
#include "ring_book.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int kMaxT = 64;          // flush_len limit (reference literal is 50)
constexpr int kMaxG = 8;
constexpr int kMaxInline = 128;    // S limit for host-pointer pushes (values travel as kernargs)
constexpr int kMaxFut = 2048;      // k_future*(T-1) limit for inline future indices
constexpr int kMaxEp = 8;          // episodes per flush launch
constexpr int kPayW = 32;          // floats per env of a vector-env step's payload [t | r | d | a(A) | ag(G)]

// ---------------------------------------------------------------- stage one transition
struct StageArgs {
  float* dst;            // record base: stage[env][t][0]
  const float* s_dev;    // device source or nullptr -> s_inl
  const float* ns_dev;
  int S, A, G, SA4, S4;
  float r, d;
  float a[16];
  float ag[kMaxG];
  float s_inl[kMaxInline];
  float ns_inl[kMaxInline];
};

__global__ __launch_bounds__(64) void her_stage_kernel(StageArgs p) {
  const int o_ns = p.SA4, o_r = p.SA4 + p.S4, RW = o_r + 2;
  for (int c = threadIdx.x; c < RW + p.G; c += 64) {
    float v = 0.f;
    if (c < p.S) v = p.s_dev ? p.s_dev[c] : p.s_inl[c];
    else if (c < p.S + p.A) v = p.a[c - p.S];
    else if (c >= o_ns && c < o_ns + p.S) v = p.ns_dev ? p.ns_dev[c - o_ns] : p.ns_inl[c - o_ns];
    else if (c == o_r) v = p.r;
    else if (c == o_r + 1) v = p.d;
    else if (c >= RW) v = p.ag[c - RW];
    p.dst[c] = v;
  }
}

// one transition per block for a whole vector-env step; the small per-env payload
// [t | r | d | a(A) | ag(G)] was uploaded in one copy
__global__ __launch_bounds__(64) void her_stage_batch_kernel(float* stage, const float* pay, const float* s_dev, int ld_s,
                                                             const float* ns_dev, int ld_ns, int env0, int flush_len, int S,
                                                             int A, int G, int SA4, int S4, int RG) {
  const int i = blockIdx.x;
  const float* pw = pay + (long long)i * kPayW;
  const int t = __float_as_int(pw[0]);
  float* dst = stage + ((long long)(env0 + i) * flush_len + t) * RG;
  const int o_ns = SA4, o_r = SA4 + S4, RW = o_r + 2;
  for (int c = threadIdx.x; c < RW + G; c += 64) {
    float v = 0.f;
    if (c < S) v = s_dev[(long long)i * ld_s + c];
    else if (c < S + A) v = pw[3 + (c - S)];
    else if (c >= o_ns && c < o_ns + S) v = ns_dev[(long long)i * ld_ns + (c - o_ns)];
    else if (c == o_r) v = pw[1];
    else if (c == o_r + 1) v = pw[2];
    else if (c >= RW) v = pw[3 + A + (c - RW)];
    dst[c] = v;
  }
}

// One vector-env step of the trainer's _process_step (src/env.py:163-201) in ONE single-block launch: the observation
// normaliser's update from [obs ; next_obs] (RunningNormalizer.update, src/utils.py:75-93 — float32 batch moments in
// numpy's order, float64 merge), then every env's transition written to its staging record with the observation columns
// normalised by the UPDATED statistics (normalize, :95-97) and the goal columns raw.
// With a goal normaliser (g_normalize, src/env.py:167-175, :222-223): its update from [dg ; next_dg ; ag ; next_ag] (the order of
// the trainer's np.concatenate), then the goal columns of both states and the pushed achieved goal normalised by it.
struct ProcArgs {
  float* stage; const float* raw; const float* pay;   // raw = [obs n*D | next_obs n*D | dg n*G | next_dg n*G | ag n*G | next_ag n*G (goal normaliser only)]
  double* mean; double* var; double* count; double clip;   // null mean: no normaliser
  double* gmean; double* gvar; double* gcount; double gclip;   // goal normaliser (null: goals stay raw)
  int update, gupdate, n, env0, flush_len, D, S, A, G, SA4, S4, RG;
  int mode, gmode;   // norm_math.h bits: float32 statistics (loaded normalisers), float64 rows (the trainer's observation batches)
};
__device__ inline void her_process_step_body(const ProcArgs& p, const float* raw, const float* pay) {
  __shared__ double s_mean[128], s_den[128], s_gmean[kMaxG], s_gden[kMaxG];
  const int n = p.n, D = p.D, G = p.G;
  const float* obs = raw; const float* nobs = obs + (size_t)n * D;
  const float* dg = nobs + (size_t)n * D; const float* ndg = dg + (size_t)n * G;
  int mode = p.mode, gmode = p.gmode;         // as of AFTER this launch's updates (float64 rows leave float64 statistics)
  if (p.update && (mode & gcrl::NORM_ROWS64)) mode &= ~gcrl::NORM_F32;
  if (p.gupdate && (gmode & gcrl::NORM_ROWS64)) gmode &= ~gcrl::NORM_F32;
  if (p.gmean && threadIdx.x >= 128 && threadIdx.x < 128 + G) {     // (a second group of threads: the two updates run side by side)
    const int j = threadIdx.x - 128;
    double m = p.gmean[j], v = p.gvar[j];
    if (p.gupdate) {
      int md = p.gmode;
      gcrl::norm_update_col(m, v, 4 * n, *p.gcount, md, [&](int i) { return dg[(size_t)i * G + j]; });   // [dg ; next_dg ; ag ; next_ag] lie in this order
      p.gmean[j] = m; p.gvar[j] = v;
    }
    s_gmean[j] = m; s_gden[j] = gcrl::norm_den(v, (gmode & gcrl::NORM_F32) != 0);
  }
  if (p.mean) {
    for (int j = threadIdx.x; j < D; j += 256) {
      double m = p.mean[j], v = p.var[j];
      if (p.update) {
        int md = p.mode;
        gcrl::norm_update_col(m, v, 2 * n, *p.count, md, [&](int i) { return obs[(size_t)i * D + j]; });   // np.concatenate([obs, next_obs]) is how the two blocks lie
        p.mean[j] = m; p.var[j] = v;
      }
      s_mean[j] = m; s_den[j] = gcrl::norm_den(v, (mode & gcrl::NORM_F32) != 0);
    }
  }
  __syncthreads();
  if (p.mean && p.update && threadIdx.x == 0) *p.count = *p.count + (double)(2 * n);   // every column has read the old count
  if (p.gmean && p.gupdate && threadIdx.x == 128) *p.gcount = *p.gcount + (double)(4 * n);
  const int o_ns = p.SA4, o_r = p.SA4 + p.S4, RW = o_r + 2, W = RW + G;
  for (int e = threadIdx.x; e < n * W; e += 256) {
    const int i = e / W, c = e - i * W;
    const float* pw = pay + (long long)i * kPayW;
    const int t = __float_as_int(pw[0]);
    float v = 0.f;
    auto state_col = [&](const float* o, const float* g, int cc) -> float {
      if (cc < D) {
        const float x = o[(size_t)i * D + cc];
        if (!p.mean) return x;
        return gcrl::norm_apply(x, s_mean[cc], s_den[cc], p.clip, gcrl::norm_apply_f32(mode));
      }
      const float x = g[(size_t)i * G + (cc - D)];
      if (!p.gmean) return x;
      return gcrl::norm_apply(x, s_gmean[cc - D], s_gden[cc - D], p.gclip, gcrl::norm_apply_f32(gmode));
    };
    if (c < p.S) v = state_col(obs, dg, c);
    else if (c < p.S + p.A) v = pw[3 + (c - p.S)];
    else if (c >= o_ns && c < o_ns + p.S) v = state_col(nobs, ndg, c - o_ns);
    else if (c == o_r) v = pw[1];
    else if (c == o_r + 1) v = pw[2];
    else if (c >= RW) {
      v = pw[3 + p.A + (c - RW)];
      if (p.gmean) v = gcrl::norm_apply(v, s_gmean[c - RW], s_gden[c - RW], p.gclip, gcrl::norm_apply_f32(gmode));   // normalize_goal(achieved_goal)
    }
    p.stage[((long long)(p.env0 + i) * p.flush_len + t) * p.RG + c] = v;
  }
}
__global__ __launch_bounds__(256) void her_process_step_kernel(ProcArgs p) { her_process_step_body(p, p.raw, p.pay); }
// the same with the step's rows and payload INSIDE the kernel arguments (round 4: a vector-env step of a few envs is < 4 KB — no
// pinned slot, no staged copy, no event: the launch itself carries the data)
constexpr int kProcInlineFloats = 896;
struct ProcArgsInline { ProcArgs p; int raw_floats; float data[kProcInlineFloats]; };
__global__ __launch_bounds__(256) void her_process_step_inline_kernel(ProcArgsInline q) { her_process_step_body(q.p, q.data, q.data + q.raw_floats); }

// ---------------------------------------------------------------- relabel + flush
struct FlushArgs {
  // the first 16 dwords are preloaded into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16 for this translation
  // unit, Makefile): what the first round of loads of a single-episode launch needs — episode 0's staging pointer and length, the
  // record geometry — so that those loads do not wait for a kernel-argument fetch first
  float* ring;
  const float* stage0; int T0, k, G, RG, RW, nep;
  long long cap, tail, skip;  // rows whose running number is < skip fell off a too-small ring
  int S, A, SA4, S4, RS;
  int reward_kind;
  float thr;
  int rng_mode;
  unsigned long long seed;
  const float* stage[kMaxEp];
  int T[kMaxEp];
  unsigned long long epi_id[kMaxEp];
  int fut_off[kMaxEp];      // first relabel slot of episode e in this launch (k*(T-1) slots per episode, draw order)
  const uint8_t* fut_ext;   // more than kMaxFut future indices in this launch: they were uploaded instead (k_future >= 42)
  const float* rew_ext;     // GCRL_REWARD_HOST: relabel rewards computed by the caller's compute_reward on the host, one per slot
  uint8_t fut[kMaxFut];
};

// grid.x = ceil(rows of the longest episode / 16), grid.y = episode.  A block writes 16 consecutive output rows of its episode
// (row n of an episode = copy n % (1+k) of step n / (1+k): the original, then its k relabels; the last step has the original
// only), one row per 16-lane group:
//   (1) exclusive scan of per-episode output-row counts (one wavefront, cross-lane shifts)
//       -> first row number of its episode (segment base of the multi-episode flush);
//   (2) ONE round of loads into LDS: the records of the (<= 3 at k = 8) steps its rows belong to, the whole achieved-goal column
//       of the episode, the future picks (and host-computed rewards) of those steps;
//   (3) every group assembles its row — relabel = goal slot of s and ns swapped for ag[f] from LDS, reward recomputed,
//       done = 0 (src/buffer.py:151-179) — from independent LDS reads (what a lane's four columns ARE is decided before the
//       barrier) and stores it as full 16-byte instructions over contiguous bytes.
// Round 4 (11.5 us per single-episode launch).  By device-clock stamps the round-3 form spent ~3 us in its loads and
// ~6 us in NINE row iterations per wave: a row is ~300 instructions of index arithmetic, branches and dependent LDS reads, and a
// lone wave per SIMD issues them at ~5 cycles each — neither the stores nor the loads mattered (removing either changed
// nothing).  Hence one row per group and no loop: 28 blocks per 442-row episode instead of 13, each re-reading ~2 KB.
// (kMulti only separates the names in a profile: single-episode launches from a vector-env step's multi-episode ones)
constexpr int kFlushRows = 16;               // rows per block
template <bool kMulti>
__global__ __launch_bounds__(256) void her_flush_kernel(FlushArgs p) {
  __shared__ float ag_lds[kMaxT * kMaxG];                              // [step][8]
  __shared__ __attribute__((aligned(16))) float rec_lds[kFlushRows * 176];   // the staged records of this block's steps, stride RG (<= 176 floats)
  __shared__ int fut_lds[kFlushRows * 64];             // future pick of (step, rep): k_future <= 64 per step
  __shared__ float rew_lds[kFlushRows * 64];
  __shared__ long long base_lds;

  const int e = kMulti ? blockIdx.y : 0;
  const int T = kMulti ? p.T[e] : p.T0;
  const int k = p.k, G = p.G, RG = p.RG, reps = 1 + k;
  const int total = T + k * (T - 1);
  const int n0 = blockIdx.x * kFlushRows;
  if (n0 >= total) return;
  const int n_hi = min(total - 1, n0 + kFlushRows - 1);
  const int i_lo = n0 / reps, i_hi = n_hi / reps;        // (uniform)
  const int nsteps = i_hi - i_lo + 1;                    // <= 16 (k = 0), 3 at k = 8
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  if (wave == 0) {
    long long rows = 0;
    if (kMulti) { if (lane < p.nep) rows = (long long)p.T[lane] + (long long)k * (p.T[lane] - 1); }
    else if (lane == 0) rows = (long long)T + (long long)k * (T - 1);
    long long incl = rows;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      long long up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == e) base_lds = incl - rows;
  }
  const float* stg = kMulti ? p.stage[e] : p.stage0;
  // everything the rows need, requested together (no divisions in these loops)
  for (int t = tid; t < T * 8; t += 256) {               // the episode's achieved-goal column, 8 slots per step
    const int st = t >> 3, q = t & 7;
    ag_lds[t] = q < G ? stg[(long long)st * RG + p.RW + q] : 0.f;
  }
  {
    const float4* src = reinterpret_cast<const float4*>(stg + (long long)i_lo * RG);   // (RG is a multiple of 16 floats)
    float4* dst = reinterpret_cast<float4*>(rec_lds);
    for (int t = tid; t < nsteps * (RG >> 2); t += 256) dst[t] = src[t];
  }
  {
    const uint8_t* fut = p.fut_ext ? p.fut_ext : p.fut;
    const int slot0 = p.fut_off[e] + i_lo * k;
    for (int t = tid; t < nsteps * 64; t += 256) {
      const int li = t >> 6, rep = t & 63, i = i_lo + li;
      int f = 0;
      if (rep < k && i < T - 1) {
        if (p.rng_mode == GCRL_RNG_CPYTHON_MT) f = fut[slot0 + li * k + rep];
        else f = i + 1 + (int)gcrl::hash_below(p.seed, p.epi_id[e], (unsigned long long)(i * k + rep), (uint32_t)(T - 1 - i));
        if (p.rew_ext) rew_lds[t] = p.rew_ext[slot0 + li * k + rep];
      }
      fut_lds[t] = f;
    }
  }
  // what this lane's four columns of a row are — the same for every row: 0 copy, 1 goal slot of s / ns (component gq), 2 the
  // reward, 3 the done flag, 4 padding
  const int grp = tid >> 4, gl = tid & 15;       // 16 groups of 16 lanes: a group owns a row
  const int o_ns = p.SA4, o_r = p.SA4 + p.S4;
  const int gs0 = p.S - G, gs1 = o_ns + p.S - G;
  int kind[3][4], gq[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 64 * j + 4 * gl + u;
      int kd = 0, q = 0;
      if (c >= gs0 && c < p.S) { kd = 1; q = c - gs0; }
      else if (c >= gs1 && c < o_ns + p.S) { kd = 1; q = c - gs1; }
      else if (c == o_r) kd = 2;
      else if (c == o_r + 1) kd = 3;
      else if (c > o_r + 1) kd = 4;
      kind[j][u] = kd; gq[j][u] = q;
    }
  // this group's row
  const int n = n0 + grp;
  int li = 0, rep = n - i_lo * reps;
  while (rep >= reps) { rep -= reps; ++li; }
  __syncthreads();
  if (n > n_hi) return;
  const long long g = base_lds + n;
  if (g < p.skip) return;                     // fell off a ring smaller than the flush
  long long phys = p.tail + g;
  if (phys >= p.cap) phys -= p.cap;
  if (phys >= p.cap) phys %= p.cap;           // (a ring smaller than the flush)
  const float* rec = rec_lds + li * RG;
  const int f = rep > 0 ? fut_lds[li * 64 + rep - 1] : 0;
  const float* agf = ag_lds + f * 8;
  // all of the row's LDS reads are independent: issued together, one wait
  float a_i[kMaxG], a_f[kMaxG];
#pragma unroll
  for (int q = 0; q < kMaxG; ++q) { a_i[q] = q < G ? rec[p.RW + q] : 0.f; a_f[q] = agf[q]; }
  float rew = rec[o_r], done = rec[o_r + 1];
  const float rew_host = (p.rew_ext && rep > 0) ? rew_lds[li * 64 + rep - 1] : 0.f;
  if (rep > 0) {
    // compute_reward(ag_i, ag_f): d = ||ag_i - ag_f||_2 in fp32, one rounding per op
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxG; ++q)
      if (q < G) { const float df = __fsub_rn(a_i[q], a_f[q]); acc = __fadd_rn(acc, __fmul_rn(df, df)); }
    const float dist = sqrtf(acc);
    rew = (p.reward_kind == GCRL_REWARD_SPARSE) ? ((dist > p.thr) ? -1.0f : -0.0f) : -dist;
    if (p.rew_ext) rew = rew_host;
    done = 0.f;
  }
  float* out = p.ring + phys * p.RS;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c4 = 64 * j + 4 * gl;
    if (c4 < p.RS) {
      const float4 r4 = *reinterpret_cast<const float4*>(rec + c4);     // (c4 + 3 < RS <= RG)
      float v[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kd = kind[j][u];
        float x = v[u];
        const float gsw = agf[gq[j][u]];                                  // (independent LDS read; gq = 0 where unused)
        x = (kd == 1 && rep > 0) ? gsw : x;
        x = kd == 2 ? rew : x;
        x = kd == 3 ? done : x;
        x = kd == 4 ? 0.f : x;
        v[u] = x;
      }
      *reinterpret_cast<float4*>(out + c4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---------------------------------------------------------------- gather (public sample)
struct GatherArgs {
  const float* ring;
  const uint32_t* idx;  // logical indices (null: computed per row from `gen`, device-RNG mode)
  gcrl::IdxGen gen;
  long long n, head, cap;
  int S, A, SA4, S4, RS;
  float *out_s, *out_a, *out_r, *out_ns, *out_d;
  int ld_s, ld_a, ld_ns;
};

// Records of <= 64 floats: a block owns 64 consecutive batch rows.  Load: 16 lanes x float4
// cover a record, so one wave-load instruction fetches four records as whole 128-B lines, 16
// records per wave in flight.  The records go through LDS; then every output field is written
// as ONE contiguous span (rows r0..r0+63 of a dense [n, S] matrix are adjacent in memory):
// lane e -> element e of the span, fully coalesced dword stores.
__global__ __launch_bounds__(256) void her_gather_kernel(GatherArgs p) {
  __shared__ float tile[64][65];  // +1: column-ish LDS reads of the store phase stay conflict-free
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, v4 = lane & 15;
  const int o_ns = p.SA4, o_r = p.SA4 + p.S4;
  const int c0 = v4 * 4;
  for (long long r0 = (long long)blockIdx.x * 64; r0 < p.n; r0 += (long long)gridDim.x * 64) {
    float4 val[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row = r0 + wave * 16 + u * 4 + sub;
      val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < p.n && c0 < p.RS) {
        const long long phys = (p.head + (long long)(p.idx ? p.idx[row] : gcrl::idxgen_at(p.gen, row))) % p.cap;
        val[u] = *reinterpret_cast<const float4*>(p.ring + phys * p.RS + c0);
      }
    }
    __syncthreads();  // previous iteration's store phase is done with the tile
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float* dst = &tile[wave * 16 + u * 4 + sub][c0];
      dst[0] = val[u].x; dst[1] = val[u].y; dst[2] = val[u].z; dst[3] = val[u].w;
    }
    __syncthreads();
    const int rows = (int)min((long long)64, p.n - r0);
    auto span = [&](float* out, int ld, int width, int col0) {
      // elements (row, c) of a [rows, width] slab; contiguous in memory when ld == width
      for (int e = threadIdx.x; e < rows * width; e += 256) {
        const int row = e / width, c = e - row * width;
        out[(r0 + row) * ld + c] = tile[row][col0 + c];
      }
    };
    span(p.out_s, p.ld_s, p.S, 0);
    span(p.out_a, p.ld_a, p.A, p.S);
    span(p.out_ns, p.ld_ns, p.S, o_ns);
    span(p.out_r, 1, 1, o_r);
    span(p.out_d, 1, 1, o_r + 1);
  }
}

// records wider than 64 floats: per-lane routing, no LDS staging
template <int kUnroll>
__global__ __launch_bounds__(256) void her_gather_wide_kernel(GatherArgs p) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const int o_ns = p.SA4, o_r = p.SA4 + p.S4;
  const int chunks = (p.RS + 63) / 64;  // 64-float pieces per record
  for (long long r0 = wave_id * (4 * kUnroll); r0 < p.n; r0 += nwaves * (4 * kUnroll)) {
    for (int ch = 0; ch < chunks; ++ch) {
      float4 val[kUnroll];
      long long row[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        row[u] = r0 + u * 4 + sub;
        val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c0 = ch * 64 + v4 * 4;
        if (row[u] < p.n && c0 < p.RS) {
          long long phys = (p.head + (long long)(p.idx ? p.idx[row[u]] : gcrl::idxgen_at(p.gen, row[u]))) % p.cap;
          val[u] = *reinterpret_cast<const float4*>(p.ring + phys * p.RS + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (row[u] >= p.n) continue;
        const float vv[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = ch * 64 + v4 * 4 + q;
          if (c < p.S) p.out_s[row[u] * p.ld_s + c] = vv[q];
          else if (c < p.S + p.A) p.out_a[row[u] * p.ld_a + (c - p.S)] = vv[q];
          else if (c >= o_ns && c < o_ns + p.S) p.out_ns[row[u] * p.ld_ns + (c - o_ns)] = vv[q];
          else if (c == o_r) p.out_r[row[u]] = vv[q];
          else if (c == o_r + 1) p.out_d[row[u]] = vv[q];
        }
      }
    }
  }
}

// ---------------------------------------------------------------- gather (update engine)
struct GatherUpdArgs {
  const float* ring;
  const uint32_t* idx;   // null: computed per row from `gen`
  gcrl::IdxGen gen;
  long long n, head, cap;
  int SA4, S4, RS, ldx;   // ldx == SA4
  float *sa, *nsa, *spa, *r, *d;
  const uint4* cp_src; uint4* cp_dst; int cp_n16;   // optional side copy (her_ring.h), all blocks share it
};

typedef float gcrl_f4 __attribute__((ext_vector_type(4)));
__device__ inline void store4_nt(float* p, float4 v) {
  gcrl_f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<gcrl_f4*>(p));
}

// index -> record -> sa | nsa | r | d (src/buffer.py:121-135 as the update engine consumes it).  One wave owns 16
// consecutive batch rows (exact grid, one pass):
//   (1) its 16 indices arrive in ONE coalesced load (lane l <- idx[r0 + l]; device-RNG mode: lane l evaluates the keyed
//       permutation) and reach the 16-lane groups by a cross-lane read; ring wrap by compare-and-subtract;
//   (2) 16 lanes x 16 B cover a record, four records per wave-load, all of a wave's loads in flight before anything else
//       (16 records per wave, ~78 K records per launch at once: the chip's random-row rate, not a latency chain);
//       lanes that would fetch only the record's padding do not load;
//   (3) the records pass through a WAVE-PRIVATE LDS tile laid out like the outputs — [16][SA4] rows of sa, [16][SA4] rows
//       of nsa (columns >= S4: zeros; the actor writes a' there before any critic reads it), r[16], d[16] — so that
//   (4) every store is a full-width 16-byte-per-lane instruction over CONTIGUOUS output bytes (sa rows r0..r0+15 are adjacent
//       in memory, so are nsa's, r's, d's; lanes of one instruction may point into different matrices), non-temporal: the
//       batch matrices are consumed once, by other kernels, much later.
// Measured (tools/gather_micro.hip, rocprofv3 durations, 77 824 PickAndPlace rows from a 1e6-row ring): round-2 kernel (per-lane
// index loads, 64-bit modulo, stores straight from the load lanes: 4 partly masked dwordx4 + 8 one-dword stores per wave)
// 8.9 us; coalesced indices 8.1; + non-temporal stores 6.9; + LDS-staged contiguous stores 6.4 us = 0.63 of 8 TB/s
// (an in-order copy of the same bytes: 6.4 / 5.2 us with plain / non-temporal stores).  Scalar (s_load) index fetches were
// slower (8.9), so were 8 / 32 / 64 rows per wave and non-temporal record loads at this launch size.
// kHead only separates the names in a profile: the call-start launch (2 batches, indices read from the pinned upload block,
// control-block side copy) from a cycle's main gather.
template <bool kHead, bool kNT = true>
__global__ __launch_bounds__(256) void her_gather_update_kernel(GatherUpdArgs p) {
  extern __shared__ float gather_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane >> 4, v4 = lane & 15;
  if (kHead)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.cp_n16; i += gridDim.x * 256) p.cp_dst[i] = p.cp_src[i];
  const int SA4 = p.SA4, S4 = p.S4, o_r = SA4 + S4;
  const int nq = 4 * SA4;                                   // 16-byte quads of a [16][SA4] tile
  float* tile = gather_lds + (size_t)w * (8 * nq + 32);     // [sa 16 x SA4 | nsa 16 x SA4 | r 16 | d 16]
  float* t_ns = tile + 4 * nq;
  float* t_rd = tile + 8 * nq;
  const long long r0 = ((long long)blockIdx.x * 4 + w) * 16;
  if (r0 >= p.n) return;
  uint32_t ph32 = 0;
  if (lane < 16 && r0 + lane < p.n) {
    unsigned long long phys = (unsigned long long)p.head + (p.idx ? p.idx[r0 + lane] : gcrl::idxgen_at(p.gen, r0 + lane));
    if (phys >= (unsigned long long)p.cap) phys -= (unsigned long long)p.cap;    // head, index < cap
    ph32 = (uint32_t)phys;
  }
  const bool full = r0 + 16 <= p.n;
  // records wider than 64 floats (state dims above ~28) take further 64-float column passes
  for (int cc = 0; cc <= o_r; cc += 64) {
    const int c0 = cc + v4 * 4;
    const bool useful = c0 <= o_r;
    float4 val[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t ph = __shfl(ph32, u * 4 + sub, 64);
      val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (useful && r0 + u * 4 + sub < p.n) val[u] = *reinterpret_cast<const float4*>(p.ring + (size_t)ph * p.RS + c0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rl = u * 4 + sub;
      const long long row = r0 + rl;
      if (!useful || row >= p.n) continue;
      if (p.spa && c0 < S4) *reinterpret_cast<float4*>(p.spa + row * p.ldx + c0) = val[u];   // layer-per-launch schedules only
      if (full) {
        if (c0 < SA4) *reinterpret_cast<float4*>(tile + rl * SA4 + c0) = val[u];
        else if (c0 < o_r) *reinterpret_cast<float4*>(t_ns + rl * SA4 + (c0 - SA4)) = val[u];
        else { t_rd[rl] = val[u].x; t_rd[16 + rl] = val[u].y; }
      } else {   // the launch's last, partial wave: straight from the load lanes
        if (c0 < SA4) *reinterpret_cast<float4*>(p.sa + row * p.ldx + c0) = val[u];
        else if (c0 < o_r) *reinterpret_cast<float4*>(p.nsa + row * p.ldx + (c0 - SA4)) = val[u];
        else { p.r[row] = val[u].x; p.d[row] = val[u].y; }
      }
    }
  }
  if (!full) return;
  const int zq = (SA4 - S4) >> 2;          // quads of an nsa row beyond the record's ns group (<= 5: action_dim <= 16)
  if (v4 < zq) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(t_ns + (u * 4 + sub) * SA4 + S4 + v4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the tile is private to this wave and a wave's LDS operations execute in order: no barrier, only the compiler is held back
  __builtin_amdgcn_wave_barrier();
  float* sa = p.sa + r0 * p.ldx;
  float* nsa = p.nsa + r0 * p.ldx;
  const bool rd_vec = ((reinterpret_cast<size_t>(p.r) | reinterpret_cast<size_t>(p.d)) & 15) == 0;
  const int Q = 2 * nq + (rd_vec ? 8 : 0);
  for (int q = lane; q < Q; q += 64) {
    const float4 v = *reinterpret_cast<const float4*>(tile + q * 4);
    float* dst = q < nq ? sa + q * 4 : q < 2 * nq ? nsa + (q - nq) * 4 : q < 2 * nq + 4 ? p.r + r0 + (q - 2 * nq) * 4 : p.d + r0 + (q - 2 * nq - 4) * 4;
    if (kNT) store4_nt(dst, v);
    else *reinterpret_cast<float4*>(dst) = v;
  }
  if (!rd_vec && lane < 32) (lane < 16 ? p.r : p.d)[r0 + (lane & 15)] = t_rd[lane];
}

// round 2's form of the same gather (per-lane index loads, stores straight from the load lanes), kept for same-box A/B runs:
// GCRL_GATHER_R2=1
template <int kUnroll>
__global__ __launch_bounds__(256) void her_gather_update_r2_kernel(GatherUpdArgs p) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const int o_r = p.SA4 + p.S4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < p.cp_n16; i += gridDim.x * 256) p.cp_dst[i] = p.cp_src[i];
  // records wider than 64 floats (state dims above ~28) take further 64-float column passes
  for (int cc = 0; cc < p.RS; cc += 64) {
    const int c0 = cc + v4 * 4;
    for (long long r0 = wave_id * (4 * kUnroll); r0 < p.n; r0 += nwaves * (4 * kUnroll)) {
      float4 val[kUnroll];
      long long row[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        row[u] = r0 + u * 4 + sub;
        val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row[u] < p.n && c0 < p.RS) {
          const long long phys = (p.head + (long long)(p.idx ? p.idx[row[u]] : gcrl::idxgen_at(p.gen, row[u]))) % p.cap;
          val[u] = *reinterpret_cast<const float4*>(p.ring + phys * p.RS + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (row[u] >= p.n) continue;
        const long long ro = row[u] * p.ldx;
        if (c0 < p.SA4) {
          *reinterpret_cast<float4*>(p.sa + ro + c0) = val[u];
          if (p.spa && c0 < p.S4) *reinterpret_cast<float4*>(p.spa + ro + c0) = val[u];
        } else if (c0 < o_r) {
          *reinterpret_cast<float4*>(p.nsa + ro + (c0 - p.SA4)) = val[u];
        } else if (c0 == o_r) {
          p.r[row[u]] = val[u].x;
          p.d[row[u]] = val[u].y;
        }
      }
    }
  }
}

// rows [first, first+n) in logical order -> contiguous records (read_rows)
__global__ void her_copy_rows_kernel(const float* ring, long long head, long long cap, int RS,
                                     long long first, long long n, float* out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * RS) return;
  long long row = t / RS;
  int c = (int)(t - row * RS);
  out[t] = ring[((head + first + row) % cap) * RS + c];
}

constexpr size_t kProfPairs = 256;

int prof_drain(gcrl_her* h) {
  if (h->prof_used == 0) return GCRL_OK;
  for (size_t i = 0; i < h->prof_used; ++i) {
    float ms = 0.f;
    GCRL_HIP(hipEventSynchronize(h->prof_b[i]));
    GCRL_HIP(hipEventElapsedTime(&ms, h->prof_a[i], h->prof_b[i]));
    const int64_t rows = h->prof_pair_rows[i];
    // a trainer cycle issues a 1-batch head launch (so that step 0 starts early) and the main gather of the other
    // batches: the statistics describe the largest launch size seen
    if (rows > h->prof_class_rows) { h->prof_class_rows = rows; h->prof_launches = 0; h->prof_rows = 0; h->prof_ms = 0.0; }
    if (rows != h->prof_class_rows) continue;
    h->prof_launches++;
    h->prof_rows += rows;
    h->prof_ms += ms;
  }
  h->prof_used = 0;
  return GCRL_OK;
}
int prof_begin(gcrl_her* h, hipStream_t st) {
  if (!h->prof) return GCRL_OK;
  if (h->prof_used == kProfPairs)
    if (int rc = prof_drain(h)) return rc;
  GCRL_HIP(hipEventRecord(h->prof_a[h->prof_used], st));
  return GCRL_OK;
}
int prof_end(gcrl_her* h, hipStream_t st, int64_t rows) {
  if (!h->prof) return GCRL_OK;
  GCRL_HIP(hipEventRecord(h->prof_b[h->prof_used], st));
  h->prof_pair_rows[h->prof_used] = rows;
  h->prof_used++;
  return GCRL_OK;
}

int ensure_index_capacity(gcrl_her* h, size_t rows) {
  if (rows <= h->slot_rows && rows <= h->idx_dev_rows) return GCRL_OK;
  GCRL_HIP(hipStreamSynchronize(h->stream));
  GCRL_HIP(hipDeviceSynchronize());
  size_t want = std::max<size_t>(rows, 4096);
  for (int i = 0; i < gcrl_her::kSlots; ++i) {
    if (h->idx_pinned[i]) GCRL_HIP(hipHostFree(h->idx_pinned[i]));
    GCRL_HIP(hipHostMalloc((void**)&h->idx_pinned[i], want * sizeof(uint32_t), hipHostMallocDefault));
  }
  if (h->idx_dev) GCRL_HIP(hipFree(h->idx_dev));
  GCRL_HIP(hipMalloc((void**)&h->idx_dev, want * sizeof(uint32_t)));
  h->slot_rows = h->idx_dev_rows = want;
  return GCRL_OK;
}

int launch_flush(gcrl_her* h, int nep, const int* envs, const int* Ts, const uint8_t* const* futs,
                 hipStream_t st, int64_t* rows_out) {
  const gcrl_her_config& c = h->cfg;
  FlushArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  fa.ring = h->ring;
  fa.cap = c.capacity;
  gcrl::RingBook book{c.capacity, h->head, h->len};   // deque(maxlen) bookkeeping (ring_book.h)
  fa.tail = book.tail();
  fa.nep = nep;
  fa.k = c.k_future; fa.S = h->S; fa.A = h->A; fa.G = h->G; fa.SA4 = h->SA4; fa.S4 = h->S4; fa.RW = h->RW; fa.RS = h->RS; fa.RG = h->RG;
  fa.reward_kind = c.reward_kind;
  fa.thr = c.reward_threshold;
  fa.rng_mode = c.rng_mode;
  fa.seed = c.seed;
  int64_t total = 0;
  int fut_used = 0, maxT = 0;
  // future indices of the launch's episodes, in draw order (src/buffer.py:146-153): inline in the kernel arguments
  // while they fit, uploaded through a pinned slot otherwise
  size_t fut_need = 0;
  for (int e = 0; e < nep; ++e) fut_need += (size_t)c.k_future * (Ts[e] - 1);
  uint8_t* fut_host = fa.fut;
  int fslot = -1;
  if (c.rng_mode == GCRL_RNG_CPYTHON_MT && fut_need > (size_t)kMaxFut) {
    const size_t cap = (size_t)kMaxEp * c.k_future * c.flush_len;
    if (!h->fut_dev) {
      GCRL_HIP(hipMalloc((void**)&h->fut_dev, cap));
      for (int i = 0; i < gcrl_her::kSlots; ++i) GCRL_HIP(hipHostMalloc((void**)&h->fut_pinned[i], cap, hipHostMallocDefault));
    }
    fslot = h->next_fut_slot;
    h->next_fut_slot = (fslot + 1) % gcrl_her::kSlots;
    GCRL_HIP(hipEventSynchronize(h->fut_ev[fslot]));
    fut_host = h->fut_pinned[fslot];
  }
  for (int e = 0; e < nep; ++e) {
    const int T = Ts[e];
    fa.stage[e] = h->stage + ((size_t)envs[e] * c.flush_len) * h->RG;
    fa.T[e] = T;
    if (e == 0) { fa.stage0 = fa.stage[0]; fa.T0 = T; }
    fa.epi_id[e] = h->episodes_flushed + e;
    fa.fut_off[e] = fut_used;
    const int nf = c.k_future * (T - 1);
    if (c.rng_mode == GCRL_RNG_CPYTHON_MT) {
      if (futs && futs[e]) std::memcpy(fut_host + fut_used, futs[e], nf);
      else if (int rc = gcrl_mt_future_indices(h->rng, T, c.k_future, fut_host + fut_used)) return rc;
    }
    fut_used += nf;
    total += T + (int64_t)c.k_future * (T - 1);
    maxT = std::max(maxT, T);
  }
  if (fslot >= 0) {
    GCRL_HIP(hipMemcpyAsync(h->fut_dev, fut_host, (size_t)fut_used, hipMemcpyHostToDevice, st));
    GCRL_HIP(hipEventRecord(h->fut_ev[fslot], st));
    fa.fut_ext = h->fut_dev;
  }
  if (c.reward_kind == GCRL_REWARD_HOST && fut_used > 0) {
    // compute_reward(ag_i, ag_f, {}) for every relabel slot (src/buffer.py:166) by the caller's function, in the
    // reference's call order; the achieved goals are the host mirror of the staging area, the picks the ones the
    // kernel will use (MT: drawn above; device RNG: the same counter hash, restated on the host)
    if (!h->reward_cb) return gcrl::fail(GCRL_ERR_STATE, "flush: reward_kind is GCRL_REWARD_HOST but no callback is set (gcrl_her_set_reward_callback)");
    const size_t cap = (size_t)kMaxEp * c.k_future * c.flush_len;
    if (!h->rew_dev) {
      GCRL_HIP(hipMalloc((void**)&h->rew_dev, cap * sizeof(float)));
      for (int i = 0; i < gcrl_her::kSlots; ++i) {
        GCRL_HIP(hipHostMalloc((void**)&h->rew_pinned[i], cap * sizeof(float), hipHostMallocDefault));
        GCRL_HIP(hipEventCreateWithFlags(&h->rew_ev[i], hipEventDisableTiming));
      }
    }
    const int rslot = h->next_rew_slot;
    h->next_rew_slot = (rslot + 1) % gcrl_her::kSlots;
    GCRL_HIP(hipEventSynchronize(h->rew_ev[rslot]));
    const int G = h->G;
    h->cb_ag.resize((size_t)fut_used * G);
    h->cb_goal.resize((size_t)fut_used * G);
    for (int e = 0; e < nep; ++e) {
      const float* ag = h->ag_mirror.data() + (size_t)envs[e] * c.flush_len * G;
      for (int i = 0; i + 1 < Ts[e]; ++i)
        for (int r = 0; r < c.k_future; ++r) {
          const int slot = fa.fut_off[e] + i * c.k_future + r;
          const int f = c.rng_mode == GCRL_RNG_CPYTHON_MT
                            ? fut_host[slot]
                            : i + 1 + (int)gcrl::hash_below(c.seed, fa.epi_id[e], (unsigned long long)(i * c.k_future + r), (uint32_t)(Ts[e] - 1 - i));
          std::memcpy(&h->cb_ag[(size_t)slot * G], ag + (size_t)i * G, sizeof(float) * G);
          std::memcpy(&h->cb_goal[(size_t)slot * G], ag + (size_t)f * G, sizeof(float) * G);
        }
    }
    if (h->reward_cb(h->cb_ag.data(), h->cb_goal.data(), fut_used, G, h->rew_pinned[rslot], h->reward_cb_user) != 0)
      return gcrl::fail(GCRL_ERR_STATE, "flush: the compute_reward callback reported a failure");
    GCRL_HIP(hipMemcpyAsync(h->rew_dev, h->rew_pinned[rslot], (size_t)fut_used * sizeof(float), hipMemcpyHostToDevice, st));
    GCRL_HIP(hipEventRecord(h->rew_ev[rslot], st));
    fa.rew_ext = h->rew_dev;
  }
  fa.skip = book.append(total);   // (head / len of the handle follow once the launch is out)
  int max_rows = 0;
  for (int e = 0; e < nep; ++e) max_rows = std::max(max_rows, Ts[e] + c.k_future * (Ts[e] - 1));
  dim3 grid((max_rows + kFlushRows - 1) / kFlushRows, nep);
  if (nep > 1) hipLaunchKernelGGL(her_flush_kernel<true>, grid, dim3(256), 0, st, fa);
  else hipLaunchKernelGGL(her_flush_kernel<false>, grid, dim3(256), 0, st, fa);
  GCRL_HIP(hipGetLastError());
  // deque(maxlen) bookkeeping: `total` rows appended, the oldest fell off the front
  h->head = book.head;
  h->len = book.len;
  h->episodes_flushed += nep;
  h->mutation_epoch++;
  *rows_out = total;
  return GCRL_OK;
}

}  // namespace

namespace gcrl {

int her_upload_indices(gcrl_her* h, int B, int M, const uint32_t* idx_host, hipStream_t st,
                       const uint32_t** host_copy) {
  const size_t rows = (size_t)B * M;
  if (h->len < B) return fail(GCRL_ERR_NOT_ENOUGH, "[ERROR] Not enough in buffer to sample");
  if (int rc = ensure_index_capacity(h, rows)) return rc;
  const int slot = h->next_slot;
  h->next_slot = (slot + 1) % gcrl_her::kSlots;
  GCRL_HIP(hipEventSynchronize(h->slot_ev[slot]));  // previous upload from this slot is done
  uint32_t* dst = h->idx_pinned[slot];
  if (idx_host) {
    for (size_t i = 0; i < rows; ++i) {
      if (idx_host[i] >= (uint64_t)h->len) return fail(GCRL_ERR_ARG, "sample: index %u out of range (len %lld)", idx_host[i], (long long)h->len);
      dst[i] = idx_host[i];
    }
  } else if (h->cfg.rng_mode == GCRL_RNG_CPYTHON_MT) {
    for (int m = 0; m < M; ++m)
      if (int rc = gcrl_mt_sample_indices(h->rng, (uint32_t)h->len, (uint32_t)B, dst + (size_t)m * B)) return rc;
  } else {
    // device-RNG mode: the gather kernels compute the indices themselves (her_ring.h idxgen_at); the host
    // restates them only when the caller wants to see them
    if (host_copy) {
      const int hb = feistel_half_bits((uint32_t)h->len);
      for (size_t i = 0; i < rows; ++i)
        dst[i] = feistel_index(h->cfg.seed, h->draws_done + i / B, (uint32_t)h->len, hb, (uint32_t)(i % B));
      *host_copy = dst;
    }
    h->last_gen = IdxGen{h->cfg.seed, h->draws_done, (uint32_t)h->len, B, feistel_half_bits((uint32_t)h->len)};
    h->draws_done += M;
    h->idx_on_device = false;
    return GCRL_OK;
  }
  h->idx_on_device = true;
  GCRL_HIP(hipMemcpyAsync(h->idx_dev, dst, rows * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  GCRL_HIP(hipEventRecord(h->slot_ev[slot], st));
  if (host_copy) *host_copy = dst;
  return GCRL_OK;
}

int her_gather_update(gcrl_her* h, const uint32_t* idx_dev, int64_t n, float* sa, float* nsa,
                      float* spa, int ldx, float* r, float* d, hipStream_t st, const void* cp_src, void* cp_dst, size_t cp_bytes) {
  if (cp_bytes % 16 != 0 || (cp_bytes && (!cp_src || !cp_dst))) return fail(GCRL_ERR_ARG, "her_gather_update: bad side copy (%zu bytes)", cp_bytes);
  if (ldx != h->SA4) return fail(GCRL_ERR_ARG, "her_gather_update: batch row stride %d != roundup(S+A,4) = %d", ldx, h->SA4);
  if (int rc = prof_begin(h, st)) return rc;
  GatherUpdArgs ga{h->ring, idx_dev, h->last_gen, n, h->head, h->cfg.capacity, h->SA4, h->S4, h->RS, ldx, sa, nsa, spa, r, d,
                   (const uint4*)cp_src, (uint4*)cp_dst, (int)(cp_bytes / 16)};
  const int blocks = (int)((n + 63) / 64);                                              // 16 rows per wave, 4 waves per block
  const size_t lds = 4 * ((size_t)32 * h->SA4 + 32) * sizeof(float);                    // <= 54 KB (record <= 160 floats)
  static const int dev_variant = std::getenv("GCRL_GATHER_R2") ? 2 : (std::getenv("GCRL_GATHER_PLAIN") ? 1 : 0);   // development A/B knobs
  if (dev_variant == 2) hipLaunchKernelGGL(her_gather_update_r2_kernel<4>, dim3((int)std::min<int64_t>((n + 63) / 64, 8192)), dim3(256), 0, st, ga);
  else if (cp_bytes) hipLaunchKernelGGL(her_gather_update_kernel<true>, dim3(blocks), dim3(256), lds, st, ga);
  else if (dev_variant == 1) hipLaunchKernelGGL((her_gather_update_kernel<false, false>), dim3(blocks), dim3(256), lds, st, ga);
  else hipLaunchKernelGGL(her_gather_update_kernel<false>, dim3(blocks), dim3(256), lds, st, ga);
  GCRL_HIP(hipGetLastError());
  return prof_end(h, st, n);
}

}  // namespace gcrl

extern "C" {

int gcrl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

gcrl_her* gcrl_her_create(const gcrl_her_config* cfg, gcrl_mt* rng) {
  auto bad = [](const char* m) -> gcrl_her* { gcrl::fail(GCRL_ERR_ARG, "gcrl_her_create: %s", m); return nullptr; };
  if (!cfg) return bad("null config");
  if (cfg->state_dim < 1 || cfg->action_dim < 1 || cfg->action_dim > 16) return bad("state_dim >= 1 and 1 <= action_dim <= 16 required");
  if (cfg->goal_dim < 1 || cfg->goal_dim > kMaxG || cfg->goal_dim > cfg->state_dim) return bad("goal_dim must be 1..8 and <= state_dim");
  if (cfg->capacity < 1 || cfg->capacity >= (1ll << 32)) return bad("capacity must be in [1, 2^32)");
  if (cfg->nenvs < 1 || cfg->k_future < 0 || cfg->k_future > 64) return bad("nenvs >= 1 and 0 <= k_future <= 64 required");
  if (cfg->flush_len < 1 || cfg->flush_len > kMaxT) return bad("flush_len must be 1..64");
  if (2 * cfg->state_dim + cfg->action_dim + 8 + cfg->goal_dim > 160) return bad("record wider than 160 floats");
  int ndev = gcrl_device_count();
  if (ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    gcrl::fail(GCRL_ERR_HIP, "gcrl_her_create: no usable HIP device (count=%d, requested %d); there is no CPU fallback", ndev, cfg->device);
    return nullptr;
  }
  gcrl_her* h = new gcrl_her;
  h->cfg = *cfg;
  h->S = cfg->state_dim; h->A = cfg->action_dim; h->G = cfg->goal_dim;
  h->SA4 = gcrl::round_up(h->S + h->A, 4);
  h->S4 = gcrl::round_up(h->S, 4);
  h->RW = h->SA4 + h->S4 + 2;
  h->RS = gcrl::round_up(h->RW, 16);
  h->RG = gcrl::round_up(h->RW + h->G, 16);
  h->staged.assign(cfg->nenvs, 0);
  h->ag_mirror.assign((size_t)cfg->nenvs * cfg->flush_len * cfg->goal_dim, 0.f);
  if (rng) { h->rng = rng; h->own_rng = false; }
  else { h->rng = gcrl_mt_create(); h->own_rng = true; gcrl_mt_seed(h->rng, cfg->seed); }
  auto ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    gcrl::fail(GCRL_ERR_HIP, "gcrl_her_create: %s failed: %s", what, hipGetErrorString(e));
    return false;
  };
  bool good = ok(hipSetDevice(cfg->device), "hipSetDevice") &&
              ok(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate") &&
              ok(hipMalloc((void**)&h->ring, (size_t)cfg->capacity * h->RS * sizeof(float)), "hipMalloc(ring)") &&
              ok(hipMalloc((void**)&h->stage, (size_t)cfg->nenvs * cfg->flush_len * h->RG * sizeof(float)), "hipMalloc(stage)");
  for (int i = 0; good && i < gcrl_her::kSlots; ++i) {
    good = ok(hipEventCreateWithFlags(&h->slot_ev[i], hipEventDisableTiming), "hipEventCreate") &&
           ok(hipEventCreateWithFlags(&h->epi_ev[i], hipEventDisableTiming), "hipEventCreate") &&
           ok(hipEventCreateWithFlags(&h->fut_ev[i], hipEventDisableTiming), "hipEventCreate") &&
           ok(hipHostMalloc((void**)&h->epi_pinned[i], (size_t)cfg->flush_len * h->RG * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
  }
  if (good) good = ok(hipMemset(h->stage, 0, (size_t)cfg->nenvs * cfg->flush_len * h->RG * sizeof(float)), "hipMemset") &&
                   ok(hipDeviceSynchronize(), "hipDeviceSynchronize");   // the first stage kernel runs on the CALLER's stream
  if (!good) { gcrl_her_destroy(h); return nullptr; }
  return h;
}

void gcrl_her_destroy(gcrl_her* h) {
  if (!h) return;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (int i = 0; i < gcrl_her::kSlots; ++i) {
    if (h->idx_pinned[i]) (void)hipHostFree(h->idx_pinned[i]);
    if (h->epi_pinned[i]) (void)hipHostFree(h->epi_pinned[i]);
    if (h->slot_ev[i]) (void)hipEventDestroy(h->slot_ev[i]);
    if (h->epi_ev[i]) (void)hipEventDestroy(h->epi_ev[i]);
  }
  for (hipEvent_t e : h->prof_a) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->prof_b) (void)hipEventDestroy(e);
  if (h->pay_dev) (void)hipFree(h->pay_dev);
  if (h->fut_dev) (void)hipFree(h->fut_dev);
  if (h->rew_dev) (void)hipFree(h->rew_dev);
  for (int i = 0; i < gcrl_her::kSlots; ++i) {
    if (h->rew_pinned[i]) (void)hipHostFree(h->rew_pinned[i]);
    if (h->rew_ev[i]) (void)hipEventDestroy(h->rew_ev[i]);
  }
  if (h->ps_dev) (void)hipFree(h->ps_dev);
  for (int i = 0; i < gcrl_her::kSlots; ++i) if (h->ps_pinned[i]) (void)hipHostFree(h->ps_pinned[i]);
  for (int i = 0; i < gcrl_her::kSlots; ++i) {
    if (h->fut_pinned[i]) (void)hipHostFree(h->fut_pinned[i]);
    if (h->fut_ev[i]) (void)hipEventDestroy(h->fut_ev[i]);
  }
  for (int i = 0; i < gcrl_her::kSlots; ++i) if (h->pay_pinned[i]) (void)hipHostFree(h->pay_pinned[i]);
  if (h->idx_dev) (void)hipFree(h->idx_dev);
  if (h->ring) (void)hipFree(h->ring);
  if (h->stage) (void)hipFree(h->stage);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->own_rng) gcrl_mt_destroy(h->rng);
  delete h;
}

int gcrl_her_set_reward_callback(gcrl_her* h, gcrl_reward_fn fn, void* user) {
  GCRL_CHECK_ARG(h, "gcrl_her_set_reward_callback: null handle");
  GCRL_CHECK_ARG(h->cfg.reward_kind == GCRL_REWARD_HOST || !fn, "gcrl_her_set_reward_callback: the ring was created with a built-in reward kind");
  h->reward_cb = fn;
  h->reward_cb_user = user;
  return GCRL_OK;
}

int64_t gcrl_her_len(const gcrl_her* h) { return h ? h->len : 0; }
int64_t gcrl_her_head(const gcrl_her* h) { return h ? h->head : 0; }
int32_t gcrl_her_staged(const gcrl_her* h, int env) {
  return (h && env >= 0 && env < h->cfg.nenvs) ? h->staged[env] : -1;
}
void* gcrl_her_stream(const gcrl_her* h) { return h ? (void*)h->stream : nullptr; }

int64_t gcrl_her_push(gcrl_her* h, int env, const float* state, int state_on_device,
                      const float* action_host, const float* next_state, int next_on_device,
                      float reward, int done, const float* dg_host, const float* ag_host,
                      void* stream) {
  GCRL_CHECK_ARG(h && state && action_host && next_state && ag_host, "gcrl_her_push: null argument");
  GCRL_CHECK_ARG(env >= 0 && env < h->cfg.nenvs, "gcrl_her_push: env %d out of range", env);
  (void)dg_host;  // the desired goal is the goal slot of `state` itself (src/env.py:177-182)
  if ((!state_on_device || !next_on_device) && h->S > kMaxInline)
    return gcrl::fail(GCRL_ERR_ARG, "gcrl_her_push: host-pointer states support state_dim <= %d", kMaxInline);
  hipStream_t st = h->pick(stream);
  const int t = h->staged[env];
  StageArgs sa;
  sa.dst = h->stage + ((size_t)env * h->cfg.flush_len + t) * h->RG;
  sa.s_dev = state_on_device ? state : nullptr;
  sa.ns_dev = next_on_device ? next_state : nullptr;
  sa.S = h->S; sa.A = h->A; sa.G = h->G; sa.SA4 = h->SA4; sa.S4 = h->S4;
  sa.r = reward;
  sa.d = done ? 1.0f : 0.0f;
  std::memcpy(sa.a, action_host, sizeof(float) * h->A);
  std::memcpy(sa.ag, ag_host, sizeof(float) * h->G);
  if (!state_on_device) std::memcpy(sa.s_inl, state, sizeof(float) * h->S);
  if (!next_on_device) std::memcpy(sa.ns_inl, next_state, sizeof(float) * h->S);
  hipLaunchKernelGGL(her_stage_kernel, dim3(1), dim3(64), 0, st, sa);
  GCRL_HIP(hipGetLastError());
  std::memcpy(&h->ag_mirror[((size_t)env * h->cfg.flush_len + t) * h->G], ag_host, sizeof(float) * h->G);
  h->staged[env] = t + 1;
  if (done || h->staged[env] >= h->cfg.flush_len) {  // src/buffer.py:117
    int64_t rows = 0;
    const int T = h->staged[env];
    if (int rc = launch_flush(h, 1, &env, &T, nullptr, st, &rows)) return rc;
    h->staged[env] = 0;
    return rows;
  }
  return 0;
}

// after a vector-env step's transitions are staged: envs that finish this step flush together, in env order (the
// reference's loop order, src/env.py:192-201), up to kMaxEp episodes per launch.  Returns ring rows appended.
static int64_t finish_vector_step(gcrl_her* h, int env0, int n, const uint8_t* dones_host, hipStream_t st) {
  int envs[kMaxEp], Ts[kMaxEp], cnt = 0;
  int64_t total = 0;
  auto flush_pending = [&]() -> int {
    if (cnt == 0) return GCRL_OK;
    int64_t rows = 0;
    if (int rc = launch_flush(h, cnt, envs, Ts, nullptr, st, &rows)) return rc;
    for (int q = 0; q < cnt; ++q) h->staged[envs[q]] = 0;
    total += rows;
    cnt = 0;
    return GCRL_OK;
  };
  for (int i = 0; i < n; ++i) {
    const int env = env0 + i;
    h->staged[env] += 1;
    if (dones_host[i] || h->staged[env] >= h->cfg.flush_len) {
      if (cnt == kMaxEp) if (int rc = flush_pending()) return rc;
      envs[cnt] = env; Ts[cnt] = h->staged[env]; ++cnt;
    }
  }
  if (int rc = flush_pending()) return rc;
  return total;
}

int64_t gcrl_her_append(gcrl_her* h, const float* state, int state_on_device, const float* action_host, float reward,
                        const float* next_state, int next_on_device, int done, void* stream) {
  GCRL_CHECK_ARG(h && state && action_host && next_state, "gcrl_her_append: null argument");
  if ((!state_on_device || !next_on_device) && h->S > kMaxInline)
    return gcrl::fail(GCRL_ERR_ARG, "gcrl_her_append: host-pointer states support state_dim <= %d", kMaxInline);
  hipStream_t st = h->pick(stream);
  gcrl::RingBook book{h->cfg.capacity, h->head, h->len};
  const int64_t phys = book.tail();
  StageArgs sa;
  sa.dst = h->ring + (size_t)phys * h->RS;     // a ring record IS the leading RW floats of a staging record
  sa.s_dev = state_on_device ? state : nullptr;
  sa.ns_dev = next_on_device ? next_state : nullptr;
  sa.S = h->S; sa.A = h->A; sa.G = 0; sa.SA4 = h->SA4; sa.S4 = h->S4;
  sa.r = reward;
  sa.d = done ? 1.0f : 0.0f;
  std::memcpy(sa.a, action_host, sizeof(float) * h->A);
  if (!state_on_device) std::memcpy(sa.s_inl, state, sizeof(float) * h->S);
  if (!next_on_device) std::memcpy(sa.ns_inl, next_state, sizeof(float) * h->S);
  hipLaunchKernelGGL(her_stage_kernel, dim3(1), dim3(64), 0, st, sa);
  GCRL_HIP(hipGetLastError());
  book.append(1);                              // deque(maxlen): at capacity the oldest row falls off
  h->head = book.head; h->len = book.len;
  h->mutation_epoch++;
  return 1;
}

int64_t gcrl_her_push_batch(gcrl_her* h, int env0, int n, const float* states_dev, int ld_s,
                            const float* actions_host, const float* next_states_dev, int ld_ns,
                            const float* rewards_host, const uint8_t* dones_host,
                            const float* achieved_goals_host, void* stream) {
  GCRL_CHECK_ARG(h && states_dev && actions_host && next_states_dev && rewards_host && dones_host && achieved_goals_host,
                 "gcrl_her_push_batch: null argument");
  GCRL_CHECK_ARG(n >= 1 && env0 >= 0 && env0 + n <= h->cfg.nenvs, "gcrl_her_push_batch: envs [%d, %d) outside [0, %d)", env0, env0 + n, h->cfg.nenvs);
  GCRL_CHECK_ARG(ld_s >= h->S && ld_ns >= h->S, "gcrl_her_push_batch: row stride smaller than state_dim");
  GCRL_CHECK_ARG(3 + h->A + h->G <= kPayW, "gcrl_her_push_batch: action_dim + goal_dim too large for the payload");
  hipStream_t st = h->pick(stream);
  if (!h->pay_dev) {
    GCRL_HIP(hipMalloc((void**)&h->pay_dev, (size_t)h->cfg.nenvs * kPayW * sizeof(float)));
    for (int i = 0; i < gcrl_her::kSlots; ++i)
      GCRL_HIP(hipHostMalloc((void**)&h->pay_pinned[i], (size_t)h->cfg.nenvs * kPayW * sizeof(float), hipHostMallocDefault));
  }
  const int slot = h->next_epi_slot;
  h->next_epi_slot = (slot + 1) % gcrl_her::kSlots;
  GCRL_HIP(hipEventSynchronize(h->epi_ev[slot]));
  float* pay = h->pay_pinned[slot];
  for (int i = 0; i < n; ++i) {
    float* pw = pay + (size_t)i * kPayW;
    const int t = h->staged[env0 + i];
    std::memcpy(&pw[0], &t, sizeof(int));
    pw[1] = rewards_host[i];
    pw[2] = dones_host[i] ? 1.0f : 0.0f;
    std::memcpy(pw + 3, actions_host + (size_t)i * h->A, sizeof(float) * h->A);
    std::memcpy(pw + 3 + h->A, achieved_goals_host + (size_t)i * h->G, sizeof(float) * h->G);
    std::memcpy(&h->ag_mirror[((size_t)(env0 + i) * h->cfg.flush_len + t) * h->G], achieved_goals_host + (size_t)i * h->G, sizeof(float) * h->G);
  }
  GCRL_HIP(hipMemcpyAsync(h->pay_dev, pay, (size_t)n * kPayW * sizeof(float), hipMemcpyHostToDevice, st));
  GCRL_HIP(hipEventRecord(h->epi_ev[slot], st));
  hipLaunchKernelGGL(her_stage_batch_kernel, dim3(n), dim3(64), 0, st, h->stage, h->pay_dev, states_dev, ld_s, next_states_dev,
                     ld_ns, env0, h->cfg.flush_len, h->S, h->A, h->G, h->SA4, h->S4, h->RG);
  GCRL_HIP(hipGetLastError());
  return finish_vector_step(h, env0, n, dones_host, st);
}

int64_t gcrl_her_process_step(gcrl_her* h, gcrl_normalizer* nz_obs, int update_stats, const float* obs_host,
                              const float* next_obs_host, int obs_dim, const float* dg_host, const float* next_dg_host,
                              const float* next_ag_host, const float* actions_host, const float* rewards_host,
                              const uint8_t* dones_host, int env0, int n, void* stream) {
  return gcrl_her_process_step_g(h, nz_obs, update_stats, nullptr, 0, obs_host, next_obs_host, obs_dim, dg_host, next_dg_host, nullptr,
                                 next_ag_host, actions_host, rewards_host, dones_host, env0, n, stream);
}

int64_t gcrl_her_process_step_g(gcrl_her* h, gcrl_normalizer* nz_obs, int update_stats, gcrl_normalizer* nz_dg, int update_goal_stats,
                                const float* obs_host, const float* next_obs_host, int obs_dim, const float* dg_host,
                                const float* next_dg_host, const float* ag_host, const float* next_ag_host, const float* actions_host,
                                const float* rewards_host, const uint8_t* dones_host, int env0, int n, void* stream) {
  GCRL_CHECK_ARG(h && obs_host && next_obs_host && dg_host && next_dg_host && next_ag_host && actions_host && rewards_host && dones_host,
                 "gcrl_her_process_step: null argument");
  GCRL_CHECK_ARG(!nz_dg || (ag_host && gcrl_normalizer_size(nz_dg) == h->G), "gcrl_her_process_step: the goal normaliser needs the state's achieved goals and size goal_dim = %d", h->G);
  GCRL_CHECK_ARG(obs_dim >= 1 && obs_dim <= 128 && obs_dim + h->G == h->S, "gcrl_her_process_step: obs_dim %d + goal_dim %d != state_dim %d (obs_dim <= 128)", obs_dim, h->G, h->S);
  GCRL_CHECK_ARG(n >= 1 && env0 >= 0 && env0 + n <= h->cfg.nenvs, "gcrl_her_process_step: envs [%d, %d) outside [0, %d)", env0, env0 + n, h->cfg.nenvs);
  GCRL_CHECK_ARG(3 + h->A + h->G <= kPayW, "gcrl_her_process_step: action_dim + goal_dim too large for the payload");
  GCRL_CHECK_ARG(!nz_obs || gcrl_normalizer_size(nz_obs) == obs_dim, "gcrl_her_process_step: observation normaliser of size %d for obs_dim %d", gcrl_normalizer_size(nz_obs), obs_dim);
  hipStream_t st = h->pick(stream);
  const int D = obs_dim, G = h->G;
  // ONE upload: raw rows [obs(n*D) | next_obs(n*D) | dg(n*G) | next_dg(n*G)], then the per-env payload [t | r | d | a | ag]
  const size_t raw = (size_t)n * (2 * D + (nz_dg ? 4 : 2) * G), need = raw + (size_t)n * kPayW;
  static_assert(sizeof(ProcArgsInline) <= 4096, "kernel arguments are limited to 4 KB");
  const bool inl = need <= (size_t)kProcInlineFloats && !std::getenv("GCRL_PROC_STAGED");
  if (!inl && need > h->ps_floats) {
    GCRL_HIP(hipDeviceSynchronize());
    if (h->ps_dev) GCRL_HIP(hipFree(h->ps_dev));
    const size_t want = std::max<size_t>(need, (size_t)h->cfg.nenvs * (2 * D + 4 * G + kPayW));
    GCRL_HIP(hipMalloc((void**)&h->ps_dev, want * sizeof(float)));
    for (int i = 0; i < gcrl_her::kSlots; ++i) {
      if (h->ps_pinned[i]) GCRL_HIP(hipHostFree(h->ps_pinned[i]));
      GCRL_HIP(hipHostMalloc((void**)&h->ps_pinned[i], want * sizeof(float), hipHostMallocDefault));
    }
    h->ps_floats = want;
  }
  ProcArgsInline qi;
  int slot = -1;
  if (!inl) {
    slot = h->next_epi_slot;
    h->next_epi_slot = (slot + 1) % gcrl_her::kSlots;
    GCRL_HIP(hipEventSynchronize(h->epi_ev[slot]));
  }
  float* pin = inl ? qi.data : h->ps_pinned[slot];
  std::memcpy(pin, obs_host, sizeof(float) * n * D);
  std::memcpy(pin + (size_t)n * D, next_obs_host, sizeof(float) * n * D);
  std::memcpy(pin + (size_t)2 * n * D, dg_host, sizeof(float) * n * G);
  std::memcpy(pin + (size_t)2 * n * D + (size_t)n * G, next_dg_host, sizeof(float) * n * G);
  if (nz_dg) {
    std::memcpy(pin + (size_t)2 * n * D + (size_t)2 * n * G, ag_host, sizeof(float) * n * G);
    std::memcpy(pin + (size_t)2 * n * D + (size_t)3 * n * G, next_ag_host, sizeof(float) * n * G);
  }
  float* pay = pin + raw;
  for (int i = 0; i < n; ++i) {
    float* pw = pay + (size_t)i * kPayW;
    const int t = h->staged[env0 + i];
    std::memcpy(&pw[0], &t, sizeof(int));
    pw[1] = rewards_host[i];
    pw[2] = dones_host[i] ? 1.0f : 0.0f;
    std::memcpy(pw + 3, actions_host + (size_t)i * h->A, sizeof(float) * h->A);
    std::memcpy(pw + 3 + h->A, next_ag_host + (size_t)i * G, sizeof(float) * G);
    std::memcpy(&h->ag_mirror[((size_t)(env0 + i) * h->cfg.flush_len + t) * G], next_ag_host + (size_t)i * G, sizeof(float) * G);
  }
  GCRL_CHECK_ARG(!(nz_dg && h->cfg.reward_kind == GCRL_REWARD_HOST), "gcrl_her_process_step: a goal normaliser together with a host-callback "
                 "compute_reward is not supported by the fused entry (the callback would need the device-normalised goals): use the separate calls");
  if (!inl) {
    GCRL_HIP(hipMemcpyAsync(h->ps_dev, pin, need * sizeof(float), hipMemcpyHostToDevice, st));
    GCRL_HIP(hipEventRecord(h->epi_ev[slot], st));
  }
  ProcArgs& pa = qi.p;
  std::memset(&pa, 0, sizeof(pa));
  qi.raw_floats = (int)raw;
  pa.stage = h->stage; pa.raw = h->ps_dev; pa.pay = h->ps_dev + raw;
  const double *mean = nullptr, *var = nullptr;
  gcrl::normalizer_view(nz_obs, &mean, &var, &pa.count, &pa.clip, &pa.mode);
  pa.mean = const_cast<double*>(mean); pa.var = const_cast<double*>(var);
  pa.update = (nz_obs && update_stats) ? 1 : 0;
  if (nz_dg) {
    const double *gm = nullptr, *gv = nullptr;
    gcrl::normalizer_view(nz_dg, &gm, &gv, &pa.gcount, &pa.gclip, &pa.gmode);
    pa.gmean = const_cast<double*>(gm); pa.gvar = const_cast<double*>(gv);
    pa.gupdate = update_goal_stats ? 1 : 0;
  }
  pa.n = n; pa.env0 = env0; pa.flush_len = h->cfg.flush_len; pa.D = D; pa.S = h->S; pa.A = h->A; pa.G = G;
  pa.SA4 = h->SA4; pa.S4 = h->S4; pa.RG = h->RG;
  if (inl) hipLaunchKernelGGL(her_process_step_inline_kernel, dim3(1), dim3(256), 0, st, qi);
  else hipLaunchKernelGGL(her_process_step_kernel, dim3(1), dim3(256), 0, st, pa);
  GCRL_HIP(hipGetLastError());
  if (pa.update) gcrl::normalizer_updated(nz_obs);
  if (pa.gupdate) gcrl::normalizer_updated(nz_dg);
  return finish_vector_step(h, env0, n, dones_host, st);
}

int64_t gcrl_her_push_episode(gcrl_her* h, int env, int T, const float* s, const float* a,
                              const float* ns, const float* r, const float* d, const float* ag,
                              const uint8_t* fut, void* stream) {
  GCRL_CHECK_ARG(h && s && a && ns && r && d && ag, "gcrl_her_push_episode: null argument");
  GCRL_CHECK_ARG(env >= 0 && env < h->cfg.nenvs, "gcrl_her_push_episode: env %d out of range", env);
  GCRL_CHECK_ARG(T >= 1 && T <= h->cfg.flush_len, "gcrl_her_push_episode: T=%d not in 1..%d", T, h->cfg.flush_len);
  if (h->staged[env] != 0) return gcrl::fail(GCRL_ERR_STATE, "gcrl_her_push_episode: env %d has %d staged transitions", env, h->staged[env]);
  hipStream_t st = h->pick(stream);
  const int slot = h->next_epi_slot;
  h->next_epi_slot = (slot + 1) % gcrl_her::kSlots;
  GCRL_HIP(hipEventSynchronize(h->epi_ev[slot]));
  float* buf = h->epi_pinned[slot];
  std::memset(buf, 0, (size_t)T * h->RG * sizeof(float));
  for (int t = 0; t < T; ++t) {
    float* rec = buf + (size_t)t * h->RG;
    std::memcpy(rec, s + (size_t)t * h->S, sizeof(float) * h->S);
    std::memcpy(rec + h->S, a + (size_t)t * h->A, sizeof(float) * h->A);
    std::memcpy(rec + h->SA4, ns + (size_t)t * h->S, sizeof(float) * h->S);
    rec[h->SA4 + h->S4] = r[t];
    rec[h->SA4 + h->S4 + 1] = d[t];
    std::memcpy(rec + h->RW, ag + (size_t)t * h->G, sizeof(float) * h->G);
  }
  std::memcpy(&h->ag_mirror[(size_t)env * h->cfg.flush_len * h->G], ag, sizeof(float) * (size_t)T * h->G);
  float* dst = h->stage + ((size_t)env * h->cfg.flush_len) * h->RG;
  GCRL_HIP(hipMemcpyAsync(dst, buf, (size_t)T * h->RG * sizeof(float), hipMemcpyHostToDevice, st));
  GCRL_HIP(hipEventRecord(h->epi_ev[slot], st));
  int64_t rows = 0;
  const uint8_t* futs[1] = {fut};
  if (int rc = launch_flush(h, 1, &env, &T, futs, st, &rows)) return rc;
  return rows;
}

int gcrl_her_sample(gcrl_her* h, int B, int M, const uint32_t* idx_host, float* out_s, int ld_s,
                    float* out_a, int ld_a, float* out_r, float* out_ns, int ld_ns,
                    float* out_d, uint32_t* drawn_idx_host, void* stream) {
  GCRL_CHECK_ARG(h && out_s && out_a && out_r && out_ns && out_d, "gcrl_her_sample: null output");
  GCRL_CHECK_ARG(B >= 1 && M >= 1, "gcrl_her_sample: B and M must be >= 1");
  GCRL_CHECK_ARG(ld_s >= h->S && ld_ns >= h->S && ld_a >= h->A, "gcrl_her_sample: row stride smaller than the row");
  hipStream_t st = h->pick(stream);
  const uint32_t* host_copy = nullptr;
  if (int rc = gcrl::her_upload_indices(h, B, M, idx_host, st, drawn_idx_host ? &host_copy : nullptr)) return rc;
  if (drawn_idx_host) std::memcpy(drawn_idx_host, host_copy, (size_t)B * M * sizeof(uint32_t));
  const long long n = (long long)B * M;
  GatherArgs ga{h->ring, h->idx_on_device ? h->idx_dev : nullptr, h->last_gen, n, h->head, h->cfg.capacity, h->S, h->A, h->SA4, h->S4, h->RS,
                out_s, out_a, out_r, out_ns, out_d, ld_s, ld_a, ld_ns};
  if (int rc = prof_begin(h, st)) return rc;
  if (h->RS <= 64) {
    int blocks = (int)std::min<long long>((n + 63) / 64, 8192);
    hipLaunchKernelGGL(her_gather_kernel, dim3(blocks), dim3(256), 0, st, ga);
  } else {
    constexpr int kUnroll = 2;
    int blocks = (int)std::min<long long>((n + 4 * 4 * kUnroll - 1) / (4 * 4 * kUnroll), 4096);
    hipLaunchKernelGGL(her_gather_wide_kernel<kUnroll>, dim3(blocks), dim3(256), 0, st, ga);
  }
  GCRL_HIP(hipGetLastError());
  return prof_end(h, st, n);
}

int gcrl_her_profile_enable(gcrl_her* h, int on) {
  GCRL_CHECK_ARG(h, "gcrl_her_profile_enable: null handle");
  if (on && h->prof_a.empty()) {
    h->prof_a.resize(kProfPairs);
    h->prof_b.resize(kProfPairs);
    h->prof_pair_rows.assign(kProfPairs, 0);
    for (size_t i = 0; i < kProfPairs; ++i) {
      GCRL_HIP(hipEventCreate(&h->prof_a[i]));
      GCRL_HIP(hipEventCreate(&h->prof_b[i]));
    }
  }
  if (int rc = prof_drain(h)) return rc;
  h->prof = on != 0;
  h->prof_launches = 0; h->prof_rows = 0; h->prof_ms = 0.0; h->prof_class_rows = 0;
  return GCRL_OK;
}

int gcrl_her_profile_read(gcrl_her* h, int64_t* launches, double* total_ms, int64_t* rows, double* device_clock_ms) {
  GCRL_CHECK_ARG(h, "gcrl_her_profile_read: null handle");
  if (int rc = prof_drain(h)) return rc;
  if (launches) *launches = h->prof_launches;
  if (total_ms) *total_ms = h->prof_ms;
  if (rows) *rows = h->prof_rows;
  if (device_clock_ms) *device_clock_ms = 0.0;   // no longer measured (include/gcrl.h)
  return GCRL_OK;
}

// ---- full resume state: ring rows in logical (oldest-first) order, staged partial episodes, counters
namespace {
struct HerStateHeader {
  uint32_t magic, version;
  int32_t S, A, G, nenvs, k_future, flush_len, RS, RG;
  int64_t capacity, len;
  uint64_t episodes_flushed, draws_done, mutation_epoch;
};
constexpr uint32_t kHerMagic = 0x52454847u;   // "GHER"
size_t her_state_bytes(const gcrl_her* h) {
  return sizeof(HerStateHeader) + (size_t)h->cfg.nenvs * sizeof(int32_t) +
         ((size_t)h->cfg.nenvs * h->cfg.flush_len * h->RG + (size_t)h->len * h->RS) * sizeof(float);
}
}  // namespace

int64_t gcrl_her_state_size(const gcrl_her* h) { return h ? (int64_t)her_state_bytes(h) : -1; }

int gcrl_her_save_state(gcrl_her* h, void* dst_host, int64_t n) {
  GCRL_CHECK_ARG(h && dst_host && n == (int64_t)her_state_bytes(h), "gcrl_her_save_state: buffer must be gcrl_her_state_size() bytes");
  GCRL_HIP(hipDeviceSynchronize());
  HerStateHeader hd{kHerMagic, 1, h->S, h->A, h->G, h->cfg.nenvs, h->cfg.k_future, h->cfg.flush_len, h->RS, h->RG,
                    h->cfg.capacity, h->len, h->episodes_flushed, h->draws_done, h->mutation_epoch};
  char* o = (char*)dst_host;
  std::memcpy(o, &hd, sizeof(hd)); o += sizeof(hd);
  for (int e = 0; e < h->cfg.nenvs; ++e) { const int32_t v = h->staged[e]; std::memcpy(o, &v, sizeof(v)); o += sizeof(v); }
  const size_t stage_bytes = (size_t)h->cfg.nenvs * h->cfg.flush_len * h->RG * sizeof(float);
  GCRL_HIP(hipMemcpy(o, h->stage, stage_bytes, hipMemcpyDeviceToHost)); o += stage_bytes;
  // logical order = physical order from `head`, wrapping once
  const int64_t first = std::min<int64_t>(h->len, h->cfg.capacity - h->head);
  GCRL_HIP(hipMemcpy(o, h->ring + (size_t)h->head * h->RS, (size_t)first * h->RS * sizeof(float), hipMemcpyDeviceToHost));
  o += (size_t)first * h->RS * sizeof(float);
  if (h->len > first) GCRL_HIP(hipMemcpy(o, h->ring, (size_t)(h->len - first) * h->RS * sizeof(float), hipMemcpyDeviceToHost));
  return GCRL_OK;
}

int gcrl_her_load_state(gcrl_her* h, const void* src_host, int64_t n) {
  GCRL_CHECK_ARG(h && src_host && n >= (int64_t)sizeof(HerStateHeader), "gcrl_her_load_state: null / short blob");
  HerStateHeader hd;
  std::memcpy(&hd, src_host, sizeof(hd));
  GCRL_CHECK_ARG(hd.magic == kHerMagic && hd.version == 1, "gcrl_her_load_state: not a replay-ring state blob");
  GCRL_CHECK_ARG(hd.S == h->S && hd.A == h->A && hd.G == h->G && hd.nenvs == h->cfg.nenvs && hd.flush_len == h->cfg.flush_len &&
                     hd.RS == h->RS && hd.RG == h->RG && hd.k_future == h->cfg.k_future,
                 "gcrl_her_load_state: the blob was saved by a ring of another shape");
  GCRL_CHECK_ARG(hd.len >= 0 && hd.len <= h->cfg.capacity, "gcrl_her_load_state: %lld saved rows do not fit capacity %lld", (long long)hd.len, (long long)h->cfg.capacity);
  const size_t stage_bytes = (size_t)h->cfg.nenvs * h->cfg.flush_len * h->RG * sizeof(float);
  const size_t want = sizeof(hd) + (size_t)h->cfg.nenvs * sizeof(int32_t) + stage_bytes + (size_t)hd.len * h->RS * sizeof(float);
  GCRL_CHECK_ARG((size_t)n == want, "gcrl_her_load_state: blob size %lld, expected %zu", (long long)n, want);
  GCRL_HIP(hipDeviceSynchronize());
  const char* o = (const char*)src_host + sizeof(hd);
  for (int e = 0; e < h->cfg.nenvs; ++e) { int32_t v; std::memcpy(&v, o, sizeof(v)); o += sizeof(v); h->staged[e] = v; }
  GCRL_HIP(hipMemcpy(h->stage, o, stage_bytes, hipMemcpyHostToDevice)); o += stage_bytes;
  GCRL_HIP(hipMemcpy(h->ring, o, (size_t)hd.len * h->RS * sizeof(float), hipMemcpyHostToDevice));
  h->head = 0; h->len = hd.len;
  h->episodes_flushed = hd.episodes_flushed; h->draws_done = hd.draws_done; h->mutation_epoch = hd.mutation_epoch + 1;
  return GCRL_OK;
}

int gcrl_her_read_rows(gcrl_her* h, int64_t first, int64_t n, float* s, float* a, float* ns,
                       float* r, float* d) {
  GCRL_CHECK_ARG(h, "gcrl_her_read_rows: null handle");
  GCRL_CHECK_ARG(first >= 0 && n >= 0 && first + n <= h->len, "gcrl_her_read_rows: range [%lld,+%lld) outside len %lld", (long long)first, (long long)n, (long long)h->len);
  if (n == 0) return GCRL_OK;
  GCRL_HIP(hipDeviceSynchronize());   // pushes / flushes run on the caller's stream, this copy on the handle's own
  float* tmp = nullptr;
  GCRL_HIP(hipMalloc((void**)&tmp, (size_t)n * h->RS * sizeof(float)));
  long long total = n * h->RS;
  hipLaunchKernelGGL(her_copy_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream,
                     h->ring, (long long)h->head, (long long)h->cfg.capacity, h->RS, (long long)first, (long long)n, tmp);
  std::vector<float> host((size_t)total);
  hipError_t e = hipMemcpyAsync(host.data(), tmp, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(tmp);
  GCRL_HIP(e);
  for (int64_t i = 0; i < n; ++i) {
    const float* rec = host.data() + (size_t)i * h->RS;
    if (s) std::memcpy(s + (size_t)i * h->S, rec, sizeof(float) * h->S);
    if (a) std::memcpy(a + (size_t)i * h->A, rec + h->S, sizeof(float) * h->A);
    if (ns) std::memcpy(ns + (size_t)i * h->S, rec + h->SA4, sizeof(float) * h->S);
    if (r) r[i] = rec[h->SA4 + h->S4];
    if (d) d[i] = rec[h->SA4 + h->S4 + 1];
  }
  return GCRL_OK;
}

}  // extern "C"
