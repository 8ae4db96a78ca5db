// gemm_tiled.h — LDS-tiled variant of the batched fp32-MFMA GEMM for throughput-sized problems
// (more than 1024 tiles of 16x16: batch 2048 x hidden 512 and the like).
//
// A 256-thread workgroup owns a 64x64 output tile; its 4 waves own 32x32 each (2x2 MFMA tiles of
// 16x16x4).  Per 32-wide k-step the two 64x32 operand tiles are fetched from global memory with
// fully coalesced 16-byte loads (along k when the operand is k-contiguous, along the row index
// otherwise — then transposed on the LDS write), held in registers for two k-steps while earlier
// steps' MFMAs run, and stored to one of two LDS buffers as [row][k] with a 4-float pad (one barrier
// per step).  Fragments come back as one ds_read_b128 per 16x16x16 chunk (same k-permutation as
// gemm_mfma.h: element q of the read feeds MFMA q); the MFMAs accumulate the TRANSPOSED tile so the
// epilogue moves 16 bytes per lane.
//
// Round-2 measurements (M=10240, N=K=512 = the 1280 tiles of TQC's 5-critic launches; the library's plain
// fp32 GEMM on the same operands: 94-97 TFLOP/s; results bitwise equal to it):
//   round 1 (k-step 64, per-step address arithmetic, branch per load, 2 barriers, 4-byte epilogue)   63 / 57 TFLOP/s (fwd / dX)
//   branch-free fetch, offsets planned once, 2 register stages + 2 LDS buffers                      73 / 71
//   + transposed accumulators, 16-byte epilogue                                                      78 / 72
//   + k-step 32 (80 VGPRs, 36 KB LDS: 4 workgroups per CU instead of 2, all 5 tiles of a CU resident)  88 / 80
//   (k-step 16: 86 / 89, TQC step equal, the quantile variant 3 % slower)
#pragma once
#include "gemm_mfma.h"

namespace gcrl {

constexpr int kTB = 64;        // block tile
constexpr int kBK = 32;        // k-step: 32 MFMAs per wave between barriers; small enough for 4 workgroups per CU (see above)
constexpr int kNV = kTB * kBK / 4 / 256;  // float4 per thread per operand tile
constexpr int kNR = 4 * kNV;              // staging registers per operand
// LDS image of an operand tile: [64 rows][32 k], NO padding; the 16-byte chunk c (= k >> 2) of row r lives at chunk position
// c ^ (r & 7) of its row (XOR swizzle).  Conflict-free for everything that touches it: the fragment reads (one ds_read_b128
// per lane: a b128 lane group of 16 lanes {li = 0-3, 12-15 | li = 4-11 of the next k-quarter} lands on 16 distinct 16-byte
// bank groups), the k-contiguous stores (8 lanes = one row's 8 chunks) and the transposing stores of the row-contiguous
// mode (tile_store: 32 lanes = 4 k x 8 rows with 8 distinct r & 7 -> 32 banks).  Round 2 padded rows to 36 floats instead:
// 36 KB per workgroup = FOUR workgroups per CU, so a launch of 1280 tiles (TQC's five critics at batch 2048: exactly five
// tiles per CU) ran four tiles per CU and then the fifth alone, one wave per SIMD; the transposing stores were 4-way bank
// conflicts (profiles/r02_gemm_tiled_sq_counters.txt).  32 KB = five resident workgroups per CU.
constexpr int kLDT = kBK;
constexpr int kRegStages = 1;   // register stages of the global -> LDS pipeline (2: round 2's form; see gemm_tiled_body)
__device__ inline int lds_at(int row, int k) { return row * kLDT + ((((k >> 2) ^ (row & 7)) << 2) | (k & 3)); }

enum { FETCH_KC = 0, FETCH_RC = 1, FETCH_GEN = 2 };

// Fetch a 64 x 64 operand tile (rows r0.., k k0..) into 16 registers per thread.
// element(row, k) = base[row*rs + k*cs]; rows >= R or k >= K read as 0; `ones_row`: that row := 1.
//
// The two vector modes are free of control flow and of per-step address arithmetic: a thread's four fragment offsets
// (bytes, at k-step 0) are computed ONCE (`FetchPlan`; rows outside the operand get an offset past the descriptor's
// extent, for which the hardware returns 0) and a k-step only adds a scalar offset to the buffer load.  Before, every
// step re-derived rows, clamps and 64-bit products and branched around each load: ~1 300 instructions per k-step for
// 64 MFMAs, issue-bound on bookkeeping (64 TFLOP/s at M=10240, N=K=512 against 94 for the library's plain fp32 GEMM;
// a lone workgroup spent 3.1 us per k-step, 0.85 us of it in MFMAs).  The launcher grants FETCH_KC / FETCH_RC only
// when whole 16-byte fragments are legal and K is a multiple of the k-step (no k tail); everything else is FETCH_GEN.
constexpr int kOob = 0x7ffffff0;   // byte offset past any descriptor extent

struct FetchPlan { int voff[kNV]; int step; bool ones[kNV][4]; bool any_one; };

// R: rows of the operand that exist in memory (for B with a synthesised ones row: N - 1)
template <int MODE>
__device__ inline FetchPlan fetch_plan(long long rs, long long cs, int r0, int R, int ones_row) {
  FetchPlan pl;
  const int tid = threadIdx.x;
  pl.any_one = false;
#pragma unroll
  for (int p = 0; p < kNV; ++p) {
    const int f = tid + 256 * p;
    if (MODE == FETCH_KC) {
      const int r = r0 + f / (kBK / 4), k = (f % (kBK / 4)) << 2;
      pl.voff[p] = r < R ? (int)(((long long)r * rs + k) * 4) : kOob;
#pragma unroll
      for (int q = 0; q < 4; ++q) { pl.ones[p][q] = r == ones_row; pl.any_one |= pl.ones[p][q]; }
    } else {
      // row-contiguous operand: a wave-load covers 4 k (lane & 3) x 16 row quads (lane >> 2) — per k a 256-byte run —
      // and the workgroup's 4 waves x kNV passes the 8 k-quads; tile_store transposes through bank-conflict-free dword stores
      const int l = tid & 63, wv = tid >> 6;
      const int k = 4 * (wv + 4 * p) + (l & 3), r = r0 + ((l >> 2) << 2);
      pl.voff[p] = r < R ? (int)(((long long)k * cs + r) * 4) : kOob;
#pragma unroll
      for (int q = 0; q < 4; ++q) { pl.ones[p][q] = r + q == ones_row; pl.any_one |= pl.ones[p][q]; }
    }
  }
  pl.step = MODE == FETCH_KC ? kBK * 4 : (int)(cs * kBK * 4);
  return pl;
}

// `in` false (a k-step past the end, issued to keep the pipeline free of branches): zeros
template <int MODE>
__device__ inline void tile_fetch(float (&reg)[kNR], const float* __restrict__ base, __amdgpu_buffer_rsrc_t rsrc, const FetchPlan& pl,
                                  long long rs, long long cs, int r0, int R, int ks, int K, int ones_row, bool in) {
  if (MODE != FETCH_GEN) {
    const int soff = in ? ks * pl.step : 0;
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? pl.voff[p] : kOob, soff, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) reg[4 * p + q] = __uint_as_float(v[q]);
    }
  } else {
    // element-wise mode (first layers with K = 25, heads with K = 1..4, unaligned operands): still one buffer load per
    // element and no branch — with `if (r < R && k < K) v = base[...]` each of a thread's 16 loads sat behind its own
    // branch and a SINGLE k-step took 50 us (TQC's first-layer and head launches cost as much as the K = 512 ones)
    if (!in) {   // (uniform)
#pragma unroll
      for (int p = 0; p < kNR; ++p) reg[p] = 0.f;
      return;
    }
    const int tid = threadIdx.x, k0 = ks * kBK;
#pragma unroll
    for (int p = 0; p < kNR; ++p) {
      const int e = tid + 256 * p, r = r0 + (e & 63), k = k0 + (e >> 6);
      const bool ok = r < R && k < K && r != ones_row;
      const int off = ok ? (int)(((long long)r * rs + (long long)k * cs) * 4) : kOob;
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
      reg[p] = (r == ones_row && k < K) ? 1.f : v;
    }
  }
}

// the synthesised ones row of a dW problem's B operand, applied when the registers go to LDS (not when they are
// requested: the select would wait for the loads one k-step early).  `in`: the tile lies inside K
template <int MODE>
__device__ inline void tile_ones(float (&reg)[kNR], const FetchPlan& pl, bool in) {
  if (MODE == FETCH_GEN || !pl.any_one) return;
#pragma unroll
  for (int p = 0; p < kNV; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) reg[4 * p + q] = (pl.ones[p][q] && in) ? 1.f : reg[4 * p + q];
}

template <int MODE>
__device__ inline void tile_store(const float (&reg)[kNR], float* __restrict__ lds) {
  const int tid = threadIdx.x;
  if (MODE == FETCH_KC) {
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      const int f = tid + 256 * p, row = f / (kBK / 4), c = f % (kBK / 4);
      *reinterpret_cast<float4*>(lds + row * kLDT + ((c ^ (row & 7)) << 2)) =
          make_float4(reg[4 * p], reg[4 * p + 1], reg[4 * p + 2], reg[4 * p + 3]);
    }
  } else if (MODE == FETCH_RC) {
    // a thread holds rows r..r+3 at ONE k (fetch_plan).  Store instruction t writes row r + ((t + rot) & 3), rot = (lane >> 3) & 3:
    // within 32 lanes (4 k x 8 row quads) the rows then show all 8 values of r & 7 -> 8 swizzled chunks x 4 k = 32 banks.
    const int l = tid & 63, wv = tid >> 6, kq = l & 3, rq = l >> 2, rot = (rq >> 1) & 3;
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      // rotate the four row values left by rot (two conditional-swap stages) so that instruction t finds its value in slot t
      float x0 = reg[4 * p], x1 = reg[4 * p + 1], x2 = reg[4 * p + 2], x3 = reg[4 * p + 3];
      const bool b0 = rot & 1, b1 = rot & 2;
      float y0 = b0 ? x1 : x0, y1 = b0 ? x2 : x1, y2 = b0 ? x3 : x2, y3 = b0 ? x0 : x3;
      const float z[4] = {b1 ? y2 : y0, b1 ? y3 : y1, b1 ? y0 : y2, b1 ? y1 : y3};
      const int c = wv + 4 * p;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = (rq << 2) + ((t + rot) & 3);
        lds[row * kLDT + (((c ^ (row & 7)) << 2) | kq)] = z[t];
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < kNR; ++p) {
      const int e = tid + 256 * p;
      lds[lds_at(e & 63, e >> 6)] = reg[p];
    }
  }
}

template <int MA, int MB>
__device__ inline void gemm_tiled_body(const GemmDesc& d, int t, float* ldsA, float* ldsB) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int M = d.M, N = d.N, K = d.K;
  const int tn = t % d.tiles_n, tm = t / d.tiles_n;
  const int m0 = tm * kTB, n0 = tn * kTB;
  const int wm = wave >> 1, wn = wave & 1;
  const long long sl = d.slot ? (long long)*d.slot : 0;
  const float* __restrict__ A = d.A + sl * d.a_slot;
  const float* __restrict__ Bm = d.B + sl * d.b_slot;
  const int ones_row = d.ones_col ? N - 1 : -1;

  v4f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  // extents for the vector modes: whole rows of the operand (KC: R rows of rs floats; RC: K rows of cs floats)
  const int n_mem = N - (d.ones_col ? 1 : 0);   // B's rows that exist in memory (the ones column is synthesised)
  // (the last element's index + 1: exact for all three modes)
  const __amdgpu_buffer_rsrc_t rsa = wave_uniform_rsrc_n(A, (long long)(M - 1) * d.a_rs + (long long)(K - 1) * d.a_cs + 1);
  const __amdgpu_buffer_rsrc_t rsb = wave_uniform_rsrc_n(Bm, (long long)(n_mem - 1) * d.b_cs + (long long)(K - 1) * d.b_rs + 1);
  // Software pipeline.  Operand tiles come from HBM / the infinity cache the first time they are touched (2-3 us
  // under load), longer than one k-step's MFMAs (64 per wave, ~1.4 us): with a single register stage every step
  // waited out the rest of that latency (a lone workgroup spent 3.1 us per k-step, 1.4 us of it computing).  Two
  // register stages keep the fetch of tile ks+3 in flight across the MFMAs of steps ks+1 and ks+2, and two LDS
  // buffers leave ONE barrier per step: while the waves read buffer ks&1, the tile of step ks+1 is written to the other.
  const int ksteps = (K + kBK - 1) / kBK;
  float ra[kRegStages][kNR], rb[kRegStages][kNR];
  const FetchPlan pla = fetch_plan<MA>(d.a_rs, d.a_cs, m0, M, -1);
  const FetchPlan plb = fetch_plan<MB>(d.b_cs, d.b_rs, n0, n_mem, ones_row);
  auto fetch = [&](float (&xa)[kNR], float (&xb)[kNR], int ks) {
    tile_fetch<MA>(xa, A, rsa, pla, d.a_rs, d.a_cs, m0, M, ks, K, -1, ks < ksteps);
    tile_fetch<MB>(xb, Bm, rsb, plb, d.b_cs, d.b_rs, n0, N, ks, K, ones_row, ks < ksteps);
  };
  auto store = [&](const float (&xa)[kNR], float (&xb)[kNR], int buf, int ks) {
    tile_ones<MB>(xb, plb, ks < ksteps);
    tile_store<MA>(xa, ldsA + buf * (kTB * kLDT));
    tile_store<MB>(xb, ldsB + buf * (kTB * kLDT));
  };
  auto compute = [&](int buf, int ks) {
    const float* la = ldsA + buf * (kTB * kLDT);
    const float* lb = ldsB + buf * (kTB * kLDT);
    const int nk = K - ks * kBK;   // k left from this step on: whole 16-k chunks of zeros are skipped (small-K problems)
#pragma unroll
    for (int kc = 0; kc < kBK / 16; ++kc) {
      if (kc * 16 >= nk) break;
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(la + (32 * wm + 16 * i + li) * kLDT + (((kc * 4 + lg) ^ (li & 7)) << 2));
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4*>(lb + (32 * wn + 16 * j + li) * kLDT + (((kc * 4 + lg) ^ (li & 7)) << 2));
      // q outer: consecutive MFMAs go to the four different accumulators (an accumulator is
      // reused every 128 cycles, above the 40-cycle dependent latency of 16x16x4 f32)
      const float av[2][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}};
      const float bv[2][4] = {{b[0].x, b[0].y, b[0].z, b[0].w}, {b[1].x, b[1].y, b[1].z, b[1].w}};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j][q], av[i][q], acc[i][j], 0, 0, 0);   // roles swapped: see the epilogue
    }
  };
  // one step: LDS[ks&1] holds tile ks; `nx` (registers) holds tile ks+1, requested two steps ago; the other register
  // stage (tile ks+2) stays in flight.  After the MFMAs: tile ks+1 -> the other LDS buffer, then request tile ks+3
  // into the registers just freed.  NOTHING here is conditional: with `if (ks + 3 < ksteps)` around the request the
  // compiler could no longer count the loads in flight and waited for the newest stage before every LDS store
  // (s_waitcnt vmcnt(0): the pipeline was one step deep again).  Steps past the end fetch zeros (tile_fetch `in`),
  // so the step count is simply rounded up to even and the tail costs at most one k-step of MFMAs on zeros.
  auto step = [&](float (&nxa)[kNR], float (&nxb)[kNR], int ks) {
    compute(ks & 1, ks);
    store(nxa, nxb, (ks + 1) & 1, ks + 1);
    fetch(nxa, nxb, ks + 3);
    __syncthreads();
  };
  if (kRegStages == 2) {
    fetch(ra[0], rb[0], 0);
    fetch(ra[kRegStages - 1], rb[kRegStages - 1], 1);
    store(ra[0], rb[0], 0, 0);
    fetch(ra[0], rb[0], 2);
    __syncthreads();
    const int kp = (ksteps + 1) & ~1;
    for (int ks = 0; ks < kp; ks += 2) {
      step(ra[kRegStages - 1], rb[kRegStages - 1], ks);
      step(ra[0], rb[0], ks + 1);
    }
  } else {
    // ONE register stage: tile ks+1 is requested a whole k-step before it goes to LDS.  With five workgroups resident per
    // CU a k-step lasts ~5 x 1024 MFMA cycles of wall time (the SIMD's matrix pipe is shared by five waves), longer than
    // an HBM round trip, and the 16 registers of the second stage are what kept the kernel at four waves per SIMD.
    fetch(ra[0], rb[0], 0);
    store(ra[0], rb[0], 0, 0);
    fetch(ra[0], rb[0], 1);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
      compute(ks & 1, ks);
      store(ra[0], rb[0], (ks + 1) & 1, ks + 1);
      fetch(ra[0], rb[0], ks + 2);
      __syncthreads();
    }
  }

  const float* __restrict__ bias = d.bias;
  const float* __restrict__ H = d.H + sl * d.h_slot;
  float* __restrict__ C = d.C + sl * d.c_slot;
  const int epi = d.epi, mul = d.mul;
  // The MFMAs ran with the operand roles swapped (B fragment as the instruction's A): the accumulator tile is the
  // transpose, i.e. register r of lane (li, lg) is C[m = li][n = 4*lg + r] — four CONSECUTIVE columns of one row per
  // lane, so bias / saved activations / results move as one 16-byte access per lane and 16x16 tile (16 rows x 64 B
  // per wave instruction) instead of four dwords each.
  float ss = 0.f;
  const bool vec_ok = (d.c_rs % 4 == 0) && (((unsigned long long)C & 15) == 0) && !d.ones_col &&
                      (mul == MUL_NONE || (d.h_rs % 4 == 0 && (((unsigned long long)H & 15) == 0))) &&
                      (!bias || (((unsigned long long)bias & 15) == 0));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + 32 * wm + 16 * i + li, nq = n0 + 32 * wn + 16 * j + 4 * lg;
      if (m >= M || nq >= N) continue;
      if (vec_ok && nq + 3 < N) {
        v4f v = acc[i][j];
        if (bias) v += *(const v4f*)(bias + nq);
        v4f h = (v4f){0.f, 0.f, 0.f, 0.f};
        if (mul != MUL_NONE) h = *(const v4f*)(H + (long long)m * d.h_rs + nq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = act_apply(v[r], epi);
          if (mul != MUL_NONE) v[r] *= act_deriv(h[r], mul);
          ss += v[r] * v[r];
        }
        *(v4f*)(C + (long long)m * d.c_rs + nq) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nq + r;
          if (n >= N) continue;
          float v = acc[i][j][r];
          if (bias) v += bias[n];
          v = act_apply(v, epi);
          if (mul != MUL_NONE) v *= act_deriv(H[(long long)m * d.h_rs + n], mul);
          ss += v * v;
          if (d.ones_col && n == N - 1) d.col_out[m] = v;
          else C[(long long)m * d.c_rs + n] = v;
        }
      }
    }
  if (d.sumsq_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) d.sumsq_out[(long long)t * 4 + wave] = ss;
  }
}

__global__ __launch_bounds__(256) void gemm_tiled_kernel(GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float ldsA[2 * kTB * kLDT];   // two buffers each (gemm_tiled_body)
  __shared__ __attribute__((aligned(16))) float ldsB[2 * kTB * kLDT];
  const int tile = xcd_tile_of((int)blockIdx.x, (int)gridDim.x);   // an XCD owns whole tile rows: each A panel enters ONE L2 (gemm_mfma.h)
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxProb; ++q)
    if (q < gb.n && tile >= gb.d[q].tile0) pi = q;
  const GemmDesc& d = gb.d[pi];
  const int t = tile - d.tile0;
  if (t >= d.ntiles) return;
  // fetch modes are per problem and wave-uniform: 3 x 3 straight-line instances
  // (vector modes need whole 16-byte fragments: extents in multiples of 4 along the vector direction)
  const int n_mem = d.N - (d.ones_col ? 1 : 0);
  const bool kfull = d.K % kBK == 0;
  const int ma = (d.a_vec && kfull) ? FETCH_KC : ((d.a_rvec && kfull && d.M % 4 == 0) ? FETCH_RC : FETCH_GEN);
  const int mb = (d.b_vec && kfull) ? FETCH_KC : ((d.b_rvec && kfull && n_mem % 4 == 0) ? FETCH_RC : FETCH_GEN);
  switch (ma * 3 + mb) {
    case 0: gemm_tiled_body<FETCH_KC, FETCH_KC>(d, t, ldsA, ldsB); break;
    case 1: gemm_tiled_body<FETCH_KC, FETCH_RC>(d, t, ldsA, ldsB); break;
    case 2: gemm_tiled_body<FETCH_KC, FETCH_GEN>(d, t, ldsA, ldsB); break;
    case 3: gemm_tiled_body<FETCH_RC, FETCH_KC>(d, t, ldsA, ldsB); break;
    case 4: gemm_tiled_body<FETCH_RC, FETCH_RC>(d, t, ldsA, ldsB); break;
    case 5: gemm_tiled_body<FETCH_RC, FETCH_GEN>(d, t, ldsA, ldsB); break;
    case 6: gemm_tiled_body<FETCH_GEN, FETCH_KC>(d, t, ldsA, ldsB); break;
    case 7: gemm_tiled_body<FETCH_GEN, FETCH_RC>(d, t, ldsA, ldsB); break;
    default: gemm_tiled_body<FETCH_GEN, FETCH_GEN>(d, t, ldsA, ldsB); break;
  }
}

}  // namespace gcrl
