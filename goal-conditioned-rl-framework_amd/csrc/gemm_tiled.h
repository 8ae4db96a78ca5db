// gemm_tiled.h — LDS-tiled variant of the batched fp32-MFMA GEMM for throughput-sized problems
// (more than 1024 tiles of 16x16: batch 2048 x hidden 512 and the like).
//
// A 256-thread workgroup owns a 64x64 output tile; its 4 waves own 32x32 each (2x2 MFMA tiles of
// 16x16x4).  Per 32-wide k-step the two 64x32 operand tiles are fetched from global memory with
// fully coalesced 16-byte loads (along k when the operand is k-contiguous, along the row index
// otherwise — then transposed on the LDS write), kept in registers while the previous step's
// MFMAs run, and stored to LDS as [row][k] with a 4-float pad.  Fragments come back as one
// ds_read_b128 per 16x16x16 chunk (same k-permutation as gemm_mfma.h: element q of the read feeds
// MFMA q).  Compared with per-wave fragment loads straight from global memory (16 rows x 64 B per
// instruction, each operand re-read by every wave that needs it) this cuts global/L2 traffic 2x
// per wave and turns every global access into whole lines.
#pragma once
#include "gemm_mfma.h"

namespace gcrl {

constexpr int kTB = 64;        // block tile
constexpr int kBK = 64;        // k-step: 64 MFMAs (0.85 us) per wave per fetched tile pair, ~ the L2 latency
constexpr int kNV = kTB * kBK / 4 / 256;  // float4 per thread per operand tile
constexpr int kNR = 4 * kNV;              // staging registers per operand
constexpr int kLDT = kBK + 4;  // LDS row stride (floats): 144 B keeps b128 alignment, spreads banks

enum { FETCH_KC = 0, FETCH_RC = 1, FETCH_GEN = 2 };

// Fetch a 64 x 32 operand tile (rows r0.., k k0..) into 8 registers per thread.
// element(row, k) = base[row*rs + k*cs]; rows >= R or k >= K read as 0; `ones_row`: that row := 1.
template <int MODE>
__device__ inline void tile_fetch(float (&reg)[kNR], const float* __restrict__ base, long long rs, long long cs, int r0,
                                  int R, int k0, int K, int ones_row) {
  const int tid = threadIdx.x;
  if (MODE == FETCH_KC) {
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      const int f = tid + 256 * p, r = r0 + f / (kBK / 4), k = k0 + ((f % (kBK / 4)) << 2);
      const float* src = base + (long long)min(r, R - 1) * rs + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < R) {
        if (k + 3 < K) v = *reinterpret_cast<const float4*>(src);
        else { if (k < K) v.x = src[0]; if (k + 1 < K) v.y = src[1]; if (k + 2 < K) v.z = src[2]; }
      }
      if (r == ones_row) { v.x = k < K ? 1.f : 0.f; v.y = k + 1 < K ? 1.f : 0.f; v.z = k + 2 < K ? 1.f : 0.f; v.w = k + 3 < K ? 1.f : 0.f; }
      reg[4 * p] = v.x; reg[4 * p + 1] = v.y; reg[4 * p + 2] = v.z; reg[4 * p + 3] = v.w;
    }
  } else if (MODE == FETCH_RC) {
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      const int f = tid + 256 * p, k = k0 + (f >> 4), r = r0 + ((f & 15) << 2);
      const float* src = base + (long long)min(k, K - 1) * cs + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K) {
        if (r + 3 < R) v = *reinterpret_cast<const float4*>(src);
        else { if (r < R) v.x = src[0]; if (r + 1 < R) v.y = src[1]; if (r + 2 < R) v.z = src[2]; }
        if (r == ones_row) v.x = 1.f;
        if (r + 1 == ones_row) v.y = 1.f;
        if (r + 2 == ones_row) v.z = 1.f;
        if (r + 3 == ones_row) v.w = 1.f;
      }
      reg[4 * p] = v.x; reg[4 * p + 1] = v.y; reg[4 * p + 2] = v.z; reg[4 * p + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int p = 0; p < kNR; ++p) {
      const int e = tid + 256 * p, r = r0 + (e & 63), k = k0 + (e >> 6);
      float v = 0.f;
      if (r < R && k < K) v = (r == ones_row) ? 1.f : base[(long long)r * rs + (long long)k * cs];
      reg[p] = v;
    }
  }
}

template <int MODE>
__device__ inline void tile_store(const float (&reg)[kNR], float* __restrict__ lds) {
  const int tid = threadIdx.x;
  if (MODE == FETCH_KC) {
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      const int f = tid + 256 * p;
      *reinterpret_cast<float4*>(lds + (f / (kBK / 4)) * kLDT + ((f % (kBK / 4)) << 2)) =
          make_float4(reg[4 * p], reg[4 * p + 1], reg[4 * p + 2], reg[4 * p + 3]);
    }
  } else if (MODE == FETCH_RC) {
#pragma unroll
    for (int p = 0; p < kNV; ++p) {
      const int f = tid + 256 * p, k = f >> 4, r = (f & 15) << 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) lds[(r + q) * kLDT + k] = reg[4 * p + q];
    }
  } else {
#pragma unroll
    for (int p = 0; p < kNR; ++p) {
      const int e = tid + 256 * p;
      lds[(e & 63) * kLDT + (e >> 6)] = reg[p];
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

  float ra[kNR], rb[kNR];
  tile_fetch<MA>(ra, A, d.a_rs, d.a_cs, m0, M, 0, K, -1);
  tile_fetch<MB>(rb, Bm, d.b_cs, d.b_rs, n0, N, 0, K, ones_row);
  const int ksteps = (K + kBK - 1) / kBK;
  for (int ks = 0; ks < ksteps; ++ks) {
    __syncthreads();  // the previous step's fragment reads are done
    tile_store<MA>(ra, ldsA);
    tile_store<MB>(rb, ldsB);
    __syncthreads();
    if (ks + 1 < ksteps) {  // next tiles travel while this step's MFMAs run
      tile_fetch<MA>(ra, A, d.a_rs, d.a_cs, m0, M, (ks + 1) * kBK, K, -1);
      tile_fetch<MB>(rb, Bm, d.b_cs, d.b_rs, n0, N, (ks + 1) * kBK, K, ones_row);
    }
#pragma unroll
    for (int kc = 0; kc < kBK / 16; ++kc) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(ldsA + (32 * wm + 16 * i + li) * kLDT + kc * 16 + 4 * lg);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4*>(ldsB + (32 * wn + 16 * j + li) * kLDT + kc * 16 + 4 * lg);
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
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
    }
  }

  const float* __restrict__ bias = d.bias;
  const float* __restrict__ H = d.H + sl * d.h_slot;
  float* __restrict__ C = d.C + sl * d.c_slot;
  const int epi = d.epi, mul = d.mul;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 32 * wm + 16 * i + 4 * lg + r, n = n0 + 32 * wn + 16 * j + li;
        if (m >= M || n >= N) continue;
        float v = acc[i][j][r];
        if (bias) v += bias[n];
        v = act_apply(v, epi);
        if (mul != MUL_NONE) v *= act_deriv(H[(long long)m * d.h_rs + n], mul);
        ss += v * v;
        if (d.ones_col && n == N - 1) d.col_out[m] = v;
        else C[(long long)m * d.c_rs + n] = v;
      }
  if (d.sumsq_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) d.sumsq_out[(long long)t * 4 + wave] = ss;
  }
}

__global__ __launch_bounds__(256) void gemm_tiled_kernel(GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float ldsA[kTB * kLDT];
  __shared__ __attribute__((aligned(16))) float ldsB[kTB * kLDT];
  const int tile = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxProb; ++q)
    if (q < gb.n && tile >= gb.d[q].tile0) pi = q;
  const GemmDesc& d = gb.d[pi];
  const int t = tile - d.tile0;
  if (t >= d.ntiles) return;
  // fetch modes are per problem and wave-uniform: 3 x 3 straight-line instances
  const int ma = d.a_vec ? FETCH_KC : (d.a_rvec ? FETCH_RC : FETCH_GEN);
  const int mb = d.b_vec ? FETCH_KC : (d.b_rvec ? FETCH_RC : FETCH_GEN);
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
