// gemm_mfma.h — batched small-GEMM kernel on the gfx950 fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32 fma chains, 64 FLOP/clk/SIMD).
//
// Every Linear layer of the reference's networks (src/model.py:18,57,63,106,114-115) and its
// backward is one problem of the form  C[M,N] = epilogue(A[M,K] . B[K,N])  with arbitrary
// element strides, so the same kernel serves
//   forward   Y  = act(X W^T + b)            A = X  (k contiguous)   B(k,n) = W[n][k]
//   dX        G' = (G W) * act'(H_prev)      A = G  (k contiguous)   B(k,n) = W[k][n]
//   dW | db   dW = G^T [X | 1]               A(m,k) = G[k][m]        B(k,n) = X[k][n], col N-1 = 1
// Up to kMaxProb independent problems (twin / ensemble critics, target and online nets of the
// same layer depth) share one launch; a wavefront owns a (16*TM)x(16*TN) tile.
//
// k-permutation: lane (i = lane&15, g = lane>>4) loads FOUR consecutive k (one 16-byte load
// when the operand is k-contiguous) at k0+4g..k0+4g+3 and issues four MFMAs, MFMA t consuming
// element t of both fragments; the instruction's own k index (lane>>4) then stands for
// k0+4g+t, and A and B agree on it, which is all the contraction needs.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "kernarg.h"

namespace gcrl {

constexpr int kMaxProb = 12;

enum { EPI_NONE = 0, EPI_LEAKY = 1, EPI_RELU = 2, EPI_TANH = 3 };
enum { MUL_NONE = 0, MUL_DLEAKY = 1, MUL_DRELU = 2, MUL_DTANH = 3 };

struct GemmDesc {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // + bias[n] before the activation (may be null)
  const float* H;     // MUL_*: multiply by act'(H[m*h_rs + n]) (H = saved post-activation)
  float* col_out;     // ones_col: column N-1 of the result goes to col_out[m] (bias gradient)
  long long a_rs, a_cs, b_rs, b_cs, c_rs, h_rs;
  int M, N, K;
  int epi, mul;
  int ones_col;  // B(k, N-1) := 1
  // batch-slot indirection: the step's kernels are replayed from a hipGraph with frozen
  // pointers, so operands that live in the pre-gathered batch array are addressed as
  // base + (*slot) * stride, `slot` pointing at the current step's batch_slot on the device
  const int* slot;
  long long a_slot, b_slot, c_slot, h_slot;
  // dW problems: sum of squares of this problem's outputs, one partial per finishing wave, at
  // sumsq_out[(tile * KSPLIT) + wave]: the global-norm clip needs ||g||, and producing the
  // partials here (fixed slots, fixed order -> deterministic) removes a reduction launch
  float* sumsq_out;
  // BatchNorm statistics of the result (launcher-checked: the KSPLIT = 4 form — 16-row tiles — or an unsplit LDS-tiled problem —
  // 64-row tiles): per-column (mean, M2) of each row tile by a local two-pass, at bn_part[tm*N + n] and
  // bn_part[(tiles_m + tm)*N + n] — the row-block partials bn_relu_apply merges (ops_sac.hip), so the producing GEMM replaces
  // the bn_stats launch
  float* bn_part;
  int a_vec, b_vec;  // 16-byte loads legal along k (filled by the launcher)
  int a_rvec, b_rvec;  // 16-byte loads legal along the row index (operand stored k-major)
  int tile0, tiles_n, ntiles;  // filled by the launcher
  // 0: the tile form follows from the problem's shape (gemm_shape_of).  Otherwise the caller's choice (1..4) — for problems
  // it ALWAYS launches in the same company (e.g. the dW problems of a whole critic ensemble in one launch: alone, a long
  // reduction into a 512 x 513 output is 72 tiles of 64x64 and wants the k-split form; twenty of them fill the chip with
  // LDS-tiled workgroups).  Still a function of the problem and the agent's configuration only, never of a launch's
  // accidental composition: a problem's summation order does not change from step to step.
  int shape_hint;
  // LDS-tiled form only: the reduction split over `ksplit` workgroups per 64x64 tile (0 / 1: none).  Every split writes its raw
  // partial tile to kpart[(tile * ksplit + s) * kTiledPartStride], takes a ticket (kticket[tile], zero between launches: the
  // last arriver resets it) and the LAST one sums the partials in index order — so the result does not depend on which
  // workgroup finishes last — and runs the epilogue.  For long reductions into few tiles (dW at batch >= 1024).
  int ksplit;
  float* kpart;
  unsigned int* kticket;
};
constexpr int kTiledPartStride = 64 * 64 + 64;   // floats per partial: the tile + the bias-gradient row sums
constexpr int kTicketStride = 32;                // tickets 128 bytes apart

struct GemmBatch {
  int n;
  int tile0[kMaxProb];   // d[q].tile0 once more, next to n: the problem lookup's ONE scalar load (filled by the launchers)
  int pad[3];
  GemmDesc d[kMaxProb];
};

// the workgroup's / wave's problem: the last one whose first tile is not beyond `tile` (branch-free over the header's table)
__device__ __forceinline__ int gemm_problem_of(const GemmBatch& gb, int tile) {
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxProb; ++q) pi += (q < gb.n && tile >= gb.tile0[q]) ? 1 : 0;
  return pi;
}
// every field the tile bodies read in front of (or at) their first operand request, in registers NOW (kernarg.h)
__device__ __forceinline__ void gemm_pin(const GemmDesc& d) {
  // ONE asm statement: volatile asms are not reordered among themselves, so a pin per field would be a round trip per field
  asm volatile("" ::"s"(d.A), "s"(d.B), "s"(d.C), "s"(d.bias), "s"(d.H), "s"(d.col_out), "s"(d.a_rs), "s"(d.a_cs), "s"(d.b_rs), "s"(d.b_cs),
               "s"(d.c_rs), "s"(d.h_rs), "s"(d.M), "s"(d.N), "s"(d.K), "s"(d.epi), "s"(d.mul), "s"(d.ones_col), "s"(d.slot), "s"(d.a_slot),
               "s"(d.b_slot), "s"(d.c_slot), "s"(d.h_slot), "s"(d.a_vec), "s"(d.b_vec), "s"(d.tile0), "s"(d.tiles_n), "s"(d.ntiles));
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// 16-byte fragment loads go through a buffer descriptor (raw_buffer_load_b128): a plain
// `*(float4*)p` next to the strided scalar path was if-converted by hipcc into four dword
// loads with a selected stride (0 global_load_dwordx4 in the ISA, 2-4x slower).
__device__ inline __amdgpu_buffer_rsrc_t wave_uniform_rsrc(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7fffffff, 0x00020000);
}

// the same with an extent in floats: loads at or past it return 0
__device__ inline __amdgpu_buffer_rsrc_t wave_uniform_rsrc_n(const float* p, long long nfloats) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  const long long bytes = nfloats * 4;
  const int nb = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffe0LL ? 0x7fffffe0LL : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, nb, 0x00020000);
}

__device__ inline float act_apply(float v, int epi) {
  switch (epi) {
    case EPI_LEAKY: return v > 0.f ? v : 0.01f * v;  // nn.LeakyReLU() default slope
    case EPI_RELU: return v > 0.f ? v : 0.f;
    case EPI_TANH: return (float)tanh((double)v);  // rounded once: only B x ac_dim outputs take this path
    default: return v;
  }
}
__device__ inline float act_deriv(float h, int mul) {
  switch (mul) {
    case MUL_DLEAKY: return h > 0.f ? 1.f : 0.01f;
    case MUL_DRELU: return h > 0.f ? 1.f : 0.f;
    case MUL_DTANH: return 1.f - h * h;
    default: return 1.f;
  }
}

// XCD-aware workgroup -> tile mapping.  Workgroups are handed to the 8 XCDs round-robin (workgroup w runs on XCD w % 8) and
// every XCD has its own L2, so with tiles taken in launch order the 8 neighbours that share an operand panel (same tile row:
// the same A rows) sit on 8 different XCDs and the panel is fetched from the infinity cache / HBM once PER XCD — measured in
// round 2 as an 8x over-fetch (TD3's dW launch: 170 MB HBM-side for ~20 MB of operands, profiles/r02_pmc_traffic_td3_*).
// Here XCD x takes the contiguous tile range [x * G/8, (x+1) * G/8): whole tile rows, usually whole problems of a batched
// launch, live in ONE L2.  A wrong guess about the placement costs speed, never correctness; the arithmetic of a tile does
// not depend on which workgroup computes it.
__device__ inline int xcd_tile_of(int w, int G) {
  const int per = G >> 3;
  return w < (per << 3) ? (w & 7) * per + (w >> 3) : w;
}

// One (16*TM)x(16*TN) tile `t` of problem `d` by the calling wave (KSPLIT = 1) / workgroup (KSPLIT = 4): the whole body of
// gemm_batch_kernel up to and including the result stores and the BatchNorm partials.  Returns false for waves / workgroups
// beyond the problem's tiles (KSPLIT = 4: after the workgroup barrier of the partial-sum exchange, which every wave reaches).
// ss_out: this lane's sum of squares of the values it stored; x_out (KSPLIT = 4 only): the finished value of this lane's element
// (row m0 + 4*(lane>>4) + wave, column n0 + (lane&15) of the tile), 0 outside the problem.  The fused dW + optimiser launch
// (dw_adam.hip) calls the SAME body, so a gradient element is the same bits whichever launch form produced it.
template <int TM, int TN, int KSPLIT>
__device__ __forceinline__ bool gemm_batch_tile(const GemmDesc& d, const int t, float& x_out, float& ss_out) {
  static_assert(KSPLIT == 1 || (TM == 1 && TN == 1), "k-split only for single 16x16 tiles");
  constexpr int NACC = (TM * TN >= 4) ? 1 : (TM * TN == 2 ? 2 : 4);
  // wave-tile forms: the MFMAs run with the operand roles swapped, so the accumulator holds the TRANSPOSED tile —
  // register r of lane (li, lg) is C[m = li][n = 4*lg + r], four consecutive columns of one row: bias, saved
  // activations and results move as 16 bytes per lane (gemm_tiled.h does the same).  The k-split form keeps the
  // plain layout (its waves exchange partial tiles through LDS and finish one element per lane).
  constexpr bool TR = KSPLIT == 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool active = t < d.ntiles;
  x_out = 0.f; ss_out = 0.f;
  if (KSPLIT == 1 && !active) return false;

  const int M = d.M, N = d.N, K = d.K;
  const int tn = active ? t % d.tiles_n : 0, tm = active ? t / d.tiles_n : 0;
  const int m0 = tm * 16 * TM, n0 = tn * 16 * TN;
  const int li = lane & 15, lg = lane >> 4;
  const long long sl = d.slot ? (long long)*d.slot : 0;
  const float* __restrict__ A = d.A + sl * d.a_slot;
  const float* __restrict__ Bm = d.B + sl * d.b_slot;
  const long long a_rs = d.a_rs, a_cs = d.a_cs, b_rs = d.b_rs, b_cs = d.b_cs;
  const bool a_vec = d.a_vec, b_vec = d.b_vec;
  const int ones_col = d.ones_col;

  const int kchunks = (K + 15) >> 4;
  int kc_beg = 0, kc_end = kchunks;
  if (KSPLIT == 4) {
    const int per = (kchunks + 3) >> 2;
    kc_beg = wave * per;
    kc_end = min(kchunks, kc_beg + per);
  }

  v4f acc[TM][TN][NACC];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[i][j][q] = (v4f){0.f, 0.f, 0.f, 0.f};

  // row / column bases of this lane's fragments (clamped: out-of-range rows are computed on
  // valid memory and dropped at the store)
  const float* ap[TM];
  const float* bp[TN];
  bool bone[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) ap[i] = A + (long long)min(m0 + 16 * i + li, M - 1) * a_rs;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + 16 * j + li;
    bone[j] = ones_col && (n == N - 1);
    const int nmax = ones_col ? N - 2 : N - 1;
    bp[j] = Bm + (long long)min(n, nmax) * b_cs;
  }

  auto mfma_chunk = [&](const float (&a)[TM][4], const float (&b)[TN][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j][q % NACC] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][q], a[i][q], acc[i][j][q % NACC], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], b[j][q], acc[i][j][q % NACC], 0, 0, 0);
  };

  // byte offsets of the fragment rows for the 16-byte path (operands are < 2 GiB)
  const __amdgpu_buffer_rsrc_t rs_a = wave_uniform_rsrc(A);
  const __amdgpu_buffer_rsrc_t rs_b = wave_uniform_rsrc(Bm);
  int aoff[TM], boff[TN];
  const int acs4 = (int)(a_cs * 4), brs4 = (int)(b_rs * 4);   // byte strides along k (operands are < 2 GiB)
#pragma unroll
  for (int i = 0; i < TM; ++i) aoff[i] = (int)((ap[i] - A) * 4);
#pragma unroll
  for (int j = 0; j < TN; ++j) boff[j] = (int)((bp[j] - Bm) * 4);
  // The ones column (B(k, N-1) := 1, the bias gradient of a dW | db problem) without a select on the loaded value: its lanes
  // request an offset past the descriptor's range (the hardware returns 0, no traffic) and OR the bits of 1.0f into what comes
  // back.  `ones ? 1 : loaded` invited the compiler to branch around the loads — and, in the fused optimiser launch (round 5), to
  // drain every load in flight (s_waitcnt vmcnt(0)) at each such branch: one memory round trip per 16-k chunk.
  unsigned int bor[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bor[j] = bone[j] ? 0x3f800000u : 0u;
    if (bone[j]) boff[j] = (int)0x80000000u;   // (+ any k offset below 2 GiB stays at or above the range of every descriptor used here)
  }

  // epilogue operands (bias, saved activations) are fetched BEFORE the k-loop so their latency
  // overlaps the fragment loads instead of adding a memory round trip after the last MFMA
  const float* __restrict__ bias = d.bias;
  const float* __restrict__ H = d.H + sl * d.h_slot;
  const int mul = d.mul;
  constexpr int NE = (KSPLIT == 4) ? 1 : 4;
  float pre_b[TN][NE], pre_h[TM][TN][NE];
  const bool epi_vec = TR && (d.c_rs % 4 == 0) && (((unsigned long long)(d.C + sl * d.c_slot)) & 15) == 0 && !ones_col &&
                       (mul == MUL_NONE || (d.h_rs % 4 == 0 && (((unsigned long long)H) & 15) == 0)) &&
                       (!bias || (((unsigned long long)bias) & 15) == 0) && (N % 4 == 0);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if constexpr (TR) {
      const int nq = n0 + 16 * j + 4 * lg;
      if (epi_vec) {
        const v4f bv = (bias && active && nq < N) ? *(const v4f*)(bias + nq) : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) pre_b[j][r] = bv[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) pre_b[j][r] = (bias && active && nq + r < N) ? bias[nq + r] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + 16 * i + li;
        if (epi_vec) {
          const v4f hv = (mul != MUL_NONE && active && m < M && nq < N) ? *(const v4f*)(H + (long long)m * d.h_rs + nq) : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) pre_h[i][j][r] = hv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pre_h[i][j][r] = (mul != MUL_NONE && active && m < M && nq + r < N) ? H[(long long)m * d.h_rs + nq + r] : 0.f;
        }
      }
    } else {
      const int n = n0 + 16 * j + li;
      pre_b[j][0] = (bias && active && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + 4 * lg + wave;
        pre_h[i][j][0] = (mul != MUL_NONE && active && m < M && n < N) ? H[(long long)m * d.h_rs + n] : 0.f;
      }
    }
  }

  if (active) {
    // Interior chunks (all 16 k inside K) go in groups of U: the group's 2*U fragment loads are
    // unconditional (addresses clamped, surplus results unused), free of control flow and
    // independent of any MFMA, so they are all in flight before the first MFMA waits: one
    // memory round trip per 64 k.  The 16-byte / strided choice is made ONCE per wave (four
    // straight-line instances of the loop): a branch per chunk made hipcc drain vmcnt(0) at
    // every join and the loop ran at one L2 latency (0.37 us) per chunk.
    const int kfull = K >> 4;
    const int kint_end = min(kc_end, kfull);
    auto interior = [&](auto va_tag, auto vb_tag) {
      constexpr bool VA = decltype(va_tag)::value, VB = decltype(vb_tag)::value;
      constexpr int U = 4;
      for (int kc = kc_beg; kc < kint_end; kc += U) {
        float a[U][TM][4], b[U][TN][4];
        const int nvalid = min(U, kint_end - kc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int kb = (min(kc + u, kint_end - 1) << 4) + (lg << 2);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if constexpr (VA) {
              const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff[i] + kb * 4, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) a[u][i][q] = __uint_as_float(v[q]);
            } else {   // (32-bit byte offsets into the descriptor: 64-bit pointer arithmetic per element was measurable, see bn_slab.hip)
              const int o = aoff[i] + kb * acs4;
#pragma unroll
              for (int q = 0; q < 4; ++q) a[u][i][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, o + q * acs4, 0, 0));
            }
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (VB) {
              const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, boff[j] + kb * 4, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) b[u][j][q] = __uint_as_float(v[q]);
            } else {
              const int o = boff[j] + kb * brs4;
#pragma unroll
              for (int q = 0; q < 4; ++q) b[u][j][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_b, o + q * brs4, 0, 0));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) b[u][j][q] = __uint_as_float(__float_as_uint(b[u][j][q]) | bor[j]);
          }
        }
        // (every load of the group is requested before its first MFMA: in the fused optimiser launch the scheduler sank the last
        // loads of a chunk below the first MFMAs and waited for each of them on its own — four round trips instead of one)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (u < nvalid) mfma_chunk(a[u], b[u]);
      }
    };
    using T1 = std::integral_constant<bool, true>;
    using T0 = std::integral_constant<bool, false>;
    if (a_vec) { if (b_vec) interior(T1{}, T1{}); else interior(T1{}, T0{}); }
    else       { if (b_vec) interior(T0{}, T1{}); else interior(T0{}, T0{}); }
    if (kc_beg <= kfull && kfull < kc_end) {
      // the one partial chunk at the end of K (first layers: K = obs + ac dims): element loads through descriptors
      // with the operands' extents, k >= K pushed past them (the hardware returns 0) — no branch per element.  With
      // `(kb + q < K) ? p[..] : 0` each load sat in its own predicated block and a K = 25 problem spent most of
      // its time here.
      constexpr int kPast = 0x7ffffff0;
      const __amdgpu_buffer_rsrc_t ta = wave_uniform_rsrc_n(A, (long long)(M - 1) * a_rs + (long long)(K - 1) * a_cs + 1);
      const int n_mem = ones_col ? N - 1 : N;
      const __amdgpu_buffer_rsrc_t tb = wave_uniform_rsrc_n(Bm, (long long)(n_mem - 1) * b_cs + (long long)(K - 1) * b_rs + 1);
      const int kb = (kfull << 4) + (lg << 2);
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int off = (kb + q < K) ? aoff[i] + (int)((long long)(kb + q) * a_cs * 4) : kPast;
          a[i][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ta, off, 0, 0));
        }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int off = (kb + q < K && !bone[j]) ? boff[j] + (int)((long long)(kb + q) * b_rs * 4) : kPast;
          const unsigned int v = __builtin_amdgcn_raw_buffer_load_b32(tb, off, 0, 0);
          b[j][q] = __uint_as_float(v | ((kb + q < K) ? bor[j] : 0u));
        }
      mfma_chunk(a, b);
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 1; q < NACC; ++q) acc[i][j][0] += acc[i][j][q];

  float* __restrict__ C = d.C + sl * d.c_slot;
  const int epi = d.epi;
  float ss = 0.f;

  auto finish = [&](float v, int m, int n, float b, float h) -> float {
    if (m >= M || n >= N) return 0.f;
    v += b;
    v = act_apply(v, epi);
    if (mul != MUL_NONE) v *= act_deriv(h, mul);
    ss += v * v;
    if (ones_col && n == N - 1) d.col_out[m] = v;
    else C[(long long)m * d.c_rs + n] = v;
    return v;
  };

  if (KSPLIT == 4) {
    // four waves hold k-partials of the same 16x16 tile: exchange through LDS, wave w
    // finishes accumulator register w (row 4*(lane>>4)+w of the tile)
    __shared__ float red[4][4][64];
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[0][0][0][r];
    __syncthreads();
    if (!active) return false;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[w][wave][lane];
    const float x = finish(v, m0 + 4 * lg + wave, n0 + li, pre_b[0][0], pre_h[0][0][0]);
    x_out = x;
    if (d.bn_part) {   // (uniform per workgroup: one problem, one tile)
      __shared__ float cs[2][4][16];
      const int rows = min(16, M - m0), n = n0 + li;
      float s = x;
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lg == 0) cs[0][wave][li] = s;
      __syncthreads();
      const float mean = (cs[0][0][li] + cs[0][1][li] + cs[0][2][li] + cs[0][3][li]) / (float)rows;
      const float df = (m0 + 4 * lg + wave < M) ? x - mean : 0.f;
      float q = df * df;
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (lg == 0) cs[1][wave][li] = q;
      __syncthreads();
      if (wave == 0 && lg == 0 && n < N) {
        const int tiles_m = (M + 15) >> 4;
        d.bn_part[(long long)tm * N + n] = mean;
        d.bn_part[(long long)(tiles_m + tm) * N + n] = cs[1][0][li] + cs[1][1][li] + cs[1][2][li] + cs[1][3][li];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = m0 + 16 * i + li, nq = n0 + 16 * j + 4 * lg;
        if (epi_vec) {
          if (m < M && nq < N) {   // (N % 4 == 0: the whole quad is inside)
            v4f v = acc[i][j][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = v[r] + pre_b[j][r];
              x = act_apply(x, epi);
              if (mul != MUL_NONE) x *= act_deriv(pre_h[i][j][r], mul);
              ss += x * x;
              v[r] = x;
            }
            *(v4f*)(C + (long long)m * d.c_rs + nq) = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) finish(acc[i][j][0][r], m, nq + r, pre_b[j][r], pre_h[i][j][r]);
        }
      }
  }
  ss_out = ss;
  return true;
}

template <int TM, int TN, int KSPLIT>
__global__ __launch_bounds__(256) void gemm_batch_kernel(GemmBatch gb) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_tile_of((int)blockIdx.x, (int)gridDim.x);
  const int wtile = (KSPLIT == 4) ? bid : bid * 4 + wave;
  const int pi = gemm_problem_of(gb, wtile);
  const GemmDesc& d = gb.d[pi];
  gemm_pin(d);
  const int t = wtile - d.tile0;
  float x, ss;
  if (!gemm_batch_tile<TM, TN, KSPLIT>(d, t, x, ss)) return;
  if (d.sumsq_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) d.sumsq_out[(long long)t * KSPLIT + (KSPLIT == 4 ? wave : 0)] = ss;
  }
}

// 1: one 16x16 tile per workgroup with K split over its waves (the form that can produce bn_part), 2-4: larger forms
int gemm_shape_of(const GemmDesc& d);

// Launch `n` problems in one grid.  shape: 0 = auto.
int launch_gemm_batch(hipStream_t st, GemmDesc* descs, int n, int shape = 0);
// What launch_gemm_batch fills in for a problem of the k-split 16x16 form (vector-load flags, tile bookkeeping with tile0 = 0), for
// kernels that call gemm_batch_tile<1, 1, 4> themselves (dw_adam.hip, ops_sac.hip heads_sample_kernel).  Returns the tile count.
int gemm_prepare_ksplit(GemmDesc& d);

}  // namespace gcrl
