// gemm_tiled.h — LDS-tiled variant of the batched fp32-MFMA GEMM for throughput-sized problems
// (more than 1024 tiles of 16x16: batch 2048 x hidden 512 and the like).
//
// A 256-thread workgroup owns a 64x64 output tile; its 4 waves own 32x32 each: ONE v_mfma_f32_32x32x2_f32 accumulator.
// Per 16-wide k-step the two 64x16 operand tiles are fetched from global memory with 16-byte loads (along k when the
// operand is k-contiguous, along the row index otherwise — then transposed on the LDS write), held in registers for two
// k-steps while earlier steps' MFMAs run, and stored to one of two LDS buffers (one barrier per step).  Fragments come
// back as one ds_read_b128 per operand and 8 k (the k-permutation of gemm_mfma.h: element q of the read feeds MFMA q);
// the MFMAs accumulate the TRANSPOSED tile so the epilogue moves 16 bytes per lane.
//
// What round 3 found, in the order it was found (M = 10240, N = K = 512: the 1280 tiles of TQC's 5-critic launches; all
// numbers by tools/gemm_micro.hip on one MI355X; round 2's kernel: 86 / 80 TFLOP/s forward / dX):
//  * the 16x16x4 instruction SUSTAINS 98 / 134 TFLOP/s on this chip at 1 / 2+ waves per SIMD, 32x32x2 142 / 152
//    (profiles/r03_mfma_issue_ceiling.txt) — yet swapping it in changed nothing, and neither did a conflict-free LDS image,
//    an XCD-aware tile order, deeper or shallower prefetch, hand-interleaving every LDS / global instruction between the
//    MFMAs, or removing the result stores: 59-62 us every time.  With ALL memory work and the barrier removed (MFMAs only)
//    the launch still took 54 us;
//  * wall-clock stamps per workgroup showed why: 256 of the 1280 workgroups started 36 us late.  A CU holds only FOUR
//    workgroups of 32 768 B of LDS, although 5 x 32 KB is exactly its 160 KB and the runtime's occupancy query answers 5
//    (tools/occupancy_probe.hip: 32 256 B -> 4 resident, 31 744 B -> 5).  So a launch of exactly five tiles per CU ran four
//    of them, then the fifth alone — the same two rounds as round 2's 36 KB kernel;
//  * hence this form: k-step 16 = 16 KB of LDS (the register stages keep the same two-k-step = 32-k prefetch distance), five
//    workgroups resident, one round;
//  * what is left (same harness at this form, gpurun_out -> profiles/r03_gemm_tiled_ablation.txt): forward 52.8 us = 102 TFLOP/s;
//    without the LDS reads 49.9, without the LDS stores 49.6, without the global fetches 50.4, without the barrier 52.6, with
//    NONE of them (MFMAs + epilogue only) 46.9 us = 115 TFLOP/s.  By the workgroup stamps the main loop itself runs at the
//    matrix pipe's rate (median 36.5 us for 1280 MFMAs per SIMD = 34 us at 2.4 GHz); the rest is the launch ramp (the median
//    workgroup enters its loop 2.8 us after the first one started) and the epilogue, which all 1280 workgroups reach together:
//    7 us from the last MFMA to the last store, 21 MB of results at once.  One round of lock-stepped workgroups cannot overlap
//    its epilogue with anything; non-temporal result stores gain 3 % alone and lose 0.7 % inside TQC's step (the next layer
//    re-reads the result).
#pragma once
#include "gemm_mfma.h"

namespace gcrl {

constexpr int kTB = 64;        // block tile
constexpr int kBK = 16;        // k-step: 8 MFMAs per wave between barriers
constexpr int kNR = kTB * kBK / 256;      // staging registers per operand and stage: ONE float4 per thread
// LDS image of an operand tile: [64 rows][16 k], no padding; the 16-byte chunk c (= k >> 2) of row r lives at chunk position
// c ^ ((r >> 2) & 3) of its 64-byte row.  Conflict-free for everything that touches it: a fragment read (ds_read_b128: the 16
// lanes of a lane group sit on 16 different rows at the same k-chunk -> row bits 0-1 pick the 64-byte quarter of the bank
// space, row bits 2-3 the chunk), the k-contiguous stores (8 lanes = two rows' 4 chunks) and the transposing stores of the
// row-contiguous mode (store_tile: 32 lanes = 4 k x 8 row quads -> 32 banks).
constexpr int kLDT = kBK;
__device__ inline int lds_at(int row, int k) { return row * kLDT + ((((k >> 2) ^ ((row >> 2) & 3)) << 2) | (k & 3)); }

enum { FETCH_KC = 0, FETCH_RC = 1, FETCH_GEN = 2 };

// Fetch a 64 x 16 operand tile (rows r0.., k k0..) into 4 registers per thread.
// element(row, k) = base[row*rs + k*cs]; rows >= R or k >= K read as 0.
//
// The two vector modes are free of control flow and of per-step address arithmetic: a thread's fragment offset (bytes, at
// k-step 0) is computed ONCE (`FetchPlan`; rows outside the operand get an offset past the descriptor's extent, for which
// the hardware returns 0) and a k-step only adds a scalar offset to the buffer load.  The launcher grants FETCH_KC /
// FETCH_RC only when whole 16-byte fragments are legal and K is a multiple of the k-step; everything else is FETCH_GEN.
constexpr int kOob = 0x7ffffff0;   // byte offset past any descriptor extent

struct FetchPlan { int voff; int step; };

// R: rows of the operand that exist in memory
template <int MODE>
__device__ inline FetchPlan fetch_plan(long long rs, long long cs, int r0, int R) {
  FetchPlan pl;
  const int tid = threadIdx.x;
  if (MODE == FETCH_KC) {     // 4 lanes x 16 B = one row's 16 k; a wave covers 16 rows
    const int r = r0 + (tid >> 2), k = (tid & 3) << 2;
    pl.voff = r < R ? (int)(((long long)r * rs + k) * 4) : kOob;
  } else {
    // row-contiguous operand: a wave-load covers 4 k (lane & 3) x 16 row quads (lane >> 2) — per k a 256-byte run — and the
    // workgroup's 4 waves the 4 k-quads; store_tile transposes through bank-conflict-free dword stores
    const int l = tid & 63, wv = tid >> 6;
    const int k = 4 * wv + (l & 3), r = r0 + ((l >> 2) << 2);
    pl.voff = r < R ? (int)(((long long)k * cs + r) * 4) : kOob;
  }
  pl.step = MODE == FETCH_KC ? kBK * 4 : (int)(cs * kBK * 4);
  return pl;
}

// `in` false (a k-step past the end, issued to keep the pipeline free of branches): zeros
template <int MODE>
__device__ inline void fetch_tile(float (&reg)[kNR], const float* __restrict__ base, __amdgpu_buffer_rsrc_t rsrc, const FetchPlan& pl,
                                  long long rs, long long cs, int r0, int R, int ks, int K, bool in) {
  if (MODE != FETCH_GEN) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? pl.voff : kOob, in ? ks * pl.step : 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) reg[q] = __uint_as_float(v[q]);
  } else {
    // element-wise mode (first layers with K = 25, heads with K = 1..4, unaligned operands): one buffer load per element,
    // no branch (with `if (r < R && k < K) v = base[...]` every load sat behind its own branch: 50 us per k-step in round 1)
    // (`in` folded into the offsets, no early return: this fetch also runs inside the pipelined k-step, where any branch
    // makes the compiler drain the loads in flight)
    const int tid = threadIdx.x, k0 = ks * kBK;
    const int rs4 = (int)(rs * 4), cs4 = (int)(cs * 4);
#pragma unroll
    for (int p = 0; p < kNR; ++p) {
      const int e = tid + 256 * p, r = r0 + (e & 63), k = k0 + (e >> 6);
      const bool ok = in && r < R && k < K;
      const int off = ok ? r * rs4 + k * cs4 : kOob;   // (32-bit: operands are < 2 GiB; same speed as the 64-bit form here, measured)
      reg[p] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
    }
  }
}

// registers -> LDS
template <int MODE>
__device__ inline void store_tile(float (&reg)[kNR], float* __restrict__ lds) {
  const int tid = threadIdx.x;
  if (MODE == FETCH_KC) {
    const int row = tid >> 2, c = tid & 3;
    *reinterpret_cast<float4*>(lds + row * kLDT + ((c ^ ((row >> 2) & 3)) << 2)) = make_float4(reg[0], reg[1], reg[2], reg[3]);
  } else if (MODE == FETCH_RC) {
    // a thread holds rows r..r+3 at ONE k (fetch_plan).  Store instruction t writes row r + ((t + rot) & 3), rot = (lane >> 4) & 3:
    // within 32 lanes (4 k x 8 row quads) the two row quads that share a swizzled chunk then differ in row bit 0 -> 32 banks.
    const int l = tid & 63, wv = tid >> 6, kq = l & 3, rq = l >> 2, rot = (rq >> 2) & 3;
    // rotate the four row values left by rot (two conditional-swap stages) so that instruction t finds its value in slot t
    const float x0 = reg[0], x1 = reg[1], x2 = reg[2], x3 = reg[3];
    const bool b0 = rot & 1, b1 = rot & 2;
    const float y0 = b0 ? x1 : x0, y1 = b0 ? x2 : x1, y2 = b0 ? x3 : x2, y3 = b0 ? x0 : x3;
    const float z[4] = {b1 ? y2 : y0, b1 ? y3 : y1, b1 ? y0 : y2, b1 ? y1 : y3};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = (rq << 2) + ((t + rot) & 3);
      lds[row * kLDT + (((wv ^ (rq & 3)) << 2) | kq)] = z[t];       // (row >> 2) & 3 == rq & 3
    }
  } else {
#pragma unroll
    for (int p = 0; p < kNR; ++p) {
      const int e = tid + 256 * p;
      lds[lds_at(e & 63, e >> 6)] = reg[p];
    }
  }
}

// hooks for tools/gemm_micro.hip (timing experiments that drop one ingredient of the k-step, wall-clock stamps at a
// workgroup's phase boundaries); never defined in the library
#ifndef GCRL_ABL_READ
#define GCRL_ABL_READ(p) (*reinterpret_cast<const float4*>(p))
#endif
#ifndef GCRL_ABL_STORE
#define GCRL_ABL_STORE(x) x
#endif
#ifndef GCRL_ABL_FETCH
#define GCRL_ABL_FETCH(x) x
#endif
#ifndef GCRL_ABL_BARRIER
#define GCRL_ABL_BARRIER() __syncthreads()
#endif
#ifndef GCRL_STAMP
#define GCRL_STAMP(i) do { } while (0)
#endif

// sum over the workgroup's 16 row slots for every column quad, in slot order (all threads get the result) — ops_sac.hip's slot_sum
__device__ inline v4f tiled_slot_sum(v4f v, v4f (*red)[16], int cq, int slot) {
  __syncthreads();
  red[slot][cq] = v;
  __syncthreads();
  v4f s = red[0][cq];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += red[i][cq];
  return s;
}

// The bias gradient of a dW problem (column N-1 of G^T [X | 1] = the row sums of the A operand over k) is NOT a synthesised
// ones column of B any more: for N - 1 = 512 that column cost a ninth tile column — 8 of 72 tiles per problem computing one
// useful column.  Instead every thread adds up the A fragments it stages anyway (4 adds per k-step), and the tiles of tile
// column 0 reduce those over the workgroup in a fixed order: db comes out of the first tile column for free.
template <int MA, int MB>
__device__ inline void gemm_tiled_body(const GemmDesc& d, int t, float* ldsA, float* ldsB) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hk = lane >> 5;
  const int M = d.M, K = d.K;
  const int N = d.N - (d.ones_col ? 1 : 0);      // columns that are tiles' work (and B's rows in memory)
  const int S = d.ksplit > 1 ? d.ksplit : 1;
  const int tiles = d.ntiles / S;
  const int ti = t % tiles, sp = t / tiles;      // split-major: the workgroups of one k-range are neighbours (they share operand panels)
  const int tn = ti % d.tiles_n, tm = ti / d.tiles_n;
  const int m0 = tm * kTB, n0 = tn * kTB;
  const int wm = wave >> 1, wn = wave & 1;
  const long long sl = d.slot ? (long long)*d.slot : 0;
  const float* __restrict__ A = d.A + sl * d.a_slot;
  const float* __restrict__ Bm = d.B + sl * d.b_slot;
  const bool bias_tile = d.ones_col && tn == 0;   // (uniform)

  v16f acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  v4f bs = (v4f){0.f, 0.f, 0.f, 0.f};             // sums over k of the A fragments this thread staged
  GCRL_STAMP(0);

  // extents for the vector modes: whole rows of the operand (KC: R rows of rs floats; RC: K rows of cs floats)
  // (the last element's index + 1: exact for all three modes)
  const __amdgpu_buffer_rsrc_t rsa = wave_uniform_rsrc_n(A, (long long)(M - 1) * d.a_rs + (long long)(K - 1) * d.a_cs + 1);
  const __amdgpu_buffer_rsrc_t rsb = wave_uniform_rsrc_n(Bm, (long long)(N - 1) * d.b_cs + (long long)(K - 1) * d.b_rs + 1);
  // this workgroup's k-steps [ks0, ks1): an even number per split, so that LDS buffer parity = step parity everywhere
  const int ksteps_all = (K + kBK - 1) / kBK;
  const int per = (((ksteps_all + S - 1) / S) + 1) & ~1;
  const int ks0 = sp * per, ksteps = min(ksteps_all, ks0 + per);
  const FetchPlan pla = fetch_plan<MA>(d.a_rs, d.a_cs, m0, M);
  const FetchPlan plb = fetch_plan<MB>(d.b_cs, d.b_rs, n0, N);
  // fragment offsets of this lane (floats): row 32*wm + r32 of the A tile, row 32*wn + r32 of the B tile; a k-step's two
  // 8-k groups: lane half hk reads chunk 2g + hk (4 consecutive k), MFMA q of the group takes element q of both reads, i.e.
  // k = 8g + q from half 0 and 8g + 4 + q from half 1 — A and B permuted alike, so every product pairs the same k
  const int rowA = 32 * wm + r32, rowB = 32 * wn + r32;
  const int offA = rowA * kLDT, offB = rowB * kLDT;
  const int swA = (rowA >> 2) & 3, swB = (rowB >> 2) & 3;
  auto fetch = [&](float (&xa)[kNR], float (&xb)[kNR], int ks) {
    fetch_tile<MA>(xa, A, rsa, pla, d.a_rs, d.a_cs, m0, M, ks, K, ks < ksteps);
    fetch_tile<MB>(xb, Bm, rsb, plb, d.b_cs, d.b_rs, n0, N, ks, K, ks < ksteps);
  };
  auto tally = [&](const float (&xa)[kNR]) {     // (steps past the end were fetched as zeros)
#pragma unroll
    for (int q = 0; q < 4; ++q) bs[q] += xa[q];
  };
  // (roles swapped in every MFMA — B fragment as the instruction's A: the accumulator holds the TRANSPOSED tile, see the epilogue)
#define GCRL_MFMA(bv, av) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0)
#define GCRL_SB() __builtin_amdgcn_sched_barrier(0)
  // the pipelined loop for every problem with a reduction worth pipelining (element-wise operands included: a dW problem's
  // head row or 27-column input — one memory round trip per k-step in the plain loop below, 16 of them per split, set the
  // length of TD3's dW launch); the plain loop for a handful of k-steps (first layers' K = 25, heads' K = 1..4)
  if ((MA != FETCH_GEN && MB != FETCH_GEN) || ksteps - ks0 >= 6) {
    // Vector modes (K a multiple of the k-step).  Software pipeline: LDS[ks & 1] holds tile ks; register stage (ks + 1) & 1
    // holds tile ks+1, requested two k-steps ago; the other stage (tile ks+2) stays in flight.  During the MFMAs of step ks:
    // tile ks+1 -> the other LDS buffer, then tile ks+3 is requested into the registers just freed.  Operand tiles come from
    // HBM / the infinity cache the first time they are touched (2-3 us under load) — two k-steps of five co-resident waves'
    // MFMAs.  NOTHING is conditional (with `if (ks + 3 < ksteps)` around a request the compiler could no longer count the
    // loads in flight and waited for the newest stage before every LDS store): steps past the end fetch zeros, the step count
    // is rounded up to even and the tail costs at most one k-step of MFMAs on zeros.
    // The step itself is HAND-ORDERED: a wave issues in order — an MFMA occupies the matrix pipe for 64 cycles and the next
    // one cannot issue before, but whatever stands BETWEEN two MFMAs issues while the first executes — so the LDS stores and
    // the global requests are dealt one per MFMA gap instead of forming phases of their own; sched_barrier pins the order.
    float ra[2][kNR], rb[2][kNR];
    auto step = [&](int ks, float (&xa)[kNR], float (&xb)[kNR]) {
      const int buf = ks & 1;
      const float* la = ldsA + buf * (kTB * kLDT) + offA;
      const float* lb = ldsB + buf * (kTB * kLDT) + offB;
      float* na = ldsA + (buf ^ 1) * (kTB * kLDT);
      float* nb = ldsB + (buf ^ 1) * (kTB * kLDT);
      const float4 a0 = GCRL_ABL_READ(la + ((hk ^ swA) << 2)), b0 = GCRL_ABL_READ(lb + ((hk ^ swB) << 2));
      const float4 a1 = GCRL_ABL_READ(la + (((2 + hk) ^ swA) << 2)), b1 = GCRL_ABL_READ(lb + (((2 + hk) ^ swB) << 2));
      GCRL_SB();
      GCRL_MFMA(b0.x, a0.x); GCRL_SB(); GCRL_ABL_STORE(store_tile<MA>(xa, na)); tally(xa); GCRL_SB();
      GCRL_MFMA(b0.y, a0.y); GCRL_SB(); GCRL_ABL_STORE(store_tile<MB>(xb, nb)); GCRL_SB();
      GCRL_MFMA(b0.z, a0.z); GCRL_SB(); GCRL_ABL_FETCH(fetch_tile<MA>(xa, A, rsa, pla, d.a_rs, d.a_cs, m0, M, ks + 3, K, ks + 3 < ksteps)); GCRL_SB();
      GCRL_MFMA(b0.w, a0.w); GCRL_SB(); GCRL_ABL_FETCH(fetch_tile<MB>(xb, Bm, rsb, plb, d.b_cs, d.b_rs, n0, N, ks + 3, K, ks + 3 < ksteps)); GCRL_SB();
      GCRL_MFMA(b1.x, a1.x); GCRL_MFMA(b1.y, a1.y); GCRL_MFMA(b1.z, a1.z); GCRL_MFMA(b1.w, a1.w);
      GCRL_ABL_BARRIER();
    };
    fetch(ra[0], rb[0], ks0);
    fetch(ra[1], rb[1], ks0 + 1);
    store_tile<MA>(ra[0], ldsA);
    store_tile<MB>(rb[0], ldsB);
    tally(ra[0]);
    fetch(ra[0], rb[0], ks0 + 2);
    __syncthreads();
    GCRL_STAMP(1);
    const int kp = ks0 + ((max(ksteps - ks0, 0) + 1) & ~1);     // (ks0 is even)
    for (int ks = ks0; ks < kp; ks += 2) {
      step(ks, ra[1], rb[1]);
      step(ks + 1, ra[0], rb[0]);
    }
    GCRL_STAMP(2);
  } else {
    // element-wise fetch (first layers with K = 25, heads, unaligned operands: a few k-steps at most): plain phases
    float ra[kNR], rb[kNR];
    fetch(ra, rb, ks0);
    store_tile<MA>(ra, ldsA + (ks0 & 1) * (kTB * kLDT));
    store_tile<MB>(rb, ldsB + (ks0 & 1) * (kTB * kLDT));
    tally(ra);
    fetch(ra, rb, ks0 + 1);
    __syncthreads();
    for (int ks = ks0; ks < ksteps; ++ks) {
      const float* la = ldsA + (ks & 1) * (kTB * kLDT) + offA;
      const float* lb = ldsB + (ks & 1) * (kTB * kLDT) + offB;
      const int nk = K - ks * kBK;   // k left from this step on: a whole 8-k group of zeros is skipped (small-K problems)
#pragma unroll
      for (int g = 0; g < kBK / 8; ++g) {
        if (g * 8 >= nk) break;
        const float4 a = *reinterpret_cast<const float4*>(la + (((2 * g + hk) ^ swA) << 2));
        const float4 b = *reinterpret_cast<const float4*>(lb + (((2 * g + hk) ^ swB) << 2));
        GCRL_MFMA(b.x, a.x); GCRL_MFMA(b.y, a.y); GCRL_MFMA(b.z, a.z); GCRL_MFMA(b.w, a.w);
      }
      store_tile<MA>(ra, ldsA + ((ks + 1) & 1) * (kTB * kLDT));
      store_tile<MB>(rb, ldsB + ((ks + 1) & 1) * (kTB * kLDT));
      tally(ra);
      fetch(ra, rb, ks + 2);
      __syncthreads();
    }
  }
#undef GCRL_MFMA
#undef GCRL_SB

  // bias gradient: db[m0 + r] = the sum over the workgroup's threads that staged row r, slot by slot in a fixed order.
  // Who staged what (fetch_plan / fetch_tile): RC — thread (wave wv, lane l) rows 4*(l >> 2) + q at k = 4*wv + (l & 3) mod 16:
  // 16 slots; KC — row tid >> 2, four consecutive k: 4 slots (tid & 3); GEN — row tid & 63 for every register: 4 slots (tid >> 6).
  float dbv = 0.f;                                 // threads 0..63: db of row m0 + tid (this split's share)
  if (bias_tile) {
    float* red = ldsA;                             // [slot][64 rows]; the main loop's last barrier has passed
    const int tid = threadIdx.x;
    int nslot;
    if (MA == FETCH_RC) {
      nslot = 16;
      *reinterpret_cast<float4*>(red + (4 * wave + (lane & 3)) * 64 + ((lane >> 2) << 2)) = make_float4(bs[0], bs[1], bs[2], bs[3]);
    } else if (MA == FETCH_KC) {
      nslot = 4;
      red[(tid & 3) * 64 + (tid >> 2)] = (bs[0] + bs[1]) + (bs[2] + bs[3]);
    } else {
      nslot = 4;
      red[(tid >> 6) * 64 + (tid & 63)] = (bs[0] + bs[1]) + (bs[2] + bs[3]);
    }
    __syncthreads();
    if (tid < 64)
      for (int i = 0; i < nslot; ++i) dbv += red[i * 64 + tid];
    __syncthreads();
  }

  if (S > 1) {
    // Split reduction: publish the raw partial, take a ticket; the last arriver sums all S partials in index order (its own
    // comes back from memory like the others': the sum is the same whoever is last) and goes on to the epilogue.
    // The partials move with AGENT-SCOPE accesses (sc1: written through to / read from the memory side of the per-XCD L2s), so
    // no cache maintenance is needed: a release / acquire fence pair at agent scope (__threadfence: buffer_wbl2 + buffer_inv
    // per wave) serialised the workgroups of a launch at ~0.15 us each — 1024 workgroups: 157 us for 9 us of arithmetic, measured.
    constexpr int kSc1 = 16;                       // cache-policy bit of the raw buffer builtins on gfx94x / gfx950
    const __amdgpu_buffer_rsrc_t rsp = wave_uniform_rsrc_n(d.kpart + (long long)ti * S * kTiledPartStride, (long long)S * kTiledPartStride);
    const int mine = sp * kTiledPartStride * 4;    // byte offsets: S * 16.25 KB per tile
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v4u v = {__float_as_uint(acc[4 * g]), __float_as_uint(acc[4 * g + 1]), __float_as_uint(acc[4 * g + 2]), __float_as_uint(acc[4 * g + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(v, rsp, mine + (((wave * 4 + g) * 64 + lane) << 4), 0, kSc1);
    }
    if (bias_tile && threadIdx.x < 64) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dbv), rsp, mine + (64 * 64 + (int)threadIdx.x) * 4, 0, kSc1);
    // EVERY wave waits for the memory side's acknowledgement of its write-through stores before the barrier that precedes the
    // ticket (a workgroup-scope release fence emits no vmcnt wait on gfx950: without this the ticket could overtake the
    // partials of waves 1-3 and the last arriver sum a previous step's values — ADVICE r3; tools/check_release_isa.py greps the ISA)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int* s_ticket = reinterpret_cast<unsigned int*>(ldsB);   // (LDS is free between the main loop and the epilogue)
    __syncthreads();
    if (threadIdx.x == 0) *s_ticket = __hip_atomic_fetch_add(d.kticket + ti * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned int ticket = *s_ticket;
    __syncthreads();                               // (the epilogue's tile will overwrite the slot)
    if (ticket != (unsigned)(S - 1)) return;       // (uniform)
    if (threadIdx.x == 0) __hip_atomic_store(d.kticket + ti * kTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    dbv = 0.f;
    // four partials' loads in flight at a time (an agent-scope load is a round trip to the memory side: ~2 us under load, and
    // one partial per round trip made the fix-up of an 8-way split 16 us long); the adds stay in index order
    for (int j0 = 0; j0 < S; j0 += 4) {
      v4u v[4][4];
      unsigned int bj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pj = (j0 + u < S) ? (j0 + u) * kTiledPartStride * 4 : kOob;     // (past the descriptor's extent: zeros)
#pragma unroll
        for (int g = 0; g < 4; ++g) v[u][g] = __builtin_amdgcn_raw_buffer_load_b128(rsp, pj == kOob ? kOob : pj + (((wave * 4 + g) * 64 + lane) << 4), 0, kSc1);
        bj[u] = 0;
        if (bias_tile && threadIdx.x < 64) bj[u] = __builtin_amdgcn_raw_buffer_load_b32(rsp, pj == kOob ? kOob : pj + (64 * 64 + (int)threadIdx.x) * 4, 0, kSc1);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[4 * g] += __uint_as_float(v[u][g][0]); acc[4 * g + 1] += __uint_as_float(v[u][g][1]);
          acc[4 * g + 2] += __uint_as_float(v[u][g][2]); acc[4 * g + 3] += __uint_as_float(v[u][g][3]);
        }
        dbv += __uint_as_float(bj[u]);
      }
    }
  }

  const float* __restrict__ bias = d.bias;
  const float* __restrict__ H = d.H + sl * d.h_slot;
  float* __restrict__ C = d.C + sl * d.c_slot;
  const int epi = d.epi, mul = d.mul;
  // Epilogue through LDS.  The MFMAs ran with the operand roles swapped (B fragment as the instruction's A): the accumulator
  // tile is the transpose — lane (r32, hk) holds row m = r32 of C, registers 4g..4g+3 its columns n = 8g + 4*hk + 0..3.
  // Stored straight from there a wave instruction writes 32 rows x 32 bytes (and reads the saved activations of a dX problem
  // the same way): 21 MB of quarter-line requests at the end of every launch, 7 of a forward launch's 52 us by the
  // workgroup stamps of tools/gemm_micro.hip.  Instead the raw tile goes to the (now free) LDS image [64][64] with the
  // 16-byte quad q of row r at position q ^ (r & 15) (both the transposed writes — 8 lanes on 8 rows at one column — and the
  // row reads are conflict-free), and after one barrier thread t finishes quad t & 15 of rows (t >> 4) + 16 i: bias, saved
  // activations and results move as whole 256-byte row segments; the bias / activation loads are issued BEFORE the barrier.
  float* tile = ldsA;                         // 2 x 64 x 16 floats of A + the same of B = 64 x 64 floats, contiguous (gemm_tiled_kernel)
  const bool vec_ok = (d.c_rs % 4 == 0) && (((unsigned long long)C & 15) == 0) &&
                      (mul == MUL_NONE || (d.h_rs % 4 == 0 && (((unsigned long long)H & 15) == 0))) &&
                      (!bias || (((unsigned long long)bias & 15) == 0));
  const int eq = threadIdx.x & 15, er = threadIdx.x >> 4;
  const int nq = n0 + 4 * eq;
  const bool quad_vec = vec_ok && nq + 3 < N;
  v4f bv = (v4f){0.f, 0.f, 0.f, 0.f}, hv[4];
  if (quad_vec) {
    if (bias) bv = *(const v4f*)(bias + nq);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hv[i] = (v4f){0.f, 0.f, 0.f, 0.f};
      const int m = m0 + er + 16 * i;
      if (mul != MUL_NONE && m < M) hv[i] = *(const v4f*)(H + (long long)m * d.h_rs + nq);
    }
  }
  {
    const int row = 32 * wm + r32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q = 8 * wn + 2 * g + hk;
      *reinterpret_cast<float4*>(tile + row * 64 + ((q ^ (row & 15)) << 2)) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
  }
  GCRL_STAMP(4);
  __syncthreads();
  GCRL_STAMP(5);
  float ss = 0.f;
  if (bias_tile && threadIdx.x < 64 && m0 + (int)threadIdx.x < M) {   // (threads of wave 0: its sum-of-squares slot takes db's share)
    d.col_out[m0 + threadIdx.x] = dbv;
    ss += dbv * dbv;
  }
  v4f keep[4];                                  // the finished quads (BatchNorm statistics below)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    keep[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int row = er + 16 * i, m = m0 + row;
    if (m >= M || nq >= N) continue;
    const float4 t4 = *reinterpret_cast<const float4*>(tile + row * 64 + ((eq ^ (row & 15)) << 2));
    if (quad_vec) {
      v4f v = (v4f){t4.x, t4.y, t4.z, t4.w};
      v += bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = act_apply(v[r], epi);
        if (mul != MUL_NONE) v[r] *= act_deriv(hv[i][r], mul);
        ss += v[r] * v[r];
      }
      *(v4f*)(C + (long long)m * d.c_rs + nq) = v;      // (non-temporal here: +3 % alone, -0.7 % inside TQC's step — the next layer re-reads it)
      keep[i] = v;
    } else {
      const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nq + r;
        if (n >= N) continue;
        float v = tv[r];
        if (bias) v += bias[n];
        v = act_apply(v, epi);
        if (mul != MUL_NONE) v *= act_deriv(H[(long long)m * d.h_rs + n], mul);
        ss += v * v;
        C[(long long)m * d.c_rs + n] = v;
      }
    }
  }
  if (d.bn_part) {   // (uniform per workgroup; the launcher admits whole-quad tiles only)
    // BatchNorm1d statistics of the tile's columns over its (up to) 64 rows — the row-block partial bn_stats_kernel (ops_sac.hip)
    // would form from the stored matrix, by the same sums in the same order (thread = column quad x row slot, rows slot + 16 i;
    // local two-pass): the actor's forward at B = 2048 loses a 4.4 us launch per hidden layer and a re-read of z
    v4f (*red)[16] = reinterpret_cast<v4f (*)[16]>(tile);     // (tiled_slot_sum's first barrier: every thread has read its tile quads)
    const int nr = min(kTB, M - m0);
    v4f s = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) if (er + 16 * i < nr) s += keep[i];
    const v4f mean = tiled_slot_sum(s, red, eq, er) / (float)nr;
    v4f q = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) if (er + 16 * i < nr) { const v4f df = keep[i] - mean; q += df * df; }
    const v4f m2 = tiled_slot_sum(q, red, eq, er);
    if (er == 0 && nq + 3 < N) {
      const int tiles_m = (M + kTB - 1) / kTB;
      *(v4f*)(d.bn_part + (long long)tm * N + nq) = mean;
      *(v4f*)(d.bn_part + (long long)(tiles_m + tm) * N + nq) = m2;
    }
  }
  GCRL_STAMP(3);
  if (d.sumsq_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) d.sumsq_out[(long long)ti * 4 + wave] = ss;
  }
}

__global__ __launch_bounds__(256, 5) void gemm_tiled_kernel(GemmBatch gb) {   // five workgroups per CU: <= 96 registers, 16 KB of LDS
  __shared__ __attribute__((aligned(16))) float lds[4 * kTB * kLDT];   // A: two buffers, B: two buffers; the epilogue's 64 x 64 tile
  float* ldsA = lds;
  float* ldsB = lds + 2 * kTB * kLDT;
  const int tile = xcd_tile_of((int)blockIdx.x, (int)gridDim.x);   // an XCD owns whole tile rows: each A panel enters ONE L2 (gemm_mfma.h)
  const int pi = gemm_problem_of(gb, tile);
  const GemmDesc& d = gb.d[pi];
  gemm_pin(d);   // (the record in registers after ONE scalar round trip instead of fourteen dependent ones: kernarg.h)
  asm volatile("" ::"s"(d.a_rvec), "s"(d.b_rvec), "s"(d.ksplit), "s"(d.kpart), "s"(d.kticket), "s"(d.sumsq_out), "s"(d.bn_part));
  const int t = tile - d.tile0;
  if (t >= d.ntiles) return;
  // fetch modes are per problem and wave-uniform: 3 x 3 straight-line instances
  // (vector modes need whole 16-byte fragments: extents in multiples of 4 along the vector direction)
  const int n_mem = d.N - (d.ones_col ? 1 : 0);   // (a dW problem's last column is the bias gradient: row sums, not a tile column)
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
