// gemm_mfma.hip — launcher of the batched fp32-MFMA GEMM (kernel: gemm_mfma.h).
#include "gemm_mfma.h"
#include <cstdio>
#include <cstdlib>
#include "gemm_tiled.h"

namespace gcrl {

namespace {
template <int TM, int TN, int KSPLIT>
int launch_shape(hipStream_t st, GemmBatch& gb) {
  int tiles = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmDesc& d = gb.d[i];
    const int tm = (d.M + 16 * TM - 1) / (16 * TM);
    d.tiles_n = (d.N + 16 * TN - 1) / (16 * TN);
    d.ntiles = tm * d.tiles_n;
    d.tile0 = tiles;
    tiles += d.ntiles;
    if (KSPLIT == 1) tiles = (tiles + 3) & ~3;  // a workgroup's 4 waves stay inside one problem
  }
  const int grid = (KSPLIT == 4) ? tiles : (tiles + 3) / 4;
  for (int i = 0; i < kMaxProb; ++i) gb.tile0[i] = i < gb.n ? gb.d[i].tile0 : 0x7fffffff;
  hipLaunchKernelGGL((gemm_batch_kernel<TM, TN, KSPLIT>), dim3(grid), dim3(256), 0, st, gb);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

// The tile shape is a function of the PROBLEM alone, never of what else shares the launch: the
// k-partition (and with it the fp32 summation order) of a problem must not change when the
// engine co-schedules it with other problems (software-pipelined steps stay bitwise equal to
// sequential ones).
//   <= 1024 tiles of 16x16 : one tile per workgroup, K split over its 4 waves (latency-bound
//                            sizes; small K just leaves waves idle in the exchange)
//   <= 8192                : one tile per wave
//   larger                 : 32x32 per wave
bool outer_ok(const GemmDesc& d);
int shape_of(const GemmDesc& d) {
  const long long tiles16 = (long long)((d.M + 15) / 16) * ((d.N + 15) / 16);
  // (tried for long reductions, i.e. dW at batch >= 1024: LDS-tiled 64x64 — 20 tiles per problem each
  // walking K alone, 2x slower; LDS-tiled 64x64 split over k-ranges of 256 plus a reduce pass — 41 + 5 us
  // vs 32 us for TD3's 12 problems at B=2048: both forms move ~150 MB through L2, i.e. are L2-bound at
  // this tile size.  Also tried: a dW-specialised kernel with 128x128 tiles, batch-major operand rows copied to
  // LDS as they lie in memory and 4-byte fragment reads — correct, but 1.8 us per 16-row stage with one
  // workgroup per CU (2x its MFMA time) and slower end to end.  The k-split 16x16 form stays.)
  if (tiles16 <= 1024) return 1;
  // a long reduction into a mid-sized output (dW of a 512-wide layer at batch 2048: 72 tiles of 64x64): the
  // LDS-tiled form has too few workgroups, each walking K alone (128 us alone vs 37 us k-split; measured.  Round 2,
  // with the faster tiled kernel and TQC's five critics batched: still 1604 vs 1578 us/step, the k-split form stays.
  // Also tried in round 2: 2x2 tiles per workgroup with the same k-split (half the operand bytes per MFMA, a quarter of
  // the workgroups): TD3 177 -> 199, TQC 1520 -> 1553 us/step.)
  if (d.K >= 1024 && (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) < 192) return 1;
  // LDS-tiled 64x64 workgroup tiles (gemm_tiled.h) once a problem alone fills most CUs with
  // them (>= 192 tiles of 64x64); in between, one 16x16 tile per wave keeps more CUs busy
  // (TD3 at B=2048, H=256: 128 tiles of 64x64 was 16% slower than 2048 wave tiles)
  // ... or when the reduction is long (dW at batch 2048: a wave walking 128 k-chunks alone took
  // 89 us; 64x64 tiles with LDS reuse are MFMA-bound there too)
  // a single 16-k chunk (head dX with K = 1..4: outer products): nothing to stage, 32x32 wave tiles (12 us vs 17 us
  // LDS-tiled at M=10240, N=512).  K = 25 first layers: the LDS-tiled form again (19 us vs 24 us) since its element-wise
  // fetch went branch-free.
  if (outer_ok(d) && (long long)d.M * d.N >= (1 << 18)) return 5;   // (small problems: launch-bound either way, the MFMA form stays)
  if (d.K <= 16) return 3;
  const long long tiles64 = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64);
  return (tiles64 >= 192 || d.K >= 1024) ? 4 : 2;
}

// Form 5 (round 4): K = 1 — the input gradient of a one-output head, G' = (g w^T) * act'(H): an outer product per problem,
// no reduction at all.  On the matrix cores (form 3) TQC's five critic heads at B = 2048, H = 512 took 23 us for 42 MB of
// traffic, twice per step; as a streaming kernel (one 16-byte quad of C per thread and pass, four passes per workgroup, all
// loads requested first) it is bound by HBM like any copy.  Same value as the MFMA path: one product rounded once (the
// instruction's padded k contribute exact zeros), then bias, activation and derivative in the same order.
constexpr int kOuterQuads = 1024;   // quads of C per workgroup
bool outer_ok(const GemmDesc& d) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return d.K == 1 && !d.ones_col && !d.sumsq_out && !d.bn_part && d.ksplit <= 1 && d.N % 4 == 0 && d.b_cs == 1 && d.c_rs % 4 == 0 && al(d.B) &&
         al(d.C) && (!d.bias || al(d.bias)) && (d.mul == MUL_NONE || (d.H && d.h_rs % 4 == 0 && al(d.H))) &&
         (!d.slot || (d.b_slot % 4 == 0 && d.c_slot % 4 == 0 && d.h_slot % 4 == 0));
}
__global__ __launch_bounds__(256) void gemm_outer_kernel(GemmBatch gb) {
  const int pi = gemm_problem_of(gb, (int)blockIdx.x);
  const GemmDesc& d = gb.d[pi];
  gemm_pin(d);
  const int nq = d.N >> 2;
  const long long total = (long long)d.M * nq;
  const long long sl = d.slot ? (long long)*d.slot : 0;
  const float* __restrict__ A = d.A + sl * d.a_slot;
  const float* __restrict__ Bm = d.B + sl * d.b_slot;
  const float* __restrict__ H = d.H ? d.H + sl * d.h_slot : nullptr;
  float* __restrict__ C = d.C + sl * d.c_slot;
  const int epi = d.epi, mul = d.mul;
  const long long q0 = (long long)((int)blockIdx.x - d.tile0) * kOuterQuads + threadIdx.x;
  float av[4];
  v4f bv[4], hv[4], biv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long qi = q0 + 256 * u;
    const bool in = qi < total;
    const int m = in ? (int)(qi / nq) : 0, n = in ? (int)(qi % nq) * 4 : 0;
    av[u] = in ? A[(long long)m * d.a_rs] : 0.f;
    bv[u] = in ? *(const v4f*)(Bm + n) : (v4f){0.f, 0.f, 0.f, 0.f};
    biv[u] = (in && d.bias) ? *(const v4f*)(d.bias + n) : (v4f){0.f, 0.f, 0.f, 0.f};
    hv[u] = (in && mul != MUL_NONE) ? *(const v4f*)(H + (long long)m * d.h_rs + n) : (v4f){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long qi = q0 + 256 * u;
    if (qi >= total) continue;
    const int m = (int)(qi / nq), n = (int)(qi % nq) * 4;
    v4f v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = __fmul_rn(av[u], bv[u][r]);
      if (d.bias) x += biv[u][r];
      x = act_apply(x, epi);
      if (mul != MUL_NONE) x *= act_deriv(hv[u][r], mul);
      v[r] = x;
    }
    *(v4f*)(C + (long long)m * d.c_rs + n) = v;
  }
}
int launch_outer(hipStream_t st, GemmBatch& gb) {
  int wgs = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmDesc& d = gb.d[i];
    d.tile0 = wgs;
    d.ntiles = (int)(((long long)d.M * (d.N >> 2) + kOuterQuads - 1) / kOuterQuads);
    wgs += d.ntiles;
  }
  for (int i = 0; i < kMaxProb; ++i) gb.tile0[i] = i < gb.n ? gb.d[i].tile0 : 0x7fffffff;
  hipLaunchKernelGGL(gemm_outer_kernel, dim3(wgs), dim3(256), 0, st, gb);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

int launch_tiled(hipStream_t st, GemmBatch& gb) {
  static const bool trace = std::getenv("GCRL_GEMM_TRACE") != nullptr;
  if (trace) {
    std::fprintf(stderr, "[tiled] %d problems:", gb.n);
    for (int i = 0; i < gb.n; ++i) {
      const GemmDesc& d = gb.d[i];
      std::fprintf(stderr, " (M=%d N=%d K=%d a:%d%d b:%d%d ones=%d)", d.M, d.N, d.K, d.a_vec, d.a_rvec, d.b_vec, d.b_rvec, d.ones_col);
    }
    std::fprintf(stderr, "\n");
  }
  int tiles = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmDesc& d = gb.d[i];
    const int tm = (d.M + kTB - 1) / kTB;
    d.tiles_n = (d.N - (d.ones_col ? 1 : 0) + kTB - 1) / kTB;      // (the bias gradient comes out of tile column 0)
    if (d.ksplit < 1) d.ksplit = 1;
    GCRL_CHECK_ARG(d.ksplit == 1 || (d.kpart && d.kticket), "launch_gemm_batch: a split reduction needs its partial and ticket arrays");
    d.ntiles = tm * d.tiles_n * d.ksplit;                          // work items: (split, tile)
    d.tile0 = tiles;
    tiles += d.ntiles;
  }
  for (int i = 0; i < kMaxProb; ++i) gb.tile0[i] = i < gb.n ? gb.d[i].tile0 : 0x7fffffff;
  hipLaunchKernelGGL(gemm_tiled_kernel, dim3(tiles), dim3(256), 0, st, gb);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}
}  // namespace

int gemm_shape_of(const GemmDesc& d) { return shape_of(d); }

int gemm_prepare_ksplit(GemmDesc& d) {
  d.a_vec = (d.a_cs == 1 && d.a_rs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
  d.b_vec = (d.b_rs == 1 && d.b_cs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
  d.a_rvec = (d.a_rs == 1 && d.a_cs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
  d.b_rvec = (d.b_cs == 1 && d.b_rs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
  d.tiles_n = (d.N + 15) / 16;
  d.ntiles = ((d.M + 15) / 16) * d.tiles_n;
  d.tile0 = 0;
  return d.ntiles;
}

int launch_gemm_batch(hipStream_t st, GemmDesc* descs, int n, int shape) {
  GCRL_CHECK_ARG(n >= 1 && n <= kMaxProb, "launch_gemm_batch: %d problems (max %d)", n, kMaxProb);
  int shapes[kMaxProb];
  for (int i = 0; i < n; ++i) {
    GemmDesc& d = descs[i];
    GCRL_CHECK_ARG(d.M >= 1 && d.N >= 1 && d.K >= 1 && d.A && d.B && d.C, "launch_gemm_batch: bad problem %d (M=%d N=%d K=%d)", i, d.M, d.N, d.K);
    GCRL_CHECK_ARG(!d.ones_col || (d.N >= 2 && d.col_out), "launch_gemm_batch: ones_col needs N >= 2 and col_out");
    d.a_vec = (d.a_cs == 1 && d.a_rs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
    d.b_vec = (d.b_rs == 1 && d.b_cs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
    d.a_rvec = (d.a_rs == 1 && d.a_cs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
    d.b_rvec = (d.b_cs == 1 && d.b_rs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
    shapes[i] = shape ? shape : (d.shape_hint ? d.shape_hint : shape_of(d));
    // BatchNorm partials out of the epilogue: the k-split 16x16 form (16-row partials) or the LDS-tiled one (64-row partials, whole
    // 16-byte column quads only)
    GCRL_CHECK_ARG(!d.bn_part || (!d.ones_col && d.c_rs >= d.N &&
                                  (shapes[i] == 1 || (shapes[i] == 4 && d.N % 4 == 0 && d.c_rs % 4 == 0 && ((uintptr_t)d.C & 15) == 0 &&
                                                      ((uintptr_t)d.bn_part & 15) == 0 && (!d.bias || ((uintptr_t)d.bias & 15) == 0) &&
                                                      d.mul == MUL_NONE && d.ksplit <= 1))),
                   "launch_gemm_batch: bn_part needs the k-split 16x16 form or an unsplit LDS-tiled problem with aligned quads");
  }
  for (int i = 0; i < n; ++i) GCRL_CHECK_ARG(shapes[i] != 5 || outer_ok(descs[i]), "launch_gemm_batch: problem %d is not an aligned K = 1 outer product", i);
  for (int s = 1; s <= 5; ++s) {  // one launch per shape present (almost always exactly one)
    GemmBatch gb;
    gb.n = 0;
    for (int i = 0; i < n; ++i)
      if (shapes[i] == s) gb.d[gb.n++] = descs[i];
    if (gb.n == 0) continue;
    int rc = s == 1 ? launch_shape<1, 1, 4>(st, gb)
             : (s == 2 ? launch_shape<1, 1, 1>(st, gb)
                       : (s == 3 ? launch_shape<2, 2, 1>(st, gb) : (s == 4 ? launch_tiled(st, gb) : launch_outer(st, gb))));
    if (rc) return rc;
  }
  return GCRL_OK;
}

}  // namespace gcrl
