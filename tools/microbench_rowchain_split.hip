// microbench: ONE dependent chain of H x H layer passes for 4 batch rows per row block, as the row-chain
// kernels run it (csrc/rowchain.h: one workgroup on one CU streams the whole [in][out] weight matrix per pass, the chained
// one-barrier form), against the same chain K-SPLIT OVER TWO WORKGROUPS ON TWO CUs: each streams half the matrix's rows (its
// half of the reduction), forms partial sums for all H outputs, and the two exchange the halves they do not finish — 2 KB
// each way — through agent-scope stores and a meeting counter (csrc/meet.h, the primitive rc_meet uses since round 3).  Layer
// l + 1 of workgroup h needs only the activation columns of ITS k-half, so one exchange per layer suffices.
//
// Variants, each its own kernel name for rocprofv3 --kernel-trace --stats:
//   chain_one_cu         the reference: rows_linear<1>(chained) per layer (what mlp_hidden does)
//   chain_two_cu         the k-split with the exchange
//   chain_two_cu_nox     the k-split WITHOUT the exchange (wrong numbers; the floor: half the stream + the in-workgroup reduction)
// Numerics of the first two are checked against a CPU double reference.
// usage: microbench_rowchain_split [B=256] [H=256] [NL=10]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "meet.h"
#include "rowchain.h"
using namespace gcrl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(kRowThreads) void chain_one_cu(const float* X, int B, int H, const float* Wt, const float* bias, int NL, float* Y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int R = 4;
  const int ldx = H + 4;
  float* xs0 = lds;
  float* xs1 = xs0 + R * ldx;
  float* part = xs1 + R * ldx;                       // two exchange buffers of the chained form
  const int r0 = blockIdx.x * R;
  for (int i = threadIdx.x; i < R * H; i += kRowThreads) {
    const int r = i / H, c = i - r * H;
    xs0[r * ldx + c] = (r0 + r < B) ? X[(long long)(r0 + r) * H + c] : 0.f;
  }
  __syncthreads();
  float* a = xs0; float* b = xs1;
  for (int l = 0; l < NL; ++l) {
    rows_linear<1>(a, ldx, H, Wt + (long long)l * H * H, H, H, bias + l * H, EPI_LEAKY, part, b, ldx,
                   l == NL - 1 ? Y + (long long)r0 * H : nullptr, H, min(R, B - r0), nullptr, 0, MUL_NONE, true, l & 1);
    float* t = a; a = b; b = t;
  }
}

constexpr int kSc1 = 16;

// grid = 2 * nblk: workgroup 2*blk + h owns reduction rows [h*H/2, (h+1)*H/2) of every layer and finishes output columns
// [h*H/2, (h+1)*H/2) — exactly the activation columns its next pass multiplies.  H <= 256, H % 8 == 0.
template <bool EXCHANGE>
__global__ __launch_bounds__(kRowThreads) void chain_two_cu_kernel(const float* X, int B, int H, const float* Wt, const float* bias, int NL, float* Y,
                                                                  float* xbuf, unsigned long long* ctr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned int s_flag;
  constexpr int R = 4;
  const int ldx = H + 4, Hh = H / 2;
  float* xs0 = lds;
  float* xs1 = xs0 + R * ldx;
  float* part = xs1 + R * ldx;                       // [4 waves][R][kRowChunk]
  const int blk = blockIdx.x >> 1, half = blockIdx.x & 1, other = half ^ 1;
  const int r0 = blk * R, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < R * H; i += kRowThreads) {
    const int r = i / H, c = i - r * H;
    xs0[r * ldx + c] = (r0 + r < B) ? X[(long long)(r0 + r) * H + c] : 0.f;
  }
  __syncthreads();
  float* a = xs0; float* b = xs1;
  // exchange buffers: [blk][parity][sender half][R][Hh] floats: what `sender` computed for the OTHER half's columns
  float* xb = xbuf + (long long)blk * 2 * 2 * R * Hh;
  const int per = ((Hh + 3) / 4 + 3) & ~3;           // this wave's share of the workgroup's k-half
  for (int l = 0; l < NL; ++l) {
    const float* M = Wt + (long long)l * H * H;
    const __amdgpu_buffer_rsrc_t rs = bounded_rsrc(M, (long long)H * H);
    const int jb = half * Hh + wave * per, je = min((half + 1) * Hh, jb + per);
    // epilogue operand first (its latency hides behind the stream): thread t finishes row t / 64... of the own half
    const int fr = tid >> 6, fc = half * Hh + 4 * (tid & 63);          // row, first column this thread finishes (own half: Hh / 4 lanes per row)
    const bool fin = 4 * (tid & 63) < Hh;
    v4f bv = fin ? *(const v4f*)(bias + l * H + fc) : (v4f){0.f, 0.f, 0.f, 0.f};
    v4f acc[1][4];
    rows_matmul_wave<1>(a, ldx, rs, H, 0, jb, je, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *(v4f*)(part + ((wave * R + i) * kRowChunk) + 4 * lane) = (v4f){acc[0][0][i], acc[0][1][i], acc[0][2][i], acc[0][3][i]};
    __syncthreads();
    // in-workgroup reduction over the four waves: thread t sums row t / 64, columns 4 * (t % 64) .. + 3 (all H columns)
    v4f p = (v4f){0.f, 0.f, 0.f, 0.f};
    const int pr = tid >> 6, pc = 4 * (tid & 63);
    if (pc < H) {
      p = *(const v4f*)(part + (0 * R + pr) * kRowChunk + pc);
#pragma unroll
      for (int w = 1; w < 4; ++w) p += *(const v4f*)(part + (w * R + pr) * kRowChunk + pc);
    }
    const bool mine = pc >= half * Hh && pc < (half + 1) * Hh;       // a column this workgroup finishes
    float* out_mine = xb + ((long long)((l & 1) * 2 + half) * R) * Hh;       // what I computed for the other half's columns
    const float* in_other = xb + ((long long)((l & 1) * 2 + other) * R) * Hh; // what the other computed for mine
    if (EXCHANGE) {
      if (pc < H && !mine) {
        const v4u v = {__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(p[3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, bounded_rsrc(out_mine, (long long)R * Hh), (pr * Hh + (pc - other * Hh)) * 4, 0, kSc1);
      }
      // own-half partial sums wait in LDS (the threads that finish a column are not the ones that summed it)
      if (pc < H && mine) *(v4f*)(b + pr * ldx + pc) = p;
      meet(ctr + (long long)blk * 16, 2u, true, &s_flag, nullptr, 0u);
      if (fin) {
        const v4u o = __builtin_amdgcn_raw_buffer_load_b128(bounded_rsrc(in_other, (long long)R * Hh), (fr * Hh + 4 * (tid & 63)) * 4, 0, kSc1);
        v4f v = *(const v4f*)(b + fr * ldx + fc);
        // fixed order: half 0's partial + half 1's partial
        v4f po = (v4f){__uint_as_float(o[0]), __uint_as_float(o[1]), __uint_as_float(o[2]), __uint_as_float(o[3])};
        v = half == 0 ? v + po : po + v;
        v += bv;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.01f * v[q];
        *(v4f*)(b + fr * ldx + fc) = v;
        if (l == NL - 1 && r0 + fr < B) *(v4f*)(Y + (long long)(r0 + fr) * H + fc) = v;
      }
    } else if (pc < H && mine) {
      v4f v = p + (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.01f * v[q];
      *(v4f*)(b + pr * ldx + pc) = v;
    }
    __syncthreads();
    float* t = a; a = b; b = t;
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 256, NL = argc > 3 ? atoi(argv[3]) : 10;
  if (H > 256 || H % 8) { printf("H <= 256 and H %% 8 == 0 required\n"); return 1; }
  std::vector<float> X((size_t)B * H), W((size_t)NL * H * H), bias((size_t)NL * H);
  srand(1);
  for (auto& v : X) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.15f;
  for (auto& v : bias) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  const int R = 4, nblk = (B + R - 1) / R;
  float *dX, *dW, *db, *dY, *dXb;
  unsigned long long* dCtr;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, bias.size() * 4)); CK(hipMalloc(&dY, X.size() * 4));
  CK(hipMalloc(&dXb, (size_t)nblk * 2 * 2 * R * (H / 2) * 4));
  CK(hipMalloc(&dCtr, (size_t)nblk * 16 * 8));
  CK(hipMemset(dCtr, 0, (size_t)nblk * 16 * 8));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  std::vector<double> cur(X.begin(), X.end()), nxt(cur.size());
  for (int l = 0; l < NL; ++l) {
    for (int r = 0; r < B; ++r)
      for (int c = 0; c < H; ++c) {
        double s = bias[l * H + c];
        for (int j = 0; j < H; ++j) s += cur[(size_t)r * H + j] * W[((size_t)l * H + j) * H + c];
        nxt[(size_t)r * H + c] = s > 0 ? s : 0.01 * s;
      }
    cur.swap(nxt);
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds1 = (size_t)(2 * R * (H + 4) + 2 * 4 * R * kRowChunk) * 4, lds2 = (size_t)(2 * R * (H + 4) + 4 * R * kRowChunk) * 4;
  CK(hipFuncSetAttribute((const void*)chain_one_cu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  for (int variant = 0; variant < 3; ++variant) {
    auto launch = [&]() {
      if (variant == 0) hipLaunchKernelGGL(chain_one_cu, dim3(nblk), dim3(kRowThreads), lds1, st, dX, B, H, dW, db, NL, dY);
      else if (variant == 1) hipLaunchKernelGGL(chain_two_cu_kernel<true>, dim3(2 * nblk), dim3(kRowThreads), lds2, st, dX, B, H, dW, db, NL, dY, dXb, dCtr);
      else hipLaunchKernelGGL(chain_two_cu_kernel<false>, dim3(2 * nblk), dim3(kRowThreads), lds2, st, dX, B, H, dW, db, NL, dY, dXb, dCtr);
    };
    CK(hipMemsetAsync(dY, 0, X.size() * 4, st));
    launch();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    std::vector<float> Y(X.size());
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < Y.size(); ++i) { maxerr = fmax(maxerr, fabs(Y[i] - cur[i])); maxref = fmax(maxref, fabs(cur[i])); }
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e0, st));
    const int reps = 300;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const char* names[3] = {"chain_one_cu      (1 WG / 4 rows, whole matrix per pass)", "chain_two_cu      (2 WGs / 4 rows, k-split + 2 KB exchange each way)",
                            "chain_two_cu_nox  (k-split, NO exchange: floor, wrong numbers)"};
    printf("%-70s grid %4d  max|err| %.3e (max|ref| %.3f)  %.2f us/launch  %.2f us/layer (hipEvent, back-to-back launches)\n", names[variant],
           variant == 0 ? nblk : 2 * nblk, maxerr, maxref, ms * 1e3 / reps, ms * 1e3 / reps / NL);
  }
  return 0;
}
