// microbench (round 4, after the two-CU split did not pay): ONE dependent chain of H x H layer passes for B batch rows as a
// WEIGHT-SLICE form — a workgroup owns the 16 x 16 output tile (row block rb, column block cb) of EVERY layer, so per layer it
// streams 16 KB of the weight matrix (its 16 columns; independent of computed data: requested before the wait) and the 16 KB
// of its row block's activations, instead of the whole 256 KB matrix per 4 rows (csrc/rowchain.h).  The price is an exchange
// per layer between the H / 16 workgroups of a row block.  Variants, each its own kernel name for rocprofv3:
//   tile_sentinel<xcd>   the activations themselves are the flags: a consumer polls its A-operand loads (agent-scope) until no
//                        dword is the sentinel 0xFFFFFFFF; a producer resets its tile of layer l - 1's buffer once layer l + 1's
//                        inputs have arrived (all readers of l - 1 are done), the last one at the start of the next launch
//   tile_counter<xcd>    meeting counter per (layer, row block) (csrc/meet.h protocol: drained stores, monotonic counter)
//   tile_nox             no wait at all (wrong numbers; the floor)
//   <xcd> = 1: the workgroups of a row block sit on ONE XCD (workgroup w runs on XCD w % 8), 0: in launch order
// usage: microbench_rowtile [B=256] [H=256] [NL=10]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr unsigned int kSent = 0xFFFFFFFFu;
constexpr int kSpinMax = 1 << 16;

__device__ inline __amdgpu_buffer_rsrc_t rsrc_of(const void* p, long long bytes) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// MODE 0: sentinel, 1: counter, 2: no exchange, 3: sentinel with a workgroup vote (__syncthreads_or) per poll iteration.  LD_AUX / ST_AUX: cache-policy bits of the exchange loads / stores (16 = sc1:
// agent scope; 1 = sc0).  WAVES: waves per workgroup (k split WAVES ways).
template <int MODE, int XCD, int LD_AUX, int ST_AUX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void tile_chain(const float* X0, int B, int H, const float* Wt, const float* bias, int NL, float* Y,
                                                         float* xb, unsigned long long* ctr, unsigned long long round, unsigned int* fail) {
  __shared__ __attribute__((aligned(16))) float part[2][WAVES][256];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncb = H >> 4, nrb = B >> 4;
  int rb, cb;
  if (XCD) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    rb = (slot / ncb) * 8 + xcd;
    cb = slot % ncb;
  } else {
    rb = blockIdx.x / ncb;
    cb = blockIdx.x % ncb;
  }
  (void)nrb;
  const long long BH = (long long)B * H;
  const int kper = H / WAVES;            // this wave's k share (multiple of 16)
  const int nt = kper >> 4;              // 16-k groups per wave
  // own tile, as wave 0 writes it: lane -> row lane >> 2, columns 4 * (lane & 3) .. + 3
  const int orow = rb * 16 + (lane >> 2), ocol = cb * 16 + 4 * (lane & 3);
  if ((MODE == 0 || MODE == 3) && wave == 0 && NL >= 2) {
    // last launch's final exchange tile (input of layer NL - 1): nobody could tell when its readers were done
    const v4u s = {kSent, kSent, kSent, kSent};
    __builtin_amdgcn_raw_buffer_store_b128(s, rsrc_of(xb + (long long)(NL - 1) * BH, BH * 4), (orow * H + ocol) * 4, 0, ST_AUX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (int l = 0; l < NL; ++l) {
    // B operand: the layer's 16 columns, rows of this wave's k share (k-permuted: MFMA (t, s) consumes k = kb + 16 t + 4 lg + s)
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(Wt + (long long)l * H * H, (long long)H * H * 4);
    float bw[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        bw[t][s] = t < nt ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, ((wave * kper + 16 * t + 4 * lg + s) * H + cb * 16 + li) * 4, 0, 0)) : 0.f;
    const v4f bv = *(const v4f*)(bias + l * H + ocol);
    // A operand: row rb * 16 + li, k = kb + 16 t + 4 lg .. + 3
    const float* Xin = l == 0 ? X0 : xb + (long long)l * BH;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(Xin, BH * 4);
    const int aoff = ((rb * 16 + li) * H + wave * kper + 4 * lg) * 4;
    v4u a[4];
    if (MODE == 1 && l > 0) {
      const unsigned long long target = (round + 1ull) * (unsigned long long)ncb;
      const unsigned long long* c = ctr + ((long long)l * (B >> 4) + rb) * 16;
      int spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kSpinMax) __builtin_amdgcn_s_sleep(1);
      if (spins >= kSpinMax && lane == 0) atomicOr(fail, 1u);
    }
    int spins = 0;
    for (;;) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < nt) {
          if (l == 0) a[t] = __builtin_amdgcn_raw_buffer_load_b128(rx, aoff + 64 * t, 0, 0);
          else a[t] = __builtin_amdgcn_raw_buffer_load_b128(rx, aoff + 64 * t, 0, LD_AUX);
        }
      if ((MODE != 0 && MODE != 3) || l == 0) break;
      bool bad = false;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < nt) bad = bad || a[t][0] == kSent || a[t][1] == kSent || a[t][2] == kSent || a[t][3] == kSent;
      if (MODE == 3) { if (!__syncthreads_or(__any(bad) ? 1 : 0)) break; }
      else if (!__any(bad)) break;
      if (++spins >= kSpinMax) { if (lane == 0) atomicOr(fail, 2u); break; }
      __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // transposed product: lane li = output row, acc[r] = column 4 lg + r
          if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][s], __uint_as_float(a[t][s]), acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][s], __uint_as_float(a[t][s]), acc0, 0, 0, 0);
        }
      }
    *(v4f*)(&part[l & 1][wave][li * 16 + 4 * lg]) = acc0 + acc1;
    __syncthreads();
    if (wave == 0) {
      if ((MODE == 0 || MODE == 3) && l >= 2) {
        // ALL waves' inputs of layer l have arrived (the barrier): every workgroup of the row block is past its reads of layer l - 1's input
        const v4u s = {kSent, kSent, kSent, kSent};
        __builtin_amdgcn_raw_buffer_store_b128(s, rsrc_of(xb + (long long)(l - 1) * BH, BH * 4), (orow * H + ocol) * 4, 0, ST_AUX);
      }
      v4f v = *(const v4f*)(&part[l & 1][0][4 * lane]);
#pragma unroll
      for (int w = 1; w < WAVES; ++w) v += *(const v4f*)(&part[l & 1][w][4 * lane]);
      v += bv;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.01f * v[q];
      const v4u o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      if (l == NL - 1) {
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_of(Y, BH * 4), (orow * H + ocol) * 4, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_of(xb + (long long)(l + 1) * BH, BH * 4), (orow * H + ocol) * 4, 0, ST_AUX);
        if (MODE == 1) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(ctr + ((long long)(l + 1) * (B >> 4) + rb) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 256, NL = argc > 3 ? atoi(argv[3]) : 10;
  if (H > 256 || H % 64 || B % 128) { printf("H <= 256, H %% 64 == 0, B %% 128 == 0 required\n"); return 1; }
  std::vector<float> X((size_t)B * H), W((size_t)NL * H * H), bias((size_t)NL * H);
  srand(1);
  for (auto& v : X) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.15f;
  for (auto& v : bias) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  float *dX, *dW, *db, *dY, *dXb;
  unsigned long long* dCtr;
  unsigned int* dFail;
  const size_t xb_floats = (size_t)(NL + 1) * B * H;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, bias.size() * 4)); CK(hipMalloc(&dY, X.size() * 4));
  CK(hipMalloc(&dXb, xb_floats * 4));
  const size_t nctr = (size_t)(NL + 1) * (B / 16) * 16;
  CK(hipMalloc(&dCtr, nctr * 8));
  CK(hipMalloc(&dFail, 4));
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
  const int grid = (B / 16) * (H / 16);
  struct V { const char* name; int id; };
  const V vs[] = {{"tile_sentinel xcd=1 sc1/sc1 4 waves", 0}, {"tile_sentinel xcd=0 sc1/sc1 4 waves", 1}, {"tile_counter  xcd=1 sc1/sc1 4 waves", 2},
                  {"tile_nox      xcd=1          4 waves", 3}, {"tile_sentinel xcd=1 sc0/plain 4 waves (same-XCD L2 only)", 4},
                  {"tile_sentinel xcd=1 sc1/sc1 8 waves", 5}, {"tile_sentinel xcd=1 sys/sys 4 waves", 6},
                  {"tile_sentinel xcd=1 sc1 loads / PLAIN stores (same-XCD L2; placement-dependent)", 7},
                  {"tile_counter  xcd=1 sc1 loads / PLAIN stores (placement-dependent)", 8},
                  {"tile_sentinel + workgroup vote per poll, xcd=1 sc1 loads / PLAIN stores", 9}};
  for (const V& v : vs) {
    unsigned long long round = 0;
    CK(hipMemset(dCtr, 0, nctr * 8));
    CK(hipMemset(dFail, 0, 4));
    CK(hipMemset(dXb, 0xFF, xb_floats * 4));
    CK(hipDeviceSynchronize());
    auto launch = [&]() {
      switch (v.id) {
        case 0: hipLaunchKernelGGL((tile_chain<0, 1, 16, 16, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 1: hipLaunchKernelGGL((tile_chain<0, 0, 16, 16, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 2: hipLaunchKernelGGL((tile_chain<1, 1, 16, 16, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 3: hipLaunchKernelGGL((tile_chain<2, 1, 0, 0, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 4: hipLaunchKernelGGL((tile_chain<0, 1, 1, 0, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 5: hipLaunchKernelGGL((tile_chain<0, 1, 16, 16, 8>), dim3(grid), dim3(512), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 7: hipLaunchKernelGGL((tile_chain<0, 1, 16, 0, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 8: hipLaunchKernelGGL((tile_chain<1, 1, 16, 0, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 9: hipLaunchKernelGGL((tile_chain<3, 1, 16, 0, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
        case 6: hipLaunchKernelGGL((tile_chain<0, 1, 17, 17, 4>), dim3(grid), dim3(256), 0, st, dX, B, H, dW, db, NL, dY, dXb, dCtr, round, dFail); break;
      }
      ++round;
    };
    CK(hipMemsetAsync(dY, 0, X.size() * 4, st));
    launch();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    std::vector<float> Y(X.size());
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < Y.size(); ++i) { maxerr = fmax(maxerr, fabs(Y[i] - cur[i])); maxref = fmax(maxref, fabs(cur[i])); }
    { unsigned int f0 = 0; CK(hipMemcpy(&f0, dFail, 4, hipMemcpyDeviceToHost));
      if (f0) { printf("%-60s FIRST LAUNCH TIMED OUT (%u), max|err| %.3e: skipped\n", v.name, f0, maxerr); continue; } }
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e0, st));
    const int reps = 300;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    // numerics again after the replays (the sentinel resets must have left every buffer ready)
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double maxerr2 = 0;
    for (size_t i = 0; i < Y.size(); ++i) maxerr2 = fmax(maxerr2, fabs(Y[i] - cur[i]));
    unsigned int f = 0;
    CK(hipMemcpy(&f, dFail, 4, hipMemcpyDeviceToHost));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s grid %4d  max|err| %.3e / %.3e after replays (max|ref| %.3f) timeouts %u  %.2f us/launch  %.2f us/layer\n", v.name, grid, maxerr, maxerr2,
           maxref, f, ms * 1e3 / reps, ms * 1e3 / reps / NL);
    fflush(stdout);
  }
  return 0;
}
