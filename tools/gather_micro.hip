// gather_micro.hip — stand-alone study of the HER sample gather (engine form: index -> record -> sa | nsa | r | d).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_micro.hip -o tools/gather_micro
//   tools/gather_micro <rows> [iters]          (run under rocprofv3 --kernel-trace --stats for the profiler's clock)
// Record layout = her_ring.h at PickAndPlace dims (S=23, A=4): RS = 64 floats, sa <- [0,28), nsa <- [28,52), (r,d) <- 52,53.
// Every variant is a separate kernel name so that the profiler's statistics separate them; every launch reads a fresh
// window of a long random index array (no launch re-reads the rows of the one before).
#include <hip/hip_runtime.h>
#ifdef ENGINE_KERNEL   // the library's own kernel, compiled into this program: -DENGINE_KERNEL -I<csrc> -I<include> + link libgcrl_hip.so
#include "her_ring.hip"
#endif

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const float* ring; const uint32_t* idx; long long n, head, cap;
  int SA4, S4, RS;
  float *sa, *nsa, *r, *d;
};

// ---- V0: the round-2 kernel (per-lane index load, 64-bit modulo, 4 groups of 4 records per wave)
template <int U>
__global__ __launch_bounds__(256) void g_v0(Args p) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  const int o_r = p.SA4 + p.S4, c0 = v4 * 4;
  for (long long r0 = wave_id * (4 * U); r0 < p.n; r0 += nwaves * (4 * U)) {
    float4 val[U]; long long row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = r0 + u * 4 + sub;
      val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row[u] < p.n && c0 < p.RS) {
        const long long phys = (p.head + (long long)p.idx[row[u]]) % p.cap;
        val[u] = *reinterpret_cast<const float4*>(p.ring + phys * p.RS + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] >= p.n) continue;
      const long long ro = row[u] * p.SA4;
      if (c0 < p.SA4) *reinterpret_cast<float4*>(p.sa + ro + c0) = val[u];
      else if (c0 < o_r) *reinterpret_cast<float4*>(p.nsa + row[u] * p.S4 + (c0 - p.SA4)) = val[u];
      else if (c0 == o_r) { p.r[row[u]] = val[u].x; p.d[row[u]] = val[u].y; }
    }
  }
}

__device__ inline float4 ld4(const float* p, bool nt) {
  if (nt) {
    float4 v;
    v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3);
    return v;
  }
  return *reinterpret_cast<const float4*>(p);
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ inline void st4(float* p, float4 v, bool nt) {
  if (nt) { f4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p)); }
  else *reinterpret_cast<float4*>(p) = v;
}
__device__ inline float4 ld4v(const float* p, bool nt) {
  if (nt) { f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const float4*>(p);
}

// ---- V1: a wave's 4*U indices arrive in ONE coalesced load (lane l <- idx[r0 + l]) and are handed to the
// 16-lane groups by a cross-lane read; wrap by compare-and-subtract; exact grid (one pass per wave); lanes that
// would fetch only padding do not load.  NTL / NTS: non-temporal record loads / batch stores.
template <int U, bool NTL, bool NTS, int THREADS>
__global__ __launch_bounds__(THREADS) void g_v1(Args p) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = (long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  const int o_r = p.SA4 + p.S4, c0 = v4 * 4;
  const long long r0 = wave_id * (4 * U);
  if (r0 >= p.n) return;
  unsigned long long phys = 0;
  if (lane < 4 * U && r0 + lane < p.n) {
    phys = (unsigned long long)p.head + p.idx[r0 + lane];
    if (phys >= (unsigned long long)p.cap) phys -= p.cap;
  }
  const uint32_t ph32 = (uint32_t)phys;
  float4 val[U];
  const bool useful = c0 <= o_r;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t ph = __shfl(ph32, u * 4 + sub, 64);
    val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (useful && r0 + u * 4 + sub < p.n) val[u] = ld4v(p.ring + (size_t)ph * p.RS + c0, NTL);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = r0 + u * 4 + sub;
    if (row >= p.n) continue;
    if (c0 < p.SA4) st4(p.sa + row * p.SA4 + c0, val[u], NTS);
    else if (c0 < o_r) st4(p.nsa + row * p.S4 + (c0 - p.SA4), val[u], NTS);
    else if (c0 == o_r) { p.r[row] = val[u].x; p.d[row] = val[u].y; }
  }
}

// ---- copy bound: the same bytes with the records read IN ORDER (no index, no randomness): what a streaming kernel of this
// size and shape costs by the same clock
template <int U>
__global__ __launch_bounds__(256) void g_copy(Args p) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int o_r = p.SA4 + p.S4, c0 = v4 * 4;
  const long long r0 = wave_id * (4 * U);
  if (r0 >= p.n) return;
  float4 val[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = r0 + u * 4 + sub;
    val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < p.n) val[u] = *reinterpret_cast<const float4*>(p.ring + ((size_t)(p.head + row) % p.cap) * p.RS + c0);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = r0 + u * 4 + sub;
    if (row >= p.n) continue;
    if (c0 < p.SA4) *reinterpret_cast<float4*>(p.sa + row * p.SA4 + c0) = val[u];
    else if (c0 < o_r) *reinterpret_cast<float4*>(p.nsa + row * p.S4 + (c0 - p.SA4)) = val[u];
    else if (c0 == o_r) { p.r[row] = val[u].x; p.d[row] = val[u].y; }
  }
}

// ---- V2: V1 with the wave's indices fetched by SCALAR loads (the wave's 4*U indices are wave-uniform addresses: one
// s_load_dwordx16 through the scalar cache instead of a vector-memory round trip), non-temporal stores.  A wave whose rows
// run past n takes the V1 form.
template <int U, int THREADS>
__global__ __launch_bounds__(THREADS) void g_v2(Args p) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)));
  const int o_r = p.SA4 + p.S4, c0 = v4 * 4;
  const long long r0 = wave_id * (4 * U);
  if (r0 >= p.n) return;
  float4 val[U];
  const bool useful = c0 <= o_r;
  const uint32_t head = (uint32_t)p.head, cap = (uint32_t)p.cap;
  if (r0 + 4 * U <= p.n) {
    uint32_t ii[4 * U];
    const uint32_t* ip = p.idx + r0;
#pragma unroll
    for (int j = 0; j < 4 * U; ++j) ii[j] = ip[j];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t mine = sub == 0 ? ii[u * 4] : sub == 1 ? ii[u * 4 + 1] : sub == 2 ? ii[u * 4 + 2] : ii[u * 4 + 3];
      uint32_t ph = head + mine;          // head, index < cap < 2^31
      if (ph >= cap) ph -= cap;
      val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (useful) val[u] = ld4v(p.ring + (size_t)ph * p.RS + c0, false);
    }
  } else {
    uint32_t ph32 = 0;
    if (lane < 4 * U && r0 + lane < p.n) { ph32 = head + p.idx[r0 + lane]; if (ph32 >= cap) ph32 -= cap; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t ph = __shfl(ph32, u * 4 + sub, 64);
      val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (useful && r0 + u * 4 + sub < p.n) val[u] = ld4v(p.ring + (size_t)ph * p.RS + c0, false);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = r0 + u * 4 + sub;
    if (row >= p.n) continue;
    if (c0 < p.SA4) st4(p.sa + row * p.SA4 + c0, val[u], true);
    else if (c0 < o_r) st4(p.nsa + row * p.S4 + (c0 - p.SA4), val[u], true);
    else if (c0 == o_r) { __builtin_nontemporal_store(val[u].x, p.r + row); __builtin_nontemporal_store(val[u].y, p.d + row); }
  }
}

// copy bound with non-temporal stores
template <int U>
__global__ __launch_bounds__(256) void g_copy_nts(Args p) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, v4 = lane & 15;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int o_r = p.SA4 + p.S4, c0 = v4 * 4;
  const long long r0 = wave_id * (4 * U);
  if (r0 >= p.n) return;
  float4 val[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = r0 + u * 4 + sub;
    val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < p.n) val[u] = *reinterpret_cast<const float4*>(p.ring + ((size_t)(p.head + row) % p.cap) * p.RS + c0);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = r0 + u * 4 + sub;
    if (row >= p.n) continue;
    if (c0 < p.SA4) st4(p.sa + row * p.SA4 + c0, val[u], true);
    else if (c0 < o_r) st4(p.nsa + row * p.S4 + (c0 - p.SA4), val[u], true);
    else if (c0 == o_r) { p.r[row] = val[u].x; p.d[row] = val[u].y; }
  }
}

// ---- V3: V1 + the wave's 16 records pass through a wave-private LDS tile so that every store instruction is a full-width
// dwordx4 over CONTIGUOUS output bytes (sa rows r0..r0+15 = 1792 B, nsa 1536 B, r 64 B, d 64 B): 4 store instructions per wave
// instead of 4 partly masked ones + 8 one-dword ones.  Lanes of one instruction may point into different arrays.
template <bool NTS>
__global__ __launch_bounds__(256) void g_v3(Args p) {
  __shared__ float tile[4][16][60];     // per wave: 16 records x (52 + r,d + pad); row stride 240 B
  const int lane = threadIdx.x & 63, sub = lane >> 4, v4 = lane & 15, w = threadIdx.x >> 6;
  const long long wave_id = (long long)blockIdx.x * 4 + w;
  const int c0 = v4 * 4;
  const long long r0 = wave_id * 16;
  if (r0 >= p.n) return;
  uint32_t ph32 = 0;
  if (lane < 16 && r0 + lane < p.n) {
    unsigned long long phys = (unsigned long long)p.head + p.idx[r0 + lane];
    if (phys >= (unsigned long long)p.cap) phys -= p.cap;
    ph32 = (uint32_t)phys;
  }
  float4 val[4];
  const bool useful = c0 <= 52;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t ph = __shfl(ph32, u * 4 + sub, 64);
    val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (useful && r0 + u * 4 + sub < p.n) val[u] = *reinterpret_cast<const float4*>(p.ring + (size_t)ph * p.RS + c0);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (useful) *reinterpret_cast<float4*>(&tile[w][u * 4 + sub][c0]) = val[u];
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes (no other wave touches this tile)
  __builtin_amdgcn_wave_barrier();
  if (r0 + 16 <= p.n) {
    float* sa = p.sa + r0 * 28; float* nsa = p.nsa + r0 * 24;
    // A: sa quads 0..63
    { const int q = lane; st4(sa + q * 4, *reinterpret_cast<const float4*>(&tile[w][q / 7][(q % 7) * 4]), NTS); }
    // B: sa quads 64..111 | nsa quads 0..15
    if (lane < 48) { const int q = 64 + lane; st4(sa + q * 4, *reinterpret_cast<const float4*>(&tile[w][q / 7][(q % 7) * 4]), NTS); }
    else { const int q = lane - 48; st4(nsa + q * 4, *reinterpret_cast<const float4*>(&tile[w][q / 6][28 + (q % 6) * 4]), NTS); }
    // C: nsa quads 16..79
    { const int q = 16 + lane; st4(nsa + q * 4, *reinterpret_cast<const float4*>(&tile[w][q / 6][28 + (q % 6) * 4]), NTS); }
    // D: nsa quads 80..95 | r (4 quads) | d (4 quads)
    if (lane < 16) { const int q = 80 + lane; st4(nsa + q * 4, *reinterpret_cast<const float4*>(&tile[w][q / 6][28 + (q % 6) * 4]), NTS); }
    else if (lane < 24) {
      const int j = (lane - 16) & 3, col = 52 + ((lane - 16) >> 2);
      const float4 v = make_float4(tile[w][4 * j][col], tile[w][4 * j + 1][col], tile[w][4 * j + 2][col], tile[w][4 * j + 3][col]);
      st4((lane < 20 ? p.r : p.d) + r0 + 4 * j, v, NTS);
    }
  } else {
    for (int u = 0; u < 4; ++u) {
      const long long row = r0 + u * 4 + sub;
      if (row >= p.n) continue;
      if (c0 < 28) st4(p.sa + row * 28 + c0, val[u], NTS);
      else if (c0 < 52) st4(p.nsa + row * 24 + (c0 - 28), val[u], NTS);
      else if (c0 == 52) { p.r[row] = val[u].x; p.d[row] = val[u].y; }
    }
  }
}

__global__ void g_empty(Args) {}
// streams through a buffer larger than the infinity cache: what 40 update steps of other traffic do between two gathers
__global__ __launch_bounds__(256) void g_thrash(float4* buf, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = buf[i]; v.x += 1.f; buf[i] = v; }
}

int main(int argc, char** argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 77824;
  const int iters = argc > 2 ? atoi(argv[2]) : 40;
  const size_t thrash_mb = argc > 3 ? (size_t)atoll(argv[3]) : 0;   // > 0: evict the caches before every launch
  const long long cap = 1000000;
  const int SA4 = 28, S4 = 24, RS = 64;
  float *ring, *sa, *nsa, *r, *d; uint32_t* idx;
  const size_t nidx = (size_t)rows * 8 + 1024;
  CK(hipMalloc(&ring, (size_t)cap * RS * 4)); CK(hipMalloc(&sa, (size_t)rows * SA4 * 4)); CK(hipMalloc(&nsa, (size_t)rows * S4 * 4));
  CK(hipMalloc(&r, rows * 4)); CK(hipMalloc(&d, rows * 4)); CK(hipMalloc(&idx, nidx * 4));
  CK(hipMemset(ring, 0, (size_t)cap * RS * 4));
  std::vector<uint32_t> h(nidx);
  uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % cap); }
  CK(hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float4* tb = nullptr; const size_t tn4 = thrash_mb * (1u << 20) / 16;
  if (thrash_mb) { CK(hipMalloc(&tb, tn4 * 16)); CK(hipMemset(tb, 0, tn4 * 16)); }
  auto thrash = [&]() { if (tb) hipLaunchKernelGGL(g_thrash, dim3(4096), dim3(256), 0, st, tb, tn4); };
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double alg = 416.0 * rows;
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 3; ++w) { thrash(); launch(w % 8); }
    CK(hipStreamSynchronize(st));
    double tot = 0, best = 1e9;
    for (int i = 0; i < iters; ++i) {
      thrash(); CK(hipEventRecord(a, st)); launch(i % 8); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms; if (ms < best) best = ms;
    }
    printf("%-28s rows %8lld  event avg %7.2f us  min %7.2f us  alg %6.2f TB/s (avg)\n", name, rows, tot / iters * 1e3, best * 1e3,
           alg / (tot / iters * 1e-3) / 1e12);
  };
  auto args = [&](int win) { return Args{ring, idx + (size_t)win * rows, rows, 123457, cap, SA4, S4, RS, sa, nsa, r, d}; };
  auto blocks = [&](int rows_per_block) { return (int)((rows + rows_per_block - 1) / rows_per_block); };
  run("empty", [&](int w) { hipLaunchKernelGGL(g_empty, dim3(1), dim3(64), 0, st, args(w)); });
  run("v0_u4 (round 2)", [&](int w) { hipLaunchKernelGGL(g_v0<4>, dim3(std::min(blocks(64), 8192)), dim3(256), 0, st, args(w)); });
  run("v1_u4", [&](int w) { hipLaunchKernelGGL((g_v1<4, false, false, 256>), dim3(blocks(64)), dim3(256), 0, st, args(w)); });
  run("v1_u4_nts", [&](int w) { hipLaunchKernelGGL((g_v1<4, false, true, 256>), dim3(blocks(64)), dim3(256), 0, st, args(w)); });
  run("v1_u4_nts_t128", [&](int w) { hipLaunchKernelGGL((g_v1<4, false, true, 128>), dim3(blocks(32)), dim3(128), 0, st, args(w)); });
  run("v1_u4_nts_t512", [&](int w) { hipLaunchKernelGGL((g_v1<4, false, true, 512>), dim3(blocks(128)), dim3(512), 0, st, args(w)); });
  run("v1_u2_nts", [&](int w) { hipLaunchKernelGGL((g_v1<2, false, true, 256>), dim3(blocks(32)), dim3(256), 0, st, args(w)); });
  run("v1_u8_nts", [&](int w) { hipLaunchKernelGGL((g_v1<8, false, true, 256>), dim3(blocks(128)), dim3(256), 0, st, args(w)); });
  run("v3 (lds, contiguous stores)", [&](int w) { hipLaunchKernelGGL(g_v3<false>, dim3(blocks(64)), dim3(256), 0, st, args(w)); });
  run("v3_nts", [&](int w) { hipLaunchKernelGGL(g_v3<true>, dim3(blocks(64)), dim3(256), 0, st, args(w)); });
#ifdef ENGINE_KERNEL
  {
    float* nsa_w; CK(hipMalloc(&nsa_w, (size_t)rows * SA4 * 4));     // the engine's nsa rows are SA4 wide
    auto eargs = [&](int win) { return GatherUpdArgs{ring, idx + (size_t)win * rows, gcrl::IdxGen{}, rows, 123457, cap, SA4, S4, RS, SA4, sa, nsa_w, nullptr, r, d, nullptr, nullptr, 0}; };
    const size_t lds = 4 * ((size_t)32 * SA4 + 32) * sizeof(float);
    run("ENGINE her_gather_update_kernel", [&](int w) { hipLaunchKernelGGL(her_gather_update_kernel<false>, dim3(blocks(64)), dim3(256), lds, st, eargs(w)); });
  }
#endif
  run("copy_u4_nts (bound)", [&](int w) { hipLaunchKernelGGL(g_copy_nts<4>, dim3(blocks(64)), dim3(256), 0, st, args(w)); });
  run("copy_u4 (in-order bound)", [&](int w) { hipLaunchKernelGGL(g_copy<4>, dim3(blocks(64)), dim3(256), 0, st, args(w)); });
  return 0;
}
