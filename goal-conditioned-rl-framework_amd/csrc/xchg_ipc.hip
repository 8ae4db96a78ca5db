// xchg_ipc.hip — the data-parallel gradient exchange as ONE kernel of the engine's own launch sequence: a two-shot
// all-reduce over IPC-mapped gradient arenas, peer to peer, that returns the clip norm's partial sums with the reduced bytes.
//
// New design — the reference is single-process; BASELINE.json's north star asks for the gradient
// all-reduce over xGMI.  Round 2-3 issued a stock ncclAllReduce of the flat block (csrc/dp_rccl.cc, still selectable) plus a
// sum-of-squares launch: a library collective on the critical path of a 58 us step, outside hipGraph capture, with the
// control-advance riders and multi-step graphs of the single-GPU path switched off around it.
//
// Shape of the exchange (world W <= 8 ranks of ONE node, one process per GPU):
//   * every rank's gradient arena (one hipMalloc: all critics | actor | log_alpha) and a small control block are exported with
//     hipIpcGetMemHandle and mapped by every peer (handles travel through whatever the host side has: torch.distributed's
//     store).  xGMI is point to point: each rank talks to each peer directly, one hop, no ring.
//   * the block is cut into 1024-float chunks, net by net; chunk c belongs to rank c mod W.  The owner reads the chunk from
//     EVERY rank's arena (W loads in flight per lane), adds them in RANK ORDER — the same sum whoever computes it, so the
//     replicas stay bitwise identical — forms the chunk's sum of squares, and writes chunk and partial back into every rank's
//     RECEIVE BUFFER / partial array.  The receive buffer (and the control block) is FINE-GRAINED device memory
//     (hipExtMallocWithFlags): a peer's stores into coarse-grained memory could hide behind stale lines of the local L2, which a
//     kernel boundary at agent scope does not invalidate for local memory — the arena itself is only ever READ remotely, after the
//     kernel boundary that follows the owner's backward pass.  The optimiser launch that follows reads the reduced gradients
//     from the receive buffer and ~N/1024 partials per net: no separate norm launch.  (A world of one: nothing moves, the
//     optimiser reads the arena, the kernel only forms the partials.)
//   * ordering by two monotonic 64-bit counters per (rank, source) pair, each on its own 128-byte line of the TARGET's control
//     block: ready[src] ("src's gradients of exchange e are complete": stream order on src, announced by its first workgroup),
//     done[src] ("src has delivered every chunk it owns for exchange e").  A rank reads peers only after their `ready`, and the
//     kernel ends only after every peer's `done`: the kernel boundary then publishes the peers' writes to the optimiser launch,
//     and a rank's next backward pass cannot overwrite gradients a peer is still reading.  The exchange number lives in device
//     memory and is advanced by the kernel itself, so the launch is capturable in hipGraphs (multi-step graphs stay on).
//   * all cross-device traffic uses system-scope (sc0 sc1) loads / stores; every storing wave drains (`s_waitcnt vmcnt(0)`)
//     before its workgroup's arrival (meet.h); waits are bounded and report through the handle's status word.
// Nothing here has been timed over xGMI (1-GPU boxes): proven bitwise against the RCCL / gloo result with two processes on one
// GPU (tests/test_gpu_dp.py), world-size-1 cost measured (profiles/r04_dp_overhead_world1.json).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "meet.h"
#include "ops.h"
#include "xchg_ipc.h"

namespace gcrl {

namespace {

constexpr int kSys = 17;                       // sc0 | sc1 of the raw buffer builtins: system scope (gfx94x / gfx950)
constexpr long long kFlagStride = 16;          // 64-bit words per flag: one 128-byte line each
constexpr int kMaxGrid = 1024;                 // workgroups of one rank per exchange
// Bound of a wait for a PEER (polls of a few us each: about a minute).  Far longer than a wait inside one launch (kMeetSpinMax,
// ~1 s): ranks reach their first exchange seconds apart (ring fill, graph capture), and a collective library would wait for ever.
constexpr int kPeerSpinMax = 1 << 24;
constexpr int kMaxSeg = 12;                    // segments of one handle (8 critics + actor + log_alpha + spare)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// control block of a rank (IPC-exported; 64-bit words):
//   ready[src]      at src * kFlagStride                      "src's gradients of exchange e are complete"
//   epoch           at kXchgMaxWorld * kFlagStride             exchanges this rank has completed (local)
//   done[src][wg]   at (kXchgMaxWorld + 1) * kFlagStride + src * kMaxGrid + wg     "workgroup wg of src has delivered its chunks of exchange e"
//   partials        floats, after the flags
constexpr long long kEpochWord = (long long)kXchgMaxWorld * kFlagStride;
constexpr long long kDoneWord = (long long)(kXchgMaxWorld + 1) * kFlagStride;
constexpr long long kFlagWords = kDoneWord + (long long)kXchgMaxWorld * kMaxGrid;

struct XArgs {
  float* arena[kXchgMaxWorld];                 // every rank's gradient arena (own: the local pointer): read only
  float* recv[kXchgMaxWorld];                  // every rank's receive buffer (fine-grained): the reduced chunks land here
  unsigned long long* ctl[kXchgMaxWorld];      // every rank's control block
  float* parts[kXchgMaxWorld];                 // every rank's partial array
  long long seg_off[kMaxSeg]; int seg_n[kMaxSeg]; int seg_c0[kMaxSeg + 1];   // segments: offset / floats / first chunk (seg_c0[nseg] = all chunks)
  int nseg, world, rank, c0, c1;               // chunks [c0, c1) take part
  unsigned int* status;
};

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// wave-uniform descriptor over [p, p + nfloats): loads past the extent return 0, stores past it are dropped (per dword)
__device__ inline __amdgpu_buffer_rsrc_t rsrc_n(const float* p, int nfloats) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(nfloats * 4), 0x00020000);
}

// workgroups rank r runs for an exchange of nch chunks: it owns chunks r, r + W, ... (the host launches exactly this many)
__host__ __device__ inline int grid_of(int nch, int r, int W) {
  const int mine = nch > r ? (nch - r + W - 1) / W : 0;
  return mine < 1 ? 1 : (mine > kMaxGrid ? kMaxGrid : mine);
}

template <bool SOLO>   // SOLO: a world of one — nobody to wait for, nothing to rewrite: the partials only
__global__ __launch_bounds__(256) void xchg_two_shot_kernel(XArgs a) {
  __shared__ float red[4];
  __shared__ unsigned int s_ok;
  const int W = SOLO ? 1 : a.world, me = SOLO ? 0 : a.rank, tid = threadIdx.x;
  constexpr bool solo = SOLO;
  unsigned long long* ctl = a.ctl[me];
  // the exchange this launch performs: device state, advanced by workgroup 0 once every workgroup of every rank — this rank's
  // included, all of which have read it by then — has reported in
  unsigned long long e = 0;
  if (!solo) e = __hip_atomic_load(ctl + kEpochWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  if (!solo && blockIdx.x == 0 && tid < W && tid != me)              // my gradients are complete (stream order): tell every peer
    st_sys(a.ctl[tid] + (long long)me * kFlagStride, e);
  if (tid == 0) s_ok = 1u;
  __syncthreads();
  if (!solo && tid < W && tid != me) {                               // every peer's gradients are complete
    int spins = 0;
    while (ld_sys(ctl + (long long)tid * kFlagStride) < e && ++spins < kPeerSpinMax) __builtin_amdgcn_s_sleep(8);
    if (spins >= kPeerSpinMax) { s_ok = 0u; if (a.status) __hip_atomic_fetch_or(a.status, (unsigned)MEET_ERR_XCHG_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
  if (!solo) __syncthreads();
  const bool ok = s_ok != 0u;
  const int wave = tid >> 6, lane = tid & 63;
  const int boff = 16 * tid;                                         // this lane's four floats of a chunk (byte offset)
  for (int c = a.c0 + me + W * (int)blockIdx.x; c < a.c1; c += W * (int)gridDim.x) {
    // chunk c of the handle's table, from the segment list in the kernel arguments (scalar work: no dependent memory access)
    int sg = 0;
#pragma unroll
    for (int i = 1; i < kMaxSeg; ++i) if (i < a.nseg && c >= a.seg_c0[i]) sg = i;
    const int first = (c - a.seg_c0[sg]) * kXchgChunk;
    const long long off = a.seg_off[sg] + first;
    const int n = min(kXchgChunk, a.seg_n[sg] - first);
    // branch-free over the 8 possible peers: a rank beyond the world gets a descriptor of extent 0 — its load returns zeros
    // without touching memory, its store is dropped — so that all W loads are in flight together (a branch per peer made the
    // compiler wait for each load before the next: W serial round trips over the fabric)
    v4u v[kXchgMaxWorld];
#pragma unroll
    for (int q = 0; q < kXchgMaxWorld; ++q)
      v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_n(a.arena[q < W ? q : 0] + off, q < W ? n : 0), boff, 0, SOLO ? 0 : kSys);
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[u] = __uint_as_float(v[0][u]);
#pragma unroll
      for (int q = 1; q < kXchgMaxWorld; ++q) {                      // rank order; (x + 0 would turn a -0 into +0: select instead)
        const float t = __fadd_rn(s[u], __uint_as_float(v[q][u]));
        s[u] = q < W ? t : s[u];
      }
    }
    if (!ok) { s[0] = s[1] = s[2] = s[3] = __builtin_nanf(""); }     // a timed-out wait must not pass for a result
    const v4u r = {__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])};
#pragma unroll
    for (int q = 0; q < kXchgMaxWorld; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(r, rsrc_n(a.recv[q < W ? q : 0] + off, (q < W && !solo) ? n : 0), boff, 0, kSys);   // (past the chunk: dropped)
    // the chunk's sum of squares, in a fixed order: lane's four (zeros past the chunk), wave tree, four waves
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (4 * tid + u < n) ss = __fadd_rn(ss, __fmul_rn(s[u], s[u]));
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) ss = __fadd_rn(ss, __shfl_xor(ss, off2, 64));
    __syncthreads();                                                 // (red[] of the previous chunk has been read)
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (tid < W) {
      const float tot = __fadd_rn(__fadd_rn(red[0], red[1]), __fadd_rn(red[2], red[3]));
      if (solo) a.parts[0][c] = tot;                                  // (the kernel boundary publishes it)
      else __hip_atomic_store(a.parts[tid] + c, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (solo) return;
  drain_stores();                                                    // every wave: its write-through (sc0 sc1) stores have been acknowledged
  __syncthreads();
  // this workgroup's chunks are everywhere: say so to every rank, itself included — one 8-byte write-through store per
  // rank, no atomic (arrivals on one line serialise at a memory round trip each: 272 tickets were most of the launch)
  if (tid < W) st_sys(a.ctl[tid] + kDoneWord + (long long)me * kMaxGrid + blockIdx.x, e);
  if (blockIdx.x != 0) return;
  // workgroup 0 keeps the launch alive until EVERY workgroup of EVERY rank has delivered: the kernel boundary then publishes the
  // peers' writes to the optimiser launch, and this rank's next backward pass cannot overwrite gradients a peer still reads
  const int nch = a.c1 - a.c0;
  bool all = true;
  for (int q = 0; q < W; ++q) {
    const int gq = grid_of(nch, q, W);
    for (int wg = tid; wg < gq; wg += 256) {
      int spins = 0;
      while (ld_sys(ctl + kDoneWord + (long long)q * kMaxGrid + wg) < e && ++spins < kPeerSpinMax) __builtin_amdgcn_s_sleep(8);
      all = all && spins < kPeerSpinMax;
    }
  }
  if (!all && a.status) __hip_atomic_fetch_or(a.status, (unsigned)MEET_ERR_XCHG_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (tid == 0) __hip_atomic_store(ctl + kEpochWord, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace
}  // namespace gcrl

using namespace gcrl;

struct gcrl_xchg {
  int rank = 0, world = 1, device = 0;
  float* arena = nullptr; long long arena_floats = 0;
  char* ctl = nullptr; size_t ctl_bytes = 0;          // control block + partial array (fine-grained, IPC-exported)
  float* recv = nullptr;                              // receive buffer [arena_floats] (fine-grained, IPC-exported; world > 1 only)
  void* peer_recv[kXchgMaxWorld] = {};
  size_t parts_off = 0;
  std::vector<long long> seg_off, seg_n;
  std::vector<int> seg_c0;                            // first chunk of segment s; seg_c0[nseg] = total
  int nchunks() const { return seg_c0.back(); }
  void* peer_arena[kXchgMaxWorld] = {};
  void* peer_ctl[kXchgMaxWorld] = {};
  bool connected = false;
  unsigned int* status = nullptr;                     // device-visible status word of the owner (may be null)
};

extern "C" {

gcrl_xchg* gcrl_xchg_create(float* arena_dev, int64_t arena_floats, const int64_t* seg_off, const int64_t* seg_n, int nseg, int rank,
                            int world, int device) {
  auto bad = [](const char* m) -> gcrl_xchg* { fail(GCRL_ERR_ARG, "gcrl_xchg_create: %s", m); return nullptr; };
  if (!arena_dev || arena_floats < 1 || !seg_off || !seg_n || nseg < 1) return bad("null / empty argument");
  if (nseg > kMaxSeg) return bad("at most 12 segments");
  if (world < 1 || world > kXchgMaxWorld || rank < 0 || rank >= world) return bad("1 <= world <= 8 and 0 <= rank < world required (one node, point-to-point xGMI)");
  if (((uintptr_t)arena_dev & 15) != 0) return bad("the arena must be 16-byte aligned");
  gcrl_xchg* x = new gcrl_xchg;
  x->rank = rank; x->world = world; x->device = device;
  x->arena = arena_dev; x->arena_floats = arena_floats;
  int nch = 0;
  for (int s = 0; s < nseg; ++s) {
    if (seg_off[s] < 0 || seg_n[s] < 1 || seg_n[s] > 0x7fffffff || seg_off[s] % 4 != 0 || seg_off[s] + seg_n[s] > arena_floats) { delete x; return bad("segment outside the arena or not 16-byte aligned"); }
    x->seg_off.push_back(seg_off[s]); x->seg_n.push_back(seg_n[s]);
    x->seg_c0.push_back(nch);
    nch += (int)((seg_n[s] + kXchgChunk - 1) / kXchgChunk);
  }
  x->seg_c0.push_back(nch);
  const size_t flags = (size_t)kFlagWords * sizeof(unsigned long long);
  x->parts_off = flags;
  x->ctl_bytes = flags + (((size_t)nch * sizeof(float) + 255) / 256) * 256;
  // fine-grained device memory for everything a PEER writes and this GPU reads (coherent without cache maintenance)
  // (NO fallback to coarse-grained memory once there is a peer — a peer's store could then hide behind a stale line of
  // this GPU's L2, i.e. stale reduced gradients or spurious timeouts, with nothing to warn about it.  The caller falls back to
  // another exchange when creation fails: src/dp.py.  A world of one has no peer and may take what it gets.)
  auto fine = [world](void** p, size_t bytes) {
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) == hipSuccess) return true;
    (void)hipGetLastError();
    return world == 1 && hipMalloc(p, bytes) == hipSuccess;
  };
  bool ok = hipSetDevice(device) == hipSuccess && fine((void**)&x->ctl, x->ctl_bytes) &&
            hipMemset(x->ctl, 0, x->ctl_bytes) == hipSuccess &&
            (world == 1 || (fine((void**)&x->recv, (size_t)arena_floats * sizeof(float)) &&
                            hipMemset(x->recv, 0, (size_t)arena_floats * sizeof(float)) == hipSuccess)) &&
            hipDeviceSynchronize() == hipSuccess;
  if (!ok) { fail(GCRL_ERR_HIP, "gcrl_xchg_create: device allocation failed (fine-grained device memory is required for the buffers peers write): %s", hipGetErrorString(hipGetLastError())); gcrl_xchg_destroy(x); return nullptr; }
  x->peer_arena[rank] = x->arena;
  x->peer_ctl[rank] = x->ctl;
  x->peer_recv[rank] = x->recv;
  x->connected = world == 1;
  return x;
}

void gcrl_xchg_destroy(gcrl_xchg* x) {
  if (!x) return;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < x->world; ++q) {
    if (q == x->rank) continue;
    if (x->peer_arena[q]) (void)hipIpcCloseMemHandle(x->peer_arena[q]);
    if (x->peer_ctl[q]) (void)hipIpcCloseMemHandle(x->peer_ctl[q]);
    if (x->peer_recv[q]) (void)hipIpcCloseMemHandle(x->peer_recv[q]);
  }
  if (x->ctl) (void)hipFree(x->ctl);
  if (x->recv) (void)hipFree(x->recv);
  delete x;
}

int gcrl_xchg_handles(gcrl_xchg* x, uint8_t* out, int64_t n) {
  GCRL_CHECK_ARG(x && out && n == GCRL_XCHG_HANDLE_BYTES, "gcrl_xchg_handles: the buffer must hold GCRL_XCHG_HANDLE_BYTES bytes");
  static_assert(3 * sizeof(hipIpcMemHandle_t) + 2 * sizeof(int64_t) + 32 <= GCRL_XCHG_HANDLE_BYTES, "handle record");
  hipIpcMemHandle_t h[3];
  std::memset(h, 0, sizeof(h));
  GCRL_HIP(hipIpcGetMemHandle(&h[0], x->arena));
  GCRL_HIP(hipIpcGetMemHandle(&h[1], x->ctl));
  if (x->recv) GCRL_HIP(hipIpcGetMemHandle(&h[2], x->recv));
  std::memset(out, 0, (size_t)n);
  std::memcpy(out, h, sizeof(h));
  const int64_t meta[2] = {(int64_t)x->arena_floats, (int64_t)x->nchunks()};
  std::memcpy(out + sizeof(h), meta, sizeof(meta));
  // the owner's PCI bus id: a peer on ANOTHER device asks hipDeviceCanAccessPeer before it maps anything (gcrl_xchg_connect)
  char bus[32] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), x->device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
  std::memcpy(out + sizeof(h) + sizeof(meta), bus, sizeof(bus));
  return GCRL_OK;
}

int gcrl_xchg_connect(gcrl_xchg* x, const uint8_t* all, int64_t n) {
  GCRL_CHECK_ARG(x && all && n == (int64_t)x->world * GCRL_XCHG_HANDLE_BYTES, "gcrl_xchg_connect: world x GCRL_XCHG_HANDLE_BYTES bytes expected");
  GCRL_HIP(hipSetDevice(x->device));
  for (int q = 0; q < x->world; ++q) {
    if (q == x->rank) continue;
    const uint8_t* rec = all + (size_t)q * GCRL_XCHG_HANDLE_BYTES;
    hipIpcMemHandle_t h[3];
    int64_t meta[2];
    std::memcpy(h, rec, sizeof(h));
    std::memcpy(meta, rec + sizeof(h), sizeof(meta));
    if (meta[0] != x->arena_floats || meta[1] != (int64_t)x->nchunks())
      return fail(GCRL_ERR_STATE, "gcrl_xchg_connect: rank %d exchanges a different layout (%lld floats / %lld chunks, here %lld / %zu): the replicas must be built alike",
                  q, (long long)meta[0], (long long)meta[1], (long long)x->arena_floats, (size_t)x->nchunks());
    // a peer on another device of this process's view: peer access must be possible at all — a clear error here (the caller falls
    // back to RCCL) instead of a fault or a hang in the first exchange
    char bus[32];
    std::memcpy(bus, rec + sizeof(h) + sizeof(meta), sizeof(bus));
    bus[31] = 0;
    int pdev = -1;
    if (bus[0] && hipDeviceGetByPCIBusId(&pdev, bus) == hipSuccess && pdev >= 0 && pdev != x->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, x->device, pdev) != hipSuccess) { (void)hipGetLastError(); can = 0; }
      if (!can) return fail(GCRL_ERR_STATE, "gcrl_xchg_connect: device %d cannot access the memory of rank %d's device %d (%s): no peer-to-peer exchange on this node", x->device, q, pdev, bus);
    } else {
      (void)hipGetLastError();   // (the peer's device is not visible to this process, or it is this device: the mapping below decides)
    }
    GCRL_HIP(hipIpcOpenMemHandle(&x->peer_arena[q], h[0], hipIpcMemLazyEnablePeerAccess));
    GCRL_HIP(hipIpcOpenMemHandle(&x->peer_ctl[q], h[1], hipIpcMemLazyEnablePeerAccess));
    GCRL_HIP(hipIpcOpenMemHandle(&x->peer_recv[q], h[2], hipIpcMemLazyEnablePeerAccess));
  }
  x->connected = true;
  return GCRL_OK;
}

int gcrl_xchg_world(const gcrl_xchg* x) { return x ? x->world : 0; }

// where the reduced gradients are after an exchange: the receive buffer (same layout as the arena) — or the arena itself in a
// world of one
const float* gcrl_xchg_result(const gcrl_xchg* x) { return x ? (x->recv ? x->recv : x->arena) : nullptr; }

void gcrl_xchg_set_status(gcrl_xchg* x, unsigned int* status_dev) { if (x) x->status = status_dev; }

int gcrl_xchg_seg_parts(const gcrl_xchg* x, int seg, const float** parts_dev, int* nparts) {
  GCRL_CHECK_ARG(x && seg >= 0 && seg + 1 < (int)x->seg_c0.size() && parts_dev && nparts, "gcrl_xchg_seg_parts: bad segment");
  *parts_dev = reinterpret_cast<const float*>(x->ctl + x->parts_off) + x->seg_c0[seg];
  *nparts = x->seg_c0[seg + 1] - x->seg_c0[seg];
  return GCRL_OK;
}

int gcrl_xchg_read(gcrl_xchg* x, int64_t first, int64_t n, float* out_host) {
  GCRL_CHECK_ARG(x && out_host && first >= 0 && n >= 0 && first + n <= x->arena_floats, "gcrl_xchg_read: range outside the arena");
  GCRL_HIP(hipDeviceSynchronize());
  GCRL_HIP(hipMemcpy(out_host, gcrl_xchg_result(x) + first, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return GCRL_OK;
}

int gcrl_xchg_get_partials(gcrl_xchg* x, int seg, float* out_host, int n) {
  const float* p = nullptr; int np = 0;
  if (int rc = gcrl_xchg_seg_parts(x, seg, &p, &np)) return rc;
  GCRL_CHECK_ARG(out_host && n >= np, "gcrl_xchg_get_partials: segment %d has %d partials, room for %d", seg, np, n);
  GCRL_HIP(hipDeviceSynchronize());
  GCRL_HIP(hipMemcpy(out_host, p, (size_t)np * sizeof(float), hipMemcpyDeviceToHost));
  return np;
}

int gcrl_xchg_allreduce(gcrl_xchg* x, int seg0, int nseg, void* stream) {
  GCRL_CHECK_ARG(x && seg0 >= 0 && nseg >= 1 && seg0 + nseg < (int)x->seg_c0.size(), "gcrl_xchg_allreduce: segments [%d, %d) outside the table", seg0, seg0 + nseg);
  if (!x->connected) return fail(GCRL_ERR_STATE, "gcrl_xchg_allreduce: the peers' handles have not been connected (gcrl_xchg_connect)");
  XArgs a;
  std::memset(&a, 0, sizeof(a));
  for (int q = 0; q < x->world; ++q) {
    a.arena[q] = (float*)x->peer_arena[q];
    a.recv[q] = (float*)x->peer_recv[q];
    a.ctl[q] = (unsigned long long*)x->peer_ctl[q];
    a.parts[q] = reinterpret_cast<float*>((char*)x->peer_ctl[q] + x->parts_off);
  }
  a.nseg = (int)x->seg_off.size();
  for (int i = 0; i < a.nseg; ++i) { a.seg_off[i] = x->seg_off[i]; a.seg_n[i] = (int)x->seg_n[i]; }
  for (int i = 0; i <= a.nseg; ++i) a.seg_c0[i] = x->seg_c0[i];
  a.world = x->world; a.rank = x->rank;
  a.c0 = x->seg_c0[seg0]; a.c1 = x->seg_c0[seg0 + nseg];
  a.status = x->status;
  const int grid = grid_of(a.c1 - a.c0, x->rank, x->world);
  hipStream_t st = stream == GCRL_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)stream;
  if (x->world == 1) hipLaunchKernelGGL(xchg_two_shot_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(xchg_two_shot_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

// Collective self-test (every rank calls it once after gcrl_xchg_connect, before the arena holds anything of value).  THREE rounds
// over the SAME addresses with a different pattern each — rank r writes (r + 1) * (round + 1) + i / 4096 into the first
// min(2048, n) floats of segment 0 (and, in the middle round, of segment 1 as well: another grid, other chunk owners) — and every
// rank must read back the rank-order sum.  The first round catches what the handle exchange cannot see (a peer mapping that
// faults, peers on different layouts, a rank that never arrives: bounded wait); the later ones what a first touch cannot (ADVICE
// r4): a stale line of the arena or of the receive buffer in a reader's caches when the same addresses are read again, an epoch or
// done flag that does not survive re-use.  The host side falls back to another exchange BEFORE training starts when any rank
// fails.  Restores what it overwrote.
int gcrl_xchg_selftest(gcrl_xchg* x, void* stream) {
  GCRL_CHECK_ARG(x, "gcrl_xchg_selftest: null handle");
  if (!x->connected) return fail(GCRL_ERR_STATE, "gcrl_xchg_selftest: not connected");
  hipStream_t st = stream == GCRL_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)stream;
  const int nseg_all = (int)x->seg_n.size();
  unsigned int* status_keep = x->status;
  static unsigned int* st_host = nullptr; static unsigned int* st_dev = nullptr;
  if (!st_host) {
    GCRL_HIP(hipHostMalloc((void**)&st_host, 64, hipHostMallocMapped));
    GCRL_HIP(hipHostGetDevicePointer((void**)&st_dev, st_host, 0));
  }
  const int W = x->world;
  std::vector<std::vector<float>> saved((size_t)std::min(2, nseg_all));
  GCRL_HIP(hipStreamSynchronize(st));
  for (size_t sg = 0; sg < saved.size(); ++sg) {
    saved[sg].resize((size_t)std::min<long long>(2048, x->seg_n[sg]));
    GCRL_HIP(hipMemcpy(saved[sg].data(), x->arena + x->seg_off[sg], saved[sg].size() * sizeof(float), hipMemcpyDeviceToHost));
  }
  int rc = GCRL_OK;
  for (int round = 0; round < 3 && rc == GCRL_OK; ++round) {
    const int nsegs = (round == 1) ? (int)saved.size() : 1;
    auto value = [&](int r, int sg, int i) { return (float)((r + 1) * (round + 1)) + (float)i / 4096.0f + (float)(16 * sg); };
    for (int sg = 0; sg < nsegs; ++sg) {
      std::vector<float> v(saved[sg].size());
      for (size_t i = 0; i < v.size(); ++i) v[i] = value(x->rank, sg, (int)i);
      GCRL_HIP(hipMemcpy(x->arena + x->seg_off[sg], v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    *st_host = 0;
    x->status = st_dev;
    rc = gcrl_xchg_allreduce(x, 0, nsegs, stream);
    x->status = status_keep;
    if (rc) break;
    GCRL_HIP(hipStreamSynchronize(st));
    if (*st_host) { rc = fail(GCRL_ERR_STATE, "gcrl_xchg_selftest: a wait for a peer timed out in round %d (status 0x%x)", round, *st_host); break; }
    for (int sg = 0; sg < nsegs && rc == GCRL_OK; ++sg) {
      std::vector<float> got(saved[sg].size());
      GCRL_HIP(hipMemcpy(got.data(), gcrl_xchg_result(x) + x->seg_off[sg], got.size() * sizeof(float), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < got.size(); ++i) {
        float want = 0.f;
        for (int r = 0; r < W; ++r) want += value(r, sg, (int)i);   // rank order, as the kernel adds
        if (got[i] != want) {
          rc = fail(GCRL_ERR_STATE, "gcrl_xchg_selftest: round %d, segment %d, element %zu is %.9g, expected %.9g (world %d)%s", round, sg, i, (double)got[i],
                    (double)want, W, round ? ": a re-read of the same addresses returned stale data" : "");
          break;
        }
      }
    }
  }
  for (size_t sg = 0; sg < saved.size(); ++sg)
    (void)hipMemcpy(x->arena + x->seg_off[sg], saved[sg].data(), saved[sg].size() * sizeof(float), hipMemcpyHostToDevice);
  return rc;
}

// after a timed-out exchange (status word): every rank's counters back to a common state.  Collective in spirit: the caller
// synchronises the ranks around it (the host side does: src/dp.py)
int gcrl_xchg_reset(gcrl_xchg* x) {
  GCRL_CHECK_ARG(x, "gcrl_xchg_reset: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  GCRL_HIP(hipMemset(x->ctl, 0, x->parts_off));
  GCRL_HIP(hipDeviceSynchronize());
  return GCRL_OK;
}

}  // extern "C"
