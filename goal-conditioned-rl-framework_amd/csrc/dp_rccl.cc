// dp_rccl.cc — gradient exchange inside the engine: an RCCL communicator owned by the library, so a
// data-parallel trainer cycle (gradient_step update steps with one or two all-reduces each) is ONE host
// call that enqueues kernels and collectives on the engine's stream — no Python round trip per exchange.
//
// New design: the reference is single-process.  One process per GPU; gradients of a
// phase live in one flat fp32 block (all critics | actor (+ log_alpha)), summed over ranks in place and
// scaled by 1/world inside the optimiser kernels.  Messages are 37 KB - 11 MB: latency-bound on
// point-to-point xGMI, hence one collective per block, never per tensor.
//
// RCCL is bound at run time (dlopen of the library the host process already uses — PyTorch-ROCm ships
// its own librccl next to its HIP runtime, and two copies of either in one process do not mix); the few
// entry points used are declared here exactly as in rccl.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;   // NCCL_UNIQUE_ID_BYTES
typedef int ncclResult_t;                               // ncclSuccess = 0
enum { kNcclFloat32 = 7, kNcclSum = 0 };

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int bind_rccl(const char* path) {
  if (g_rccl.lib) return GCRL_OK;
  const char* tries[] = {path, "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* t : tries) {
    if (!t || !*t) continue;
    h = dlopen(t, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return gcrl::fail(GCRL_ERR_STATE, "gcrl_dp: cannot load RCCL (%s): %s", path ? path : "librccl.so", dlerror());
  auto sym = [&](const char* n) { return dlsym(h, n); };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.Broadcast)
    return gcrl::fail(GCRL_ERR_STATE, "gcrl_dp: the loaded RCCL lacks an entry point");
  g_rccl.lib = h;
  return GCRL_OK;
}

int nccl_fail(const char* what, ncclResult_t r) {
  return gcrl::fail(GCRL_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

}  // namespace

struct gcrl_dp {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int gcrl_dp_unique_id(uint8_t* id128_out, const char* rccl_path) {
  GCRL_CHECK_ARG(id128_out, "gcrl_dp_unique_id: null output");
  if (int rc = bind_rccl(rccl_path)) return rc;
  ncclUniqueId id;
  if (ncclResult_t r = g_rccl.GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", r);
  std::memcpy(id128_out, id.internal, sizeof(id.internal));
  return GCRL_OK;
}

gcrl_dp* gcrl_dp_create(int rank, int world, const uint8_t* id128, int device, const char* rccl_path) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) { gcrl::fail(GCRL_ERR_ARG, "gcrl_dp_create: bad rank %d / world %d", rank, world); return nullptr; }
  if (bind_rccl(rccl_path)) return nullptr;
  if (hipSetDevice(device) != hipSuccess) { gcrl::fail(GCRL_ERR_HIP, "gcrl_dp_create: hipSetDevice(%d) failed", device); return nullptr; }
  ncclUniqueId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  gcrl_dp* d = new gcrl_dp;
  d->rank = rank; d->world = world;
  if (ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank)) {
    nccl_fail("ncclCommInitRank", r);
    delete d;
    return nullptr;
  }
  return d;
}

void gcrl_dp_destroy(gcrl_dp* d) {
  if (!d) return;
  if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
  delete d;
}

int gcrl_dp_world(const gcrl_dp* d) { return d ? d->world : 0; }

int gcrl_dp_allreduce_sum(gcrl_dp* d, float* buf_dev, int64_t n, void* stream) {
  GCRL_CHECK_ARG(d && buf_dev && n >= 0, "gcrl_dp_allreduce_sum: bad arguments");
  if (n == 0) return GCRL_OK;
  hipStream_t st = stream == GCRL_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)stream;
  if (ncclResult_t r = g_rccl.AllReduce(buf_dev, buf_dev, (size_t)n, kNcclFloat32, kNcclSum, d->comm, st)) return nccl_fail("ncclAllReduce", r);
  return GCRL_OK;
}

int gcrl_dp_broadcast(gcrl_dp* d, float* buf_dev, int64_t n, int root, void* stream) {
  GCRL_CHECK_ARG(d && buf_dev && n >= 0 && root >= 0 && root < d->world, "gcrl_dp_broadcast: bad arguments");
  if (n == 0) return GCRL_OK;
  hipStream_t st = stream == GCRL_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)stream;
  if (ncclResult_t r = g_rccl.Broadcast(buf_dev, buf_dev, (size_t)n, kNcclFloat32, root, d->comm, st)) return nccl_fail("ncclBroadcast", r);
  return GCRL_OK;
}

}  // extern "C"
