// meet.h — workgroups of ONE launch that meet in memory (bn_slab.hip: the row groups of a slab; rowchain.hip: the role
// workgroups of a row block), and the release every such publication needs (gemm_tiled.h: the split-dW ticket).
//
// Protocol (round 4; rounds 1-3 used an arrival counter + a generation word, with two defects the advisor found in the
// ISA: no wait between a wave's agent-scope stores and the arrival, and a relaxed generation load ordered before the
// arriving add by program order only):
//   * the data moves with agent-scope (sc1) stores / loads: written through to / read from the memory side of the per-XCD L2s;
//   * EVERY storing wave drains its stores (`s_waitcnt vmcnt(0)`: a write-through store is acknowledged by the memory side)
//     BEFORE the workgroup barrier that precedes the arrival — a workgroup-scope release fence emits no such wait on gfx950;
//   * ONE 64-bit counter per meeting point, monotonic, never reset: the arrival `t = fetch_add(ctr, 1)` itself names the
//     round (t / arrivals), and a waiter polls until ctr >= (t / arrivals + 1) * arrivals.  No generation word, no reset
//     store, nothing a reordering could break; a launch (or graph replay) always adds exactly `arrivals` per point, so the
//     counter is a multiple of `arrivals` between launches;
//   * the wait is bounded (kMeetSpinMax polls of ~1 us).  A timed-out waiter returns false — the caller poisons its result
//     with NaN — AND sets a bit of the host-visible status word: the next host synchronisation of the owning handle turns
//     it into GCRL_ERR_STATE (the reference raises on any failed step, src/agent.py:659-699), zeroes the counters and
//     clears the word, so the launch after the error works again.
// Residency: a waiting workgroup must never keep an awaited one off the chip — the launchers admit the meeting forms only
// when every workgroup of the launch is resident at once on a device this process has to itself (meet_capacity below).
#pragma once
#include <hip/hip_runtime.h>

namespace gcrl {

constexpr int kMeetSpinMax = 1 << 20;
// bits of the status word
enum { MEET_ERR_BN_SLAB = 1, MEET_ERR_ROWCHAIN = 2, MEET_ERR_XCHG_READY = 4, MEET_ERR_XCHG_DONE = 8, MEET_ERR_DW_ADAM = 16 };

// this wave's global stores (agent- or system-scope write-through ones in particular) have been acknowledged
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// All threads of the workgroup call this after their publishing stores.  `ctr`: the meeting point's counter (its own
// 128-byte line: same-line agent-scope atomics serialise at a memory round trip each).  Returns false on a timed-out wait.
__device__ inline bool meet(unsigned long long* ctr, unsigned int arrivals, bool wait, unsigned int* s_flag, unsigned int* status,
                            unsigned int err_bit) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int ok = 1;
    if (wait) {
      const unsigned long long target = (t / arrivals + 1ull) * arrivals;
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kMeetSpinMax) __builtin_amdgcn_s_sleep(4);
      if (spins >= kMeetSpinMax) {
        ok = 0;
        if (status) __hip_atomic_fetch_or(status, err_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// Producers / consumers (round 4, TD3's critic phase at batch 2048: more workgroups than the chip holds at once).  PRODUCER
// workgroups arrive and never wait; CONSUMER workgroups wait for the `producers` arrivals of THIS launch without arriving
// themselves — each consumer counts its own launches in a private word (`my_round`; only it ever writes it), so the counter stays
// a multiple of `producers` between launches.  No consumer waits for another consumer, and the launcher gives every producer a
// LOWER workgroup index than any consumer: every XCD dispatches its workgroups in index order, so a consumer's producers were
// dispatched before it on whatever XCD they run and — never waiting — complete.  Deadlock-free without all workgroups being
// resident at once (the bounded wait and the status word stay all the same).
__device__ inline void meet_produce(unsigned long long* ctr) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool meet_consume(const unsigned long long* ctr, unsigned long long* my_round, unsigned int producers, unsigned int* s_flag,
                                    unsigned int* status, unsigned int err_bit) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long r = __hip_atomic_load(my_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (r + 1ull) * producers;
    unsigned int ok = 1;
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kMeetSpinMax) __builtin_amdgcn_s_sleep(4);
    if (spins >= kMeetSpinMax) {
      ok = 0;
      if (status) __hip_atomic_fetch_or(status, err_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(my_round, r + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// Process-wide switch: GCRL_SHARED_GPU=1 in the environment, or gcrl_set_shared_device(1) — the device is shared with other
// processes / streams that hold CUs (two ranks on one GPU in the tests, a collective library's kernels), so no launch may
// contain a wait for another workgroup of itself.
bool meet_device_shared();
void meet_set_device_shared(bool on);
// another process holds the per-device presence lock of this library (abi_misc.hip): counts the device as shared from then on
bool meet_probe_device(int device);
// workgroups of `kernel` (block threads, dynamic LDS bytes) that are resident at once on the current device, with the
// headroom the occupancy query needs (it over-reports: 5 where 4 are resident at 32 KB of LDS); 0 when the
// device is shared or the query fails
long long meet_capacity(const void* kernel, int threads, size_t lds_bytes);

}  // namespace gcrl
