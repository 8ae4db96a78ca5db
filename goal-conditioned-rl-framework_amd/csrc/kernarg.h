// kernarg.h — getting a kernel's argument records into registers in ONE scalar round trip.
//
// Left to itself the compiler fetches every field of a by-value argument struct where it is first used: a kernel that picks its
// problem record at run time (gemm_mfma.h, ops.hip, dw_adam.hip) started with 7-14 DEPENDENT s_load / s_waitcnt pairs — the fused
// optimiser launch spent 2.2 us there before its first operand request (round 5, tools/scalar_front.py counts them).  GCRL_PIN(x)
// makes x an input of an empty volatile asm: the value must be in a scalar register HERE, so the loads of everything pinned in a row
// are issued together and waited for once; the later uses are the same SSA values (a pointer stays a kernel-argument pointer, i.e.
// global memory: laundering it through an asm OUTPUT would turn its accesses into flat ones).
#pragma once

#define GCRL_PIN(x) asm volatile("" ::"s"(x))

// Every 64-byte line of the kernel's argument segment requested at once (scalar loads, one wait): afterwards the lines sit in the
// CU's scalar cache, which starts every launch empty.  For kernels that read hundreds of argument fields all along a dependent
// chain (rowchain.hip: 207 s_loads in ~110 groups; each first touch of a line was a trip to the L2 or beyond, ~0.3-1 us, in the
// middle of the chain).  BYTES: size of the kernel's (single, by-value) argument struct.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  typedef const unsigned int __attribute__((address_space(4))) cu32;
  cu32* kp = (cu32*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned int acc = 0;
#pragma unroll
  for (int i = 0; i < BYTES / 4; i += 16) acc |= kp[i];
  asm volatile("" ::"s"(acc));
}
