// common.h — error plumbing shared by all translation units of libgcrl_hip.so.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/gcrl.h"

namespace gcrl {

// thread-local last error message (gcrl_last_error)
std::string& last_error();
int fail(int status, const char* fmt, ...);

}  // namespace gcrl

#define GCRL_CHECK_ARG(cond, ...)                          \
  do {                                                     \
    if (!(cond)) return gcrl::fail(GCRL_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define GCRL_HIP(call)                                                                  \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess)                                                              \
      return gcrl::fail(GCRL_ERR_HIP, "%s failed: %s (%s:%d)", #call,                   \
                        hipGetErrorString(e__), __FILE__, __LINE__);                    \
  } while (0)
