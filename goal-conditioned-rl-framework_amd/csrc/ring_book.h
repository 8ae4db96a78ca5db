// ring_book.h — the replay ring's deque(maxlen) bookkeeping, host-only (no HIP): where a flush's rows land and what falls off.
//
// Replaces what collections.deque(maxlen=max_len) does for the reference's buffers (src/buffer.py:95, :8-20): `append` past the
// capacity drops the OLDEST row.  Logical index j (what random.sample indexes) lives at physical row (head + j) mod capacity.
// Kept apart from her_ring.hip so that the arithmetic can be built with -fsanitize=address,undefined and driven without a GPU
// (csrc/Makefile `asan` target, tests/test_host_abi.py, tools/asan_host_check.sh): the C entry points below are test hooks.
#pragma once
#include <cstdint>

namespace gcrl {

struct RingBook {
  int64_t cap = 0, head = 0, len = 0;

  int64_t tail() const { return (head + len) % cap; }                 // physical row the next append writes
  int64_t phys(int64_t logical) const { return (head + logical) % cap; }
  // A flush appends `rows` rows at tail(), tail() + 1, ... (mod cap) in order.  When rows > cap the first rows - cap of them are
  // overwritten by later ones of the same flush: the kernel skips them (returns that count).  Advances head / len as
  // deque(maxlen) would after `rows` appends.
  int64_t append(int64_t rows) {
    const int64_t skip = rows > cap ? rows - cap : 0;
    int64_t newlen = len + rows;
    if (newlen > cap) {
      head = (head + (newlen - cap)) % cap;
      newlen = cap;
    }
    len = newlen;
    return skip;
  }
};

}  // namespace gcrl
