// ring_book.cc — host-only test hook of ring_book.h (no HIP): drives the deque(maxlen) bookkeeping through a sequence of appends
// so that tests (and the sanitizer build: `make asan`) can hold it against collections.deque without a GPU.
#include "ring_book.h"

#include "common.h"

extern "C" {

// After appends[0..n) rows have been appended to a ring of `capacity` rows that held `len0` rows with head `head0`: writes, per
// append i, the physical row its first row went to (tail_out[i]) and how many of its rows were overwritten inside the same
// append (skip_out[i]); returns the final head / len.  Mirrors what gcrl_her_push* / gcrl_her_append do on the handle.
int gcrl_ringbook_sim(int64_t capacity, int64_t head0, int64_t len0, const int64_t* appends, int n, int64_t* tail_out, int64_t* skip_out,
                      int64_t* head_out, int64_t* len_out) {
  GCRL_CHECK_ARG(capacity >= 1 && head0 >= 0 && head0 < capacity && len0 >= 0 && len0 <= capacity && (n == 0 || appends) && head_out && len_out,
                 "gcrl_ringbook_sim: bad arguments");
  gcrl::RingBook b{capacity, head0, len0};
  for (int i = 0; i < n; ++i) {
    GCRL_CHECK_ARG(appends[i] >= 0, "gcrl_ringbook_sim: negative append");
    if (tail_out) tail_out[i] = b.tail();
    const int64_t skip = b.append(appends[i]);
    if (skip_out) skip_out[i] = skip;
  }
  *head_out = b.head;
  *len_out = b.len;
  return GCRL_OK;
}

}  // extern "C"
