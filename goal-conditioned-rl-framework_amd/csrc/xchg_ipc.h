// xchg_ipc.h — the peer-to-peer gradient exchange of xchg_ipc.hip as the engine sees it.
#pragma once
#include "../../include/gcrl.h"

namespace gcrl {
constexpr int kXchgMaxWorld = 8;       // ranks of one node (point-to-point xGMI: 7 links per GPU)
constexpr int kXchgChunk = 1024;       // floats per chunk: one 16-byte access per lane of a 256-thread workgroup
}  // namespace gcrl

extern "C" {
void gcrl_xchg_set_status(gcrl_xchg* x, unsigned int* status_dev);                       // (engine-internal: the owner's status word)
const float* gcrl_xchg_result(const gcrl_xchg* x);   // base of the reduced gradients (arena layout): the fine-grained receive buffer, or the arena (world 1)
int gcrl_xchg_seg_parts(const gcrl_xchg* x, int seg, const float** parts_dev, int* nparts);   // where segment `seg`'s sum-of-squares partials land
}
