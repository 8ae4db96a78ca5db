"""Timeline of rowtile_ddpg_kernel (development tool): needs a library built by tools/rt_stamps_build.sh
(GCRL_HIP_LIB=tools/abl/libgcrl_rtstamps<cb>.so); prints the device-clock stamps (10 ns ticks) that workgroup (row block 0,
column block <cb>) of each role left during the LAST launch of a run of headline steps: W = a wait has ended, R = the four
waves' partial sums are reduced, A = the tile is stored, drained and its arrival sent."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ddpg_pickplace_b256")
a = ap.parse_args()
w = dict(bench.WORKLOADS[a.workload], cap=100_000)
args = argparse.Namespace(no_graph=False, pipeline=-1, rng="engine")
agent, _, _ = bench.build_agent(w, args, 0, 0)
for c in range(5):
    agent.update_many(1 + 40 * c, 40)
torch.cuda.synchronize()
import gcrl_amd  # noqa: E402
lib = gcrl_amd._ffi.lib
out = (C.c_uint64 * 192)()
fn = lib.gcrl_debug_rt_stamps
fn.restype = C.c_int
assert fn(out) == 0
st = np.array(list(out), dtype=np.int64).reshape(3, 64)
base = min(int(st[r][0]) for r in range(3) if st[r][0])
for role, name in ((0, "P"), (1, "KT"), (2, "KO")):
    if not st[role][0]:
        continue
    print("role", name)
    prev = st[role][0]
    for i in range(64):
        t = st[role][i]
        if not t or t < base:
            continue
        print(f"  stamp {i:2d} at {(t - base) / 100:7.2f} us   (+{(t - prev) / 100:5.2f})")
        prev = t
