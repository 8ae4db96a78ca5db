"""Section timeline of rowchain_ddpg_kernel (development tool): needs a library built with -DGCRL_RC_STAMPS
(GCRL_HIP_LIB=<that .so>); prints the device-clock stamps (10 ns ticks) the first K-role and the first P-role workgroup
left at their section boundaries during the LAST launch of a run of headline steps."""
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
out = (C.c_uint64 * 64)()
fn = lib.gcrl_debug_rc_stamps
fn.restype = C.c_int
assert fn(out) == 0
st = np.array(list(out), dtype=np.int64).reshape(2, 32)
names_k = {0: "start", 1: "prologue", 2: "target actor hidden", 3: "its head + action", 4: "target critic hidden", 5: "its head + y", 6: "critic hidden",
           7: "head, loss, head bwd", 8: "critic dX chain"}
names_p = {0: "start", 1: "prologue", 2: "actor hidden", 3: "actor head + action", 4: "critic hidden", 6: "Q head + head bwd", 7: "critic dX chain",
           9: "da head, tanh', actor head bwd", 10: "actor dX chain"}
for role, names in ((0, names_k), (1, names_p)):
    t0 = st[role][0]
    print("role", "K" if role == 0 else "P")
    prev = t0
    for i in sorted(names):
        t = st[role][i]
        print(f"  {names[i]:32s} at {(t - t0) / 100:7.2f} us   (+{(t - prev) / 100:5.2f})")
        prev = t
