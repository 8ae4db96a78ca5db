"""Headline network (DDPG PickAndPlace H=256 L=3) at batch sizes 16 .. 512: if a step costs the same with 4 + 4
workgroups as with 64 + 64, the row-chain kernel is bound per CU (its own request queue and dependency chain), not by
contention between workgroups for L2 / HBM."""
import argparse
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402

w0 = dict(bench.WORKLOADS["ddpg_pickplace_b256"], cap=100000)
args = argparse.Namespace(no_graph=False, pipeline=-1, rng="engine")
import os
for B in [int(x) for x in os.environ.get("GCRL_SWEEP_B", "16,32,64,128,256,512").split(",")]:
    agent, _, _ = bench.build_agent(w0, args, 0, 0, batch=B)
    el, _, _, _ = bench.timed_region(agent, None, w0, 2000, 200)
    print(f"B={B:4d}  workgroups per phase {-(-B // 4):3d}   {2000 / el:8.1f} steps/s   {1e6 * el / 2000:6.2f} us/step", flush=True)
    del agent   # (refcount frees it: no cyclic collection needed)
    torch.cuda.synchronize()
