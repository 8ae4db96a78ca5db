#!/usr/bin/env python3
"""Measurement aid: HER gather kernel bandwidth vs rows per launch (HIP-event bracketed, on the
launch stream).  Algorithmic bytes = 2*(2S+A+2)*4 per sampled row."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcrl_amd  # noqa: E402
from gcrl_amd._ffi import check, lib  # noqa: E402
from gcrl_amd.src.synthetic import synthetic_episode  # noqa: E402


def main():
    out = []
    for name, S, A, k in [("pickplace", 23, 4, 8), ("reach", 10, 3, 4)]:
        cap = 1_000_000
        buf = gcrl_amd.HERBuffer(cap, 50, 8, k_future=k, rng="engine", seed=1)
        gen = np.random.default_rng(0)
        pool = []
        for _ in range(32):
            s, a, ns, r, d, dg, ag = zip(*synthetic_episode(gen, 50, S, A))
            pool.append((np.array(s), np.array(a), np.array(ns), np.array(r, np.float32), np.zeros(50, np.float32), np.array(ag)))
        n_eps = -(-cap // (50 + k * 49))
        for ep in range(n_eps):
            buf.push_episode(ep % 8, *pool[ep % 32])
        torch.cuda.synchronize()
        R = (2 * S + A + 2) * 4
        for B, M in [(256, 1), (256, 40), (1024, 40), (2048, 40), (2048, 160), (4096, 256)]:
            for _ in range(3):
                buf.sample(B, num_batches=M)
            torch.cuda.synchronize()
            check(lib.gcrl_her_profile_enable(buf.handle, 1))
            for _ in range(20):
                buf.sample(B, num_batches=M)
            launches, ms, rows, dms = C.c_int64(), C.c_double(), C.c_int64(), C.c_double()
            check(lib.gcrl_her_profile_read(buf.handle, C.byref(launches), C.byref(ms), C.byref(rows), C.byref(dms)))
            check(lib.gcrl_her_profile_enable(buf.handle, 0))
            us = ms.value * 1e3 / launches.value
            dus = dms.value * 1e3 / launches.value
            gbs = 2 * R * B * M / (us * 1e-6) / 1e9
            dgbs = 2 * R * B * M / (dus * 1e-6) / 1e9
            out.append(dict(task=name, rows=B * M, event_us=round(us, 2), alg_GBs=round(gbs, 1), frac_8TBs=round(gbs / 8000, 4),
                            kernel_us=round(dus, 2), kernel_GBs=round(dgbs, 1), kernel_frac=round(dgbs / 8000, 4)))
            print(out[-1], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
