#!/usr/bin/env python3
"""Summary of the clock stamps a -DGCRL_OF_STAMPS build of csrc/dw_adam.hip leaves (tools/of_stamps.sh): per launch and net, when the
workgroups started, finished their gradient tile, published their norm slot, had every slot, and ended — microseconds from the launch's
first stamp (100 MHz constant-rate clock: 10 ns resolution)."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
per = 2 * 2048 * 8
names = ["start", "operands requested", "tile done", "slot published", "all slots seen", "stepped"]
for li in range(raw.size // per):
    a = raw[li * per:(li + 1) * per].reshape(2, 2048, 8).astype(np.int64)
    t0 = a[:, :, 0][a[:, :, 0] > 0].min()
    print(f"launch {li}:")
    for net in range(2):
        used = a[net, :, 0] > 0
        n = int(used.sum())
        if n == 0:
            continue
        rows = []
        for k in range(5):
            v = (a[net, used, k] - t0) / 100.0
            rows.append(f"{['start', 'pre-loads issued', 'tile done + slot published', 'all slots seen', 'stepped'][k]:>28}: min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
        print(f"  net {net} ({n} workgroups)")
        print("\n".join("    " + r for r in rows))
