#!/bin/bash
# Development build of the library with clock stamps inside the forward BatchNorm slab launch (csrc/bn_slab.hip, -DGCRL_SLAB_STAMPS), a SAC
# cfg-5 run without hipGraphs that leaves nine launches' per-workgroup stamps in gpurun_out/slab_stamps.bin, and their summary.
set -e
cd "$(dirname "$0")/.."
make -s -C goal-conditioned-rl-framework_amd/csrc OBJDIR=../build_stamps OUT=../libgcrl_hip_stamps.so FLAGS_bn_slab=-DGCRL_SLAB_STAMPS -j8
mkdir -p gpurun_out
GCRL_HIP_LIB=$PWD/goal-conditioned-rl-framework_amd/libgcrl_hip_stamps.so GCRL_SLAB_STAMPS=$PWD/gpurun_out/slab_stamps.bin GCRL_SLAB_STAMPS_AT=${AT:-1500} \
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-profiler --no-graph --workload sac_slide_b512 --steps 1500 --warmup 200 > /dev/null
python - <<'PY'
import numpy as np
raw = np.fromfile("gpurun_out/slab_stamps.bin", dtype=np.uint64).reshape(-1, 1024, 8).astype(np.int64)
names = ["start", "GEMM done", "own statistics", "exchange done", "stored"]
for li, a in enumerate(raw):
    used = a[:, 0] > 0
    t0 = a[used, 0].min()
    print(f"launch {li}: {int(used.sum())} workgroups")
    for k, nm in enumerate(names):
        v = (a[used, k] - t0) / 100.0
        print(f"  {nm:>16}: min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us")
PY
