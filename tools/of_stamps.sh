#!/bin/bash
# Development build of the library with clock stamps inside the fused dW + optimiser launch (csrc/dw_adam.hip, -DGCRL_OF_STAMPS), a
# headline run without hipGraphs that leaves eight launches' per-workgroup stamps in gpurun_out/of_stamps.bin, and their summary
# (tools/of_stamps.py).  Extra environment (poll pacing knobs ...) is passed through.
set -e
cd "$(dirname "$0")/.."
make -s -C goal-conditioned-rl-framework_amd/csrc OBJDIR=../build_stamps OUT=../libgcrl_hip_stamps.so FLAGS_dw_adam=-DGCRL_OF_STAMPS FLAGS_agent=-DGCRL_OF_STAMPS -j8
mkdir -p gpurun_out
GCRL_HIP_LIB=$PWD/goal-conditioned-rl-framework_amd/libgcrl_hip_stamps.so GCRL_OF_STAMPS=$PWD/gpurun_out/of_stamps.bin GCRL_OF_STAMPS_AT=${AT:-1500} \
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-profiler --no-graph --steps 3000 --warmup 200 > /dev/null
python tools/of_stamps.py gpurun_out/of_stamps.bin
