#!/bin/bash
# builds tools/microbench against the already-built objects of the library
set -e
cd "$(dirname "$0")/.."
P=goal-conditioned-rl-framework_amd
make -C $P/csrc -j8 >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $P/csrc -c tools/microbench.hip -o /tmp/microbench.o
hipcc --offload-arch=gfx950 /tmp/microbench.o $P/build/gemm_mfma.o $P/build/lr_sched.o -o tools/microbench
