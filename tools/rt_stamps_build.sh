#!/bin/bash
# development tool: tools/abl/libgcrl_rtstamps<cb>.so = the library with device-clock stamps in rowtile_ddpg_kernel (workgroup
# (row block 0, column block <cb>) of each role), read back by tools/rt_stamps.py.  The product build carries no stamps.
set -e
cd "$(dirname "$0")/.."
P=goal-conditioned-rl-framework_amd
make -C $P/csrc -j8 >/dev/null
mkdir -p tools/abl
for cb in ${@:-0}; do
  (cd $P/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DGCRL_RT_STAMPS=$cb -c rowtile.hip -o /tmp/rowtile_st$cb.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/libgcrl_rtstamps$cb.so $(ls $P/build/*.o | grep -v rowtile) /tmp/rowtile_st$cb.o -ldl
  echo built tools/abl/libgcrl_rtstamps$cb.so
done
