#!/bin/bash
# the weight-slice (16 x 16 tile per workgroup) form of a dependent chain of layer passes, measured (tools/microbench_rowtile.hip)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-rowtile}
mkdir -p $out
cd $GRAFT_REPO_ROOT
[ -x tools/microbench_rowtile ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/microbench_rowtile tools/microbench_rowtile.hip
[ -x tools/xcc_census ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o tools/xcc_census tools/xcc_census.hip
./tools/xcc_census > $out/microbench.txt
for B in 256 128; do
  echo "== B=$B H=256 NL=10" >> $out/microbench.txt
  timeout -k 10 200 ./tools/microbench_rowtile $B 256 10 >> $out/microbench.txt
done
cat $out/microbench.txt
