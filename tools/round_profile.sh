#!/bin/bash
# end-of-milestone evidence: default bench line (with CPU leg), every workload's line, rocprofv3 kernel stats
set -e
tag=${1:-x}
mkdir -p gpurun_out/$tag
timeout -k 10 400 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
for w in ddpg_reach_b256 ddpg_reach_b1024 td3_pickplace_b2048 sac_slide_b512 tqc_push_b2048; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 1000 --warmup 120 2>/dev/null | tail -1 > gpurun_out/$tag/bench_$w.json
  echo "$w $(grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_$w.json)"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2000 --warmup 200 > $GRAFT_REPO_ROOT/gpurun_out/$tag/prof.log 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-200 gpurun_out/$tag/bench_default.json
