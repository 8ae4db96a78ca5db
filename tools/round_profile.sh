#!/bin/bash
# End-of-milestone evidence, everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bench_default.json        the driver's command line (python bench.py --steps 20 --warmup 5) AND the default run
#   bench_<workload>.json     every other workload's line
#   prof_<workload>/          rocprofv3 --kernel-trace --stats summaries (headline, TD3, SAC, TQC)
#   pmc_<workload>/summary.json   HBM / L2 traffic counters per kernel (tools/pmc_traffic.sh, separate --pmc passes)
set -e
tag=${1:-x}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
for w in ddpg_reach_b256 ddpg_reach_b1024 td3_pickplace_b2048 sac_slide_b512 tqc_push_b2048 tqc_quantile_push_b2048; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 1000 --warmup 120 2>/dev/null | tail -1 > $out/bench_$w.json
  echo "$w $(grep -o '"value": [0-9.]*' $out/bench_$w.json)"
done
root=$PWD
cd /tmp && export TMPDIR=/tmp
for w in ddpg_pickplace_b256 td3_pickplace_b2048 sac_slide_b512 tqc_push_b2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o p -- python3 $root/bench.py --no-cpu-baseline --workload $w --steps 2000 --warmup 200 > $root/$out/prof_$w.log 2>&1
done
cd $root
find $out -name "*kernel_trace.csv" -delete      # tens of MB each; the stats summaries are what is kept
for w in ddpg_pickplace_b256 td3_pickplace_b2048; do
  GRAFT_REPO_ROOT=$root bash tools/pmc_traffic.sh $tag/pmc_$w --workload $w > $out/pmc_$w.log 2>&1
done
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
cut -c1-260 $out/bench_driver_cmd.json
cut -c1-260 $out/bench_default.json
