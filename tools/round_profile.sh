#!/bin/bash
# End-of-milestone evidence, everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bench_driver_cmd.json     the driver's command line (python bench.py --steps 20 --warmup 5)
#   bench_default.json        the default run
#   bench_<workload>.json     every other workload's line (own rocprofv3 child: profiler.kernel_stats inside the line)
#   prof_<workload>/          the child's rocprofv3 --kernel-trace --stats summary (headline, TD3, SAC, TQC)
#   pmc_<workload>/summary.json   HBM / L2 traffic counters per kernel (tools/pmc_traffic.sh, separate --pmc passes)
set -e
tag=${1:-x}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
echo "driver cmd: $(grep -o '"value": [0-9.]*' $out/bench_driver_cmd.json | head -1)"
timeout -k 10 500 python bench.py --keep-profile $out/prof_ddpg_pickplace_b256 > $out/bench_default.json 2> $out/bench_default.err
echo "default: $(grep -o '"value": [0-9.]*' $out/bench_default.json | head -1)"
for w in ddpg_reach_b256 ddpg_reach_b1024 td3_pickplace_b2048 sac_slide_b512 tqc_push_b2048 tqc_quantile_push_b2048; do
  timeout -k 10 500 python bench.py --no-cpu-baseline --workload $w --steps 1000 --warmup 120 --keep-profile $out/prof_$w 2>/dev/null | tail -1 > $out/bench_$w.json
  echo "$w $(grep -o '"value": [0-9.]*' $out/bench_$w.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$w.json | head -1)"
done
root=$PWD
for w in ddpg_pickplace_b256 td3_pickplace_b2048; do
  GRAFT_REPO_ROOT=$root bash tools/pmc_traffic.sh $tag/pmc_$w --workload $w > $out/pmc_$w.log 2>&1 || echo "pmc $w failed"
done
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
cut -c1-300 $out/bench_driver_cmd.json
