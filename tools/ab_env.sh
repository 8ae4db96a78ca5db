#!/bin/bash
# same-box A/B of environment knobs: tools/ab_env.sh <workload> <steps> "" "GCRL_NO_OPT_FUSE=1" ...   (alternates the settings, 3 rounds;
# an empty string is the default build)
w=$1; steps=$2; shift 2
for r in 1 2 3; do
  for kv in "$@"; do
    v=$(env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps $steps --warmup 200 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
    echo "$w round $r [${kv:-default}] $v"
  done
done
