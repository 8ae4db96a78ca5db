#!/bin/bash
# same-box A/B of library builds: tools/ab.sh <workload> <steps> libA.so libB.so ...   (alternates the builds, 3 rounds)
w=$1; steps=$2; shift 2
for r in 1 2 3; do
  for lib in "$@"; do
    v=$(GCRL_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps $steps --warmup 200 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
    echo "$w round $r $lib $v"
  done
done
