#!/bin/bash
# same-box A/B of environment knobs on the DRIVER's command line (20 timed steps, 5 warm-up): tools/ab_env_driver.sh "" "GCRL_HEAD_BATCHES=4" ...
# (alternates the settings, ROUNDS rounds (default 5); an empty string is the default build; prints each run and the per-setting mean)
R=${ROUNDS:-5}
for r in $(seq 1 $R); do
  for kv in "$@"; do
    v=$(env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --steps 20 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | grep -o '[0-9.]*$')
    echo "driver line round $r [${kv:-default}] $v"
  done
done | tee /tmp/ab_env_driver.$$
awk -F'[][]' '{split($3,a," "); s[$2]+=a[1]; n[$2]++} END {for (k in s) printf "mean [%s] %.2f us/step over %d runs\n", k, 1000*s[k]/n[k], n[k]}' /tmp/ab_env_driver.$$
rm -f /tmp/ab_env_driver.$$
