#!/bin/bash
# soak runs: tools/soak.sh > profiles/rNN_soak.txt   (one MI355X; every step's metrics finite, no in-kernel wait timed out — a timeout
# raises GCRL_ERR_STATE and the bench run fails)
echo "# soak runs at HEAD, one MI355X (python bench.py --no-cpu-baseline --no-profiler --workload W --steps N --warmup 400): every step's metrics finite,"
echo "# no in-kernel wait timed out (a timeout would have raised GCRL_ERR_STATE); in_kernel_meetings as reported in the bench line"
for spec in ddpg_pickplace_b256:400000 ddpg_reach_b1024:200000 sac_slide_b512:200000 td3_pickplace_b2048:60000; do
  w=${spec%%:*}; n=${spec##*:}
  line=$(timeout -k 10 500 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps $n --warmup 400 2>/dev/null | tail -1)
  python - "$w" "$n" "$line" <<'PY'
import json, math, sys
w, n, line = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(line)
    fin = all(math.isfinite(float(x)) for x in d.get("last_metrics", []))
    print(w, n, "steps", round(1000 * d["ms_per_step"], 2), "us/step", "finite" if fin else "NOT FINITE", d["config"].get("in_kernel_meetings"))
except Exception as e:
    print(w, n, "FAILED", repr(e), line[-300:])
PY
done
