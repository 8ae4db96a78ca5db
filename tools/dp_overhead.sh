#!/bin/bash
# World-size-1 cost of the data-parallel code path on one GPU: the default step; the DP path with the engine's own
# peer-to-peer exchange kernel (IPC arenas: the exchange is a launch of the step's sequence); with the in-engine RCCL exchange
# (one native call per trainer cycle); with the per-exchange Python loop.
set -e
out=gpurun_out/${1:-dp_overhead}
mkdir -p $out
for w in ddpg_pickplace_b256 sac_slide_b512; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps 2000 --warmup 200 2>/dev/null | tail -1 > $out/${w}_single.json
  GCRL_FORCE_DP=1 GCRL_DP_EXCHANGE=ipc timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps 2000 --warmup 200 2>/dev/null | tail -1 > $out/${w}_dp_ipc_native.json
  GCRL_FORCE_DP=1 GCRL_DP_EXCHANGE=rccl timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps 2000 --warmup 200 2>/dev/null | tail -1 > $out/${w}_dp_rccl_native.json
  GCRL_FORCE_DP=1 GCRL_DP_EXCHANGE=python timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps 2000 --warmup 200 2>/dev/null | tail -1 > $out/${w}_dp_python.json
done
python3 - $out <<'PY'
import glob, json, sys
res = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    if f.endswith("summary.json"):
        continue
    try:
        d = json.loads(open(f).read())
        res[f.split("/")[-1][:-5]] = dict(steps_per_s=round(d["value"], 1), us_per_step=round(1e3 * d["ms_per_step"], 2), dp_exchange=d.get("dp_exchange"))
    except Exception as e:
        res[f] = str(e)
json.dump(res, open(sys.argv[1] + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
