#!/bin/bash
# Memory-traffic counters of a bench run's kernels: separate --pmc passes (kernel trace only, never with
# other trace domains), plain launches (--no-graph: the counter tool cannot follow graph replays).
#   tools/pmc_traffic.sh <tag> [bench.py args...]
# Per kernel and launch: FETCH_SIZE (x2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md §HBM),
# WRITE_SIZE, and the L2 request / hit / miss and L1->L2 read-request counts when the tool lists them.
set -e
tag=${1:-pmc}; shift || true
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
pass() {  # name, counters...
  name=$1; shift
  have=""
  for c in "$@"; do
    if grep -qw "$c" $out/counters_available.txt; then have="$have $c"; fi
  done
  if [ -z "$have" ]; then echo "pass $name: none of [$*] listed, skipped"; return 0; fi
  echo "pass $name:$have"
  rocprofv3 --kernel-trace --pmc $have --output-format csv -d $out/$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-profiler --steps 200 --warmup 40 $BENCH_ARGS > $out/$name.log 2>&1
}
BENCH_ARGS="$*"
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
pass l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
cd $GRAFT_REPO_ROOT
wl=$(echo "$BENCH_ARGS" | sed -n 's/.*--workload \([a-z0-9_]*\).*/\1/p'); wl=${wl:-ddpg_pickplace_b256}
python3 - "$out" "$wl" <<'PY'
import collections, csv, glob, json, sys
out, BENCH_WORKLOAD = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gcrl::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        if "her_gather_update" in k and r.get("Grid_Size"):   # 64 batch rows per 256-thread block: rows of THIS launch
            acc[k]["_rows"] += float(r["Grid_Size"]) / 4.0
            cnt[k]["_rows"] += 1
res = {}
for k in acc:
    n = max(cnt[k].values())
    if n < 5:
        continue
    d = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    d["launches_seen"] = n
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:   # KB per launch
        d["hbm_bytes_per_launch_corrected"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    res[k] = d
for k, d in res.items():
    if "_rows" in d:   # bench.py scales the per-row figure to its own launch size; rows come from the launches' own grids
        d["rows_per_launch"] = d.pop("_rows")
        if "hbm_bytes_per_launch_corrected" in d:
            d["hbm_bytes_per_row"] = d["hbm_bytes_per_launch_corrected"] / d["rows_per_launch"]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0)):
    print(k, {c: round(v, 1) for c, v in d.items()})
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
