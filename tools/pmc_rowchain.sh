#!/bin/bash
# SQ counters of the default bench's kernels (one --pmc pass, kernel trace only), summary per kernel name
set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profiler --steps 400 --warmup 80 > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc_sq
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-48:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k in acc:
    if cnt[k] < 50: continue
    print(k, "launches", cnt[k], {c: round(v / cnt[k], 1) for c, v in acc[k].items()})
PY
