#!/bin/bash
# Hardware counters of the LDS-tiled GEMM at TQC's launch size (tools/gemm_micro), separate --pmc passes, kernel trace only.
#   tools/pmc_gemm.sh <tag> [gemm_micro args...]
set -e
tag=${1:-pmc_gemm}; shift || true
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o p -- $root/tools/gemm_micro $ARGS > $out/$name.log 2>&1 || echo "pass $name failed"; }
ARGS="$*"
pass sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_LEVEL_WAVES
pass sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass sq3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD
pass tc1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass tc2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
cd $root
python3 - "$out" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_tiled" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
d = {c: acc[c] / cnt[c] for c in acc}
for c in sorted(d):
    print("%-40s %16.0f   (%d launches)" % (c, d[c], cnt[c]))
g = lambda k: d.get(k, float("nan"))
print("MFMA pipe busy / (4 SIMD-groups?): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
print("wave cycles: waiting on any instruction %.3f, waiting (no instruction ready) %.3f, issuing %.3f" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
print("average VMEM instruction latency (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD+WR) = %.0f clk" % (g("SQ_INST_LEVEL_VMEM") / (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"))))
print("average LDS instruction latency (SQ_INST_LEVEL_LDS / SQ_INSTS_LDS) = %.0f clk" % (g("SQ_INST_LEVEL_LDS") / g("SQ_INSTS_LDS")))
print("L1->L2 read latency (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ) = %.0f clk; L2 hit rate %.3f" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
