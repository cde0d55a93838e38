#!/bin/bash
# PMC passes for the ICP kernels (one rocprofv3 run per counter group; --kernel-trace only, as the guide prescribes).
# usage: tools/pmc_passes.sh <outdir> <backend> <iters>
OUT=$1; BACKEND=${2:-lbvh}; ITERS=${3:-5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TAGRAM0_REQ_sum TCP_TCP_LATENCY_sum" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python tools/prof_run.py $BACKEND $ITERS > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.csv", "w", newline="") as f:
    wr = csv.writer(f)                                     # (template kernels have commas in their names: quoted)
    wr.writerow(["kernel", "counter", "mean_per_launch", "launches"])
    for k in sorted(agg):
        if "icpdev" not in k: continue
        for c in sorted(agg[k]):
            v = agg[k][c]; wr.writerow([k, c, "%.6g" % (sum(v) / len(v)), len(v)])
print(open("$OUT/summary.csv").read())
PY
