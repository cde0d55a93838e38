#!/bin/bash
# Per-dispatch PMC values of the fused matcher for the first launches of one icp_run (which iteration is bound by what).
# usage: tools/pmc_per_dispatch.sh <outdir> <iters>
OUT=$1; ITERS=${2:-6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TAGRAM0_REQ_sum TCP_TCP_LATENCY_sum" \
         "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python tools/prof_run.py lbvh $ITERS > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for d in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "k_knn_bvh_post" not in r["Kernel_Name"]: continue
        per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in per.items():
        v.sort()
        for n, (did, val) in enumerate(v): rows[n][c] = val
with open("$OUT/per_dispatch.csv", "w") as f:
    cs = sorted({c for n in rows for c in rows[n]})
    f.write("launch," + ",".join(cs) + "\n")
    for n in sorted(rows): f.write(str(n) + "," + ",".join("%.6g" % rows[n].get(c, float("nan")) for c in cs) + "\n")
print(open("$OUT/per_dispatch.csv").read())
PY
