#!/bin/bash
# Regenerates everything under profiles/<round>/ on a GPU box: bench lines, rocprofv3 kernel stats, PMC passes, traffic_latest.json.
# usage (through gpurun):  bash tools/refresh_profiles.sh gpurun_out/r03p    -> copy the results into profiles/r03/
OUT=${1:-gpurun_out/r03p}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 "$@" > $OUT/$name.json 2> $OUT/$name.err; rc=$?; echo "$name rc=$rc" >> $OUT/status.txt; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit 1; fi; }
run bench_lbvh python bench.py --steps 20 --warmup 5
run bench_lbvh_noincremental python bench.py --no-incremental --no-cpu-baseline
run bench_lbvh_separate env ICP_HIP_MERGE=0 python bench.py --no-cpu-baseline --no-extras
run bench_lbvh_separate_nostageevents env ICP_HIP_MERGE=0 python bench.py --no-cpu-baseline --no-extras --stage-timing 0
run bench_lbvh_loop env ICP_HIP_PERSIST=1 ICP_HIP_LOOP_WAVESLEEP=6 python bench.py --no-cpu-baseline --no-extras --per-iteration
run bench_lbvh_loop_nostageevents env ICP_HIP_PERSIST=1 ICP_HIP_LOOP_WAVESLEEP=6 python bench.py --no-cpu-baseline --no-extras --stage-timing 0
run bench_lbvh_hybrid10_nostageevents env ICP_HIP_PERSIST=1 ICP_HIP_LOOP_FROM=10 ICP_HIP_LOOP_WAVESLEEP=6 python bench.py --no-cpu-baseline --no-extras --stage-timing 0
run bench_lbvh_alltimed python bench.py --stage-timing 1 --no-cpu-baseline
run bench_lbvh_nostageevents python bench.py --stage-timing 0 --no-cpu-baseline
run bench_brute python bench.py --knn brute --steps 2 --no-cpu-baseline
run bench_batch16 python bench.py --pairs 16 --steps 3 --warmup 1
run bench_batch16_shared python bench.py --pairs 16 --steps 3 --warmup 1 --shared-scans
run bench_batch44_shared python bench.py --pairs 44 --steps 2 --warmup 1 --shared-scans
run bench_lbvh_2pairs python bench.py --resident-pairs 2 --no-cpu-baseline
echo "== kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/kt.log 2>&1 || exit 1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/lbvh_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktb -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --knn brute > $OUT/ktb.log 2>&1 || exit 1
find $OUT/ktb -name "*kernel_stats.csv" -exec cp {} $OUT/brute_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktp -- python bench.py --pairs 16 --steps 2 --warmup 1 --shared-scans > $OUT/ktp.log 2>&1 || exit 1
find $OUT/ktp -name "*kernel_stats.csv" -exec cp {} $OUT/batch16_kernel_stats.csv \;
echo "== pmc passes"
timeout -k 10 900 bash tools/pmc_passes.sh $OUT/pmc lbvh 50 > $OUT/pmc.log 2>&1 || exit 1
cp $OUT/pmc/summary.csv $OUT/lbvh_pmc_summary.csv
python - <<PY
import csv, json
rows = {(r["kernel"], r["counter"]): float(r["mean_per_launch"]) for r in csv.DictReader(open("$OUT/lbvh_pmc_summary.csv"))}
k = [kk for (kk, c) in rows if "k_knn_bvh_post_ring<3" in kk][0]
f, w = rows[(k, "FETCH_SIZE")], rows[(k, "WRITE_SIZE")]
# MI355X_MICROARCH.md, HBM section: both counters are in KB; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B -> doubled
json.dump({"knn": "lbvh", "n_src": 370488, "kernel": "k_knn_bvh_post_ring<3, false>", "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
           "fetch_size_kb_raw": f, "write_size_kb": w, "fetch_correction": "x2 (gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B against 128-B requests)",
           "source": "profiles/r03/lbvh_pmc_summary.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only "
                     "(tools/pmc_passes.sh lbvh 50), the 50 launches of one icp_run averaged"}, open("$OUT/traffic_latest.json", "w"))
print(open("$OUT/traffic_latest.json").read())
PY
if [ -f icp-variants_amd/lib/libicp_hip_times.so ]; then      # development build with per-wave phase stamps (ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1), built before the call
  echo "== wave phase times"
  ICP_HIP_MERGE=0 ICP_HIP_LIB=icp-variants_amd/lib/libicp_hip_times.so timeout -k 10 300 python tools/dev_wave_times.py 1 3 6 12 20 45 > $OUT/wave_phase_times.txt 2> $OUT/wave_phase_times.err
  ICP_HIP_LIB=icp-variants_amd/lib/libicp_hip_times.so timeout -k 10 300 python tools/dev_ring_times.py 3 12 30 40 > $OUT/ring_phase_times.txt 2> $OUT/ring_phase_times.err
  for it in 2 12 30 40; do ICP_HIP_PERSIST=1 ICP_HIP_LOOP_WAVESLEEP=6 ICP_HIP_DBG_ITER=$it ICP_HIP_LIB=icp-variants_amd/lib/libicp_hip_times.so timeout -k 10 300 python tools/dev_loop_times.py; done > $OUT/loop_phase_times.txt 2> $OUT/loop_phase_times.err
fi
rm -rf $OUT/kt $OUT/ktb $OUT/ktp $OUT/pmc/p*/
echo done
