#!/bin/bash
# Regenerates everything under profiles/<round>/ on a GPU box: bench lines, rocprofv3 kernel stats, PMC passes.
# usage (through gpurun):  bash tools/refresh_profiles.sh gpurun_out/r01    -> copy the results into profiles/r01/
set -o pipefail
OUT=${1:-gpurun_out/r01}; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_lbvh.json 2> $OUT/bench_lbvh.err || exit 1
python bench.py --no-incremental --no-cpu-baseline > $OUT/bench_lbvh_noincremental.json 2>> $OUT/bench_lbvh.err || exit 1
python bench.py --stage-timing 1 --no-cpu-baseline > $OUT/bench_lbvh_alltimed.json 2>> $OUT/bench_lbvh.err || exit 1
python bench.py --stage-timing 0 --no-cpu-baseline > $OUT/bench_lbvh_nostageevents.json 2>> $OUT/bench_lbvh.err || exit 1
python bench.py --knn brute --steps 2 --no-cpu-baseline > $OUT/bench_brute.json 2>> $OUT/bench_lbvh.err || exit 1
python bench.py --pairs 16 --steps 2 --warmup 1 > $OUT/bench_batch16.json 2>> $OUT/bench_lbvh.err || exit 1
python bench.py --resident-pairs 2 --no-cpu-baseline > $OUT/bench_lbvh_2pairs.json 2>> $OUT/bench_lbvh.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1 || exit 1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/lbvh_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktb -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --knn brute > $OUT/ktb.log 2>&1 || exit 1
find $OUT/ktb -name "*kernel_stats.csv" -exec cp {} $OUT/brute_kernel_stats.csv \;
bash tools/pmc_passes.sh $OUT/pmc lbvh 50 > $OUT/pmc.log 2>&1 || exit 1
cp $OUT/pmc/summary.csv $OUT/lbvh_pmc_summary.csv
rm -rf $OUT/kt $OUT/ktb $OUT/pmc/p*/
echo done
