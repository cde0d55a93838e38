#!/bin/bash
# Runs a list of GPU steps in one gpurun call; stops at the first step that was killed at its time limit (exit 124 / 137):
# a killed GPU step says something is wrong, nothing else may start on that box.  Other failures are recorded and the batch goes on.
# usage: bash tools/gpu_batch.sh <outdir> <<< "name|timeout_s|command" lines on stdin
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
while IFS='|' read -r name limit cmd; do
  [ -z "$name" ] && continue
  echo "== $name (limit ${limit}s): $cmd"
  timeout -k 10 $limit bash -c "$cmd" > $OUT/$name.log 2> $OUT/$name.err
  rc=$?
  echo "== $name rc=$rc"
  echo "$name rc=$rc" >> $OUT/batch_status.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping the batch"; exit 1; fi
done
exit 0
