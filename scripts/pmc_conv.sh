#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_conv.sh TAG N Ci T H W Co k s p [ops]
# HBM bytes per launch of one conv geometry (FETCH_SIZE and WRITE_SIZE passes over scripts/bench_conv.py, each its
# own process, no trace domains): the per-geometry companion of scripts/pmc_check.sh, whose per-kernel rows average
# every geometry a kernel runs in the step.
TAG=$1; shift
GEOM=("${@:1:9}")
OPS=${10:-wgrad}
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv_$TAG
mkdir -p $OUT
for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
  name=${pass%%:*}; ctr=${pass##*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$name -o p -- python3 scripts/bench_conv.py "${GEOM[@]}" --ops $OPS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
done
python3 scripts/pmc_summary.py $OUT --top 8
