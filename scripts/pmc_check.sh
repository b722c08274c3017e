#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_check.sh TAG [bench.py arguments, e.g. --model 50]
# three separate rocprofv3 counter passes over a short bench run (each its own process: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; no trace domains are combined with --pmc), CSV output under gpurun_out/pmc_TAG/
TAG=${1:-x}
shift
EXTRA=("$@")
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline "${EXTRA[@]}" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && run grbm GRBM_GUI_ACTIVE
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt && head -30 $OUT/summary.txt
