#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/gpu_check.sh TAG [pytest -k expr]
# runs the GPU tests, the bench (with the per-conv table) and a rocprof kernel trace; outputs under gpurun_out/
TAG=${1:-x}
KEXPR=${2:-}
set -o pipefail
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then KARG=(-k "$KEXPR"); else KARG=(); fi
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread "${KARG[@]}" > gpurun_out/t_$TAG.log 2>&1
rc=$?
tail -4 gpurun_out/t_$TAG.log
[ $rc -ne 0 ] && { grep -n "^E " gpurun_out/t_$TAG.log | head -20; exit $rc; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --profile-kernels --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
cat gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1 || exit 1
python3 scripts/prof_summary.py gpurun_out/prof_$TAG/p_results.db --steps 13 --top 40 > gpurun_out/prof_$TAG.txt
