#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh TAG
# Everything the profiles/ directory of a round holds: kernel traces (R3D-18 and R3D-50), the bench lines with the
# per-conv tables, the four PMC passes + the joined roofline table, the other BASELINE configs, host enqueue time.
TAG=${1:-x}
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
step() { echo "== $1" >&2; }
step "bench R3D-18" && timeout -k 10 300 python3 bench.py --profile-kernels > $O/bench.json 2> $O/conv_table.txt || exit 1
step "bench R3D-50" && timeout -k 10 300 python3 bench.py --model 50 --profile-kernels --no-cpu-baseline > $O/bench_r3d50.json 2> $O/conv_table_r3d50.txt || exit 1
step "trace R3D-18" && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace18 -o p -- python3 bench.py --steps 27 --warmup 3 --no-cpu-baseline > $O/trace18.log 2>&1 || exit 1
python3 scripts/prof_summary.py $O/trace18/p_results.db --steps 30 --top 60 --csv $O/kernel_stats.csv > $O/kernel_stats.txt
step "trace R3D-50" && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace50 -o p -- python3 bench.py --model 50 --steps 27 --warmup 3 --no-cpu-baseline > $O/trace50.log 2>&1 || exit 1
python3 scripts/prof_summary.py $O/trace50/p_results.db --steps 30 --top 70 --csv $O/kernel_stats_r3d50.csv > $O/kernel_stats_r3d50.txt
step "pmc" && bash scripts/pmc_check.sh $TAG > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
python3 scripts/pmc_summary.py gpurun_out/pmc_$TAG --csv $O/pmc_per_kernel.csv > $O/pmc_summary.txt
python3 scripts/roofline_table.py $O/kernel_stats.csv $O/pmc_per_kernel.csv > $O/roofline_table.txt
step "pmc R3D-50" && bash scripts/pmc_check.sh ${TAG}_r50 --model 50 > $O/pmc_r50.log 2>&1 || { tail -5 $O/pmc_r50.log; exit 1; }
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_r50 --csv $O/pmc_per_kernel_r3d50.csv > $O/pmc_summary_r3d50.txt
python3 scripts/roofline_table.py $O/kernel_stats_r3d50.csv $O/pmc_per_kernel_r3d50.csv > $O/roofline_table_r3d50.txt
step "other configs" && {
  timeout -k 10 300 python3 bench.py --model 50 --classes 400 --eval --no-cpu-baseline 2>/dev/null
  timeout -k 10 300 python3 bench.py --model 18 --eval --no-cpu-baseline 2>/dev/null
  timeout -k 10 400 python3 bench.py --model 50 --clip-len 64 --crop 224 --classes 400 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null
  timeout -k 10 400 python3 bench.py --model 50 --clip-len 64 --crop 224 --classes 400 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null
  # config 5 as specified: 30 views (10 clips x 3 crops) per video through valid.py's loader and validate()
  timeout -k 10 300 python3 bench.py --model 18 --eval --views 30 --no-cpu-baseline 2>/dev/null
  timeout -k 10 300 python3 bench.py --model 50 --classes 400 --eval --views 30 --no-cpu-baseline 2>/dev/null
  # the reference's default optimizer; the deterministic-reduction mode; the RCCL bucket path on one rank
  timeout -k 10 300 python3 bench.py --optimizer adam --no-cpu-baseline 2>/dev/null
  VC_DETERMINISTIC=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null
  VC_DDP_FORCE_COMM=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null
} > $O/bench_other_configs.jsonl
step "host enqueue" && { timeout -k 10 200 python3 scripts/cpu_enqueue.py 18 && timeout -k 10 200 python3 scripts/cpu_enqueue.py 50; } > $O/cpu_enqueue.txt 2>&1
step "timeline" && timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl -o p -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline > $O/tl.log 2>&1 && python3 scripts/step_timeline.py $O/tl/p_results.db --marker pack_dgrad_batched_kernel > $O/step_timeline.txt
rm -rf $O/tl
rm -rf $O/trace18 $O/trace50 gpurun_out/pmc_$TAG/*/p_results.db gpurun_out/pmc_${TAG}_r50/*/p_results.db 2>/dev/null
ls $O
