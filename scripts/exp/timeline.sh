# ordered kernel timeline of one steady R3D-18 training step
export TMPDIR=/tmp
O=gpurun_out/tl
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t -o p -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline "$@" > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
tail -1 $O/bench.log | cut -c1-200
python3 scripts/step_timeline.py $O/t/p_results.db --marker pack_dgrad_batched_kernel > $O/timeline.txt
tail -5 $O/timeline.txt
rm -rf $O/t
