# ordered kernel timeline of one R3D-18 training step through the data-parallel wrapper (one rank, RCCL bucket path)
export TMPDIR=/tmp
export VC_DDP_FORCE_COMM=1
O=gpurun_out/tl_ddp
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t -o p -- python3 scripts/cpu_enqueue.py 18 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
tail -1 $O/run.log
python3 scripts/step_timeline.py $O/t/p_results.db > $O/timeline.txt
tail -5 $O/timeline.txt
rm -rf $O/t
