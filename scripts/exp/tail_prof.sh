# per-kernel durations of the "long tail" conv geometries (strided, layer3/4)
export TMPDIR=/tmp
O=gpurun_out/tail
mkdir -p $O
i=0
for g in "64 64 8 28 28 128 3 2 1" "64 128 4 14 14 256 3 2 1" "64 256 2 7 7 512 3 2 1" "64 256 2 7 7 256 3 1 1" "64 512 1 4 4 512 3 1 1" "64 64 8 28 28 128 1 2 0"; do
  i=$((i+1))
  echo "== $g" 
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/t$i -o p -- python3 scripts/bench_conv.py $g --ops fwd,dgrad,dacc,wgrad > $O/t$i.log 2>&1 || { tail -5 $O/t$i.log; exit 1; }
  grep planner $O/t$i.log
  python3 scripts/prof_summary.py $O/t$i/p_results.db --steps 1 --top 12 | cut -c1-150
done
rm -rf $O/t*/
