export TMPDIR=/tmp
O=gpurun_out/dsprof
mkdir -p $O
i=0
for g in "64 1024 2 7 7 2048 1 2 0" "64 512 4 14 14 1024 1 2 0" "64 256 8 28 28 512 1 2 0"; do
  i=$((i+1))
  echo "== $g"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/t$i -o p -- python3 scripts/bench_conv.py $g --ops fwd,fwdns,dgrad,wgrad > $O/t$i.log 2>&1 || { tail -5 $O/t$i.log; exit 1; }
  grep planner $O/t$i.log
  python3 scripts/prof_summary.py $O/t$i/p_results.db --steps 1 --top 8 | cut -c1-150
done
rm -rf $O/t*/
