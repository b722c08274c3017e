# per-kernel times of the stride-2 3x3x3 data gradients (layer2.0 / 3.0 / 4.0 conv1), one rocprof trace each
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in "64 64 8 28 28 128" "64 128 4 14 14 256" "64 256 2 7 7 512"; do
  tag=$(echo $g | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sd_$tag -o t -- python3 $R/scripts/bench_conv.py $g 3 2 1 --ops ${OPS:-fwdns,dgrad} > $R/gpurun_out/sd_$tag.log 2>&1
  grep planner $R/gpurun_out/sd_$tag.log
  python3 $R/scripts/prof_summary.py $R/gpurun_out/sd_$tag/t_results.db --top 8 | cut -c1-60,110-180
done
