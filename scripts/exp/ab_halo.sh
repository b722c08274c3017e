set -e
B=video-classification_amd/vc3d/libvc3d_base.so
for g in "64 64 8 28 28 64 3 1 1" "64 128 4 14 14 128 3 1 1" "64 256 2 7 7 256 3 1 1"; do
  echo "== $g base"; VC3D_LIB=$PWD/$B timeout -k 10 120 python scripts/bench_conv.py $g --ops fwd,dgrad
  echo "== $g new";  timeout -k 10 120 python scripts/bench_conv.py $g --ops fwd,dgrad
done
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -x -q -k "conv3d or no_alloc or pointwise or dgrad" --timeout 120 --timeout-method thread 2>&1 | tail -3
