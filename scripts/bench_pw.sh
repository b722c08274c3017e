#!/bin/bash
# 1x1x1 (pointwise) conv shapes of R3D-50 at batch 64 through scripts/bench_conv.py
set -o pipefail
for g in "64 64 8 28 28 256" "64 256 8 28 28 64" "64 256 8 28 28 128" "64 128 4 14 14 512" "64 512 4 14 14 128" "64 256 2 7 7 1024" "64 1024 2 7 7 256" "64 512 1 4 4 2048" "64 2048 1 4 4 512"; do
  echo "== $g"
  timeout -k 10 120 python scripts/bench_conv.py $g 1 1 0 || exit 1
done
