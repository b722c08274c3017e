# generic weight-gradient kernel: register prefetch depth (VC_WGRAD_STAGES) and the 256-co tile, per shape
for g in "64 64 8 28 28 256 1 1 0" "64 256 8 28 28 64 1 1 0" "64 128 4 14 14 512 1 1 0" "64 512 4 14 14 128 1 1 0" "64 256 2 7 7 1024 1 1 0" "64 1024 2 7 7 256 1 1 0" "64 512 1 4 4 2048 1 1 0" "64 64 8 28 28 128 3 2 1" "64 128 4 14 14 256 3 2 1" "64 256 2 7 7 512 3 2 1" "64 128 8 28 28 128 3 2 1"; do
  for st in 1 2 3; do
    r=$(VC_WGRAD_STAGES=$st timeout -k 10 120 python3 scripts/bench_conv.py $g --ops wgrad | tail -1) || exit 1
    echo "[$g] stages=$st $r"
  done
done
r=$(VC_WGRAD_NO_256=1 timeout -k 10 120 python3 scripts/bench_conv.py 64 64 8 28 28 256 1 1 0 --ops wgrad | tail -1); echo "[64 64 8 28 28 256 no256] $r"
