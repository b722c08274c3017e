# implicit-GEMM kernel: depth of the operand ring (VC_IGEMM_NBUF) per shape, forward and data gradient
for g in "64 64 8 28 28 128 3 2 1" "64 128 4 14 14 256 3 2 1" "64 256 2 7 7 512 3 2 1" "64 64 8 28 28 128 1 2 0" "64 256 2 7 7 512 1 2 0" "64 128 8 28 28 128 3 2 1" "64 256 4 14 14 256 3 2 1" "64 512 2 7 7 512 3 2 1"; do
  for nb in 2 3 4; do
    r=$(VC_IGEMM_NBUF=$nb timeout -k 10 120 python3 scripts/bench_conv.py $g --ops fwdns,dgrad 2>/dev/null | tail -1) || exit 1
    echo "[$g] nbuf=$nb $r"
  done
done
