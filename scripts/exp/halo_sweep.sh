# plan sweep of the halo kernel at the small-plane geometries (layers 3 and 4 of R3D-18)
export VC_HALO_VERBOSE=1
echo "== layer4 512->512 1x4x4"
timeout -k 10 200 python3 scripts/bench_conv.py 64 512 1 4 4 512 3 1 1 --ops fwd,dgrad \
  --force 4,16,16,8 --force 4,16,16,4 --force 4,16,16,2 --force 8,32,16,8 --force 8,32,16,4 --force 4,8,16,8 --force 4,8,16,4 --force 8,16,16,8 --force 8,16,16,4 --force 4,4,16,8 --force 4,4,16,4 2>&1 | grep -v "^halo plan.*NW" | sort -u | head -40 || exit 1
echo "== layer3 256->256 2x7x7"
timeout -k 10 200 python3 scripts/bench_conv.py 64 256 2 7 7 256 3 1 1 --ops fwd,dgrad \
  --force 4,5,49,4 --force 4,5,49,2 --force 4,5,49,1 --force 8,10,49,4 --force 8,10,49,2 --force 8,10,49,1 --force 4,4,49,4 --force 4,4,49,2 --force 8,8,49,4 --force 8,8,49,2 --force 4,2,49,4 --force 4,2,49,4,64,1 --force 8,2,49,4,64,1 2>&1 | grep -v "^halo plan.*NW" | sort -u | head -40 || exit 1
echo "== plans"
VC_HALO_VERBOSE=1 timeout -k 10 100 python3 scripts/bench_conv.py 64 512 1 4 4 512 3 1 1 --ops fwd,dgrad 2>&1 | grep "halo plan" | sort | uniq -c
VC_HALO_VERBOSE=1 timeout -k 10 100 python3 scripts/bench_conv.py 64 256 2 7 7 256 3 1 1 --ops fwd,dgrad 2>&1 | grep "halo plan" | sort | uniq -c
