#!/bin/bash
# Timing experiment: build libvc3d variants with parts of the halo tap loop removed (results are wrong on
# purpose) to see which resource bounds the loop.  Usage: scripts/halo_ablate.sh build | run
set -e
cd "$(dirname "$0")/../video-classification_amd/csrc"
VARIANTS="${VARIANTS:-0 32}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DVC_HALO_ABLATE=$v -c halo_conv.hip -o /tmp/halo_ab$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o igemm_conv.o pw_conv.o /tmp/halo_ab$v.o wgrad_conv.o wgrad9_conv.o stem_wgrad.o stem_fwd.o elementwise.o lstm.o -o ../vc3d/libvc3d_ab$v.so ) &
  done
  wait
  ls -la ../vc3d/*.so
else
  cd ../..
  for v in $VARIANTS; do
    echo "== ablate $v"
    if [ -n "$STAMPS" ]; then
      VC3D_LIB=$PWD/video-classification_amd/vc3d/libvc3d_ab$v.so timeout -k 10 120 python scripts/halo_stamps.py 64 64 8 28 28 64 2>&1 | tail -9
      VC_HALO_NO_2WG=1 VC3D_LIB=$PWD/video-classification_amd/vc3d/libvc3d_ab$v.so timeout -k 10 120 python scripts/halo_stamps.py 64 64 8 28 28 64 2>&1 | tail -7
    else
      VC3D_LIB=$PWD/video-classification_amd/vc3d/libvc3d_ab$v.so timeout -k 10 120 python scripts/bench_conv.py 64 64 8 28 28 64 3 1 1 --ops fwd,dgrad --force 8,4,112,1,64,0 --force 8,4,112,1,64,1 2>&1 | tail -3
    fi
  done
fi
