#!/bin/bash
# Timing probes of conv3_wino3_kernel: variants of libfluidnet_hip.so with pieces of the kernel's phase compiled out
# (-DW3_ABL=bits: 1 phase barrier, 2 MFMAs, 4 input transform, 8 halo fetch + store, 16 weight DMA, 32 operand reads, 64 epilogue;
# 128 every halo load from one cache-resident 8 KB, 256 every full-tile store into 64 KB per workgroup -- the same instructions
# without their memory traffic; bits 2 and 32 leave the MFMA operands undefined and the compiler then drops MFMAs: not usable).
#   here (no GPU):   tools/wino3_ablation.sh build 0 1 2 ...      -> variants/libfluidnet_hip_w3_<bits>.so
#   GPU box:         tools/wino3_ablation.sh run 2d|3d            -> one line per variant
set -u
cd "$(dirname "$0")/.."
V=variants
if [ "$1" = build ]; then
  shift; mkdir -p $V
  for b in "$@"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -Wno-unused-value -DW3_ABL=$b \
        -c fluidnet_cxx_amd/csrc/fnx_cnn.hip -o $V/fnx_cnn_w3_$b.o 2>/dev/null &&
      objs=$(ls fluidnet_cxx_amd/build/*.o | grep -v "fnx_cnn.o" | grep -v "hip-amdgcn\|host-x86") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libfluidnet_hip_w3_$b.so $objs $V/fnx_cnn_w3_$b.o && rm $V/fnx_cnn_w3_$b.o && echo built $b ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
else
  cp fluidnet_cxx_amd/libfluidnet_hip.so /tmp/libfluidnet_hip.keep
  for f in $(ls $V/libfluidnet_hip_w3_*.so | sort -t_ -k4 -n); do
    b=${f##*_}; b=${b%.so}
    cp $f fluidnet_cxx_amd/libfluidnet_hip.so
    echo -n "W3_ABL=$b: "; timeout 300 python tools/cnn_mode_time.py $2 fp32 2>&1 | grep "fp32:" | sed 's/.*forward/forward/'
  done
  cp /tmp/libfluidnet_hip.keep fluidnet_cxx_amd/libfluidnet_hip.so
fi
