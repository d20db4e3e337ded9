#!/bin/bash
# Timing probes of conv3_wbf_kernel (FNX_PRECISION_BF16X6): variants of libfluidnet_hip.so with pieces of the kernel's phase
# compiled out (-DWB_ABL=bits: 1 split+store units, 2 raw loads + combos, 4 MFMAs, 8 B reads, 16 A loads, 32 halo DMA).
#   here (no GPU):   tools/wbf_ablation.sh build 0 3 4 8 16 32 59 63      -> gpurun_out/../variants/libfluidnet_hip_<bits>.so
#   GPU box:         tools/wbf_ablation.sh run 2d|3d                      -> one line per variant
set -u
cd "$(dirname "$0")/.."
V=variants
if [ "$1" = build ]; then
  shift; mkdir -p $V
  for b in "$@"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -Wno-unused-value -DWB_ABL=$b \
        -c fluidnet_cxx_amd/csrc/fnx_cnn.hip -o $V/fnx_cnn_$b.o 2>/dev/null &&
      objs=$(ls fluidnet_cxx_amd/build/*.o | grep -v "fnx_cnn.o" | grep -v "hip-amdgcn\|host-x86") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libfluidnet_hip_$b.so $objs $V/fnx_cnn_$b.o && rm $V/fnx_cnn_$b.o && echo built $b ) &
  done
  wait
else
  cp fluidnet_cxx_amd/libfluidnet_hip.so /tmp/libfluidnet_hip.keep
  for f in $V/libfluidnet_hip_*.so; do
    b=${f##*_}; b=${b%.so}
    cp $f fluidnet_cxx_amd/libfluidnet_hip.so
    echo -n "WB_ABL=$b: "; timeout 300 python tools/cnn_mode_time.py $2 bf16x6 2>&1 | grep "bf16x6:" | sed 's/.*conv_bf16/conv_bf16/'
  done
  cp /tmp/libfluidnet_hip.keep fluidnet_cxx_amd/libfluidnet_hip.so
fi
