#!/bin/bash
# A/B of two builds of libfluidnet_hip.so on ONE box (boxes differ by a few per cent, so do runs minutes apart):
#   here:     [UNIT=fnx_jacobi] [ABFLAGS=-DX=1] tools/ab_libs.sh build <name> [git-rev]   -> variants/libfluidnet_hip_<name>.so from the working
#             tree's (or the revision's) csrc/$UNIT.hip (default fnx_cnn) and the other objects of the current build
#   GPU box:  tools/ab_libs.sh run <rounds> <name>... -- <command>     (round-robin over the builds; prints the command's output)
set -u
cd "$(dirname "$0")/.."
V=variants
if [ "$1" = build ]; then
  mkdir -p $V
  U=${UNIT:-fnx_cnn}
  extra=""; [ $U != fnx_cnn ] && [ $U != fnx_slab ] && [ $U != fnx_peer ] && extra="-ffp-contract=off"    # (build.py's per-unit flags)
  src=fluidnet_cxx_amd/csrc/$U.hip
  if [ $# -ge 3 ]; then git show $3:$src > fluidnet_cxx_amd/csrc/.ab_$2.hip; src=fluidnet_cxx_amd/csrc/.ab_$2.hip; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -Wno-unused-value $extra ${ABFLAGS:-} \
      -c $src -o $V/${U}_$2.o 2>/dev/null &&
  objs=$(ls fluidnet_cxx_amd/build/*.o | grep -v "/$U.o" | grep -v "hip-amdgcn\|host-x86") &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libfluidnet_hip_$2.so $objs $V/${U}_$2.o && echo built $2
  rm -f $V/${U}_$2.o fluidnet_cxx_amd/csrc/.ab_$2.hip
else
  n=$2; shift 2
  names=()
  while [ "$1" != "--" ]; do names+=("$1"); shift; done
  shift
  cp fluidnet_cxx_amd/libfluidnet_hip.so /tmp/libfluidnet_hip.keep
  for i in $(seq $n); do
    for v in "${names[@]}"; do
      cp $V/libfluidnet_hip_$v.so fluidnet_cxx_amd/libfluidnet_hip.so
      echo -n "$v: "; "$@" 2>&1 | tail -n +1
    done
  done
  cp /tmp/libfluidnet_hip.keep fluidnet_cxx_amd/libfluidnet_hip.so
fi
