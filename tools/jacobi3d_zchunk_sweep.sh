#!/bin/bash
# GPU experiment: 2-sweep pass time of the 512x512x64 Jacobi against the plane chunk per wave (FNX_JACOBI_ZCHUNK; default:
# as many equal chunks as fit one resident set = 22) -- latency- or throughput-bound?
for Z in 8 11 13 16 20 22 24 32 64; do
  echo -n "zchunk $Z: "; FNX_JACOBI_ZCHUNK=$Z python tools/jacobi3d_time.py 64 512 512 100 2>&1 | tail -1
done
