#!/bin/bash
# GPU experiment: 2-sweep pass time of the 3D Jacobi against the solver's footprint (2 x p + div + mask = 13 B/cell),
# 512x512 planes, D swept across the 256 MiB Infinity Cache.   bash tools/jacobi3d_footprint_sweep.sh > gpurun_out/j3d_sweep.txt
for D in 24 32 40 48 56 60 64 68 72 80 96 128 192; do
  python tools/jacobi3d_time.py $D 512 512 100 2>&1 | tail -1
done
