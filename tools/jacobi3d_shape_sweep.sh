#!/bin/bash
# GPU experiment: does the 2-sweep pass of the 3D Jacobi care about power-of-two row / plane strides?  (us per pass; cells differ)
for S in "64 512 512" "64 512 480" "64 512 540" "64 500 540" "64 520 480" "60 500 540" "64 528 496"; do
  python tools/jacobi3d_time.py $S 100 2>&1 | tail -1
done
