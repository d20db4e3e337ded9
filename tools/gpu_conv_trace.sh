#!/bin/bash
# Usage (GPU box): tools/gpu_conv_trace.sh <tag> <mode> [D H W] -> gpurun_out/<tag>_<mode>_layers.txt
set -u
tag=$1; mode=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out/ct_${tag}_$mode
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace -d $out -o r --output-format csv -- python tools/cnn3d_layers_probe.py $mode "$@" > $out/run.log 2>&1
python3 tools/show_conv_trace.py $out/r_kernel_trace.csv > gpurun_out/${tag}_${mode}_layers.txt 2>&1
rm -f $out/r_kernel_trace.csv
