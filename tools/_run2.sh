python -m pytest tests/test_parity_gpu.py -q -m gpu -k "jacobi or slab or simulate or axis or ops_vs" -x 2>&1 | tail -2
python tools/jacobi3d_time.py 64 512 512 100 2>&1 | tail -1
python tools/jacobi3d_time.py 256 256 256 100 2>&1 | tail -1
python tools/jacobi3d_time.py 60 500 540 100 2>&1 | tail -1
FNX_JACOBI_QUAD=0 python tools/jacobi3d_time.py 64 512 512 100 2>&1 | tail -1
