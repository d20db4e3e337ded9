cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05_pytest_gpu.log 2>&1; tail -5 gpurun_out/r05_pytest_gpu.log
bash tools/gpu_refresh.sh r05h > gpurun_out/r05_refresh.log 2>&1; tail -3 gpurun_out/r05_refresh.log
ls gpurun_out/prof_r05h gpurun_out/pmc_r05h | head -50
