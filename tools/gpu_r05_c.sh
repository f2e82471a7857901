#!/bin/bash
# round 5, GPU call C: the file level paths after the one-mapping writer + cached rings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_cli_gpu.py -x -q -m gpu > gpurun_out/r05c_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05c_pytest.log
tail -4 gpurun_out/r05c_pytest.log
timeout 600 python tools/gpu_io_sweep.py 60 quick > gpurun_out/r05c_io_sweep.log 2>&1; grep -v amdgpu.ids gpurun_out/r05c_io_sweep.log | tail -12
