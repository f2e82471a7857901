#!/bin/bash
# round 5, GPU call D: sweep with the final defaults + CLI timing detail, then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/gpu_io_sweep.py 60 quick > gpurun_out/r05d_io_sweep.log 2>&1; grep -v amdgpu.ids gpurun_out/r05d_io_sweep.log | tail -14
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05d_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05d_pytest.log
tail -6 gpurun_out/r05d_pytest.log
