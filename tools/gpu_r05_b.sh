#!/bin/bash
# round 5, GPU call B: where the file level add spends its time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/gpu_io_sweep.py 60 quick > gpurun_out/r05b_io_sweep.log 2>&1; grep -v amdgpu.ids gpurun_out/r05b_io_sweep.log | tail -12
