#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 300 python tools/gpu_sharded_prof.py 60 > $O/sharded_prof.txt 2>&1; grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|socket" $O/sharded_prof.txt | tail -20
timeout 2400 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc $?"; tail -12 $O/gputests.log
