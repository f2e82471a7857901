#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "multi_context or sharded" > $O/shard_tests.log 2>&1; echo "shard tests rc $?"; tail -30 $O/shard_tests.log
timeout 300 python tools/gpu_sharded_prof.py 60 > $O/sharded_prof.txt 2>&1; grep -v amdgpu.ids $O/sharded_prof.txt | tail -20
timeout 900 python -m pytest tests/test_cli_gpu.py -m gpu -q > $O/cli_tests.log 2>&1; echo "cli tests rc $?"; tail -8 $O/cli_tests.log
