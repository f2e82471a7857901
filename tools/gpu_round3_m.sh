#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export GPU_MAX_HW_QUEUES=16
python tools/gpu_variants.py 2>&1 | grep -v amdgpu.ids | tail -18
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "viterbi or decode or get or golden" 2>&1 | tail -3
