#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
python tools/gpu_debug_determinism.py 2>&1 | grep -v amdgpu.ids | tail -12
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "multi_context or two_ranks" 2>&1 | tail -3; done
timeout 900 python -m pytest tests/test_gpu_fullsize_ref.py -m gpu -q -k "config4" > $O/config4.log 2>&1; echo "config4 rc $?"; tail -12 $O/config4.log | cut -c1-400
timeout 900 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2> $O/bench_clips.err; tail -c 1500 $O/bench_clips.json; tail -3 $O/bench_clips.err
