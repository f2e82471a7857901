#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
bash tools/profile_bench.sh r03 > $O/profile.log 2>&1; tail -25 $O/profile.log
timeout 900 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2> $O/bench_clips.err; tail -c 700 $O/bench_clips.json
timeout 2400 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc $?"; tail -8 $O/gputests.log | cut -c1-300
