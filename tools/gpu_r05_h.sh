#!/bin/bash
# round 5, GPU call H (the round's last kernel commit): the whole GPU suite, the bench lines of configs[1] / [3] / [4], the stand-alone kernel
# durations of one group of 64 clips -> gpurun_out/prof_r05h/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r05h
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/gpu_tests.log; cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json; echo
timeout 600 python bench.py --config 8h --steps 8 --warmup 5 > $O/bench_8h.json 2>/dev/null; tail -c 200 $O/bench_8h.json; echo
timeout 600 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2>/dev/null; tail -c 500 $O/bench_clips.json; echo
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -o s -- python $R/tools/gpu_clips_one_group.py 3 > $O/clips_one_group.log 2>&1)
cp $(find $O/c4 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_clips_one_group.csv 2>/dev/null; rm -rf $O/c4; tail -1 $O/clips_one_group.log
timeout 200 python tools/gpu_clip_keys_ab.py 1024 4 2>&1 | tail -7 > $O/clip_keys_ab.txt; cat $O/clip_keys_ab.txt
