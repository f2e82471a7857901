#!/bin/bash
# the measurements behind profiles/rNN: rocprofv3 passes (tools/profile_bench.sh), the bench lines of every configuration, the sharded path
# with one rank, the A / B of the kernel variants.   usage (GPU box, repository root): tools/gpu_final.sh r04
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
bash tools/profile_bench.sh $TAG > $O/profile.log 2>&1; tail -14 $O/profile.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json
timeout 600 python bench.py --steps 10 --warmup 3 --sharded --no-e2e --no-cpu-baseline --no-detect-speed-config > $O/bench_n1_sharded.json 2>/dev/null
timeout 900 python bench.py --config 8h --steps 8 --warmup 12 > $O/bench_8h.json 2>/dev/null; tail -c 300 $O/bench_8h.json
timeout 900 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2>/dev/null; tail -c 600 $O/bench_clips.json
python tools/gpu_variants.py 2>&1 | grep -v amdgpu.ids > $O/variants.txt; cat $O/variants.txt
timeout 300 python tools/gpu_sharded_prof.py 60 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket" > $O/sharded_prof.txt; cat $O/sharded_prof.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2 -o s -- python $R/tools/gpu_config2_prof.py > $O/config2_detect_speed.log 2>&1)
cp $(find $O/cfg2 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_config2_detect_speed.csv 2>/dev/null; rm -rf $O/cfg2; tail -1 $O/config2_detect_speed.log | cut -c1-400
timeout 900 python tools/gpu_tie_census.py 3.6 96 > $O/census.log 2>&1; echo "census rc $?"; tail -1 $O/census.log | cut -c1-600
timeout 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 $O/gpu_tests.log
git -C $R rev-parse HEAD > $O/COMMIT 2>/dev/null || true
