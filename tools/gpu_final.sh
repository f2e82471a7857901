#!/bin/bash
# the measurements behind profiles/rNN in ONE GPU call: rocprofv3 passes of the headline configuration (tools/profile_bench.sh), the bench
# lines of every configuration, stand-alone kernel stats + counter passes of configs[2] and configs[4], the three-way tie census, the multi-GPU
# protocol on one device, the whole GPU suite.   usage (GPU box, repository root): tools/gpu_final.sh r05
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
bash tools/profile_bench.sh $TAG > $O/profile.log 2>&1; tail -14 $O/profile.log
# the bench lines below shall carry the traffic of THESE passes: put it where bench.py looks (the box's copy of profiles/; the container copies the
# same files to profiles/$TAG afterwards)
mkdir -p profiles/$TAG && cp $O/traffic.json profiles/$TAG/traffic.json && \
  python -c "import sys; sys.argv = ['x']; import bench; print (bench.hip_sources_digest())" > profiles/$TAG/HIP_SOURCES_SHA256
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json
timeout 600 python bench.py --steps 10 --warmup 3 --sharded --no-e2e --no-cpu-baseline --no-detect-speed-config > $O/bench_n1_sharded.json 2>/dev/null
timeout 900 python bench.py --config 8h --steps 8 --warmup 5 > $O/bench_8h.json 2>/dev/null; tail -c 300 $O/bench_8h.json
timeout 900 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2>/dev/null; tail -c 600 $O/bench_clips.json
timeout 600 python bench.py --gpus 2 --same-device --steps 5 --warmup 2 > $O/bench_same_device_2.json 2>/dev/null; tail -c 300 $O/bench_same_device_2.json
timeout 300 python tools/gpu_sharded_prof.py 60 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket" > $O/sharded_prof.txt; cat $O/sharded_prof.txt
# configs[2] / configs[4]: stand-alone kernel durations and counters
prof () { (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 "$@"); }
prof --kernel-trace --stats --output-format csv -d $O/c2 -o s -- python $R/tools/gpu_config2_one_lane.py 3 > $O/config2_one_lane.log 2>&1
cp $(find $O/c2 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_config2_one_lane.csv 2>/dev/null; rm -rf $O/c2; tail -1 $O/config2_one_lane.log
for p in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $p | cut -d" " -f1)
  prof --pmc $p --output-format csv -d $O/c2pmc_$n -o s -- python $R/tools/gpu_config2_one_lane.py 1 > $O/config2_pmc_$n.log 2>&1
done
python tools/pmc_table.py $(find $O/c2pmc_SQ_INSTS_VALU -name "*counter_collection.csv") > $O/config2_pmc_valu.txt 2>&1
python tools/pmc_traffic.py $(find $O/c2pmc_FETCH_SIZE $O/c2pmc_WRITE_SIZE -name "*counter_collection.csv") > $O/config2_traffic.json 2>/dev/null
rm -rf $O/c2pmc_*
prof --kernel-trace --stats --output-format csv -d $O/c4 -o s -- python $R/tools/gpu_clips_one_group.py 3 > $O/clips_one_group.log 2>&1
cp $(find $O/c4 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_clips_one_group.csv 2>/dev/null; rm -rf $O/c4; tail -1 $O/clips_one_group.log
timeout 200 python tools/gpu_clip_keys_ab.py 1024 4 2>&1 | grep -v amdgpu > $O/clip_keys_ab.txt; cat $O/clip_keys_ab.txt
[ -x tools/lds_b128_probe ] && tools/lds_b128_probe > $O/lds_b128_probe.txt 2>&1
timeout 300 python tools/gpu_io_sweep.py 60 quick > $O/io_sweep.log 2>&1; cp gpurun_out/io_sweep.json $O/io_sweep.json; tail -1 $O/io_sweep.log | cut -c1-900
timeout 300 python tools/gpu_first_calls.py > $O/first_calls.txt 2>&1; grep -v amdgpu $O/first_calls.txt
timeout 1200 python tools/gpu_census_three_way.py 1 14 > $O/census_three_way.log 2>&1; echo "census rc $?"; tail -1 $O/census_three_way.log | cut -c1-1500; cp gpurun_out/census_three_way.json $O/ 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 $O/gpu_tests.log; cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
git -C $R rev-parse HEAD > $O/COMMIT 2>/dev/null || true       # (no .git on the GPU box: written again, from the container, when the files are copied to profiles/)
python -c "import sys; sys.argv = ['x']; import bench; print (bench.hip_sources_digest())" > $O/HIP_SOURCES_SHA256
