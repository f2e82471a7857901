#!/bin/bash
# resampler kernels: parity (zita restatement, VResampler, speed detection) + configs[2] timing under rocprofv3
mkdir -p gpurun_out/r04d
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speed.py -q -m gpu -x -k "resample or other_rates or roundtrip_48k or speed" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04d/cfg2 -o s -- python $GRAFT_REPO_ROOT/tools/gpu_config2_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r04d/config2.log 2>&1
grep workload $GRAFT_REPO_ROOT/gpurun_out/r04d/config2.log | cut -c1-600
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04d/cfg2 -name "s_kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-140
