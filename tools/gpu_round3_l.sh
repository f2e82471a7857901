#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
python tools/gpu_variants.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "sync or silence or refine or search or fuzz or decode or get" > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log | cut -c1-300
timeout 900 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2> $O/bench_clips.err; python -c "
import json; d=json.loads(open('gpurun_out/r3l/bench_clips.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['clips_with_payload'], d['config']['clip_batch_config'])"
