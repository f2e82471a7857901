#!/bin/bash
# round 5, GPU call J: `add` with a key per clip, the tables of all keys first -- equality tests, A/B against "tables beside the clips", the clips bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "add_batch or key_tables or multi_context or clip_batch" 2>&1 | tail -3
timeout 120 python tools/gpu_clip_keys_ab.py 1024 4 2>&1 | tail -8
timeout 200 python bench.py --config clips --steps 3 --warmup 1 2>/dev/null > gpurun_out/bench_clips_j.json; python -c "
import json
e = json.loads(open('gpurun_out/bench_clips_j.json').read().strip().splitlines()[-1]); c = e['config']['clip_batch_config']
print(e['value'], e['ms_per_step'], e['config']['clips_with_payload'], {k: v for k, v in c.items() if k != 'kernels_one_group_of_64_clips'})
"
