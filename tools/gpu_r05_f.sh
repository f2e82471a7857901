#!/bin/bash
# round 5, GPU call F: clip path changes (K7 padding skip, trimmed padded copies) -- clip tests against the reference + the clips bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize_ref.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "clip or config4 or batch or fuzz or silence" 2>&1 | tail -3
timeout 600 python bench.py --config clips --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
e = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = e['config']['clip_batch_config']
print(e['value'], e['ms_per_step'], e['config']['clips_with_payload'], {k: v for k, v in c.items() if k != 'kernels_one_group_of_64_clips'})
for k in c['kernels_one_group_of_64_clips']: print('  ', k['scope'], k['ms_per_call'], k['frac'])
"
