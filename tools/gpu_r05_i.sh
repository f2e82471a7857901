#!/bin/bash
# round 5, GPU call I: `add` of a batch of clips in one launch per stage -- equality tests, the clip tests, the per-key A/B, the clips bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "add_batch or key_tables or clip_batch or multi_context or fuzz" 2>&1 | tail -5
timeout 300 python tools/gpu_clip_keys_ab.py 1024 4 2>&1 | tail -6
timeout 600 python bench.py --config clips --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
e = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = e['config']['clip_batch_config']
print(e['value'], e['ms_per_step'], e['config']['clips_with_payload'], {k: v for k, v in c.items() if k != 'kernels_one_group_of_64_clips'})
for k in c['kernels_one_group_of_64_clips']: print('  ', k['scope'], k['ms_per_call'], k['frac'])
"
