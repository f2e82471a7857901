#!/bin/bash
# round 5, GPU call E: K14 with the column loop software-pipelined -- speed tests + configs[2] timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_speed.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/gpu_config2_prof.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k != 'kernels_one_lane'})
for k in d['kernels_one_lane'][:6]: print('  ', k['scope'], k['scopes_per_call'], k['avg_ms'], k['ms_per_call'])
"
