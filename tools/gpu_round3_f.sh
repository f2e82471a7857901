#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests -m gpu -q -k "viterbi or multi_context or several_devices or sharded or add_parity or streaming" > $O/tests_a.log 2>&1; echo "tests rc $?"; tail -12 $O/tests_a.log
python tools/gpu_add_only.py 10 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stft_roofline'], d['kernels_ms_per_step_alone'], d['config'], d.get('parity'))
PY
