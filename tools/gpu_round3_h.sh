#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stft_roofline']['frac'], d['kernels_ms_per_step_alone']['viterbi_kernel'], d['config']['payload_matches'], d['e2e'].get('payload_matches'), d['detect_speed_config'].get('payload_matches'), d.get('parity'))
PY
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_multi_context_get_equals_single > $O/gputests.log 2>&1; echo "gpu tests rc $?"; tail -15 $O/gputests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k multi_context > $O/multi.log 2>&1; echo "multi rc $?"; tail -30 $O/multi.log | cut -c1-300
