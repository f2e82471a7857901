#!/bin/bash
# round 4, last GPU call: the whole GPU suite and the bench line at the final commit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 420 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1 < /dev/null; echo "gpu tests rc $?"; tail -3 $O/gpu_tests.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -1
timeout 240 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("viterbi_form", {}).get("one_launch_kernel"), d["roofline"]["kernel"], d["roofline"]["frac"])
print(d.get("cpu_baseline"))
print(d["config"].get("detect_speed_config"))
PY
