#!/bin/bash
# round 4, last GPU call: the whole GPU suite and the bench lines at the final commit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
timeout 600 python bench.py --config 8h --steps 8 --warmup 12 > $O/bench_8h.json 2>/dev/null
timeout 300 python tools/gpu_variants.py 2>&1 | grep -v amdgpu.ids > $O/variants.txt
python - <<PY
import json
for f in ("bench_n1.json", "bench_8h.json"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("viterbi_form"), (d.get("traffic_source") or {}).get("profile_dir"))
    if f == "bench_n1.json": print(d["kernels_ms_per_step_alone"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
