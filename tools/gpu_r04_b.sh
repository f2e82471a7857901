#!/bin/bash
# round 4, second GPU call: one-launch Viterbi with the coherent data path, the whole GPU suite, A / B of the Viterbi variants, the add
# slab sweep, the tie census, configs[3] against the compiled reference, the bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "viterbi" > $O/viterbi_tests.log 2>&1; echo "viterbi tests rc $?"; tail -3 $O/viterbi_tests.log
if ! grep -q "passed" $O/viterbi_tests.log || grep -q "failed" $O/viterbi_tests.log; then echo "STOP: viterbi tests"; exit 1; fi
timeout 400 python tools/gpu_variants.py 2>&1 | grep -v amdgpu.ids | grep "viterbi one launch" > $O/variants.txt; cat $O/variants.txt
timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -12 $O/gpu_tests.log
timeout 300 python tools/gpu_add_slabs.py > $O/add_slab_sweep.txt 2>&1; grep slab $O/add_slab_sweep.txt
timeout 900 python tools/gpu_tie_census.py 2.5 96 > $O/census.log 2>&1; echo "census rc $?"; grep -v amdgpu $O/census.log | tail -8
timeout 900 python tools/gpu_8h_vs_ref.py 8 > $O/8h.log 2>&1; echo "8h rc $?"; tail -3 $O/8h.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"; python - <<PY
import json
try:
    d = json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["kernel"], d["roofline"]["frac"])
    print(d["kernels_ms_per_step_alone"]); print(d["cpu_baseline"]); print(d["parity"])
except Exception as e:
    print("bench line unreadable", e)
PY
