#!/bin/bash
# round 4, third GPU call: Viterbi with the wave-wide walk back and the choice by launch cost, protocol overhead on one device, 8 h record
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "viterbi" > $O/viterbi_tests.log 2>&1; echo "viterbi tests rc $?"; tail -3 $O/viterbi_tests.log
if ! grep -q "passed" $O/viterbi_tests.log || grep -q "failed" $O/viterbi_tests.log; then echo "STOP: viterbi tests"; exit 1; fi
timeout 400 python tools/gpu_variants.py 2>&1 | grep -v amdgpu.ids | grep "viterbi one launch\|probe" > $O/variants.txt; cat $O/variants.txt
timeout 300 python tools/gpu_sharded_prof.py 60 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket" > $O/sharded_prof.txt; cat $O/sharded_prof.txt
timeout 300 python bench.py --gpus 2 --same-device --steps 5 --warmup 2 > $O/bench_same_device_2.json 2> $O/bench_same_device_2.err; echo "same-device rc $?"; tail -c 700 $O/bench_same_device_2.json; tail -3 $O/bench_same_device_2.err
timeout 900 python tools/gpu_8h_vs_ref.py 8 > $O/8h.log 2>&1; echo "8h rc $?"; tail -3 $O/8h.log | cut -c1-1500
timeout 400 python bench.py --steps 20 --warmup 5 --no-e2e --no-detect-speed-config > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"; python - <<PY
import json
try:
    d = json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["kernel"], d["roofline"]["frac"], d["viterbi_form"])
    print(d["kernels_ms_per_step_alone"])
except Exception as e:
    print("bench line unreadable", e)
PY
