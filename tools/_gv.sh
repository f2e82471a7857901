mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r3w/tests.log 2>&1; echo rc $?; grep -E "passed|failed" gpurun_out/r3w/tests.log | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-detect-speed-config --no-cpu-baseline > gpurun_out/r3w/bench.json 2>gpurun_out/r3w/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3w/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['kernels_ms_per_step_alone']); print(d.get('parity'))
PY
