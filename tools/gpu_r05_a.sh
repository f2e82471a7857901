#!/bin/bash
# round 5, GPU call A: the file-level tests after the I/O rework, the I/O sweep, first-call census, a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_cli_gpu.py tests/test_gpu_multidevice.py -x -q -m gpu > gpurun_out/r05a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a_pytest.log
tail -5 gpurun_out/r05a_pytest.log
timeout 600 python tools/gpu_io_sweep.py > gpurun_out/r05a_io_sweep.log 2>&1; tail -3 gpurun_out/r05a_io_sweep.log
timeout 600 python tools/gpu_first_calls.py > gpurun_out/r05a_first_calls.log 2>&1; tail -5 gpurun_out/r05a_first_calls.log
timeout 900 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; tail -c 3000 gpurun_out/r05a_bench.json
