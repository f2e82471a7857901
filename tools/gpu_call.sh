#!/bin/bash
# One parametrised GPU call instead of a script per call (rounds 1 - 5 left tools/gpu_rNN_[a-j].sh behind: `git log` has them; what each
# call measured is in profiles/rNN/README.md).  Run on a GPU box from the repository root; everything lands in gpurun_out/<tag>/.
#   tools/gpu_call.sh <tag> <step> [<step> ...]
# steps:  tests            the whole GPU suite (pytest -m gpu)
#         tests:<expr>     pytest -m gpu -k <expr>
#         bench            bench.py, default flags (configs[1])          bench8h | benchclips   the other configurations
#         k4s              tools/gpu_k4s_forms.py + tools/gpu_k4s_alone.py (forms of the refinement's sliding DFT)
#         census           tools/gpu_census_three_way.py (this library against both builds of the reference, identical bytes)
#         io               tools/gpu_io_sweep.py 60 quick + tools/gpu_first_calls.py
#         k14              tools/gpu_speed_compare_ab.py (forms of K14 inside configs[2])        stagger   tools/gpu_stagger.py
#         final            tools/gpu_final.sh <tag>: the measurements behind profiles/<tag>
set -u
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16 TMPDIR=/tmp
for step in "$@"; do
  case $step in
    tests)      timeout 2300 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/gpu_tests.log; cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null ;;
    tests:*)    timeout 1500 python -m pytest tests -q -m gpu -k "${step#tests:}" > $O/gpu_tests_k.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/gpu_tests_k.log ;;
    bench)      timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json; echo ;;
    bench8h)    timeout 900 python bench.py --config 8h --steps 8 --warmup 5 > $O/bench_8h.json 2>/dev/null; tail -c 300 $O/bench_8h.json; echo ;;
    benchclips) timeout 900 python bench.py --config clips --steps 3 --warmup 1 > $O/bench_clips.json 2>/dev/null; tail -c 600 $O/bench_clips.json; echo ;;
    k4s)        timeout 300 python tools/gpu_k4s_forms.py 60 3,4,5 2>&1 | grep -v amdgpu > $O/k4s_forms.txt; tail -8 $O/k4s_forms.txt
                K4S_LD=64 timeout 200 python tools/gpu_k4s_alone.py 4,5 12750 1,33,65 0,8 2>&1 | grep -v amdgpu > $O/k4s_alone.txt; cat $O/k4s_alone.txt ;;
    census)     timeout 1800 python tools/gpu_census_three_way.py 1 14 > $O/census_three_way.log 2>&1; echo "census rc $?"; tail -1 $O/census_three_way.log | cut -c1-1500; cp gpurun_out/census_three_way.json $O/ 2>/dev/null ;;
    io)         timeout 300 python tools/gpu_io_sweep.py 60 quick > $O/io_sweep.log 2>&1; cp gpurun_out/io_sweep.json $O/ 2>/dev/null; tail -1 $O/io_sweep.log | cut -c1-900
                timeout 300 python tools/gpu_first_calls.py 2>&1 | grep -v amdgpu > $O/first_calls.txt; tail -5 $O/first_calls.txt ;;
    k14)        timeout 600 python tools/gpu_speed_compare_ab.py 2>&1 | grep -v amdgpu > $O/speed_compare_ab.txt; cat $O/speed_compare_ab.txt | cut -c1-420 ;;
    stagger)    timeout 600 python tools/gpu_stagger.py 2>&1 | grep -v amdgpu > $O/chunk_stagger.txt; cat $O/chunk_stagger.txt ;;
    final)      bash tools/gpu_final.sh $TAG ;;
    *)          echo "unknown step $step" ;;
  esac
done
