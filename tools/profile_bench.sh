#!/bin/bash
# tools/profile_bench.sh -- the rocprofv3 passes behind profiles/rNN (run on a GPU box from the repository root):
#   kernel trace + stats of bench.py with the chunks on concurrent lanes and on ONE lane (stand-alone durations),
#   separate --pmc passes on one lane, as MI355X_MICROARCH.md prescribes (never combined with a trace):
#     FETCH_SIZE | WRITE_SIZE                         -> HBM traffic per launch (tools/pmc_traffic.py -> traffic.json)
#     SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES            -> instructions per launch
#     SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
#     SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
#     GRBM_GUI_ACTIVE                                 -> effective shader clock of every kernel = cycles / stand-alone duration
# usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
# (--viterbi-form chain: what the untraced run picks on a box with cheap launches; under the tracer a launch costs several times more and
# the probe would choose the one-launch kernel -- profiles/rNN must show the kernels the bench line's step runs)
B="python $R/bench.py --no-e2e --no-cpu-baseline --no-detect-speed-config --viterbi-form ${VITERBI_FORM:-chain}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lanes -o s -- $B --steps 5 --warmup 3 > $OUT/lanes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_lane -o s -- $B --steps 5 --warmup 3 --lanes 1 > $OUT/one_lane.log 2>&1
pass () { rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$1 -o s -- $B --steps 2 --warmup 1 --lanes 1 > $OUT/pmc_$1.log 2>&1; }
pass FETCH_SIZE
pass WRITE_SIZE
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
pass SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass GRBM_GUI_ACTIVE
cd $R
# summaries (small text files; the raw per-dispatch CSVs of the two traffic passes are kept as well)
cp $(find $OUT/lanes -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_lanes.csv 2>/dev/null
cp $(find $OUT/one_lane -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_one_lane.csv 2>/dev/null
python tools/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv") > $OUT/traffic.json 2>/dev/null
for p in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE; do
  python tools/pmc_table.py $(find $OUT/pmc_$p -name "*counter_collection.csv") > $OUT/pmc_$p.txt 2>&1
done
python tools/clock_table.py $OUT/rocprofv3_kernel_stats_one_lane.csv $OUT/pmc_GRBM_GUI_ACTIVE.txt > $OUT/effective_clock.txt 2>&1
for p in FETCH_SIZE WRITE_SIZE; do cp $(find $OUT/pmc_$p -name "*counter_collection.csv" | head -1) $OUT/rocprofv3_pmc_$p.csv 2>/dev/null; done
rm -rf $OUT/lanes $OUT/one_lane $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU $OUT/pmc_SQ_ACTIVE_INST_VALU $OUT/pmc_SQ_LDS_BANK_CONFLICT $OUT/pmc_GRBM_GUI_ACTIVE
ls -la $OUT | head -30
cat $OUT/effective_clock.txt
for f in lanes one_lane; do tail -n 1 $OUT/$f.log | cut -c1-200; done
