#!/bin/bash
# tools/profile_bench.sh -- the rocprofv3 passes behind profiles/rNN (run on a GPU box from the repository root):
#   kernel trace + stats of bench.py with the chunks on concurrent lanes and on ONE lane (stand-alone durations),
#   two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on one lane, as MI355X_MICROARCH.md prescribes.
# usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/*.csv
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-e2e --no-cpu-baseline --no-detect-speed-config"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lanes -o s -- $B --steps 5 --warmup 3 > $OUT/lanes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_lane -o s -- $B --steps 5 --warmup 3 --lanes 1 > $OUT/one_lane.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o s -- $B --steps 3 --warmup 2 --lanes 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o s -- $B --steps 3 --warmup 2 --lanes 1 > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
for f in lanes one_lane; do tail -n 1 $OUT/$f.log | cut -c1-300; done
