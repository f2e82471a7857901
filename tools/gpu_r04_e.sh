#!/bin/bash
# configs[2] alone under rocprofv3 (kernel stats of the round's final state)
O=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2 -o s -- python $GRAFT_REPO_ROOT/tools/gpu_config2_prof.py > $O/config2.log 2>&1 < /dev/null
grep workload $O/config2.log | cut -c1-700
f=$(find $O/cfg2 -name "s_kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; head -14 "$f" | cut -c1-150; fi
rm -rf $O/cfg2
