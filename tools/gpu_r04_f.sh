#!/bin/bash
# configs[4] (1024 clips, a key per clip) under rocprofv3: kernel stats incl. K16 (frame_mod_table_kernel)
O=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/clips -o s -- python $GRAFT_REPO_ROOT/bench.py --config clips --steps 2 --warmup 1 > $O/bench_clips_traced.json 2> $O/err.log < /dev/null
f=$(find $O/clips -name "s_kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats_clips.csv; head -16 "$f" | cut -c1-150; fi
rm -rf $O/clips
cut -c1-400 $O/bench_clips_traced.json | tail -1
