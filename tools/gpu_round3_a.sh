#!/bin/bash
# first GPU pass of round 3: K2 before / after (time + SQ_INSTS_VALU), GPU tests, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
echo "== K2 after" > $O/k2.txt
python tools/gpu_add_only.py 10 >> $O/k2.txt 2>&1
cp audiowmark_amd/libawm_hip.so /tmp/new.so
cp tools/_before/libawm_hip.so audiowmark_amd/libawm_hip.so
echo "== K2 before" >> $O/k2.txt
python tools/gpu_add_only.py 10 >> $O/k2.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/pmc_before -o s -- python $R/tools/gpu_add_only.py 2 > $O/pmc_before.log 2>&1 )
cp /tmp/new.so audiowmark_amd/libawm_hip.so
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/pmc_after -o s -- python $R/tools/gpu_add_only.py 2 > $O/pmc_after.log 2>&1 )
python tools/pmc_table.py $(find $O/pmc_before -name "*counter_collection.csv") > $O/pmc_before.txt 2>&1
python tools/pmc_table.py $(find $O/pmc_after -name "*counter_collection.csv") > $O/pmc_after.txt 2>&1
rm -rf $O/pmc_before $O/pmc_after
cat $O/k2.txt
grep add_mix $O/pmc_before.txt $O/pmc_after.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc $?" ; tail -5 $O/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
