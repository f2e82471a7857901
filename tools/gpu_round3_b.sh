#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=16
python tools/gpu_debug_cli_parity.py > $O/cli_parity.txt 2>&1
cp audiowmark_amd/libawm_hip.so /tmp/new.so
cp tools/_before/libawm_hip.so audiowmark_amd/libawm_hip.so
echo "== K2 before" > $O/k2.txt
python tools/gpu_add_only.py 10 >> $O/k2.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/pmc_before -o s -- python $R/tools/gpu_add_only.py 2 > $O/pmc_before.log 2>&1 )
cp /tmp/new.so audiowmark_amd/libawm_hip.so
echo "== K2 after" >> $O/k2.txt
python tools/gpu_add_only.py 10 >> $O/k2.txt 2>&1
python tools/pmc_table.py $(find $O/pmc_before -name "*counter_collection.csv") > $O/pmc_before.txt 2>&1
rm -rf $O/pmc_before
cat $O/k2.txt | grep -v amdgpu.ids
grep add_mix $O/pmc_before.txt
cat $O/cli_parity.txt | grep -v amdgpu.ids | head -60
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc $?" ; tail -15 $O/gputests.log
