#!/bin/bash
# usage: tools/res_usage.sh audiowmark_amd/csrc/hip/viterbi.hip [filter]
# registers / scratch / LDS / occupancy of every kernel of one gfx950 translation unit (no GPU needed): the compiler's
# -Rpass-analysis=kernel-resource-usage remarks, one line per kernel
src="$1"; filt="${2:-.}"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 | python3 -c "
import sys, re
cur = {}
keys = {'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'TotalSGPRs': 'sgpr', 'ScratchSize [bytes/lane]': 'scratch', 'Occupancy [waves/SIMD]': 'occ', 'LDS Size [bytes/block]': 'lds'}
def flush():
    if cur:
        print(' '.join('%s=%s' % kv for kv in cur.items()))
for l in sys.stdin:
    m = re.search(r'(?:Function )?Name: (\S+)', l)
    if m:
        flush(); cur = {'name': m.group(1)}; continue
    for k, short in keys.items():
        m = re.search(re.escape(k) + r': (\S+)', l)
        if m:
            cur[short] = m.group(1)
flush()
" | c++filt | grep -E "$filt"
