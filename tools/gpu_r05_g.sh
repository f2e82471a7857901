#!/bin/bash
# round 5, GPU call G: K16g (the per-clip-key tables of `get` built on the device) -- table identity, the clip tests, the per-key A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "key_tables or clip_batch or multi_context" 2>&1 | tail -5
timeout 300 python tools/gpu_clip_keys_ab.py 1024 2 3 4 2>&1 | tail -12
