#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for pf in 2 0; do echo "PF=$pf"; PA_ATTN_FWD_PF=$pf timeout 300 python tools/attn_bench.py 2>&1 | grep "B'"; done
echo "BWD_WAVES=3"; PA_ATTN_BWD_WAVES=3 timeout 300 python tools/attn_bench.py 2>&1 | grep "B'"
PA_ATTN_FWD_PF=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attn" 2>&1 | tail -3
