#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3 4; do PAINTER_AMD_DDP_SELFTEST=1 timeout 300 python tools/ddp_selftest.py 2>&1 | grep "^world" | cut -c1-250; done
ISO_MODE=inplace ISO_JITTER=pg ISO_N=600 timeout 300 python tools/race_iso.py 2>&1 | grep "^mode"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
