#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 256 128 256 128 64; do PA_WGRAD_WGS=$w timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad WGs $w', d['value'], d['ms_per_step'])"; done
