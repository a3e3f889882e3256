#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/race_iso_attn.py 2>&1 | grep "^differing"
