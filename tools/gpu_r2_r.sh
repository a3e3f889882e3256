#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r_prof -o r -- python tools/attn_bench.py 2 > gpurun_out/r_attn.log 2>&1
rm -f gpurun_out/r_prof/*kernel_trace.csv
grep "B'" gpurun_out/r_attn.log
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r_prof/r_kernel_stats.csv')))
for r in rows[:16]:
    print("%-90s n=%5s avg %8.1f us"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
P
