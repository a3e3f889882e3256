#!/bin/bash
# rocprofv3 kernel summary of tools/attn_bench.py (per-kernel average durations of the attention kernels, all arrangements)
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/attn_prof -o ap -- env PYTHONPATH=$R python $R/tools/attn_bench.py 1 > $R/gpurun_out/attn_prof.log 2>&1)
f=$(ls gpurun_out/attn_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls gpurun_out/attn_prof/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("a3::", "a4::", "attn", "relpos", "slab")):
        print("%9.1f us avg  %5d calls  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"][:110]))
PY
