#!/bin/bash
# One short GPU call for the SegGPT pre-/post-processing row: parity tests, per-frame measurement, kernel trace.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_seggpt_io_gpu.py -q -m gpu > gpurun_out/n3_tests.log 2>&1
timeout 60 python tools/seggpt_io_bench.py > gpurun_out/n3_bench.json 2> gpurun_out/n3_bench.err
rm -rf gpurun_out/n3_prof
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/n3_prof -o n3 -- python tools/seggpt_io_bench.py --iters 20 --cpu-iters 1 > gpurun_out/n3_prof.log 2>&1
rm -f gpurun_out/n3_prof/*kernel_trace.csv
tail -40 gpurun_out/n3_tests.log; cat gpurun_out/n3_bench.json; tail -3 gpurun_out/n3_bench.err
