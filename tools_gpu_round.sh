#!/bin/bash
# one GPU visit: tests, smoke, bench, rocprof kernel trace, PMC traffic passes.  Everything lands in gpurun_out/;
# copy what should be judged into profiles/.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/tests.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 ${BENCH_FLAGS:-} > gpurun_out/bench.log 2>&1
rm -rf gpurun_out/prof gpurun_out/pmc_f gpurun_out/pmc_w
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/roofline_traffic.json > gpurun_out/traffic.log 2>&1
rm -f gpurun_out/pmc_f/*kernel_trace.csv gpurun_out/pmc_w/*kernel_trace.csv
timeout 600 python tools/seggpt_bench.py > gpurun_out/seggpt.log 2>&1
# per-frame pre+post processing, end-to-end video loop, one step's input batch; kernel trace of both
timeout 120 python tools/seggpt_io_bench.py > gpurun_out/n3_bench.json 2>/dev/null
timeout 180 python tools/seggpt_video_bench.py > gpurun_out/n3_video.json 2>/dev/null
timeout 120 python tools/pair_pipeline_bench.py > gpurun_out/n2_bench.json 2>/dev/null
rm -rf gpurun_out/n3_prof gpurun_out/n2_prof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/n3_prof -o n3 -- python tools/seggpt_io_bench.py --iters 20 --cpu-iters 1 > gpurun_out/n3_prof.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/n2_prof -o n2 -- python tools/pair_pipeline_bench.py --iters 10 --cpu-iters 1 > gpurun_out/n2_prof.log 2>&1
rm -f gpurun_out/n3_prof/*kernel_trace.csv gpurun_out/n2_prof/*kernel_trace.csv
tail -5 gpurun_out/tests.log; tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log; tail -2 gpurun_out/seggpt.log; cat gpurun_out/n3_bench.json gpurun_out/n3_video.json gpurun_out/n2_bench.json
