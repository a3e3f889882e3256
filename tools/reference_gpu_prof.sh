export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ref_prof -o ref -- env PYTHONPATH=$R python $R/tools/reference_gpu_prof.py > $R/gpurun_out/ref_prof.log 2>&1)
tail -2 gpurun_out/ref_prof.log | cut -c1-400
ls gpurun_out/ref_prof | head
