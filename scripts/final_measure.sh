# Round-end evidence on one MI355X: GPU parity (QUICK=1: conv kernels + the end-to-end scenarios only), the bench line, the
# rocprofv3 kernel statistics of the same bench command and the conv sweep against MIOpen.  Run through gpurun from the repo
# root; results land in gpurun_out/ (copy what is judged into profiles/).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
if [ "$QUICK" = "1" ]; then
  timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -x -q -k "conv or pipe or scenario or full" > gpurun_out/gpu_tests.log 2>&1
else
  timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
fi
tail -2 gpurun_out/gpu_tests.log
timeout 240 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1200 gpurun_out/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
cd $R
f=$(ls gpurun_out/prof/*/bench_kernel_stats.csv gpurun_out/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" gpurun_out/kernel_stats.csv 2>/dev/null; head -6 gpurun_out/kernel_stats.csv | cut -c1-120
rm -rf gpurun_out/prof
timeout 60 python scripts/kbench.py --conv > gpurun_out/kbench_conv.json 2> /dev/null
