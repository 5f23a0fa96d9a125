# Round-end evidence on one MI355X: GPU parity suite, the bench line, and the rocprofv3 kernel statistics of the same
# bench command.  Run through gpurun from the repo root; results land in gpurun_out/ (copy what is judged into profiles/).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout 240 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
cd $R
f=$(ls gpurun_out/prof/*/bench_kernel_stats.csv gpurun_out/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" gpurun_out/kernel_stats.csv 2>/dev/null; head -12 gpurun_out/kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof
