# round-3 GPU call i: mid-round checkpoint -- GPU suite without the two 5-minute full-width oracle cases, bench (default run incl. the CPU
# sample), rocprofv3 kernel stats of a bench run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
(timeout 900 python -m pytest tests -q -x -m gpu -k "not fullwidth" -s) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
(timeout 500 python bench.py --steps 3 --warmup 1) > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
head -c 300 $O/bench_prof.json; echo; head -8 $O/kernel_stats.csv | cut -c1-150
