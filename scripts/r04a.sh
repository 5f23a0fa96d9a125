# Round 4, first GPU call: the -m gpu suite (incl. the new cfg2_8f full-width case and the two-stream test), a baseline bench with the
# CPU sample at k = 2, the in-situ PMC traffic pass over one job, and the kernel-stats profile of the round's starting point.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
(timeout 700 python -m pytest tests -m gpu -x -q -s --durations=15) > $O/gpu_tests.log 2>&1
tail -22 $O/gpu_tests.log
(timeout 400 python bench.py --steps 3 --warmup 1 --cpu-k 2) > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
bash scripts/pmc_job.sh r04a_pmc_job 50 2>&1 | tail -6
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -4 $O/kernel_stats.csv | cut -c1-120
rm -rf $O/prof
