# Round 4 closing evidence on one MI355X: the complete -m gpu suite, the bench line (CPU sample at k = 2), the in-situ PMC traffic passes
# over one job, the kernel-stats profile.  Results land in gpurun_out/r04h/; what is judged is copied into profiles/ as r04_*_final.*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
(timeout 1100 python -m pytest tests -m gpu -q -s --durations=12) > $O/gpu_tests.log 2>&1
tail -18 $O/gpu_tests.log
bash scripts/pmc_job.sh r04h_pmc_job 50 2>&1 | tail -3
cp $R/gpurun_out/r04h_pmc_job.json $R/profiles/r04_pmc_job.json 2>/dev/null   # the bench line's `traffic` reads it
(timeout 500 python bench.py --steps 5 --warmup 2 --cpu-k 2) > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -6 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
