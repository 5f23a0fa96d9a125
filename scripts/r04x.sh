# Round 4, CLOSING build (adds fz_lora_pair at the 64^2 level): the complete -m gpu suite, the in-situ PMC passes, the bench line
# (CPU sample k = 2), the kernel-stats profile.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x; mkdir -p $O
(timeout 560 python -m pytest tests -m gpu -q -s --durations=10) > $O/gpu_tests.log 2>&1
tail -16 $O/gpu_tests.log; grep "peer transport, us per" $O/gpu_tests.log
bash scripts/pmc_job.sh r04x_pmc_job 50 2>&1 | tail -2
cp $R/gpurun_out/r04x_pmc_job.json $R/profiles/r04_pmc_job.json 2>/dev/null
(timeout 400 python bench.py --steps 5 --warmup 2 --cpu-k 2) > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo; tail -2 $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -4 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
(timeout 120 python bench.py --frames 16 --latent-size 64 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe) > $O/bench_16f_64.json 2> $O/bench_16f_64.err; head -c 200 $O/bench_16f_64.json; echo
