# Round 4 closing numbers on the final build: in-situ PMC passes, the bench line (CPU sample k = 2), kernel stats, the larger BASELINE shapes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
bash scripts/pmc_job.sh r04n_pmc_job 50 2>&1 | tail -2
cp $R/gpurun_out/r04n_pmc_job.json $R/profiles/r04_pmc_job.json 2>/dev/null   # the bench line's `traffic` reads it
(timeout 500 python bench.py --steps 5 --warmup 2 --cpu-k 2) > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo; tail -2 $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -4 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
for cfg in "16 64" "24 64" "32 72"; do set -- $cfg
  (timeout 400 python bench.py --frames $1 --latent-size $2 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe) > $O/bench_${1}f_$2.json 2> $O/bench_${1}f_$2.err
  python -c "import json; d=json.load(open('$O/bench_${1}f_$2.json')); print('$1 f x latent $2:', round(d['ms_per_step']), 'ms', round(d['value'],3), 'frames/s, arena', round(d['config']['arena_GB'],1), 'GB')"
done
