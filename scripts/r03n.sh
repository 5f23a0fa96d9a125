# round-3 GPU call n: per-launch durations of the flash kernel inside a bench run (why is it 7 % slower in situ than in the A/B harness?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench.json 2> $O/bench.err
cd $R
f=$(ls $O/prof/*/bench_kernel_trace.csv $O/prof/bench_kernel_trace.csv 2>/dev/null | head -1)
head -2 "$f" | cut -c1-400
python scripts/flash_trace_stats.py "$f" | tee $O/flash_trace_stats.txt
rm -rf $O/prof
