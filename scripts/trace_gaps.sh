# kernel timeline of one bench run (rocprofv3 --kernel-trace), reduced on the box to the gap statistics scripts/gap_stats.py prints
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-breakdown > $O/bench.json 2> $O/bench.err
cd $R
f=$(ls $O/prof/*/bench_kernel_trace.csv $O/prof/bench_kernel_trace.csv 2>/dev/null | head -1)
python scripts/gap_stats.py "$f" > $O/gap_stats.txt 2>&1
rm -rf $O/prof
cat $O/gap_stats.txt
