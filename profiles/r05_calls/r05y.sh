R=$GRAFT_REPO_ROOT
cd /tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-box --no-split-mask > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
f=$(ls /tmp/prof/*/bench_kernel_stats.csv /tmp/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/kernel_stats.csv
python scripts/kstats.py $O/kernel_stats.csv 3 30 > $O/kstats.txt; head -30 $O/kstats.txt
