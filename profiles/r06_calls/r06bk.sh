R=$GRAFT_REPO_ROOT; A=$R/$O
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-split-mask --no-box > $A/bench_prof.json 2> $A/bench_prof.err; cd $R
f=$(ls $A/prof/*/bench_kernel_stats.csv $A/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $A/kernel_stats.csv; rm -rf $A/prof; python scripts/kstats.py $A/kernel_stats.csv 3 70 > $A/kstats.txt; head -16 $A/kstats.txt
