R=$GRAFT_REPO_ROOT; A=$R/$O
bash scripts/pmc_job.sh r06cb_pmc 50 --blend-th 0.959197 > $A/pmc.log 2>&1; cp gpurun_out/r06cb_pmc.json $A/pmc_job.json 2>/dev/null; cp gpurun_out/r06cb_pmc.json profiles/r06_pmc_job.json 2>/dev/null; tail -3 $A/pmc.log
(timeout 600 python bench.py) > $A/bench_default.json 2> $A/bench_default.err; head -c 300 $A/bench_default.json; echo
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-split-mask --no-box > $A/bench_prof.json 2> $A/bench_prof.err; cd $R
f=$(ls $A/prof/*/bench_kernel_stats.csv $A/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $A/kernel_stats.csv; rm -rf $A/prof; python scripts/kstats.py $A/kernel_stats.csv 3 70 > $A/kstats.txt; head -8 $A/kstats.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $A/gpu_tests_full.log 2>&1; tail -6 $A/gpu_tests_full.log
