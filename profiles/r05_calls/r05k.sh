R=$GRAFT_REPO_ROOT
bash scripts/pmc_job.sh r05_pmc_job 50 --blend-th 0.959197 2>&1 | tail -6
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-box --no-split-mask > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
f=$(ls /tmp/prof/*/bench_kernel_stats.csv /tmp/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/kernel_stats.csv
python scripts/kstats.py $O/kernel_stats.csv 3 30 > $O/kstats.txt; head -45 $O/kstats.txt
cp gpurun_out/r05_pmc_job.json $O/ 2>/dev/null; ls -la gpurun_out/r05_pmc_job.json
