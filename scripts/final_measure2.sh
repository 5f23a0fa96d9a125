# Round-3 closing evidence on one MI355X, after the flash single-source frames / one-launch GroupNorm / oracle thread-pool changes
# (results land in gpurun_out/final2/; what is judged is copied into profiles/ as r03_*_final2.*).  The full -m gpu suite last ran
# at commit 6d36932 (profiles/r03_gpu_tests_final.log, 204 green); here: every kernel test + the pipeline tests incl. both full-width
# oracle comparisons (the modules the later changes touch), with durations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final2; mkdir -p $O $R/build_tmp
(timeout 420 python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -x -q -s --durations=12) > $O/gpu_tests.log 2>&1
tail -16 $O/gpu_tests.log
rm -f $R/build_tmp/flash_ab
bash scripts/pmc_flash.sh 1 flash_d40_final2 > /dev/null 2>&1; cp $R/gpurun_out/pmc/flash_d40_final2.json $O/pmc_flash_d40_final2.json 2>/dev/null
cp $O/pmc_flash_d40_final2.json $R/profiles/r03_pmc_flash_d40_final2.json 2>/dev/null   # the bench line's `traffic` reads it
(timeout 300 python bench.py --steps 5 --warmup 2) > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -4 $O/kernel_stats.csv | cut -c1-120
rm -rf $O/prof
cd /tmp
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $O/gnprof -o gn -- python $R/scripts/gn_ab.py run > $O/gn_run.log 2>&1
cd $R
f=$(ls $O/gnprof/*/gn_kernel_trace.csv $O/gnprof/gn_kernel_trace.csv 2>/dev/null | head -1)
python scripts/gn_ab.py report "$f" > $O/gn_ab.txt 2>&1; rm -rf $O/gnprof
(timeout 100 $R/build_tmp/flash_ab) > $O/flash_ab.txt 2>&1; tail -6 $O/flash_ab.txt
