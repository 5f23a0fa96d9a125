# Round-end evidence on one MI355X (run through gpurun from the repo root; results land in gpurun_out/final/, copy what is
# judged into profiles/ as r03_*_final.*):
#   QUICK=1 : kernel tests only instead of the full -m gpu suite (the two full-width oracle comparisons take ~10 min)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O $R/build_tmp
if [ "$QUICK" = "1" ]; then
  (timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py tests/test_clip_text_gpu.py tests/test_cli_gpu.py -x -q -s) > $O/gpu_tests.log 2>&1
else
  (timeout 1700 python -m pytest tests -x -q -s -m gpu) > $O/gpu_tests.log 2>&1
fi
tail -2 $O/gpu_tests.log
(timeout 500 python bench.py --steps 5 --warmup 2 $BENCH_EXTRA) > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -5 $O/kernel_stats.csv | cut -c1-120
rm -rf $O/prof
# PMC of the shipped flash variant (index 1 of scripts/flash_ab.hip built with -DFLASH_AB_OLD) + the A/B table itself
rm -f $R/build_tmp/flash_ab
bash scripts/pmc_flash.sh 1 flash_d40_final > /dev/null 2>&1; cp $R/gpurun_out/pmc/flash_d40_final.json $O/pmc_flash_d40_final.json 2>/dev/null
(timeout 120 $R/build_tmp/flash_ab) > $O/flash_ab.txt 2>&1; tail -12 $O/flash_ab.txt
