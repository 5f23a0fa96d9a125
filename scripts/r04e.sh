# Round 4, fifth GPU call: flash exp-split A/B (NPOLY of 32 exponentials per lane and tile on packed FMAs instead of v_exp_f32), then the
# kernel-stats profile of the current build.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O $R/build_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFLASH_AB_POLY -o $R/build_tmp/flash_ab_poly $R/scripts/flash_ab.hip > $O/build.log 2>&1
for i in 1 2; do (timeout 120 $R/build_tmp/flash_ab_poly) > $O/flash_poly_$i.txt 2>&1; cat $O/flash_poly_$i.txt; done
(timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "flash") > $O/k.log 2>&1; tail -2 $O/k.log
cd /tmp; export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
