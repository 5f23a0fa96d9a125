# Round 4 closing evidence, second half (the first bench of scripts/r04h.sh died in the roofline bookkeeping: an all-ones blend mask reads no
# stored row -> division by zero; fixed): PMC passes over one job, then the bench line; plus the flash segment timeline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O $R/build_tmp
bash scripts/pmc_job.sh r04i_pmc_job 50 2>&1 | tail -3
cp $R/gpurun_out/r04i_pmc_job.json $R/profiles/r04_pmc_job.json 2>/dev/null   # the bench line's `traffic` reads it
(timeout 500 python bench.py --steps 5 --warmup 2 --cpu-k 2) > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo; tail -3 $O/bench.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFZ_FLASH_TIMING -o $R/build_tmp/flash_timing $R/scripts/flash_timing.hip > $O/ft_build.log 2>&1
(timeout 60 $R/build_tmp/flash_timing) > $O/flash_timing.txt 2>&1; cat $O/flash_timing.txt
