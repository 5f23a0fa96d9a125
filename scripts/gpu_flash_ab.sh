# A/B of the flash kernel variants + kernel micro-benchmarks on one MI355X (run through gpurun from the repo root).
TAG=${1:-flash_ab}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O $R/build_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFLASH_AB_OLD -o $R/build_tmp/flash_ab $R/scripts/flash_ab.hip > $O/flash_ab_build.log 2>&1
(timeout 120 $R/build_tmp/flash_ab) > $O/flash_ab.txt 2>&1; cat $O/flash_ab.txt
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash") > $O/ktests_flash.log 2>&1; tail -3 $O/ktests_flash.log
(timeout 200 python scripts/kbench.py) > $O/kbench.json 2> $O/kbench.err; tail -c 1500 $O/kbench.json
(timeout 100 python scripts/kbench.py --tconv) > $O/kbench_tconv.json 2>> $O/kbench.err; cat $O/kbench_tconv.json
