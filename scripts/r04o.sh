R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O $R/build_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFLASH_AB_KHM -o $R/build_tmp/flash_ab_khm $R/scripts/flash_ab.hip > $O/build.log 2>&1
for i in 1 2; do (timeout 120 $R/build_tmp/flash_ab_khm) > $O/flash_khm_$i.txt 2>&1; cat $O/flash_khm_$i.txt; done
