# build_tmp/libfz_trials.so: the kernel library with csrc/igemm.hip compiled -DFZ_IGEMM_TRIALS (the fz_igemm_trial_* switches of
# scripts/ab_lib_flag.py), the other translation units as shipped.  Never part of the product.
set -e
mkdir -p build_tmp/variant_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Iinclude -DFZ_IGEMM_TRIALS -c fatezero_amd/csrc/igemm.hip -o build_tmp/variant_obj/igemm_trials.o 2>/dev/null
objs=$(ls fatezero_amd/build/hip/*.o | grep -v "/igemm.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/libfz_trials.so build_tmp/variant_obj/igemm_trials.o $objs
ls -la build_tmp/libfz_trials.so
