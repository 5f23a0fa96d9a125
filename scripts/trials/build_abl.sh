# Timing-only ablations of the igemm K loop (never shipped).  Usage, from the repo root:
#   mkdir -p build_tmp/abl_src && cp fatezero_amd/csrc/*.hip fatezero_amd/csrc/*.h build_tmp/abl_src/ && \
#   sed -i 's#"../../include/fatezero_hip.h"#"fatezero_hip.h"#' build_tmp/abl_src/*.hip && patch build_tmp/abl_src/igemm.hip scripts/trials/igemm_ablations.patch
#   bash scripts/trials/build_abl.sh build_tmp/libfz_abl1.so -DFZ_IGEMM_ABL=1     # bits: 1 no DMA in the loop, 2 no MFMA, 4 no fragment reads,
#   FZ_TRIAL_LIB=$PWD/build_tmp/libfz_abl1.so python scripts/tile_trial.py 0        #   8 no barrier, 16 A operand from a hot 1 KB page, 32 B too, 64 rotated K start
# build_abl.sh <out.so> <flags>: variant of the kernel library from build_tmp/abl_src (igemm ablations, timing only)
out=$1; shift
od=build_tmp/abl_obj_$(basename $out .so); mkdir -p $od
for f in build_tmp/abl_src/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -w -Iinclude -Ibuild_tmp/abl_src "$@" -c $f -o $od/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $od/*.o
