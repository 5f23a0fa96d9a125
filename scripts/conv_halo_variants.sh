# Trial builds of csrc/conv_halo.hip for scripts/trials/ch_timing.py (never shipped): that one file compiled with trial flags, linked with the
# shipped objects of the other translation units -> build_tmp/libfz_ch_<name>.so.  usage: conv_halo_variants.sh name="-DFLAG -DFLAG" ...
set -e
mkdir -p build_tmp/chv
mk() { name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Iinclude "$@" -c fatezero_amd/csrc/conv_halo.hip -o build_tmp/chv/ch_$name.o 2>/dev/null
  objs=$(ls fatezero_amd/build/hip/*.o | grep -v conv_halo)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/libfz_ch_$name.so build_tmp/chv/ch_$name.o $objs
}
mk timing -DCH_TIMING &
for v in "$@"; do mk "${v%%=*}" ${v#*=} & done
wait
ls -la build_tmp/libfz_ch_*.so
