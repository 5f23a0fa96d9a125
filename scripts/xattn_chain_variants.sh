# Trial builds of csrc/xattn_chain.hip for scripts/xattn_chain_ab.py (never shipped): that one file compiled with a trial flag, linked with the
# shipped objects of the other translation units -> build_tmp/libfz_xc_<name>.so
set -e
mkdir -p build_tmp/xcv
mk() { name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Iinclude "$@" -c fatezero_amd/csrc/xattn_chain.hip -o build_tmp/xcv/xc_$name.o 2>/dev/null
  objs=$(ls fatezero_amd/build/hip/*.o | grep -v xattn_chain)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/libfz_xc_$name.so build_tmp/xcv/xc_$name.o $objs
}
mk timing -DXC_TIMING &
mk dbgxn -DXC_DEBUG_XN &
for v in "$@"; do mk "${v%%=*}" ${v#*=} & done
wait
ls -la build_tmp/libfz_xc_*.so
