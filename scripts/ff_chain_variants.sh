# Trial builds of csrc/ff_chain.hip for scripts/ff_chain_ab.py (never shipped): that one file compiled with a trial flag, linked with the
# shipped objects of the other translation units -> build_tmp/libfz_ff_<name>.so
set -e
mkdir -p build_tmp/ffv
mk() { name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Iinclude "$@" -c fatezero_amd/csrc/ff_chain.hip -o build_tmp/ffv/ff_$name.o 2>/dev/null
  objs=$(ls fatezero_amd/build/hip/*.o | grep -v ff_chain)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/libfz_ff_$name.so build_tmp/ffv/ff_$name.o $objs
}
mk nodma -DFC_TRIAL_NODMA &
mk nogelu -DFZ_GELU_TRIAL_IDENTITY &
mk timing -DFC_TIMING &
mk noprio -DFC_TRIAL_NOPRIO &
mk gate0 -DFC_GATE_UP=0 &
wait
ls -la build_tmp/libfz_ff_*.so
