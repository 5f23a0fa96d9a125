# build an emulator variant with extra -D flags into build_tmp/<name>.so and run the PP emu tests against it
name=$1; shift
mkdir -p build_tmp/emu_$name
for f in fatezero_amd/csrc/*.hip fatezero_amd/csrc/fz_emu.cpp; do
  /opt/rocm/lib/llvm/bin/clang++ -x c++ -DFZ_EMU -O2 -std=c++17 -fPIC -march=native -Wno-unknown-attributes -Wno-unused-value "$@" -c $f -o build_tmp/emu_$name/$(basename $f).o &
done
wait
/opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -o build_tmp/libemu_$name.so build_tmp/emu_$name/*.o -lpthread
