# build_variant.sh <out.so> <extra hipcc flags...>: the kernel library with extra -D flags, for same-box A/B runs (never shipped)
out=$1; shift
mkdir -p build_tmp/variant_obj
for f in fatezero_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Iinclude "$@" -c $f -o build_tmp/variant_obj/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out build_tmp/variant_obj/*.o
