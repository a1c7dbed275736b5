#!/bin/bash
# build a kernel variant of librrt.so: tools/build_variant.sh <name> <extra hipcc flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
mkdir -p build/var_$name ../../variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result"
rm -f build/var_$name/*.o ../../variants/librrt_$name.so
pids=""
for f in kernels.hip photon.hip; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o build/var_$name/${f%.hip}.o & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "build of variant $name FAILED"; exit 1; }; done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../../variants/librrt_$name.so build/runtime.o build/host_scene.o build/host_io.o build/scene_file.o build/bvh_build.o build/bvh_lbvh.o build/bvh_sah.o build/var_$name/kernels.o build/var_$name/photon.o build/noalloc.o
echo built variants/librrt_$name.so
