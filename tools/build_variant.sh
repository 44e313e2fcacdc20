#!/bin/bash
# A/B or diagnostic build of the library with extra -D flags: tools/build_variant.sh <name> "<flags>"
# -> lasgun_amd/liblasgun_hip_<name>.so (select it with LASGUN_HIP_LIB=...; delete it afterwards: variants are not shipped)
set -eu
name=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/lgbuild_$name; mkdir -p "$b"
src=$root/lasgun_amd/csrc
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $flags -c "$src/host.cpp" -o "$b/host.o" &
for f in capi launch accel devmem; do
  g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $flags -c "$src/$f.cpp" -o "$b/$f.o" &
done
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $flags -c "$src/multi.cpp" -o "$b/multi.o" &
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $flags -c "$src/tune.cpp" -o "$b/tune.o" &
for k in k_mega k_wavefront k_queue k_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c "$src/$k.hip" -o "$b/$k.o" 2>&1 | grep -v hip-link || true &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$root/lasgun_amd/liblasgun_hip_$name.so" "$b/host.o" "$b/capi.o" "$b/launch.o" "$b/accel.o" "$b/devmem.o" "$b/tune.o" "$b/multi.o" "$b/k_mega.o" "$b/k_wavefront.o" "$b/k_queue.o" "$b/k_probe.o" -ldl 2>&1 | grep -v hip-link || true
ls -la "$root/lasgun_amd/liblasgun_hip_$name.so"
