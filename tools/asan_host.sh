#!/bin/bash
# Host side (scene description, OBJ reader, HLBVH builder, flattening, C ABI) under AddressSanitizer + UBSan on the CPU:
# builds lasgun_amd/csrc/{host,capi,launch,accel,devmem,tune,multi}.cpp with -fsanitize=address,undefined, links them with the device object, and
# runs tests/test_host.py plus the flattening of 600 random / adversarial scenes (reference trees, and fast trees with their wide records) and the
# full-size configs through it.
# (GPU sanitizers are not available on the pool; the kernels are covered by the parity suite and the fuzz campaign.)
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/lg_asan; mkdir -p "$b"
src=$root/lasgun_amd/csrc
[ -f "$src/k_mega.o" ] || make -C "$src" >/dev/null
for f in host capi launch accel devmem tune multi; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$src/$f.cpp" -o "$b/$f.o" &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$b/liblasgun_hip_asan.so" "$b/host.o" "$b/capi.o" "$b/launch.o" "$b/accel.o" "$b/devmem.o" "$b/tune.o" "$b/multi.o" "$src/k_mega.o" "$src/k_wavefront.o" "$src/k_queue.o" "$src/k_probe.o" -ldl -fsanitize=address,undefined 2>&1 | grep -v hip-link || true
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LASGUN_HIP_LIB="$b/liblasgun_hip_asan.so"
cd "$root"
python3 -m pytest tests/test_host.py -x -q
python3 - <<'PY'
import lasgun_amd as la
G, S, n = la.api, la.scenes, 0
for seed in range(150):
    for gen in (S.random_scene, S.adversarial_scene, S.adversarial_mesh_scene, S.adversarial_prune_scene, S.progression_soup_scene):
        try:
            G.host_build_dump(gen(G, seed)); n += 1
            r = G.host_check_wide_records(gen(G, seed))  # the fast trees and their wide records too
            assert r["violations"] == 0 and r["deepest_stack"] <= r["reserved_stack"], (gen.__name__, seed, r)
            r = G.host_check_strips(gen(G, seed))  # the pruned walk's runs and triangle strips
            assert r["violations"] == 0, (gen.__name__, seed, r)
        except la.LasgunError:
            pass
for b in (lambda: S.mesh_scene(G), lambda: S.mixed_scene(G), lambda: S.spheres_scene(G), lambda: S.kitchen_sink_scene(G), lambda: S.instanced_scene(G), lambda: S.tie_mesh_scene(G), lambda: S.exotic_obj_scene(G)):
    G.host_build_dump(b()); n += 1
print("flattened", n, "scenes under ASan + UBSan: no report")
PY
