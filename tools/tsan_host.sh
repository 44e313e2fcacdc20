#!/bin/bash
# The host builder's threaded part (host.cpp, build_chunks: the fat leaves' runs, records and strips, one leaf per task) under
# ThreadSanitizer on the CPU: builds lasgun_amd/csrc/{host,capi,launch,accel,devmem,tune,multi}.cpp with -fsanitize=thread, links them with the device
# objects and flattens the mesh scenes with eight threads.  (Memory errors: tools/asan_host.sh.)
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/lg_tsan; mkdir -p "$b"
src=$root/lasgun_amd/csrc
[ -f "$src/k_mega.o" ] || make -C "$src" >/dev/null
for f in host capi launch accel devmem tune multi; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$src/$f.cpp" -o "$b/$f.o" &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$b/liblasgun_hip_tsan.so" "$b/host.o" "$b/capi.o" "$b/launch.o" "$b/accel.o" "$b/devmem.o" "$b/tune.o" "$b/multi.o" "$src/k_mega.o" "$src/k_wavefront.o" "$src/k_queue.o" "$src/k_probe.o" -ldl -fsanitize=thread 2>&1 | grep -v "hip-link\|not currently supported" || true
export LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS=halt_on_error=1 LASGUN_HIP_LIB="$b/liblasgun_hip_tsan.so" LASGUN_HOST_THREADS=8
cd "$root"
python3 - <<'PY'
import lasgun_amd as la
G, S = la.api, la.scenes
for b in (lambda: S.mesh_scene(G, 224, 224, "glass"), lambda: S.mixed_scene(G), lambda: S.tie_mesh_scene(G), lambda: S.exotic_obj_scene(G)):
    sc = b(); G.host_build_dump(sc); assert G.host_check_strips(sc)["violations"] == 0
for seed in range(40):
    for gen in (S.progression_soup_scene, S.adversarial_mesh_scene, S.adversarial_prune_scene):
        try:
            sc = gen(G, seed); G.host_build_dump(sc); assert G.host_check_strips(sc)["violations"] == 0
        except la.LasgunError:
            pass
print("threaded host build under ThreadSanitizer: no report")
PY
