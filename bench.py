#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

metric  : Mrays/s (primary + shadow) at 4096x4096, bit-exact RGBA8 vs the CPU oracle
workload: configs[2] -- "4096x4096, 1024 random spheres (deep BVH), 1 spp" (the headline config;
          lasgun_amd.scenes.spheres_scene, SplitMix64 seed 0x1A560001)
step    : one full pass of the hot path = one 4096x4096 frame.  The flattened scene / BVH is resident in HBM
          before the timed region; the film stays in HBM (no PCIe in `value`).
N ranks : one process per GPU (torch.distributed.run); the SAME film is cut into 64-row blocks dealt round-robin
          (rank r renders blocks r, r+N, ... into a compact tile) and ONE RCCL gather per frame brings the tiles
          to rank 0 -> total work is fixed -> "strong" scaling.

ONE policy at every N (N = 1 included): consecutive frames rotate over FRAMES_IN_FLIGHT HIP streams, each with
its own launch context in the library, so the kernels of the following frames fill the tails of a frame's
closest, shadow and shade passes and the gaps between its launches (and, for N > 1, the gather of frame k
overlaps the renders of the frames behind it).  `value` is that
pipelined throughput over exactly K steps; `latency_ms` is one frame (render + gather) issued alone with a
full synchronisation around it, at the same N -- so the numbers at N = 1, 2, 4, 8 are like for like.

Prints ONE JSON line on rank 0.
  roofline      the DOMINANT kernel (by HIP events on its launch stream, three further frames issued one after the
                other).  `bound` names the resource that binds it: f64 VALU issue ("valu_f64": algorithmic unfused f64
                operations of the reference's algorithm per launch, from the kernels' own deterministic counters,
                over the kernel's average duration, against 256 CUs x 4 SIMDs x 16 f64 lanes x 2.4 GHz).  The
                SURVEY 8(d) byte figure is kept beside it (`hbm_algorithmic`: those bytes are served by the LDS-resident
                scene and never reach HBM) together with the box's measured HBM copy rate and the LDS fraction (`lds`).
  traffic       HBM bytes per launch of that kernel from rocprofv3 PMC passes of THIS source (profiles/rNN_pmc.json
                records the hash of the device sources it was collected on); null when they have changed since.
  cpu_baseline  the CPU oracle (a port: the Rust reference cannot be built here) timed on a bounded strided sample
                of the same frame on this box's host cores (rank 0, N = 1 only).
  bit_exact     the TIMED frame (the last of the K steps, as gathered on rank 0) compared byte for byte with the film the
                cpu_baseline leg renders on the oracle, on every pixel of that sample (the whole film when the host is fast
                enough: n = 1); a mismatch makes the run exit non-zero after the line is printed.
  roofline_mesh the same bookkeeping for configs[3] (100k-triangle mesh, glass + mirror, 4096^2) from two untimed frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of streams
# that share a queue run one after the other.  The frames in flight (FRAMES_IN_FLIGHT below) need a queue each beside the
# default stream, the accel's own stream and RCCL's: with 4 queues the four frame streams pair up and 4 frames in flight are
# worth no more than 2 (7.57 ms per frame); with 8 they overlap (7.23 ms), and with RCCL's streams beside them 12 or more
# are needed (1/8-size frames with a gather each: 1.30 ms per frame with 8 queues, 1.18 with 12, 1.20 with 16).  Read by
# the runtime at initialisation, so it is set before torch is imported; an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LDS_PEAK_GBS = 150000.0    # aggregate ds_read_b128 rate, every CU streaming (MI355X_MICROARCH.md, section LDS)
# algorithmic bytes per unit of traversal work: SURVEY.md section 8(d)'s per-unit figures (restated in DESIGN.md
# "Roofline bookkeeping"): 56 B per node whose bounds are tested, 32 B per sphere / 48 B per cuboid / 48 B per triangle
# tested, 256 B (m, minv) per nested-BVH entry, 4 B material id per accepted closest hit, 4 B RGBA per primary ray.
BYTES_NODE, BYTES_SPHERE, BYTES_CUBOID, BYTES_TRI, BYTES_ACCEL_ENTRY, BYTES_HIT, BYTES_PIXEL = 56, 32, 48, 48, 256, 4, 4
# Unfused f64 operations the reference's algorithm needs per test, lower bounds (miss paths): slab test 26
# (cuboid.rs:104-121), sphere 26 up to the discriminant test (sphere.rs:30-69), cuboid 26, triangle 36 up to the
# edge-function signs (triangle.rs:186-230), accel entry 2 x (36 + 3 divisions).  Peak: 256 CUs x 4 SIMDs x 16 f64
# lanes x 2.4 GHz = 39.3 T unfused ops/s (-ffp-contract=off is part of the parity contract: no FMA).
FLOPS_NODE, FLOPS_SPHERE, FLOPS_CUBOID, FLOPS_TRI, FLOPS_ACCEL_ENTRY = 26, 26, 26, 36, 78
VALU_F64_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12
BLOCK_ROWS = 64
# frames in flight (the library keeps up to eight launch contexts per accel); measured on one GPU with a hardware queue per
# stream, ms per frame at N = 1 / one rank's share at N = 8: 1 stream 7.80 / 1.17, 2 streams 7.56 / 1.07, 4 streams 7.22 / 0.96
FRAMES_IN_FLIGHT = int(os.environ.get("LASGUN_BENCH_FRAMES", "4"))  # (the variable: A/B only)

# --workload: which BASELINE.json config a step renders.  The default is the one the metric is quoted on (configs[2], 4096^2); configs[4] is
# BASELINE's sharded case -- "8192x8192, mixed mesh+sphere scene, row-tile sharded across 8 GPUs with RCCL gather" -- through the SAME
# harness: same partition, same single gather per frame, same in-run verification (gathered film == rank 0's own film == oracle sample).
WORKLOADS = {
    "configs2": {"size": 4096, "build": lambda scenes, api: scenes.spheres_scene(api),
                 "desc": "configs[2]: %dx%d, Cornell shell + 1024 random plastic spheres (SplitMix64 0x1A560001), 1 spp, 1 point light"},
    "configs4": {"size": 8192, "build": lambda scenes, api: scenes.mixed_scene(api),
                 "desc": "configs[4]: %dx%d, mixed: config 3's 1024 spheres + the 100,352-triangle torus (plastic) in the Cornell shell, 1 spp, 1 point light"},
}


def algorithmic_bytes(st):
    return (BYTES_NODE * st["nodes_tested"] + BYTES_SPHERE * st["spheres_tested"] + BYTES_CUBOID * st["cuboids_tested"]
            + BYTES_TRI * st["triangles_tested"] + BYTES_ACCEL_ENTRY * st["accel_entries"] + BYTES_HIT * st["hits"]
            + BYTES_PIXEL * st["primary_rays"])


def algorithmic_flops(st):
    return (FLOPS_NODE * st["nodes_tested"] + FLOPS_SPHERE * st["spheres_tested"] + FLOPS_CUBOID * st["cuboids_tested"]
            + FLOPS_TRI * st["triangles_tested"] + FLOPS_ACCEL_ENTRY * st["accel_entries"])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Distinct (package, core) pairs among the CPUs this process may run on (`cores` / `threads` count hardware THREADS: SMT siblings)."""
    try:
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(os.cpu_count() or 1)
        seen = set()
        for c in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(width, height, gpu_film=None, target_seconds=15.0, build=None):
    """Time the CPU oracle on a bounded strided sample {k + i*n} of the same frame (all host cores); with `gpu_film`
    (the timed frame, (h, w, 4) uint8 on the host) also compare every pixel of that sample byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    from lasgun_amd import scenes
    o = oracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    acc = o.Accel((build or WORKLOADS["configs2"]["build"])(scenes, o))
    film = o.Film(width, height)
    area = width * height

    def run(n):
        o.stats_reset()
        t0 = time.perf_counter()
        o.capture_subset_mt(0, n, acc, film, cores)
        dt = time.perf_counter() - t0
        st = o.stats_read()
        return dt, st["primary_rays"] + st["shadow_rays"], st["primary_rays"]

    dt, rays, _ = run(max(1, area // 32768))  # calibration: ~32k pixels
    rate = rays / dt
    want_pixels = min(area, max(65536, int(rate * target_seconds / 2.0)))
    n = max(1, area // want_pixels)
    dt, rays, pixels = run(n)
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "threads": cores, "physical_cores": physical_cores(), "cpu": cpu_model(), "kind": "port",
           "sample": "capture_subset(0, n=%d) of the same %dx%d frame: %d pixels, %d rays in %.2f s on %d threads"
                     % (n, width, height, pixels, rays, dt, cores)}
    check = None
    if gpu_film is not None:
        want = film.pixels().reshape(-1, 4)[::n]
        got = gpu_film.reshape(-1, 4)[::n]
        bad = int((want != got).sum())
        check = {"bit_exact": bad == 0, "mismatched_bytes": bad, "checked_pixels": int(want.shape[0]),
                 "checked": "the timed frame against the oracle film of the cpu_baseline leg, pixels {0, n, 2n, ...}, n = %d" % n}
    return out, check


def oracle_sample_check(width, height, gpu_film, n, build=None):
    """Pixels {0, n, 2n, ...} of `gpu_film` ((h, w, 4) uint8 on the host) against the CPU oracle, byte for byte (untimed)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    from lasgun_amd import scenes
    o = oracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    film = o.Film(width, height)
    o.capture_subset_mt(0, n, o.Accel((build or WORKLOADS["configs2"]["build"])(scenes, o)), film, cores)
    want = film.pixels().reshape(-1, 4)[::n]
    got = gpu_film.reshape(-1, 4)[::n]
    bad = int((want != got).sum())
    return {"bit_exact": bad == 0, "mismatched_bytes": bad, "checked_pixels": int(want.shape[0]),
            "checked": "the timed (gathered) frame against the oracle, pixels {0, n, 2n, ...}, n = %d" % n}


def profiled_traffic(kernel_name, world, size):
    """HBM bytes per launch of `kernel_name` from the committed PMC passes, only if they were collected on these sources."""
    try:
        import glob
        import lasgun_amd
        sha = lasgun_amd.device_source_sha16()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_pmc.json")), reverse=True):  # the newest round's passes first
            pmc = json.load(open(path))
            t = pmc["kernels"].get(kernel_name)
            if t and pmc.get("device_source_sha16") == sha and world == 1 and size == 4096:
                # FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM): doubled; both in KB
                return (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0, "profiles/%s @ device sources %s" % (os.path.basename(path), sha)
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def profiled_clock(kernel_name):
    """The clock (GHz) the chip held under `kernel_name`, from the committed GRBM_GUI_ACTIVE passes (profiles/rNN_pmc.json: the counter summed
    over the 8 XCDs / 8 / the dispatch's duration, median over dispatches), only if they were collected on these sources; else None."""
    try:
        import glob
        import lasgun_amd
        sha = lasgun_amd.device_source_sha16()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_pmc.json")), reverse=True):
            pmc = json.load(open(path))
            c = (pmc.get("clock_ghz") or {}).get(kernel_name)
            if c and pmc.get("device_source_sha16") == sha:
                return float(c["median"]), "profiles/%s @ device sources %s (GRBM_GUI_ACTIVE / 8 / duration, %d dispatches, %.3f - %.3f GHz)" % (
                    os.path.basename(path), sha, c["dispatches"], c["min"], c["max"])
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None, None


def at_clock(block, kernel_name):
    """`clock_ghz` and `frac_at_clock` beside a roofline block's nominal-clock `frac`: the same achieved rate against the peak at the clock
    the chip actually held under this kernel (the nominal peak assumes 2.4 GHz, which no loaded MI355X holds)."""
    ghz, src = profiled_clock(kernel_name)
    block["clock_ghz"] = ghz
    block["clock_source"] = src
    block["frac_at_clock"] = (block["achieved"] / (VALU_F64_PEAK_TOPS * ghz / 2.4)) if ghz else None
    return block


def mesh_roofline(G, la, stream):
    """configs[3] (generated 100k-triangle torus of glass + mirror sphere, recursion 3, 4096^2) on this GPU, two untimed
    frames after a warm-up: the triangle-test side of the path (the reference's 254-triangle leaves), in the organisation and
    traversal mode the accel picks by default for such a scene (the pruned reference walk; the organisation and the direction its tiles are
    claimed in as MEASURED: `organisation` in the result says which).  Counters from the counting instantiation of the same walk."""
    size = 4096
    acc = G.Accel(la.scenes.mesh_scene(G, 224, 224, "glass"))
    film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def frame_ms(reps=2):
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ms = frame_ms()
    default_film = film.clone()
    ran_as = G.last_organisation(acc) or "?"
    kernel_of = {"megakernel": "lg::trace_kernel<false, false, false, true, 1024>", "queue": "lg::queue_kernel<false, true>", "wavefront": "lg::wf_trace_kernel<...> (level by level)"}
    G.profile_enable(acc, True)
    G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    kinds = {k: v[0] / v[1] for k, v in G.profile_read_kinds(acc).items() if v[1]}
    G.profile_read(acc)
    G.profile_enable(acc, False)
    kernel_ms = kinds.get("trace_kernel", ms) if ran_as.split(",")[0] != "wavefront" else ms  # HIP events around the persistent kernel on its launch stream
    st = G.capture_stats(acc, size, size, 0, size)
    rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
    flops = algorithmic_flops(st)
    tops = flops / (kernel_ms * 1e-3) / 1e12
    # the other organisations on the same frame (same bytes: checked), and the same frame as the reference walks it (no node or
    # run skipped): its work, and how long this build's plain walk takes over it
    G.set_streaming(acc, 0)
    ms_mega = frame_ms()
    same = bool(torch.equal(film, default_film))
    G.set_streaming(acc, 3)
    ms_queue = frame_ms()
    same = same and bool(torch.equal(film, default_film))
    G.set_streaming(acc, 1)
    G.set_prune(acc, False)
    st_ref = G.capture_stats(acc, size, size, 0, size)
    ms_plain = frame_ms(1)
    G.set_prune(acc, None)
    flops_ref = algorithmic_flops(st_ref)
    del film, default_film
    torch.cuda.empty_cache()
    return {"workload": "configs[3]: 4096x4096, 100,352-triangle torus (glass) in a transformed group + mirror sphere in the Cornell shell, recursion 3",
            "ms_per_frame": ms, "value": rays / ms / 1e3, "unit": "Mrays/s", "rays_per_frame": rays,
            "bound": "valu_f64", "achieved": tops, "peak": VALU_F64_PEAK_TOPS, "frac": tops / VALU_F64_PEAK_TOPS,
            "organisation": ran_as, "kernel": kernel_of.get(ran_as.split(",")[0], ran_as), "kernel_ms_avg": kernel_ms, "algorithmic_flops_per_frame": flops,
            "work_per_frame": {k: st[k] for k in ("nodes_tested", "spheres_tested", "cuboids_tested", "triangles_tested", "accel_entries", "hits")},
            "traversal": "reference tree, pruned walk (lg_accel_set_prune default for a scene with a big mesh); organisation and tile direction as measured (`organisation`)",
            "megakernel": {"ms_per_frame": ms_mega, "film_identical": same}, "queue": {"ms_per_frame": ms_queue, "film_identical": same},
            "plain_walk": {"ms_per_frame": ms_plain, "algorithmic_flops_per_frame": flops_ref, "frac": flops_ref / (ms_plain * 1e-3) / 1e12 / VALU_F64_PEAK_TOPS,
                           "triangles_tested": st_ref["triangles_tested"], "nodes_tested": st_ref["nodes_tested"]},
            "reference_work_rate_frac": flops_ref / (ms * 1e-3) / 1e12 / VALU_F64_PEAK_TOPS,
            "note": "byte-identical to the oracle in tests/test_gpu_configs.py.  `frac` prices the tests the pruned walk still makes against the unfused f64 rate "
                    "(profiles/r05_config4_pmc.txt: either persistent kernel issues the same 1.52e10 VALU wave-instructions, ~75 % of the issue slots at ~60 % lane use; round 5's A/Bs -- DESIGN.md 3.5 -- "
                    "moved the frame by the ORDER of its tiles, not by its instruction count: 8 % fewer VALU instructions bought 0.6 %, tiles from the middle row outwards 10 %; a triangle reached through the strips of a kept run is priced at the reference's 36 operations "
                    "although the strip answers its sign test with ~20); `plain_walk` is the same frame with every test the reference makes, "
                    "`reference_work_rate_frac` that work over the pruned walk's time (what skipping buys, not a roofline fraction)"}


def mixed_roofline(G, la, stream):
    """configs[4] (8192x8192, config 3's 1024 spheres + the 100,352-triangle torus, plastic) on ONE GPU, whole: the bookkeeping of `roofline` for
    BASELINE's sharded config -- frame time, Mrays/s, the dominant kernel by HIP events on its launch stream with its algorithmic f64 operations
    (the counting instantiation of the same walk, per kind of ray), the organisation that ran."""
    size = 8192
    acc = G.Accel(la.scenes.mixed_scene(G))
    film = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def frame_ms(reps=2):
        G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ms = frame_ms()
    ran_as = G.last_organisation(acc) or "?"
    G.profile_enable(acc, True)
    G.capture_rows_device(acc, size, size, 0, size, film.data_ptr(), row0=0, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    raw = {k: v for k, v in G.profile_read_kinds(acc).items() if v[1]}  # {kind: (total ms of the frame, launches)}: a film this size is cut into chunks
    kinds = {k: v[0] / v[1] for k, v in raw.items()}
    G.profile_read(acc)
    G.profile_enable(acc, False)
    st = G.capture_stats(acc, size, size, 0, size)
    rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
    if ran_as.split(",")[0] == "wavefront" and kinds:
        dom = max(raw, key=lambda k: raw[k][0])
        dom_ms, launches = kinds[dom], int(raw[dom][1])
        dst = dict(G.capture_stats_kind(acc, size, size, 2 if "shadow" in dom else 1, 0, size)) if dom.startswith("trace<") else st
        kernel = "lg::wf_trace_kernel<false, %s, false, %s, true, false>" % ("true" if "shadow" in dom else "false", "false" if "shadow" in dom else "true") if dom.startswith("trace<") else dom
    else:
        dom, dom_ms, dst, launches = "trace_kernel", kinds.get("trace_kernel", ms), st, int(raw.get("trace_kernel", (0, 1))[1])
        kernel = {"megakernel": "lg::trace_kernel<false, false, false, true, 1024>", "queue": "lg::queue_kernel<false, true>"}.get(ran_as.split(",")[0], ran_as)
    flops, fl_frame = algorithmic_flops(dst) / max(launches, 1), algorithmic_flops(st)  # per launch of the dominant kernel (the frame's work of that kind over its launches)
    tops = flops / (dom_ms * 1e-3) / 1e12
    del film
    torch.cuda.empty_cache()
    return {"workload": WORKLOADS["configs4"]["desc"] % (size, size) + "; ONE GPU, the whole film (BASELINE shards it over 8)",
            "ms_per_frame": ms, "value": rays / ms / 1e3, "unit": "Mrays/s", "rays_per_frame": rays,
            "bound": "valu_f64", "achieved": tops, "peak": VALU_F64_PEAK_TOPS, "frac": tops / VALU_F64_PEAK_TOPS,
            "frame_frac": fl_frame / (ms * 1e-3) / 1e12 / VALU_F64_PEAK_TOPS,
            "organisation": ran_as, "kernel": kernel, "kernel_ms_avg": dom_ms, "kernel_launches_per_frame": launches, "kernels_ms_avg": kinds,
            "algorithmic_flops_per_launch": flops, "algorithmic_flops_per_frame": fl_frame, "algorithmic_bytes_per_frame": algorithmic_bytes(st),
            "traffic": None,
            "work_per_frame": {k: st[k] for k in ("nodes_tested", "spheres_tested", "cuboids_tested", "triangles_tested", "accel_entries", "hits")},
            "traversal": "reference tree, pruned walk (default for a scene with a mesh of >= 4096 triangles)",
            "note": "byte-identical to the oracle on strided samples in tests/test_gpu_configs.py; rocprofv3 summary of the same frame: profiles/r06_config5_*"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(nproc, argv):
    """`python3 bench.py --gpus N` (no launcher, WORLD_SIZE unset): one process per GPU through torch.distributed.run, as a child of this
    process -- which has not touched the GPU -- with the same arguments; the child's stdout (rank 0's JSON line) is this process's
    stdout and its exit code this process's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)


def library_collective(args):
    """--collective library: the product's own multi-device path, timed like the torch one.  ONE process, lg_multi_* (multi.cpp): an accel per
    device, every device renders its 64-row blocks, ONE grouped ncclSend / ncclRecv brings them into the root device's film (what lg_capture
    does on a multi-GPU box, lib.rs:55-104's fan-out).  K synchronous frames after W warm-ups; the film is compared with one device's own."""
    os.environ.setdefault("LASGUN_AUTOTUNE", "2")
    import lasgun_amd as la
    G = la.api
    ndev = G.device_count()
    if ndev < 1:
        raise SystemExit("bench.py --collective library: no HIP device")
    devices = [i % ndev for i in range(args.gpus)]  # fewer devices than ranks: same-device shares (a rehearsal; LASGUN_MULTI_FORCE_RCCL=1 sends them through RCCL)
    wl = WORKLOADS[args.workload]
    w = h = args.size or wl["size"]
    scene = wl["build"](la.scenes, G)
    t0 = time.perf_counter()
    m = G.Multi(scene, devices, BLOCK_ROWS)
    create_s = time.perf_counter() - t0
    torch.cuda.set_device(devices[0])
    film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):
        m.capture_device(w, h, film.data_ptr())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.capture_device(w, h, film.data_ptr())  # synchronous: every device's render + the exchange
    elapsed = time.perf_counter() - t0
    st = G.capture_stats(m.accel(0), w, h)
    rays = st["primary_rays"] + st["shadow_rays"]
    acc = G.Accel(scene)
    ref = torch.zeros_like(film)
    G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0)
    G.synchronize(acc)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0)
    G.synchronize(acc)
    single_ms = (time.perf_counter() - t1) / args.steps * 1e3
    same = bool(torch.equal(film, ref))
    out = {"metric": "Mrays/s (primary+shadow) at 4096x4096; bit-exact RGBA8 vs CPU", "value": rays * args.steps / elapsed / 1e6, "unit": "Mrays/s",
           "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": wl["desc"] % (w, h), "workload_key": args.workload, "rays_per_frame": rays,
                      "parallelism": "ONE process, lg_multi_capture_device: 64-row blocks dealt round-robin over %d share(s) on device(s) %s, one grouped RCCL exchange per frame, frames one after the other"
                                     % (args.gpus, sorted(set(devices)))},
           "collective": "library", "uses_rccl": bool(m.uses_rccl), "rccl_ranks": m.ranks if m.uses_rccl else 0, "distinct_devices": len(set(devices)),
           "multi_create_s": create_s, "single_device_ms_per_step": single_ms, "gathered_equals_single_gpu": same,
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    if not same:
        sys.exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="configs2", choices=sorted(WORKLOADS),
                    help="BASELINE.json config a step renders: configs2 = the headline (4096^2, 1024 spheres; default), configs4 = 8192^2 mixed mesh + spheres")
    ap.add_argument("--size", type=int, default=None, help="film side in pixels (default: the workload's own, 4096 / 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fast-mode / host-film / rate-probe extras (profiling runs)")
    ap.add_argument("--sequential", action="store_true", help="A/B: frames one after the other on one stream (no overlap)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs than ranks (tiles staged through host memory)")
    ap.add_argument("--all-gather", action="store_true",
                    help="the end-of-frame collective is ONE all-gather (every rank ends with the film) instead of ONE gather to rank 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1 with a process group and a real gather: RCCL rehearsal on a 1-GPU box")
    ap.add_argument("--collective", default="torch", choices=["torch", "library"],
                    help="torch = one process per GPU, torch.distributed gather (the driver's contract); library = ONE process drives the N devices "
                         "through the product's own lg_multi_capture_device (multi.cpp: one grouped RCCL send / recv into the root's film)")
    args = ap.parse_args()

    if args.collective == "library":
        return library_collective(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` without a launcher: start one process per GPU as a CHILD, before this process has made any
        # GPU call (importing torch does not initialise the device; a process that has must never be replaced by another), hand
        # its stdout -- rank 0's one JSON line -- through, and exit with its code.
        return spawn_ranks(args.gpus, sys.argv[1:])

    exit_code = 0
    # stdout carries ONE line, the JSON record: whatever libraries print there (RCCL's version banner at communicator
    # creation, gloo's connection notes) is sent to stderr by pointing file descriptor 1 at it until the record is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    os.environ.setdefault("LASGUN_AUTOTUNE", "2")  # measure a kind of launch at its FIRST launch (the library's default: at its second), so that no timed frame holds a measurement
    import lasgun_amd as la
    G = la.api
    ndev = torch.cuda.device_count()
    dev = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)
    G.set_device(dev)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from lasgun_amd.distributed import InterleavedGather, gather_tiles, interleave_ok, row_tile

    wl = WORKLOADS[args.workload]
    if args.size is None:
        args.size = wl["size"]
    headline = args.workload == "configs2"
    w = h = args.size
    scene = wl["build"](la.scenes, G)  # replicated on every GPU
    t0 = time.perf_counter()
    acc = G.Accel(scene)  # host HLBVH build + flatten + upload (outside the timed region, reported below)
    accel_build_s = time.perf_counter() - t0
    # LASGUN_NO_LDS_SCENE=1 (A/B): the traversal kernels read the scene tables through L1/L2 instead of LDS
    lds_scene = G.set_lds_scene(acc, not os.environ.get("LASGUN_NO_LDS_SCENE")) and not os.environ.get("LASGUN_NO_LDS_SCENE")
    balanced = interleave_ok(world, h, BLOCK_ROWS)
    y0, y1 = row_tile(rank, world, h)
    cur_stream = torch.cuda.current_stream()
    if balanced:
        ig = InterleavedGather(w, h, rank, world, BLOCK_ROWS, "cuda" if args.backend == "nccl" else "cpu", always_gather=args.force_dist,
                               buffers=FRAMES_IN_FLIGHT + 1, all_ranks=args.all_gather)
        cuda_tile = torch.zeros((h // world, w, 4), dtype=torch.uint8, device="cuda") if args.backend != "nccl" else None
        frame_streams = [torch.cuda.Stream() for _ in range(FRAMES_IN_FLIGHT)]
        overlap = [not args.sequential]

        def step():
            s = frame_streams[ig.k % FRAMES_IN_FLIGHT] if overlap[0] else cur_stream
            with torch.cuda.stream(s):
                t = ig.tile()  # makes `s` wait for the gather that last read this buffer
                if t.is_cuda:
                    G.capture_interleaved_device(acc, w, h, BLOCK_ROWS, world, rank, t.data_ptr(), stream=s.cuda_stream)
                else:  # gloo rehearsal: stage through host memory
                    G.capture_interleaved_device(acc, w, h, BLOCK_ROWS, world, rank, cuda_tile.data_ptr(), stream=s.cuda_stream)
                    t.copy_(cuda_tile)
                ig.submit()  # the gather (N > 1) is ordered after this frame's render on `s`

        def finish():
            return ig.finish()
    else:  # a height the 64-row blocks do not divide: contiguous row tiles, frames one after the other
        tile = torch.zeros((y1 - y0, w, 4), dtype=torch.uint8, device="cuda")
        last = [None]
        overlap = [False]

        def step():
            G.capture_rows_device(acc, w, h, y0, y1, tile.data_ptr(), stream=cur_stream.cuda_stream)
            last[0] = gather_tiles(tile.cpu() if (args.backend == "gloo" and multi) else tile, w, h, rank, world)

        def finish():
            return last[0]

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if balanced:  # set-up, not a step: every stream's launch context allocates its queues on first use
        for _ in range(FRAMES_IN_FLIGHT):
            step()
        finish()
        fence()
    for _ in range(args.warmup):
        step()
    finish()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    full = finish()  # waits for the last gather (inside the timed region)
    fence()
    elapsed = time.perf_counter() - t0
    timed_film = full.cpu().numpy().copy() if (rank == 0 and full is not None) else None  # the frame `value` was measured on

    # the same K steps twice more (not `value`: the spread between back-to-back blocks of the same run, reported beside it)
    repeats = [elapsed]
    for _ in range(2):
        for _ in range(args.warmup):  # (the film's D2H copy above left the GPU idle: the same warm-up as before the first block)
            step()
        finish()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        finish()
        fence()
        repeats.append(time.perf_counter() - t1)

    # one frame at a time (render + gather, nothing else in flight): latency at this N
    was = overlap[0]
    overlap[0] = False
    lat = []
    for _ in range(3):
        fence()
        t1 = time.perf_counter()
        step()
        finish()
        fence()
        lat.append(time.perf_counter() - t1)
    latency_ms = min(lat) * 1e3
    # per-kernel durations for the roofline: HIP events around every kernel of three frames issued one after the other
    G.profile_enable(acc, True)
    for _ in range(3):
        step()
    finish()
    fence()
    kernel_ms, launches = G.profile_read(acc)   # device time of whole frames (all kernels of a frame)
    kinds = G.profile_read_kinds(acc)           # streaming pipeline: HIP events around each kernel
    G.profile_enable(acc, False)
    overlap[0] = was

    # deterministic work counters of this rank's share (untimed, counting kernel variant)
    def share_stats(kind=None):
        if not balanced:
            return G.capture_stats(acc, w, h, y0, y1) if kind is None else G.capture_stats_kind(acc, w, h, kind, y0, y1)
        if world == 1:
            return G.capture_stats(acc, w, h, 0, h) if kind is None else G.capture_stats_kind(acc, w, h, kind, 0, h)
        tot = None
        for g in range(h // (BLOCK_ROWS * world)):
            yb = (g * world + rank) * BLOCK_ROWS
            part = G.capture_stats(acc, w, h, yb, yb + BLOCK_ROWS) if kind is None else G.capture_stats_kind(acc, w, h, kind, yb, yb + BLOCK_ROWS)
            tot = part if tot is None else {k: tot[k] + part[k] for k in tot}
        return tot

    st = share_stats()
    keys = sorted(st)
    rdev = "cuda" if args.backend == "nccl" else "cpu"
    vec = torch.tensor([st[k] for k in keys] + [0], dtype=torch.float64, device=rdev)
    tmax = torch.tensor([elapsed, kernel_ms / max(launches, 1), latency_ms] + repeats, dtype=torch.float64, device=rdev)
    # per rank: ms per timed step as this rank saw it, and its own share rendered alone (no collective): what the gather adds to a frame
    render_only_ms = latency_ms
    if multi and balanced:
        alone = []
        for _ in range(3):
            fence()
            t1 = time.perf_counter()
            G.capture_interleaved_device(acc, w, h, BLOCK_ROWS, world, rank, (cuda_tile if cuda_tile is not None else ig.tiles[0]).data_ptr(), stream=cur_stream.cuda_stream)
            torch.cuda.synchronize()
            alone.append(time.perf_counter() - t1)
        render_only_ms = min(alone) * 1e3
    per_rank = torch.zeros((max(world, 1), 3), dtype=torch.float64, device=rdev)
    per_rank[rank if multi else 0] = torch.tensor([elapsed / args.steps * 1e3, latency_ms, render_only_ms], dtype=torch.float64, device=rdev)
    if multi:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    per_rank = per_rank.tolist()
    total = {k: int(v) for k, v in zip(keys, vec.tolist())}
    elapsed, latency_ms = float(tmax[0]), float(tmax[2])
    repeats = [float(x) for x in tmax[3:]]
    # the metric counts primary + shadow rays; neither workload has a specular material, so there are no secondary rays
    assert total["secondary_rays"] == 0, "the bench workloads must not cast secondary rays"
    rays = total["primary_rays"] + total["shadow_rays"]

    # N > 1 (always, no switch): the TIMED frame as gathered on rank 0 against the same frame rendered by rank 0's GPU alone --
    # every byte -- so that a multi-GPU number never comes without a correctness statement (lib.rs:110-162: a pixel's value does
    # not depend on the partition).  The oracle is consulted too, on a bounded sample (below); a mismatch exits non-zero.
    gathered_ok = None
    if rank == 0 and multi:
        ref = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0, stream=cur_stream.cuda_stream)
        torch.cuda.synchronize()
        gathered_ok = bool(torch.equal(torch.from_numpy(timed_film).to("cuda"), ref))
        print("verify: gathered %d-rank film %s single-GPU film" % (world, "==" if gathered_ok else "!="), file=sys.stderr)
        del ref

    extras = rank == 0 and world == 1 and not args.no_extras and not args.force_dist
    e2e_ms = fast_info = probes = mesh_info = mixed_info = None
    if extras:
        # PCIe-inclusive figure (never `value`): lg_capture into a HOST film = host BVH build + upload + render + 64 MiB D2H
        film = G.Film(w, h)
        e2e_ms = float("inf")
        for _ in range(3):  # best of 3: the first call also pays the first touch of the 64 MiB host film
            t1 = time.perf_counter()
            G.capture(scene, film)
            e2e_ms = min(e2e_ms, (time.perf_counter() - t1) * 1e3)
        # opt-in fast mode, reported beside the headline (never as `value`): same frame, film compared byte for byte
        ref_film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        G.capture_rows_device(acc, w, h, 0, h, ref_film.data_ptr(), row0=0, stream=cur_stream.cuda_stream)
        torch.cuda.synchronize()
        try:
            G.set_mode(acc, True)
            fast_film = torch.zeros_like(ref_film)
            G.capture_rows_device(acc, w, h, 0, h, fast_film.data_ptr(), row0=0, stream=cur_stream.cuda_stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                G.capture_rows_device(acc, w, h, 0, h, fast_film.data_ptr(), row0=0, stream=cur_stream.cuda_stream)
            torch.cuda.synchronize()
            fms = (time.perf_counter() - t1) / 3 * 1e3
            fast_info = {"ms_per_step": fms, "value": rays / fms / 1e3, "unit": "Mrays/s",
                         "identical_to_reference_traversal": bool(torch.equal(ref_film, fast_film)),
                         "note": "lg_accel_set_mode(1): binned-SAH tree + pruning, winner checked against the reference tree's boxes; "
                                 "verified, not proven, identical (DESIGN.md); frames one after the other"}
        finally:
            G.set_mode(acc, False)
        del ref_film
        torch.cuda.empty_cache()
        probes = {"hbm_copy_GBps": G.probe_rate("hbm_copy"), "lds_read_GBps": G.probe_rate("lds_read")}
        mesh_info = mesh_roofline(G, la, cur_stream) if (args.size == 4096 and headline) else None
        mixed_info = mixed_roofline(G, la, cur_stream) if (args.size == 4096 and headline) else None

    if rank == 0:
        value = rays * args.steps / elapsed / 1e6
        frame_ms = kernel_ms / max(launches, 1)
        per_kernel = {k: v[0] / v[1] for k, v in kinds.items() if v[1] > 0}
        if per_kernel and (len(per_kernel) > 1 or "trace_kernel" not in per_kernel):  # pipeline: closest trace (+ frame), shadow trace, shade
            dom = max(kinds, key=lambda k: kinds[k][0])  # the kind the frame spends most time in
            dom_ms = per_kernel[dom]
            dst = dict(share_stats(2 if "shadow" in dom else 1))
            dst["primary_rays"] = 0  # the RGBA write belongs to the shade kernel, not to a traversal kernel
            # a film cut into chunks (8192^2: two) launches a kind once per chunk: the frame's work of that kind over its launches is one launch's
            per_frame = max(1, round(kinds[dom][1] / 3.0))  # (three frames were profiled)
            if per_frame > 1:
                dst = {k: v / per_frame for k, v in dst.items()}
            # wavefront pipeline: lg::wf_trace_kernel<FAST, SHADOW, scene tables resident in LDS>, lg::wf_shade_kernel
            # wavefront pipeline: lg::wf_trace_kernel<FAST, SHADOW, scene tables resident in LDS, level-0 closest pass>, lg::wf_shade_kernel<KIND, L0>
            shadow = "shadow" in dom
            pruned = "true" if G.get_prune(acc) else "false"  # (LASGUN_PRUNE / lg_accel_set_prune: the PRUNE template argument of the kernel that ran)
            # (the sixth argument: the refilling shadow pass -- level 0 of a big launch over an LDS-resident scene, k_wavefront.hip: launch_wf_trace)
            refill = "true" if (shadow and lds_scene and pruned == "false" and os.environ.get("LASGUN_REFILL", "0") == "1" and w * h >= (1 << 20) * world) else "false"
            kernel_name = ("lg::wf_trace_kernel<false, %s, %s, %s, %s, %s>" % ("true" if shadow else "false", "true" if lds_scene else "false", "false" if shadow else "true", pruned, refill)
                           if dom.startswith("trace<") else "lg::wf_shade_kernel<0, true>")
        else:  # megakernel (a share too small for the pipeline, e.g. a small --size over many ranks)
            dom_ms, dst, kernel_name = frame_ms, st, "lg::trace_kernel<false, false, %s, %s, 1024>" % ("true" if lds_scene else "false", "true" if G.get_prune(acc) else "false")
            per_kernel = {"trace_kernel": frame_ms}
        dom_bytes, dom_flops = algorithmic_bytes(dst), algorithmic_flops(dst)
        secs = dom_ms * 1e-3
        traffic, traffic_src = profiled_traffic(kernel_name, world, args.size) if headline else (None, None)
        tops = dom_flops / secs / 1e12
        gbs = dom_bytes / secs / 1e9
        hbm_measured = probes["hbm_copy_GBps"] if probes else None
        lds_measured = probes["lds_read_GBps"] if probes else None
        out = {
            "metric": "Mrays/s (primary+shadow) at 4096x4096; bit-exact RGBA8 vs CPU",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "latency_ms": latency_ms,
            "value_repeats": [rays * args.steps / r / 1e6 for r in repeats],  # [0] is `value`; the other two: the same K steps again
            "value_single_frame": rays / latency_ms / 1e3,  # Mrays/s of ONE frame issued alone (render + gather), nothing else in flight
            "config": {"workload": wl["desc"] % (w, h), "workload_key": args.workload,
                       "rays_per_frame": rays, "primary": total["primary_rays"], "shadow": total["shadow_rays"],
                       "parallelism": ("64-row blocks dealt round-robin over %d rank(s)%s; consecutive frames %s"
                                       % (world, " + 1 RCCL gather per frame" if multi else "",
                                          "overlap, %d in flight" % FRAMES_IN_FLIGHT if overlap[0] else "one after the other")) if balanced
                       else ("contiguous row tiles x%d + 1 gather, frames one after the other" % world),
                       "accel_build_s": accel_build_s, "host_film_capture_ms": e2e_ms,
                       "work_per_frame": {k: total[k] for k in ("nodes_tested", "spheres_tested", "cuboids_tested", "triangles_tested", "accel_entries", "hits")}},
            "roofline": {"bound": "valu_f64", "achieved": tops, "peak": VALU_F64_PEAK_TOPS, "unit": "TFLOP/s", "frac": tops / VALU_F64_PEAK_TOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms_avg": dom_ms,
                         "algorithmic_flops_per_launch": dom_flops, "algorithmic_bytes_per_launch": dom_bytes,
                         "frame_device_ms": frame_ms, "kernels_ms_avg": per_kernel,
                         "scene_tables": "LDS" if lds_scene else "L1/L2",
                         "hbm_algorithmic": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "x_peak": gbs / HBM_PEAK_GBS,
                                             "measured_copy_peak": hbm_measured,
                                             "x_measured_copy_peak": (gbs / hbm_measured) if hbm_measured else None,
                                             "note": "SURVEY 8(d) bytes (node / primitive records the traversal demands) over the kernel's duration; "
                                                     "the scene is resident in LDS (or L2), so these bytes never reach HBM and the ratio is not a fraction"},
                         "lds": {"achieved": gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": gbs / LDS_PEAK_GBS,
                                 "measured_read_peak": lds_measured, "frac_of_measured": (gbs / lds_measured) if lds_measured else None},
                         "note": "unit = unfused f64 operations (no FMA: -ffp-contract=off is part of the parity contract); counts are lower bounds "
                                 "of the reference's algorithm (DESIGN.md, Roofline bookkeeping)"},
        }
        at_clock(out["roofline"], kernel_name)
        if mesh_info is not None:
            at_clock(mesh_info, mesh_info["kernel"])
        if mixed_info is not None:
            at_clock(mixed_info, mixed_info["kernel"])
        if fast_info is not None:
            out["fast_mode"] = fast_info
        if mesh_info is not None:
            out["roofline_mesh"] = mesh_info
        if mixed_info is not None:
            out["roofline_mixed"] = mixed_info
        check = None
        if world == 1 and not args.no_cpu_baseline and not args.force_dist and not args.no_extras:
            out["cpu_baseline"], check = cpu_baseline(w, h, timed_film, build=wl["build"])
        elif multi and not args.no_cpu_baseline:
            # N > 1: no CPU baseline figure (rank 0 at N = 1 only), but the gathered frame still meets the oracle on a bounded
            # sample (every 64th pixel: a fraction of a second of host time)
            check = oracle_sample_check(w, h, timed_film, 64 if headline else 1024, build=wl["build"])
        if multi:
            out["gathered_equals_single_gpu"] = gathered_ok
            out["rccl_ranks"] = dist.get_world_size()
            out["collective_backend"] = args.backend
            out["collective"] = "torch"
            # per rank: ms per timed step (frames overlapped), one frame alone with its gather, the rank's share alone without it;
            # gather_ms = what the collective adds to a frame issued alone, slowest rank
            out["per_rank_ms"] = [{"rank": r, "ms_per_step": v[0], "latency_ms": v[1], "render_only_ms": v[2]} for r, v in enumerate(per_rank)]
            out["gather_ms"] = max(v[1] for v in per_rank) - max(v[2] for v in per_rank)
        out["bit_exact"] = check["bit_exact"] if check else None       # null: the oracle leg did not run (--no-cpu-baseline)
        out["mismatched_bytes"] = check["mismatched_bytes"] if check else None
        out["bit_exact_check"] = check
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if check is not None and not check["bit_exact"]:
            exit_code = 3  # the metric says "bit-exact RGBA8 vs CPU": a frame that is not, is not a result
        if gathered_ok is False:
            exit_code = 4  # the gathered film is not the film
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
