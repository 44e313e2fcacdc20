#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

metric  : Mrays/s (primary + shadow) at 4096x4096, bit-exact RGBA8 vs the CPU oracle
workload: configs[2] -- "4096x4096, 1024 random spheres (deep BVH), 1 spp" (the headline config;
          lasgun_amd.scenes.spheres_scene, SplitMix64 seed 0x1A560001)
step    : one full pass of the hot path = one 4096x4096 frame (lg_capture_rows_device per rank
          + ONE RCCL gather of the RGBA8 row tiles when N > 1).  The flattened scene / BVH is
          resident in HBM before the timed region; the film stays in HBM (no PCIe in `value`).
N > 1   : one process per GPU (torch.distributed.run); rank r renders row tile r of the SAME
          4096x4096 film, so total work is fixed -> "strong" scaling.

Prints ONE JSON line on rank 0.  `roofline` is computed from the trace kernel's own deterministic
work counters (bytes of BVH-node / primitive records its traversal demands) over the kernel's
average duration measured with HIP events on the launch stream; `cpu_baseline` times the CPU
oracle (a port: the Rust reference cannot be built here) on a bounded strided sample of the same
frame on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy peak
# algorithmic bytes per unit of traversal work: SURVEY.md section 8(d)'s per-unit figures (restated in DESIGN.md
# "Roofline bookkeeping"): 56 B per node whose bounds are tested, 32 B per sphere / 48 B per cuboid / 48 B per triangle
# tested, 256 B (m, minv) per nested-BVH entry, 4 B material id per accepted closest hit, 4 B RGBA per primary ray.
# (The +36 B of vertex normals per accepted smoothed-triangle hit is left out: it is not counted separately.)
BYTES_NODE, BYTES_SPHERE, BYTES_CUBOID, BYTES_TRI, BYTES_ACCEL_ENTRY, BYTES_HIT, BYTES_PIXEL = 56, 32, 48, 48, 256, 4, 4


def algorithmic_bytes(st):
    return (BYTES_NODE * st["nodes_tested"] + BYTES_SPHERE * st["spheres_tested"] + BYTES_CUBOID * st["cuboids_tested"]
            + BYTES_TRI * st["triangles_tested"] + BYTES_ACCEL_ENTRY * st["accel_entries"] + BYTES_HIT * st["hits"]
            + BYTES_PIXEL * st["primary_rays"])


# The resource that actually binds the traversal kernels is f64 VALU issue (the scene is cache / LDS
# resident).  Unfused f64 operations the reference's algorithm needs per test, lower bounds (miss paths):
# slab test 26 (cuboid.rs:104-121), sphere 26 up to the discriminant test (sphere.rs:30-69), cuboid 26,
# triangle 36 up to the edge-function signs (triangle.rs:186-230), accel entry 2 x (36 + 3 divisions).
# Peak: 256 CUs x 4 SIMDs x 16 f64 lanes x 2.4 GHz = 39.3 T unfused ops/s (-ffp-contract=off is part of
# the parity contract, so an FMA's second flop is not available).
FLOPS_NODE, FLOPS_SPHERE, FLOPS_CUBOID, FLOPS_TRI, FLOPS_ACCEL_ENTRY = 26, 26, 26, 36, 78
VALU_F64_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12


def algorithmic_flops(st):
    return (FLOPS_NODE * st["nodes_tested"] + FLOPS_SPHERE * st["spheres_tested"] + FLOPS_CUBOID * st["cuboids_tested"]
            + FLOPS_TRI * st["triangles_tested"] + FLOPS_ACCEL_ENTRY * st["accel_entries"])


def cpu_baseline(width, height, target_seconds=15.0):
    """Time the CPU oracle on a bounded strided sample {k + i*n} of the same frame (all host cores)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    from lasgun_amd import scenes
    o = oracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    acc = o.Accel(scenes.spheres_scene(o))
    film = o.Film(width, height)
    area = width * height

    def run(n):
        o.stats_reset()
        t0 = time.perf_counter()
        o.capture_subset_mt(0, n, acc, film, cores)
        dt = time.perf_counter() - t0
        st = o.stats_read()
        return dt, st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"], st["primary_rays"]

    dt, rays, _ = run(max(1, area // 32768))  # calibration: ~32k pixels
    rate = rays / dt
    want_pixels = min(area, max(65536, int(rate * target_seconds / 2.0)))
    n = max(1, area // want_pixels)
    dt, rays, pixels = run(n)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "capture_subset(0, n=%d) of the same %dx%d frame: %d pixels, %d rays in %.2f s on %d threads"
                      % (n, width, height, pixels, rays, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs than ranks (tiles staged through host memory)")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1 through the N>1 code path (process group, interleaved tile, async gather): RCCL rehearsal on a 1-GPU box")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

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

    w = h = args.size
    scene = la.scenes.spheres_scene(G)  # replicated on every GPU
    t0 = time.perf_counter()
    acc = G.Accel(scene)  # host HLBVH build + flatten + upload (outside the timed region, reported below)
    accel_build_s = time.perf_counter() - t0
    if os.environ.get("LASGUN_PACKET"):  # A/B: one tree walk per wavefront
        G.set_packet(acc, os.environ["LASGUN_PACKET"] == "1")
    # LASGUN_NO_LDS_SCENE=1 (A/B): the traversal kernels read the scene tables through L1/L2 instead of LDS
    lds_scene = G.set_lds_scene(acc, not os.environ.get("LASGUN_NO_LDS_SCENE")) and not os.environ.get("LASGUN_NO_LDS_SCENE")
    stream = torch.cuda.current_stream().cuda_stream
    BLOCK_ROWS = 64
    balanced = multi and interleave_ok(world, h, BLOCK_ROWS)
    y0, y1 = row_tile(rank, world, h)
    if balanced:
        # rank r renders the 64-row blocks {r, r+N, r+2N, ...} (even load) into a compact tile; the
        # gather of frame k overlaps the render of frame k+1 (two tile buffers)
        ig = InterleavedGather(w, h, rank, world, BLOCK_ROWS, "cuda" if args.backend == "nccl" else "cpu", always_gather=True)
        cuda_tile = torch.zeros((h // world, w, 4), dtype=torch.uint8, device="cuda")
        # consecutive frames alternate between two streams (each with its own launch context in the library): the
        # primary pass of frame k+1 fills the tails of frame k's shadow and shade passes -- at 1/8 of a frame per
        # rank those tails are 15-20 % of a rank's render time
        frame_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        overlap = [True]

        def step():
            s = frame_streams[ig.k % 2] if overlap[0] else torch.cuda.current_stream()
            with torch.cuda.stream(s):
                t = ig.tile()  # makes `s` wait for the gather that last read this buffer
                if t.is_cuda:
                    G.capture_interleaved_device(acc, w, h, BLOCK_ROWS, world, rank, t.data_ptr(), stream=s.cuda_stream)
                else:  # gloo rehearsal: stage through host memory
                    G.capture_interleaved_device(acc, w, h, BLOCK_ROWS, world, rank, cuda_tile.data_ptr(), stream=s.cuda_stream)
                    t.copy_(cuda_tile)
                ig.submit()  # the gather is ordered after this frame's render on `s`

        def finish():
            return ig.finish()
    else:
        tile = torch.zeros((y1 - y0, w, 4), dtype=torch.uint8, device="cuda")
        last = [None]

        def step():
            G.capture_rows_device(acc, w, h, y0, y1, tile.data_ptr(), stream=stream)
            if args.backend == "gloo" and multi:
                last[0] = gather_tiles(tile.cpu(), w, h, rank, world)
            else:
                last[0] = gather_tiles(tile, w, h, rank, world)

        def finish():
            return last[0]

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish()
    fence()
    if not balanced:
        G.profile_enable(acc, True)  # HIP events around every kernel of the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    full = finish()  # waits for the last gather (inside the timed region)
    fence()
    elapsed = time.perf_counter() - t0
    if balanced:
        # the timed frames overlap each other, which stretches every kernel's wall time; the per-kernel durations
        # for the roofline come from a few more frames run one after the other on one stream
        overlap[0] = False
        G.profile_enable(acc, True)
        for _ in range(3):
            step()
        finish()
        fence()
    kernel_ms, launches = G.profile_read(acc)   # device time of whole frames (all kernels of a frame)
    kinds = G.profile_read_kinds(acc)           # streaming pipeline: HIP events around each kernel
    G.profile_enable(acc, False)

    # deterministic work counters of this rank's tile (untimed, counting kernel variant)
    if balanced:  # counters of the rows this rank owns: sum over its 64-row blocks
        st = None
        for g in range(h // (BLOCK_ROWS * world)):
            yb = (g * world + rank) * BLOCK_ROWS
            part = G.capture_stats(acc, w, h, yb, yb + BLOCK_ROWS)
            st = part if st is None else {k: st[k] + part[k] for k in st}
    else:
        st = G.capture_stats(acc, w, h, y0, y1)
    keys = sorted(st)
    rdev = "cuda" if args.backend == "nccl" else "cpu"
    vec = torch.tensor([st[k] for k in keys] + [0], dtype=torch.float64, device=rdev)
    tmax = torch.tensor([elapsed, kernel_ms / max(launches, 1)], dtype=torch.float64, device=rdev)
    if multi:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total = {k: int(v) for k, v in zip(keys, vec.tolist())}
    elapsed = float(tmax[0])
    kernel_ms_avg = float(tmax[1])  # slowest rank's average launch: the one that bounds the frame
    rays = total["primary_rays"] + total["shadow_rays"] + total["secondary_rays"]

    # PCIe-inclusive figure (never `value`): lg_capture into a HOST film = host BVH build + upload + render + 64 MiB D2H
    e2e_ms = None
    if rank == 0 and world == 1:
        film = G.Film(w, h)
        e2e_ms = float("inf")
        for _ in range(3):  # best of 3: the first call also pays the first touch of the 64 MiB host film
            t0 = time.perf_counter()
            G.capture(scene, film)
            e2e_ms = min(e2e_ms, (time.perf_counter() - t0) * 1e3)

    if rank == 0 and multi and os.environ.get("LASGUN_BENCH_VERIFY"):
        ref = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(full.to("cuda"), ref), "gathered film differs from the single-GPU film"
        print("verify: gathered %d-rank film == single-GPU film" % world, file=sys.stderr)

    # opt-in fast mode, reported beside the headline (never as `value`): same frame, film compared byte for byte
    fast_info = None
    if rank == 0 and world == 1:
        ref_film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        G.capture_rows_device(acc, w, h, 0, h, ref_film.data_ptr(), row0=0, stream=stream)
        torch.cuda.synchronize()
        try:
            G.set_mode(acc, True)
            fast_film = torch.zeros_like(ref_film)
            G.capture_rows_device(acc, w, h, 0, h, fast_film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                G.capture_rows_device(acc, w, h, 0, h, fast_film.data_ptr(), row0=0, stream=stream)
            torch.cuda.synchronize()
            fms = (time.perf_counter() - t0) / 3 * 1e3
            fast_info = {"ms_per_step": fms, "value": rays / fms / 1e3, "unit": "Mrays/s",
                         "identical_to_reference_traversal": bool(torch.equal(ref_film, fast_film)),
                         "note": "lg_accel_set_mode(1): binned-SAH tree + pruning, winner checked against the reference tree's boxes; verified, not proven, identical (DESIGN.md)"}
        finally:
            G.set_mode(acc, False)

    if rank == 0:
        value = rays * args.steps / elapsed / 1e6
        # roofline of the DOMINANT kernel of this rank's frame
        frame_ms = kernel_ms / max(launches, 1)
        per_kernel = {k: v[0] / v[1] for k, v in kinds.items() if v[1] > 0}
        if per_kernel:  # streaming pipeline: K1 primary trace (+ shading frame), K2 shadow trace, K3 shade
            dom = max(per_kernel, key=per_kernel.get)
            dom_ms = per_kernel[dom]
            kind = 2 if "shadow" in dom else 1
            if balanced:
                dst = None
                for g in range(h // (BLOCK_ROWS * world)):
                    yb = (g * world + rank) * BLOCK_ROWS
                    part = G.capture_stats_kind(acc, w, h, kind, yb, yb + BLOCK_ROWS)
                    dst = part if dst is None else {k: dst[k] + part[k] for k in dst}
            else:
                dst = G.capture_stats_kind(acc, w, h, kind, y0, y1)
            dst = dict(dst)
            dst["primary_rays"] = 0  # the RGBA write belongs to the shade kernel, not to a traversal kernel
            my_bytes = algorithmic_bytes(dst)
            tail = ", true, false>" if lds_scene else ", false, false>"  # <FAST, SHADOW, scene tables resident in LDS, FIXUP>
            kernel_name = "lg::" + dom.replace("<primary>", "<false, false" + tail).replace("<shadow>", "<false, true" + tail)
            dom_flops = algorithmic_flops(dst)
        else:  # megakernel (scenes with glass / mirror, small films)
            dom_ms, my_bytes, kernel_name, per_kernel = frame_ms, algorithmic_bytes(st), "lg::trace_kernel<false, false>", {"trace_kernel": frame_ms}
            dom_flops = algorithmic_flops(st)
        achieved = my_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KB)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_final_pmc.json")))
            t = pmc["kernels"].get(kernel_name)
            if t and world == 1 and (w, h) == (4096, 4096):
                traffic = (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "Mrays/s (primary+shadow) at 4096x4096; bit-exact RGBA8 vs CPU",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: %dx%d, Cornell shell + 1024 random plastic spheres (SplitMix64 0x1A560001), 1 spp, 1 point light" % (w, h),
                       "rays_per_frame": rays, "primary": total["primary_rays"], "shadow": total["shadow_rays"],
                       "secondary": total["secondary_rays"], "parallelism": ("64-row blocks interleaved over %d ranks + 1 RCCL gather per frame; consecutive frames overlap on two streams" % world) if balanced else ("row-tiles x%d + 1 gather" % world),
                       "accel_build_s": accel_build_s, "host_film_capture_ms": e2e_ms,
                       "work_per_frame": {k: total[k] for k in ("nodes_tested", "spheres_tested", "cuboids_tested", "triangles_tested", "accel_entries", "hits")}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": kernel_name, "kernel_ms_avg": dom_ms,
                         "algorithmic_bytes_per_launch": my_bytes,
                         "frame_device_ms": frame_ms, "kernels_ms_avg": per_kernel,
                         "scene_tables": "LDS" if lds_scene else "L1/L2",
                         "valu_f64": {"achieved": dom_flops / (dom_ms * 1e-3) / 1e12, "peak": VALU_F64_PEAK_TOPS, "unit": "T unfused f64 ops/s",
                                      "frac": dom_flops / (dom_ms * 1e-3) / 1e12 / VALU_F64_PEAK_TOPS, "algorithmic_flops_per_launch": dom_flops},
                         "note": "algorithmic bytes = node/primitive records the traversal demands (DESIGN.md); the 160 KB scene is resident in "
                                 "LDS / L2, so they never reach HBM (frac exceeds 1); the binding resource is f64 VALU issue: see valu_f64"},
        }
        if fast_info is not None:
            out["fast_mode"] = fast_info
        if world == 1 and not args.no_cpu_baseline and not args.force_dist:
            out["cpu_baseline"] = cpu_baseline(w, h)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
