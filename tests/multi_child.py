"""Child process of the multi-device tests: ONE lg_multi_* scenario per process, always started with a time limit by the test
that runs it (tests/test_gpu_multi_device.py, tests/test_gpu_configs.py) -- a collective that does not complete (ncclCommInitAll on
the first box with 8 distinct GPUs, a send without its receive) then costs one failed test, not a hung suite.  Nothing here is an
exec of a process that already touched the GPU: the tests start `python tests/multi_child.py ...` from scratch.

The scenario: a film of --scene at --w x --h rendered as len(--devices) shares (64-row blocks dealt round-robin with
--block-rows 64, contiguous row tiles with 0) through lg_multi_capture_device / lg_multi_capture (and the all-gather form with
--all-gather), compared with the single-device film, byte for byte, and with the CPU oracle: the whole film for the small scenes,
a strided sample of --sample pixels for the BASELINE configs.  Reference: src/lib.rs:55-104 (the fan-out these replace),
:110-162,152 (a pixel's value does not depend on the partition).  Prints MULTI_CHILD_OK as its last line."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kitchen_sink")
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--block-rows", type=int, default=64)
    ap.add_argument("--w", type=int, default=256)
    ap.add_argument("--h", type=int, default=1024)
    ap.add_argument("--force-rccl", action="store_true", help="LASGUN_MULTI_FORCE_RCCL=1: a repeated device's shares go through ncclSend / ncclRecv to self")
    ap.add_argument("--all-gather", action="store_true")
    ap.add_argument("--sample", type=int, default=0, help="oracle check on a strided sample of this many pixels instead of the whole film")
    ap.add_argument("--host-film", action="store_true", help="also lg_multi_capture into a host film")
    ap.add_argument("--repeats", type=int, default=2)
    args = ap.parse_args()
    if args.force_rccl:
        os.environ["LASGUN_MULTI_FORCE_RCCL"] = "1"
    import numpy as np
    import torch
    import lasgun_amd as la
    from oracle_lib import oracle
    G, S = la.api, la.scenes
    builders = {
        "kitchen_sink": lambda api: S.kitchen_sink_scene(api),
        "cornell_glass": lambda api: S.cornell_scene(api, "glass"),
        "config3": lambda api: S.spheres_scene(api),
        "config4": lambda api: S.mesh_scene(api, 224, 224, "glass"),
        "config5": lambda api: S.mixed_scene(api),
    }
    build = builders[args.scene]
    devices = [int(x) for x in args.devices.split(",")]
    w, h = args.w, args.h
    distinct = len(set(devices)) > 1
    # ---- the single-device film (device 0), in HBM
    G.set_devices([devices[0]])
    acc = G.Accel(build(G))
    one = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:%d" % devices[0])
    torch.cuda.synchronize(devices[0])
    G.capture_rows_device(acc, w, h, 0, h, one.data_ptr(), row0=0)
    G.synchronize(acc)
    # ---- against the oracle
    o = oracle()
    nthreads = max(1, min(64, len(os.sched_getaffinity(0))))
    if args.sample:
        n = max(1, (w * h) // args.sample)
        idx = np.arange(n // 3, w * h, n, dtype=np.uint64)
        want, _ = o.capture_pixels(o.Accel(build(o)), w, h, idx, radiance=False, nthreads=nthreads)
        got = one.view(-1, 4)[torch.from_numpy(idx.astype(np.int64)).to(one.device)].cpu().numpy()
        assert np.array_equal(got, want), ("single-device film vs oracle sample", int((got != want).sum()))
    else:
        f = o.Film(w, h)
        o.capture_subset_mt(0, 1, o.Accel(build(o)), f, min(32, nthreads))
        assert np.array_equal(one.cpu().numpy(), f.pixels()), "single-device film vs oracle"
    print("single-device film ok", flush=True)
    # ---- the shares through lg_multi_*
    m = G.Multi(build(G), devices, args.block_rows)
    assert m.ranks == len(devices)
    if args.force_rccl or distinct:
        assert m.uses_rccl or os.environ.get("LASGUN_CAPTURE_NO_RCCL"), "the RCCL exchange was expected"
    else:
        assert not m.uses_rccl  # one distinct device: RCCL is not even loaded
    dev = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda:%d" % devices[0])
    torch.cuda.synchronize(devices[0])
    for rep in range(args.repeats):  # tiles and communicators are reused
        m.capture_device(w, h, dev.data_ptr())
        assert torch.equal(dev, one), ("gathered film != single-device film", rep, int((dev != one).sum().item()))
        dev.fill_(9)
        torch.cuda.synchronize(devices[0])
    print("gathered film ok", flush=True)
    if args.host_film:
        host = G.Film.new_with_output(w, h, np.full((h, w, 4), 9, np.uint8))
        m.capture(host)
        assert np.array_equal(host.pixels(), one.cpu().numpy()), "host film"
    if args.all_gather and m.uses_rccl and distinct:
        bufs = [torch.full((h, w, 4), 7, dtype=torch.uint8, device="cuda:%d" % d) for d in devices]
        for d in devices:
            torch.cuda.synchronize(d)
        m.capture_device_all(w, h, [b.data_ptr() for b in bufs])
        for d, b in zip(devices, bufs):
            assert np.array_equal(b.cpu().numpy(), one.cpu().numpy()), ("all-gather", d)
    m.close()
    print("MULTI_CHILD_OK ranks=%d rccl=%d" % (len(devices), int(bool(args.force_rccl or distinct))), flush=True)


if __name__ == "__main__":
    main()
