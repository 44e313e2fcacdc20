"""BASELINE.json's configs[3] and configs[4] at FULL size on the GPU against the CPU oracle.

configs[3]: 4096x4096, ~100k-triangle OBJ mesh (torus, 100,352 triangles) in glass / metal + a mirror sphere:
            254-triangle reference leaves (bvh.rs:187,289) and specular recursion (integrate.rs:69-79,82-132).
configs[4]: 8192x8192, the same mesh (plastic) + config 3's 1024 spheres.

The oracle cannot render 16.7 / 67.1 Mpixel films in test time, so each film is checked on
  (a) a strided sample {k + i*n} of 16384 pixels -- RGBA8 byte for byte (libm oracle) and f64 radiance bit for bit
      (portable-trig oracle) --, in every kernel organisation (megakernel, streaming pipeline, fast mode): the
      FULL film is rendered on the device in each and sampled;
  (b) the committed 64x64 crop goldens (tests/golden/config4_*, config5_*; oracle outputs, make_golden.py);
  (c) size-independent properties: the organisations' full films are byte-identical to each other, alpha = 255,
      ray accounting.
"""
import os

import numpy as np
import pytest

import lasgun_amd as la
from golden_cases import CROPS
from lasgun_amd import scenes as S
from oracle_lib import oracle
from test_gpu_parity import bits

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = la.api

FULL = {
    "config4_glass": (lambda api: S.mesh_scene(api, 224, 224, "glass"), 4096, 1019, 331),
    "config4_metal": (lambda api: S.mesh_scene(api, 224, 224, "metal"), 4096, 1031, 77),
    "config5_mixed": (lambda api: S.mixed_scene(api), 8192, 4099, 1234),
}
# (label, streaming, wavefront, fast).  streaming 0 = megakernel; 2 = streamed wherever the organisation exists for the scene:
# the wavefront pipeline takes every scene (glass / mirror: level by level)
# prune: the pruned form of the reference walk (lg_accel_set_prune; these scenes' default) against the plain one
ORGANISATIONS = [("megakernel", 0, True, False, False), ("wavefront", 2, True, False, False),
                 ("megakernel-pruned", 0, True, False, True), ("wavefront-pruned", 2, True, False, True),
                 ("megakernel-fast", 0, True, True, False), ("wavefront-fast", 2, True, True, False),
                 ("queue", 3, True, False, False), ("queue-pruned", 3, True, False, True)]


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_config_vs_oracle_sample(name):
    import torch
    builder, size, n, k = FULL[name]
    w = h = size
    o = oracle()
    oacc = o.Accel(builder(o))
    idx = np.arange(k, w * h, n, dtype=np.uint64)
    nthreads = max(1, min(64, len(os.sched_getaffinity(0))))
    o.set_trig_mode(0)
    want_rgba, _ = o.capture_pixels(oacc, w, h, idx, radiance=False, nthreads=nthreads)  # libm: what the Rust binary calls
    o.set_trig_mode(1)
    try:
        want_rgba_p, want_rad = o.capture_pixels(oacc, w, h, idx, nthreads=nthreads)    # portable trig: bit-comparable
    finally:
        o.set_trig_mode(0)
    assert np.array_equal(want_rgba, want_rgba_p)
    acc = G.Accel(builder(G))
    film = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
    first = None
    for label, streaming, wavefront, fast, prune in ORGANISATIONS:
        G.set_streaming(acc, streaming)
        G.set_mode(acc, fast)
        G.set_prune(acc, prune)
        film.zero_()
        torch.cuda.synchronize()  # the fill runs on torch's stream, the render on the accel's own
        G.capture_rows_device(acc, w, h, 0, h, film.data_ptr(), row0=0)
        G.synchronize(acc)
        flat = film.view(-1, 4)
        got = flat[torch.from_numpy(idx.astype(np.int64)).cuda()].cpu().numpy()
        assert np.array_equal(got, want_rgba), (name, label, int((got != want_rgba).sum()))
        # the same pixels through the pixel-list entry point: bytes again, and radiance bits
        rgba, rad = G.capture_pixels(acc, w, h, idx)
        assert np.array_equal(rgba, want_rgba), (name, label)
        assert np.array_equal(bits(rad), bits(want_rad)), (name, label)
        assert bool((flat[:, 3] == 255).all())
        if first is None:
            first = film.clone()
        else:
            assert torch.equal(film, first), (name, label)  # whole film, organisation against organisation
    G.set_streaming(acc, 1)
    G.set_mode(acc, False)
    G.set_prune(acc, None)
    # ray accounting on two 32-row bands: through the torus (config 4 glass: its refractions) and through the mirror sphere
    a, b = G.capture_stats(acc, w, h, h // 2, h // 2 + 32), G.capture_stats(acc, w, h, (h * 11) // 16, (h * 11) // 16 + 32)
    st = {key: a[key] + b[key] for key in a}
    assert st["primary_rays"] == 64 * w and st["shadow_rays"] == st["hits"]
    if name == "config5_mixed":
        assert st["secondary_rays"] == 0
    else:
        assert st["secondary_rays"] > 0


@pytest.mark.parametrize("name", list(CROPS))
def test_full_size_config_crop_goldens(name):
    builder, w, h, x0, y0, cw, ch = CROPS[name]
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    acc = G.Accel(builder(G))
    for label, streaming, wavefront, fast, prune in ORGANISATIONS:
        G.set_streaming(acc, streaming)
        G.set_mode(acc, fast)
        G.set_prune(acc, prune)
        rgba, rad = G.capture_rect(acc, w, h, x0, y0, x0 + cw, y0 + ch)
        assert np.array_equal(rgba, z["rgba"]), (name, label)
        assert np.array_equal(bits(rad), bits(z["radiance"])), (name, label)


def test_rect_and_pixel_list_agree_with_the_film():
    """lg_capture_rect / lg_capture_pixels address the same pixels as the film (ragged sizes, every organisation)."""
    w, h = 203, 117
    acc = G.Accel(S.kitchen_sink_scene(G))
    for streaming in (0, 2, 3):
        G.set_streaming(acc, streaming)
        film = G.Film(w, h)
        G.capture_subset(0, 1, acc, film)
        full = film.pixels()
        frad = G.capture_radiance(acc, w, h)
        rgba, rad = G.capture_rect(acc, w, h, 5, 9, 198, 110)
        assert np.array_equal(rgba, full[9:110, 5:198]) and np.array_equal(bits(rad), bits(frad[9:110, 5:198]))
        rng = np.random.default_rng(3)
        idx = rng.permutation(w * h)[:1000].astype(np.uint64)
        rgba, rad = G.capture_pixels(acc, w, h, idx)
        assert np.array_equal(rgba, full.reshape(-1, 4)[idx.astype(np.int64)])
        assert np.array_equal(bits(rad), bits(frad.reshape(-1, 3)[idx.astype(np.int64)]))
    with pytest.raises(la.LasgunError):
        G.capture_pixels(acc, w, h, [w * h])
    with pytest.raises(la.LasgunError):
        G.capture_rect(acc, w, h, 0, 0, w + 1, h)


def test_capture_subset_from_concurrent_threads():
    """The reference's own threading pattern (lib.rs:67-103): n threads call capture_subset(i, n, &accel, film) on
    ONE accel and ONE film at the same time.  Every pixel must arrive, none may be lost to another call's copy."""
    import threading
    w, h, n = 320, 200, 8
    o = oracle()
    want = o.Film(w, h)
    o.capture_subset_mt(0, 1, o.Accel(S.cornell_scene(o, "glass")), want, 8)
    acc = G.Accel(S.cornell_scene(G, "glass"))
    for streaming in (0, 2, 3):
        G.set_streaming(acc, streaming)
        for _ in range(3):
            buf = np.full((h, w, 4), 9, np.uint8)
            film = G.Film.new_with_output(w, h, buf)
            errs = []

            def work(i):
                try:
                    G.capture_subset(i, n, acc, film)
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
            ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            assert not errs, errs
            assert np.array_equal(buf, want.pixels())


@pytest.mark.parametrize("devices, w, h, block_rows", [([0, 0], 256, 256, 64), ([0, 0, 0], 203, 160, 64), ([0, 0, 0, 0], 128, 512, 64),
                                                       ([0, 0], 128, 256, 0), ([0], 64, 64, 64)])
def test_multi_device_capture_gathers_the_film_on_the_root(devices, w, h, block_rows):
    """lg_multi_*: every rank renders its share (64-row blocks dealt round-robin, or contiguous row tiles) on its device, the
    shares are gathered in the ROOT's device memory.  A 1-GPU box names its device several times (device-local copies); the
    same film must come out as from a single-device capture -- device film and host film."""
    import torch
    scene = S.kitchen_sink_scene(G)
    one = G.Film(w, h)
    G.capture(scene, one)
    m = G.Multi(scene, devices, block_rows)
    assert m.ranks == len(devices) and not m.uses_rccl  # one distinct device: RCCL is not even loaded
    dev = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):  # tiles are reused
        m.capture_device(w, h, dev.data_ptr())
        assert np.array_equal(dev.cpu().numpy(), one.pixels())
    host = G.Film.new_with_output(w, h, np.full((h, w, 4), 9, np.uint8))
    m.capture(host)
    assert np.array_equal(host.pixels(), one.pixels())
    G.set_mode(m.accel(len(devices) - 1), True)  # per-rank settings through the rank's accel: fast mode on the last rank
    m.capture(host)
    assert np.array_equal(host.pixels(), one.pixels())
    m.close()


def test_multi_device_capture_through_rccl_on_one_gpu():
    """The RCCL leg on a 1-GPU box: LASGUN_MULTI_FORCE_RCCL=1 sends the shares of a repeated device through
    ncclSend / ncclRecv to self inside ONE ncclGroupStart / ncclGroupEnd (communicator from ncclCommInitAll, dlopen'ed
    librccl).  In a child process with a time limit: a collective that does not complete must not hang the suite."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "os.environ['LASGUN_MULTI_FORCE_RCCL'] = '1'\n"
            "import numpy as np, torch, lasgun_amd as la\n"
            "G = la.api; S = la.scenes\n"
            "scene = S.cornell_scene(G, 'glass'); w, h = 160, 256\n"
            "one = G.Film(w, h); G.capture(scene, one)\n"
            "for devices, b in (([0, 0], 64), ([0, 0, 0, 0], 64), ([0, 0], 0)):\n"
            "    m = G.Multi(scene, devices, b); assert m.uses_rccl\n"
            "    dev = torch.full((h, w, 4), 9, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize()\n"
            "    m.capture_device(w, h, dev.data_ptr()); m.capture_device(w, h, dev.data_ptr())\n"
            "    assert np.array_equal(dev.cpu().numpy(), one.pixels()), devices\n"
            "    m.close()\n"
            "print('rccl gather ok')\n") % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "rccl gather ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("block_rows, force_rccl", [(64, False), (0, False), (64, True), (0, True)])
def test_config5_as_eight_shares(block_rows, force_rccl):
    """BASELINE configs[4] -- "8192x8192 mixed mesh + spheres, row-tile sharded across 8 GPUs with RCCL gather" -- through the library's
    own sharded path on a 1-GPU box: lg_multi_* over devices [0] * 8, as 8 shares of interleaved 64-row blocks and as 8 contiguous row
    tiles, gathered by device-local copies and (LASGUN_MULTI_FORCE_RCCL=1) by ncclSend / ncclRecv inside one group.  The gathered
    8192^2 film == the single-device film, every byte, == the CPU oracle on a 16384-pixel strided sample.  In a child process with a
    time limit (tests/multi_child.py).  The same scenario over DISTINCT devices: tests/test_gpu_multi_device.py.
    Reference: src/lib.rs:55-104 (fan-out), :152 (the partition never changes a pixel)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tests", "multi_child.py"), "--scene", "config5", "--devices", ",".join(["0"] * 8),
           "--block-rows", str(block_rows), "--w", "8192", "--h", "8192", "--sample", "16384", "--repeats", "1"]
    if force_rccl:
        cmd.append("--force-rccl")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "MULTI_CHILD_OK ranks=8" in p.stdout, (p.stdout[-800:], p.stderr[-3000:])


def test_wavefront_bands_on_internal_streams_leave_the_film_unchanged():
    """lg_accel_set_wf_split: a 2048^2 launch cut into 1 / 2 / 4 / 8 bands rendered on internal streams (fork from and join
    into the caller's stream), glass scene included: byte-identical films, and identical to the megakernel's."""
    import torch
    w = h = 2048
    for build in (lambda: S.spheres_scene(G), lambda: S.cornell_scene(G, "glass")):
        acc = G.Accel(build())
        G.set_streaming(acc, 0)
        ref = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0)
        G.synchronize(acc)
        G.set_streaming(acc, 2)
        for bands in (1, 2, 4, 8, 0):
            G.set_wf_split(acc, bands)
            film = torch.full((h, w, 4), 7, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            G.capture_rows_device(acc, w, h, 0, h, film.data_ptr(), row0=0, stream=s.cuda_stream)
            s.synchronize()
            assert torch.equal(film, ref), bands


def test_capture_into_a_host_film_in_row_bands_is_the_same_film():
    """capture() / capture_subset(0, 1) of a film of 2^22 pixels and more renders in row bands on the accel's internal streams, each
    band followed by its own copy home (lg_capture_subset): the host film equals the device film of one launch, for a film whose height
    is not a multiple of the band size, in the megakernel, level by level and in the queue organisation; and equals the oracle's on a sample."""
    import torch
    w, h = 2048, 2056 + 8 * 3  # 260 rows of tiles: bands of 65 rows of tiles
    for build, orgs in ((lambda api: S.cornell_scene(api, "glass"), (0, 2, 3)), (lambda api: S.spheres_scene(api), (1,))):
        scene = build(G)
        acc = G.Accel(scene)
        ref = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        G.capture_rows_device(acc, w, h, 0, h, ref.data_ptr(), row0=0)
        G.synchronize(acc)
        want = ref.cpu().numpy()
        for org in orgs:
            G.set_streaming(acc, org)
            buf = np.full((h, w, 4), 9, np.uint8)
            G.capture_subset(0, 1, acc, G.Film.new_with_output(w, h, buf))
            assert np.array_equal(buf, want), org
        film = G.Film(w, h)
        G.capture(scene, film)  # lib.rs:55-104: rebuilds the accel, then the same path
        assert np.array_equal(film.pixels(), want)
        o = oracle()
        idx = np.arange(977, w * h, 4099, dtype=np.uint64)
        want_s, _ = o.capture_pixels(o.Accel(build(o)), w, h, idx, radiance=False, nthreads=max(1, min(64, len(os.sched_getaffinity(0)))))
        assert np.array_equal(want.reshape(-1, 4)[idx.astype(np.int64)], want_s.reshape(-1, 4))


@pytest.mark.parametrize("w, h, block_rows", [(128, 256, 64), (96, 80, 0)])
def test_multi_device_all_gather_form_single_rank(w, h, block_rows):
    """lg_multi_capture_device_all, the all-gather form (every rank's device ends with the whole film), at the one size a 1-GPU
    box can run it -- one rank, one device: interleaved blocks (put in row order by the strided copies) and a contiguous tile
    (rendered in place).  A repeated device is refused: a communicator has one rank per device."""
    import torch
    scene = S.kitchen_sink_scene(G)
    one = G.Film(w, h)
    G.capture(scene, one)
    m = G.Multi(scene, [0], block_rows)
    dev = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        m.capture_device_all(w, h, [dev.data_ptr()])
        assert np.array_equal(dev.cpu().numpy(), one.pixels())
    m.close()
    m2 = G.Multi(scene, [0, 0], block_rows)
    with pytest.raises(la.LasgunError):
        m2.capture_device_all(w, h, [dev.data_ptr(), dev.data_ptr()])
    m2.close()


def test_the_pruned_walks_leaf_records_are_built_when_it_will_run():
    """The pruned walk is the default from 4096 triangles in a mesh (below that it measured slower, and its leaf records are up to half the
    accel build: profiles/r05_prune_threshold.jsonl); lg_accel_set_prune(1) on an accel built without the records builds them then, and the
    film is the same either way."""
    if os.environ.get("LASGUN_PRUNE"):
        pytest.skip("LASGUN_PRUNE overrides the defaults this test is about")
    w, h = 160, 120
    small = G.Accel(S.mesh_scene(G, 24, 16, "glass"))  # 768 triangles
    assert not G.get_prune(small)
    plain = G.Film(w, h)
    G.capture_subset(0, 1, small, plain)
    G.set_prune(small, True)
    assert G.get_prune(small)
    pruned = G.Film(w, h)
    G.capture_subset(0, 1, small, pruned)
    assert np.array_equal(plain.pixels(), pruned.pixels())
    r = G.audit_prune(small, w, h)
    assert r["violations"] == 0 and r["skipped_runs"] > 0, r  # (the records are there: runs of a leaf get skipped)
    G.set_prune(small, None)
    assert not G.get_prune(small)
    again = G.Film(w, h)
    G.capture_subset(0, 1, small, again)
    assert np.array_equal(plain.pixels(), again.pixels())
    big = G.Accel(S.mesh_scene(G, 48, 48, "metal"))  # 4608 triangles
    assert G.get_prune(big)


def test_capture_takes_every_visible_device_by_default_and_prune_switch():
    """A process that never names a device (like a program written against the reference) gets every visible one from
    lg_capture (lib.rs:58-62: every core); and lg_accel_set_prune takes -1 / 0 / 1 only, the film is the same in each."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
            "os.environ['LASGUN_DEBUG'] = '1'\n"
            "import numpy as np, lasgun_amd as la\n"
            "from oracle_lib import oracle\n"
            "G = la.api; S = la.scenes; o = oracle()\n"
            "film = G.Film(96, 64); G.capture(S.cornell_scene(G, 'glass'), film)   # no set_device / set_devices before\n"
            "want = o.render(S.cornell_scene(o, 'glass'), (96, 64)).pixels()\n"
            "assert np.array_equal(film.pixels(), want)\n"
            "print('default devices ok')\n") % (root, root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "default devices ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    acc = G.Accel(S.mesh_scene(G, 32, 24, "glass"))
    films = []
    for v in (None, False, True, -1):
        G.set_prune(acc, v)
        f = G.Film(80, 60)
        G.capture_subset(0, 1, acc, f)
        films.append(f.pixels())
    assert all(np.array_equal(films[0], x) for x in films[1:])
    with pytest.raises(la.LasgunError):
        if G.call("accel_set_prune", acc.h, 2):
            raise la.LasgunError(G.last_error())


def _bench_line(extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
    return json.loads(lines[0])


def test_bench_gpus_2_without_a_launcher():
    """`python3 bench.py --gpus 2` as the driver would type it, with NO torch.distributed.run in front: bench.py starts its ranks itself (a child
    process, before the parent has touched the GPU), stdout is rank 0's one JSON line and the exit code the child's.  gloo on this box's one GPU
    (both ranks render on device 0, tiles staged through host memory): the gathered film == one GPU's film == the oracle sample.
    Reference counterpart: the fan-out over threads, src/lib.rs:55-104."""
    d = _bench_line(["--gpus", "2", "--backend", "gloo", "--size", "1024", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo" and d["collective"] == "torch"
    assert d["gathered_equals_single_gpu"] is True and d["bit_exact"] is True
    assert [r["rank"] for r in d["per_rank_ms"]] == [0, 1] and all(r["ms_per_step"] > 0.0 and r["render_only_ms"] > 0.0 for r in d["per_rank_ms"])
    assert "gather_ms" in d and d["scaling"] == "strong" and d["value"] > 0.0


def test_bench_library_collective_on_same_device_shares():
    """`bench.py --gpus 2 --collective library`: the product's own exchange (lg_multi_capture_device, multi.cpp) timed by the same script --
    here two shares on device 0; the film equals the single-device film."""
    d = _bench_line(["--gpus", "2", "--collective", "library", "--size", "1024", "--steps", "3", "--warmup", "1"])
    assert d["collective"] == "library" and d["n_gpus"] == 2 and d["gathered_equals_single_gpu"] is True and d["value"] > 0.0
