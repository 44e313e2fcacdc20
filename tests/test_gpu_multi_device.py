"""Multi-device paths over DISTINCT devices.  On a 1-GPU box every test here is skipped (the same paths run there against a
repeated device: tests/test_gpu_configs.py, test_gpu_parity.py); on a node with 2, 4 or 8 GPUs they light up by themselves:
lg_multi_capture / lg_multi_capture_device / lg_multi_capture_device_all and the default-device lg_capture over 2 / 4 / 8 devices
against the single-device film and the CPU oracle -- interleaved 64-row blocks and contiguous row tiles, the RCCL exchange
(ncclCommInitAll, one grouped send / recv into the root's film, all-gather form) and the LASGUN_CAPTURE_NO_RCCL=1 fall-through
(one host thread and one D2H copy per device).  Reference: src/lib.rs:55-104 (the fan-out these replace), :110-162 (a pixel's
value does not depend on the partition)."""
import os
import subprocess
import sys

import pytest

import lasgun_amd as la

pytestmark = pytest.mark.gpu
G = la.api
S = la.scenes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ndev():
    try:
        return G.device_count()
    except la.LasgunError:
        return 0


def need(n):
    if ndev() < n:
        pytest.skip("needs %d distinct HIP devices, this box has %d" % (n, ndev()))


SCENES = {"kitchen_sink": lambda api: S.kitchen_sink_scene(api), "cornell_glass": lambda api: S.cornell_scene(api, "glass")}


def run_multi_child(args, env_extra=None, timeout=600):
    """One lg_multi_* scenario in a FRESH process with a time limit (tests/multi_child.py): every test that initialises RCCL goes
    through here -- a hung ncclCommInitAll / exchange is one failed test, never a hung pytest."""
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_child.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0 and "MULTI_CHILD_OK" in p.stdout, (args, p.stdout[-800:], p.stderr[-3000:])
    return p.stdout


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("block_rows, w, h", [(64, 256, 1024), (0, 203, 336)])
@pytest.mark.parametrize("scene", sorted(SCENES))
def test_multi_capture_over_distinct_devices(scene, block_rows, w, h, n):
    """Every rank renders its share on ITS device, one grouped RCCL exchange (or device-to-device copies) puts the shares into the
    root's film: device film, host film and the all-gather form, against the single-device film and the oracle -- in a child
    process with a time limit (tests/multi_child.py)."""
    need(n)
    run_multi_child(["--scene", scene, "--devices", ",".join(str(d) for d in range(n)), "--block-rows", block_rows, "--w", w, "--h", h,
                     "--host-film", "--all-gather"])


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("block_rows", [64, 0])
def test_config5_sharded_over_distinct_devices(block_rows, n):
    """BASELINE configs[4] -- 8192^2, mesh + 1024 spheres -- as n shares on n DISTINCT devices with one RCCL gather: the gathered
    film equals the single-device film, which equals the oracle on a 16384-pixel sample (the 1-GPU rehearsal of the same thing
    with a repeated device: tests/test_gpu_configs.py::test_config5_as_eight_shares)."""
    need(n)
    run_multi_child(["--scene", "config5", "--devices", ",".join(str(d) for d in range(n)), "--block-rows", block_rows, "--w", 8192, "--h", 8192,
                     "--sample", 16384, "--repeats", 1], timeout=900)


def run_child(code, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env)


CHILD = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
         "os.environ['LASGUN_DEBUG'] = '1'\n"
         "import numpy as np, lasgun_amd as la\n"
         "from oracle_lib import oracle\n"
         "G = la.api; S = la.scenes; o = oracle()\n"
         "w, h = %d, %d\n"
         "%s\n"
         "film = G.Film(w, h); G.capture(S.cornell_scene(G, 'glass'), film)\n"
         "want = o.render(S.cornell_scene(o, 'glass'), (w, h)).pixels()\n"
         "assert np.array_equal(film.pixels(), want), int((film.pixels() != want).sum())\n"
         "film2 = G.Film(w, h); G.capture(S.cornell_scene(G, 'glass'), film2)   # communicators are cached per device list\n"
         "assert np.array_equal(film2.pixels(), want)\n"
         "print('capture over', G.device_count(), 'devices ok')\n")


@pytest.mark.parametrize("no_rccl", [False, True])
@pytest.mark.parametrize("w, h", [(256, 1024), (517, 640), (131, 200)])
def test_default_capture_splits_over_every_visible_device(w, h, no_rccl):
    """A process that names no device: lg_capture takes every visible one for films of 2^18 pixels and more (lib.rs:58-62: every
    core) -- interleaved blocks and contiguous tiles, over RCCL and over the per-device fall-through -- and the current device
    for a smaller film; fresh process each."""
    need(2)
    p = run_child(CHILD % (ROOT, ROOT, w, h, ""), {"LASGUN_CAPTURE_NO_RCCL": "1"} if no_rccl else None)
    assert p.returncode == 0 and "devices ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


@pytest.mark.parametrize("n", [2, 4, 8])
def test_capture_over_a_named_device_list(n):
    """lg_set_devices([0 .. n-1]) + lg_capture, fresh process, over RCCL and without."""
    need(n)
    for env in (None, {"LASGUN_CAPTURE_NO_RCCL": "1"}):
        p = run_child(CHILD % (ROOT, ROOT, 192, 512, "G.set_devices(list(range(%d)))" % n), env)
        assert p.returncode == 0 and "devices ok" in p.stdout, (env, p.stdout[-500:], p.stderr[-3000:])


def test_bench_gathered_film_is_verified_at_n_ranks():
    """bench.py's own launch line on as many GPUs as the box has (at most 4 here): the record carries
    gathered_equals_single_gpu = true, the RCCL rank count and a bit_exact statement against the oracle."""
    need(2)
    import json
    n = min(ndev(), 4)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29650 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--size", "1024",
           "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and line["rccl_ranks"] == n and line["gathered_equals_single_gpu"] is True and line["bit_exact"] is True
