"""LASGUN_TUNE_FILE: the table of measured kernel-organisation choices persists across processes (csrc/tune.cpp; include/lasgun_hip.h, lg_tune_*).
No GPU needed: the table is host state, and what an entry MEANS is only looked at when a launch uses it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json
sys.path.insert(0, %r)
import lasgun_amd as la
G = la.api
cmd = sys.argv[1]
if cmd == "import":
    G.tune_import([((1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 2**64 - 1), 66), ((7,) * 12, 1)])
elif cmd == "clear":
    G.tune_clear()
print(json.dumps(G.tune_export()))
"""


def run(cmd, path):
    env = dict(os.environ, LASGUN_TUNE_FILE=path)
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT, cmd], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    return [(tuple(k), c) for k, c in json.loads(p.stdout.strip().splitlines()[-1])]


def test_choices_survive_the_process_through_the_file(tmp_path):
    path = str(tmp_path / "tune.txt")
    assert run("show", path) == []                       # no file yet
    want = [((1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 2 ** 64 - 1), 66), ((7,) * 12, 1)]
    assert sorted(run("import", path)) == sorted(want)
    assert os.path.exists(path) and len(open(path).read().splitlines()) == 2
    assert sorted(run("show", path)) == sorted(want)     # another process: read back from the file
    with open(path, "a") as f:
        f.write("not a line of the table\n3 3 3 3 3 3 3 3 3 3 3 3 999\n")  # garbage and an impossible choice are ignored
    assert sorted(run("show", path)) == sorted(want)
    assert run("clear", path) == [] and open(path).read() == ""
    assert run("show", path) == []
