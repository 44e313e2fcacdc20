#!/usr/bin/env python3
"""The all-configs table of DESIGN.md section 3.5 from a directory of tools/bench_configs.py outputs (tools/final_profile.sh writes them):
python tools/configs_table.py gpurun_out/final4"""
import json
import os
import sys
d = sys.argv[1]
NAMES = {"1a": "1a README sphere 512²", "1b": "1b simple.rs 512², 9 spp", "2P": "2P Cornell plastic 512²", "2G": "2G Cornell glass 512² (recursion 3)",
         "3 ": "3 1024 spheres 4096² (**headline**, one frame at a time)", "4 ": "4 100k-triangle torus glass + mirror 4096²", "4m": "4m same mesh, metal",
         "5 ": "5 spheres + mesh 8192²"}


def load(name):
    out = {}
    p = os.path.join(d, "configs_%s.jsonl" % name)
    if os.path.exists(p):
        for l in open(p):
            r = json.loads(l)
            out[r["config"][:2]] = r
    return out


par, mega, wf, q, unp, fast = (load(n) for n in ("parity", "megakernel", "wavefront", "queue", "unpruned", "fast"))


def org_of(r):
    k = r["kernels_ms"]
    return "queue" if "trace_kernel" in k else "wavefront" if "trace<closest>" in k else "megakernel"


def ms(t, key):
    return ("%.3f" if t[key]["ms"] < 3 else "%.2f" if t[key]["ms"] < 20 else "%.1f") % t[key]["ms"] if key in t else "—"


print("| config | **default** | default Mrays/s | megakernel | wavefront pipeline | queue organisation | plain walk (default organisation) | fast mode |")
print("|---|---|---|---|---|---|---|---|")
for key, title in NAMES.items():
    r = par[key]
    print("| %s | **%s** (%s) | %d | %s | %s | %s | %s | %s |" % (title, ms(par, key), org_of(r), round(r["Mrays_s"]), ms(mega, key), ms(wf, key), ms(q, key), ms(unp, key), ms(fast, key)))
