#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in the built device objects (lasgun_amd/csrc/k_*.o), from the code objects' own metadata
(llvm-readelf --notes on the gfx950 code object unbundled from each object file): one line per kernel, sorted by scratch.
usage: python tools/kernel_resources.py [substring ...] > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    out = os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + out], stderr=subprocess.DEVNULL)
    return out


def kernels(co):
    text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    rows = []
    for block in re.split(r"\n\s*- \.agpr_count:", text)[1:]:
        block = ".agpr_count:" + block
        get = lambda k, d="": (re.search(r"\.%s:\s*(\S+)" % re.escape(k), block) or [None, d])[1]  # noqa: E731
        sym = get("symbol").strip("'")
        name = subprocess.run(["c++filt", sym.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
        rows.append({"kernel": name, "vgpr": int(get("vgpr_count", "0")), "agpr": int(get("agpr_count", "0")), "sgpr": int(get("sgpr_count", "0")),
                     "vgpr_spill": int(get("vgpr_spill_count", "0")), "sgpr_spill": int(get("sgpr_spill_count", "0")),
                     "scratch_bytes_per_lane": int(get("private_segment_fixed_size", "0")), "lds_bytes_static": int(get("group_segment_fixed_size", "0")),
                     "max_wg": int(get("max_flat_workgroup_size", "0"))})
    return rows


def main():
    only = sys.argv[1:]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "lasgun_amd", "csrc", "k_*.o"))):
            for r in kernels(code_object(obj, tmp)):
                r["object"] = os.path.basename(obj)
                rows.append(r)
    rows.sort(key=lambda r: (-r["scratch_bytes_per_lane"], r["kernel"]))
    print("%-9s %5s %5s %6s %6s %8s  %s" % ("object", "vgpr", "sgpr", "vspill", "sspill", "scratch", "kernel"))
    for r in rows:
        if only and not any(o in r["kernel"] for o in only):
            continue
        print("%-9s %5d %5d %6d %6d %8d  %s" % (r["object"].replace(".o", ""), r["vgpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes_per_lane"], r["kernel"]))


if __name__ == "__main__":
    main()
