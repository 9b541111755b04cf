#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage from the device assembly of libsgym_hip.so's source.

    python tools/kernel_resources.py [file.s]     (without an argument: compiles csrc/sgym_hip.hip to /tmp/isa/sgym.s first)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-array-bounds "
         "-Wno-bitwise-instead-of-logical -Wno-unused-command-line-argument -mllvm --disable-promote-alloca-to-lds -S --cuda-device-only").split()


def emit(path, extra=()):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", path,
                           os.path.join(ROOT, "scenario_gym_amd", "csrc", "sgym_hip.hip")])


def table(path):
    txt = open(path).read()
    meta = txt[txt.find("amdhsa.kernels"):]
    rows = []
    for k in meta.split("  - .agpr_count")[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        rows.append((name, g("vgpr_count"), int(re.match(r":\s+(\d+)", k).group(1)), g("sgpr_count"),
                     g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = re.sub(r"\(sg::Params.*", "", n)
        print(f"{n[:72]:72s} vgpr={r[1]:3d} agpr={r[2]:3d} sgpr={r[3]:3d} lds={r[4]:6d} scratch={r[5]}")


if __name__ == "__main__":
    p = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/sgym.s"
    if len(sys.argv) <= 1:
        emit(p)
    table(p)
