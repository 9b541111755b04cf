#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage read from the code object inside the built libsgym_hip.so.

    python tools/kernel_resources.py [path/to/lib.so]       prints one line per kernel
    from tools.kernel_resources import table                 -> {demangled name: dict(vgpr, agpr, sgpr, lds, scratch)}

(the table kernels of the rollout have to stay within 168 VGPRs -- three wavefronts per SIMD -- and control_kernel within 128:
tests/test_host_api.py checks it on every build.)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def table(lib=None):
    lib = lib or os.path.join(ROOT, "scenario_gym_amd", "lib", "libsgym_hip.so")
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "x.so")])
        # one offload bundle per object the library was linked from (csrc/Makefile: one object per kernel family)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        notes = ""
        for i, a in enumerate(starts):
            part, co = os.path.join(d, f"fat{i}.bin"), os.path.join(d, f"dev{i}.co")
            with open(part, "wb") as f:
                f.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = {}
    for k in notes.split("  - .agpr_count")[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        rows[name] = dict(vgpr=g("vgpr_count"), agpr=int(re.match(r":\s+(\d+)", k).group(1)), sgpr=g("sgpr_count"),
                          lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"))
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"\(sg::Params.*", "", n): v for n, v in zip(names, rows.values())}


if __name__ == "__main__":
    for n, r in table(sys.argv[1] if len(sys.argv) > 1 else None).items():
        print(f"{n[:72]:72s} vgpr={r['vgpr']:3d} agpr={r['agpr']:3d} sgpr={r['sgpr']:3d} lds={r['lds']:6d} scratch={r['scratch']}")
