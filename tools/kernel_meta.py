#!/usr/bin/env python3
"""kernel_meta.py <lib.so> [substr] : VGPRs / AGPRs / SGPRs / scratch / static LDS / max workgroup of every kernel in the
gfx950 code object of a built library (from the code object's AMDGPU metadata note)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    tmp = tempfile.mkdtemp()
    dst = os.path.join(tmp, "lib.so")
    subprocess.check_call(["cp", os.path.realpath(lib), dst])
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, capture_output=True)
    notes = "".join(subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, co)], text=True)
                    for co in sorted(os.listdir(tmp)) if "gfx950" in co)          # one code object per translation unit
    subprocess.call(["rm", "-rf", tmp])
    out = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        name = g("name")
        dem = subprocess.check_output(["c++filt", name], text=True).strip()
        dem = re.sub(r"^void ", "", dem).split("(")[0]
        out.append(dict(name=dem, vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                        spill=g("vgpr_spill_count"), wg=g("max_flat_workgroup_size")))
    return out


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = [k for k in kernels(sys.argv[1]) if sub in k["name"]]
    print(f"{len(ks)} kernels")
    for k in sorted(ks, key=lambda k: k["name"]):
        print(f"v{k['vgpr']:>4} a{k['agpr']:>3} s{k['sgpr']:>4} scratch{k['scratch']:>6} spill{k['spill']:>4} lds{k['lds']:>7} wg{k['wg']:>5}  {k['name']}")
