#!/usr/bin/env python3
"""Constant-time audit of the secret-handling kernels in libbee2hip.so (bign_sign_kernels.hip): disassemble
the gfx950 code object and list, per kernel,
  * every conditional branch with the instruction(s) that computed its condition,
  * every vector-memory instruction (global_ / scratch_ / buffer_ / flat_) with its address operand,
  * the scalar loads (table reads) and the LDS instruction count.
The script classifies nothing by itself beyond "condition comes from v_cmp (per-lane data)" vs "s_cmp (scalar)";
profiles/r02_sign_ct_audit.txt is its output plus the hand-written reading of every flagged line.
usage: python tools/ct_audit.py [path/to/libbee2hip.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = ("bign_mulbase_ct_kernel", "bign_mulbase_lds_kernel", "bign_mulbase_coop_kernel", "bign_sign_nonce_kernel", "bign_sign_kcheck_kernel", "bign_sign_tail_kernel",
           "bign_generic_mulbase_kernel", "bign_generic_sign_tail_kernel")


def disasm(lib):
    tmp = tempfile.mkdtemp()
    name = os.path.join(tmp, "lib.so")
    subprocess.check_call(["cp", lib, name])
    subprocess.check_call([f"{LLVM}/llvm-objdump", "--offloading", name], cwd=tmp, stdout=subprocess.DEVNULL)
    return "\n".join(subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, co)], text=True)
                     for co in sorted(os.listdir(tmp)) if "amdgcn" in co)          # one code object per translation unit


def demangle(n):
    return subprocess.check_output(["c++filt", n], text=True).strip()


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bee2_amd", "lib", "libbee2hip.so")
    text = disasm(lib)
    funcs = re.split(r"\n(?=[0-9a-f]{16} <)", text)
    callees = {}
    for f in funcs:
        m = re.match(r"[0-9a-f]{16} <(\S+)>:", f)
        if m:
            callees[m.group(1)] = f
    for f in funcs:
        m = re.match(r"[0-9a-f]{16} <(\S+)>:", f)
        if not m or not any(k in m.group(1) for k in KERNELS):
            continue
        lines = [l.split("//")[0].strip() for l in f.split("\n")[1:] if l.strip()]
        print(f"=== {demangle(m.group(1))}   ({len(lines)} instructions)")
        ops = [l.split()[0] for l in lines if l]
        def count(pfx):
            return sum(1 for o in ops if o.startswith(pfx))
        print(f"    s_load: {count('s_load')}  global_load: {count('global_load')}  global_store: {count('global_store')}  "
              f"scratch: {count('scratch_')}  ds_: {count('ds_')}  v_mad_u64_u32: {count('v_mad_u64_u32')}  "
              f"s_swappc/s_setpc (calls/returns): {count('s_swappc') + count('s_setpc')}")
        nscalar = 0
        for i, l in enumerate(lines):
            if not l.startswith("s_cbranch"):
                continue
            if l.startswith("s_cbranch_scc"):
                nscalar += 1                       # SCC comes from SALU compares of SGPRs: wavefront-uniform, never lane data
                continue
            # vcc / exec branches: find the vector compare(s) that fed the mask
            src = []
            for j in range(i - 1, max(-1, i - 40), -1):
                o = lines[j].split()[0]
                if o.startswith("v_cmp") or o.startswith("v_cmpx"):
                    src.append(lines[j])
                    break
                if o.startswith("s_cbranch") or o.startswith("s_branch"):
                    break
            print(f"    lane-mask branch @{i:6d}: {l:<26s} <- {src[0] if src else '(mask set earlier: ' + lines[i - 1] + ')'}")
        print(f"    scalar (SCC) branches: {nscalar}")
        vm = [l for l in lines if re.match(r"(global|scratch|buffer|flat)_", l)]
        for l in vm:
            print(f"    vmem: {l}")
    print("=== callees reached by s_swappc from these kernels are listed by name in the kernel bodies' s_getpc/s_add sequences;")
    for n in callees:
        if "fe_inv" in n or "fe_mul_call" in n or "fe_sqr_call" in n:
            lines = [l.split("//")[0].strip() for l in callees[n].split("\n")[1:] if l.strip()]
            br = [l for l in lines if l.startswith("s_cbranch")]
            print(f"    {demangle(n)[:90]}: {len(lines)} instructions, {len(br)} conditional branches: {sorted(set(b.split()[0] for b in br))}")


if __name__ == "__main__":
    main()
