#!/usr/bin/env python3
"""Constant-time audit of the HOST signing path (bee2_amd/csrc/host_bign_ct.hpp), static and dynamic.

static   the header is compiled twice -- g++ -O2 and the product's own host compiler and flags (hipcc's clang, -O3) -- into
         tests/hostshim/host_bign_ct_shim.cpp's C view; every function of namespace bee2hip::hostct (and the shim entries
         they are inlined into) is disassembled and EVERY conditional jump is listed with the instruction that set its
         flags and the source line it belongs to (-g line table).  A branch is acceptable when its condition is public: a
         loop counter, a window of the public exponent p - 2, the code a caller sees (bad key / bad one-time key), the
         rejection loop of algorithm 6.3.3 (as in the reference).  The listing is short enough to read in full;
         profiles/r04_host_ct_audit.txt holds it with the reading of every line.
dynamic  dudect-style: k G for five classes of scalars (random, 1..16, q-1-i, 2^j + 1, all-ones) interleaved, cycles by
         rdtsc around each call, the 10 % slowest samples of each class cropped, class means compared (Welch t).  A
         multiplier that skipped zero windows, left early on small scalars or branched on a digit would differ by
         several per cent; a constant-time one stays inside the noise.

usage: python tools/ct_audit_x86.py [--dynamic-only] [--samples N]"""
import ctypes
import math
import os
import random
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "hostshim", "host_bign_ct_shim.cpp")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
LLVM = "/opt/rocm/lib/llvm/bin"
# (name, compile command, disassembler): GNU objdump does not read clang's DWARF 5 line table, llvm-objdump does
COMPILERS = (("hipcc host pass (clang -O3, the product's flags)", [f"{LLVM}/clang++", "-O3", "-g", "-std=c++17", "-fPIC"],
              [f"{LLVM}/llvm-objdump", "-d", "-l", "-C", "--no-show-raw-insn"]),
             ("g++ -O2 (what tests/test_host_bign_ct.py builds)", ["g++", "-O2", "-g", "-std=c++17", "-fPIC"],
              ["objdump", "-d", "-l", "-C", "--no-show-raw-insn"]))
SECRET_FUNCS = ("hostct::", "hc_sign", "hc_pubkey_calc", "hc_time")


CORE = ("FieldCt<", "::mul_base(", "::mod_q(", "::sub_mod_q(", "jac_madd<")     # the arithmetic that sees secrets
CORE_PUBLIC_BRANCH_LINES = ("for (int bit = 64 * N - 8",)                        # a^(p - 2): the exponent's windows are public


def violations(compiler_index=0, carry_only=False):
    """-> (list of offending lines, number of core functions looked at) for one compiler, or (None, 0) when it is not
    installed.  Offending = inside a CORE function: any conditional jump on the carry / overflow flag, and any FORWARD
    conditional jump that is not on one of the public lines -- i.e. everything except loop back-edges."""
    name, cmd, dis = COMPILERS[compiler_index]
    tmp = tempfile.mkdtemp()
    obj = os.path.join(tmp, "shim.o")
    try:
        subprocess.check_call(cmd + ["-c", "-o", obj, SHIM])
    except (subprocess.CalledProcessError, FileNotFoundError):
        return None, 0
    text = subprocess.check_output(dis + [obj], text=True)
    src_lines = open(os.path.join(ROOT, "bee2_amd", "csrc", "host_bign_ct.hpp")).read().split("\n")
    bad, seen = [], 0
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        m = re.match(r"[0-9a-f]{16} <(.+)>:", f)
        if not m or "hostct::" not in m.group(1) or not any(k in m.group(1) for k in CORE):
            continue
        seen += 1
        src = None
        for l in f.split("\n")[1:]:
            if re.match(r"^(; )?/.*:\d+", l.strip()):
                src = l.strip().lstrip("; ")
                continue
            mm = re.match(r"^\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", l)
            if not mm or not re.match(r"^j(?!mp)", mm.group(2)):
                continue
            addr, op, arg = mm.groups()
            tgt = re.match(r"(0x)?([0-9a-f]+)", arg.strip())
            forward = not (tgt and int(tgt.group(2), 16) < int(addr, 16))
            line_text = ""
            lm = re.search(r"host_bign_ct\.hpp:(\d+)", src or "")
            if lm:
                line_text = src_lines[int(lm.group(1)) - 1]
            public = any(k in line_text for k in CORE_PUBLIC_BRANCH_LINES) or "wipe" in line_text or ("volatile" in line_text)
            # carry_only: for a compiler whose loops are rotated (guard jumps ahead of the body) and which adds stack-protector
            # checks, only the flag class is decisive: a jump on CF / OF can only come from arithmetic on data
            if re.match(r"^j(ae|b|c|nc|o|no|nae|nb)$", op) or (forward and not public and not carry_only):
                bad.append(f"{m.group(1)[:80]} @{addr}: {op} {'forward' if forward else 'backward'}  {src}  | {line_text.strip()[:80]}")
    return bad, seen


def static_audit():
    for name, cmd, dis in COMPILERS:
        tmp = tempfile.mkdtemp()
        obj = os.path.join(tmp, "shim.o")
        try:
            subprocess.check_call(cmd + ["-c", "-o", obj, SHIM])
        except (subprocess.CalledProcessError, FileNotFoundError) as e:
            print(f"=== {name}: not available here ({e})")
            continue
        text = subprocess.check_output(dis + [obj], text=True)
        print(f"=== {name}")
        funcs = re.split(r"\n(?=[0-9a-f]{16} <)", text)
        total = 0
        for f in funcs:
            m = re.match(r"[0-9a-f]{16} <(.+)>:", f)
            if not m or not any(k in m.group(1) for k in SECRET_FUNCS):
                continue
            lines = f.split("\n")[1:]
            insns, src = [], None
            for l in lines:
                if re.match(r"^(; )?/.*:\d+", l.strip()):
                    src = l.strip().lstrip("; ").replace(ROOT + "/", "").replace("tests/hostshim/../../", "")
                    continue
                mm = re.match(r"^\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", l)
                if mm:
                    insns.append((mm.group(1), mm.group(2), mm.group(3), src))
            jcc = [i for i, x in enumerate(insns) if re.match(r"^j(?!mp)", x[1])]
            calls = sorted({x[2].split("<")[-1].rstrip(">") for x in insns if x[1].startswith("call")})
            print(f"--- {m.group(1)[:150]}   ({len(insns)} instructions, {len(jcc)} conditional jumps)")
            if calls:
                print(f"    calls: {', '.join(c[:60] for c in calls)}")
            for i in jcc:
                addr, op, arg, s = insns[i]
                k = i - 1
                while k >= 0 and not re.match(r"^(cmp|test|sub|add|and|or|dec|inc|shr|shl|sar|neg|xor|bt)", insns[k][1]):
                    k -= 1
                setter = f"{insns[k][1]} {insns[k][2]}" if k >= 0 else "?"
                tgt = re.match(r"(0x)?([0-9a-f]+)", arg.strip())
                back = "backward (loop)" if tgt and int(tgt.group(2), 16) < int(addr, 16) else "forward"
                print(f"    {addr}: {op:5s} {back:15s} flags from [{setter[:48]:48s}]  {s or ''}")
            total += len(jcc)
        print(f"=== {name}: {total} conditional jumps in the secret-handling functions\n")


def dynamic_audit(samples):
    import orclib
    orc = orclib.load()
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "libhostbignct.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, SHIM])
    lib = ctypes.CDLL(so)
    assert lib.hc_init(orc.beltH()) == 1
    lib.hc_time_pubkey_calc.restype = ctypes.c_uint64
    rnd = random.Random(404)
    worst = 0.0
    for l in (128, 192, 256):
        no = l // 4
        src = open(os.path.join(ROOT, "bee2_amd", "csrc", "bign_curves.inc")).read()
        m = re.search(r"k_bign%d_q\[[^\]]*\]\s*=\s*\{([^}]*)\}" % l, src)
        q = int.from_bytes(bytes(int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", m.group(1))), "little")
        classes = {
            "random": lambda: rnd.randrange(1, q),
            "small 1..16": lambda: rnd.randrange(1, 17),
            "q-1-i": lambda: q - 1 - rnd.randrange(16),
            "2^j+1": lambda: (1 << rnd.randrange(8, 2 * l - 1)) + 1,
            "all-ones windows": lambda: (1 << (2 * l - 1)) - 1 - rnd.randrange(4),
        }
        names = list(classes)
        t = {n: [] for n in names}
        pub = ctypes.create_string_buffer(2 * no)
        for _ in range(200):                                      # warm caches, tables, branch predictors
            lib.hc_time_pubkey_calc(ctypes.c_size_t(l), classes["random"]().to_bytes(no, "little"), pub)
        for _ in range(samples):
            for n in rnd.sample(names, len(names)):               # interleaved, order shuffled every round
                t[n].append(lib.hc_time_pubkey_calc(ctypes.c_size_t(l), classes[n]().to_bytes(no, "little"), pub))
        stats = {}
        for n in names:
            v = sorted(t[n])[: int(len(t[n]) * 0.9)]              # crop the slow tail (interrupts, migrations)
            mean = sum(v) / len(v)
            var = sum((x - mean) ** 2 for x in v) / (len(v) - 1)
            stats[n] = (mean, var, len(v))
        base = stats["random"]
        print(f"l = {l}: cycles per k G (rdtsc, {samples} samples per class, slowest 10 % cropped)")
        for n in names:
            mean, var, cnt = stats[n]
            tval = (mean - base[0]) / math.sqrt(var / cnt + base[1] / base[2]) if n != "random" else 0.0
            rel = mean / base[0] - 1.0
            worst = max(worst, abs(rel))
            print(f"    {n:18s} mean {mean:10.0f}  vs random {100 * rel:+6.2f} %   Welch t {tval:+7.2f}")
    print(f"largest class-mean deviation from the random class: {100 * worst:.2f} %")
    return worst


if __name__ == "__main__":
    n = int(sys.argv[sys.argv.index("--samples") + 1]) if "--samples" in sys.argv else 2000
    if "--dynamic-only" not in sys.argv:
        static_audit()
    dynamic_audit(n)
