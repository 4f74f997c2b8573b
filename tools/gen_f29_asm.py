#!/usr/bin/env python3
"""Generate bee2_amd/csrc/bign_fe29_asm.inc: GF(2^256 - 189) multiplication / squaring on nine signed 29-bit limbs as ONE
inline-asm block each (gfx950), for bign_fe29.hpp.

Why generated assembly (round 6, profiles/r06_f29_asm_ab.txt): the C++ form of the same arithmetic compiles to 190-217
instructions per multiplication -- LLVM's reassociation sums a column's products first and adds the incoming carry last (one
v_lshl_add_u64 per column), splits columns into partial chains that start from a zero addend, and moves 64-bit accumulators
about (16-26 v_mov) -- and single v_mad statements in inline asm make its hazard recogniser put an s_nop behind every one
(inline asm is treated as a possible SDWA producer).  One block per multiplication is exactly the instructions the
arithmetic needs, in a fixed order:

  phase A  columns L .. 2L-2 of the product on their own carry chain (accumulator A, from zero) -> limbs h[0 .. L-2] of the
           high half, masked, and h[L-1] = what is left (signed, < 2^31); they are parked in the OUTPUT registers;
  phase B  columns 0 .. L-1: acc = carry + sum a[i] b[k-i] + h[k] FOLD  (2^(B L) = FOLD mod p) -> r[k] = acc & M, carry = acc >> B
           -- the fold costs one multiply-add per column and no pass of its own;
  final    the carry out of column L-1 (34 bits, weight 2^(B L) again) times FOLD into r[0], r[1];
  scale    (the _k forms) r <- K r for a per-lane K in {1, 2, 3, 4, 8}: a second carry chain t = r[k] K + cy.

Instruction count (L = 9): multiplication 91 v_mad_i64_i32 + 18 v_and + 18 v_ashrrev_i64 + 9 = 136; squaring 55 + 45 = 100;
scaling + 31.  The two phases are interleaved instruction by instruction (phase A runs two columns ahead), so a lone wavefront
never issues a multiply-add that depends on the one before it.

Accumulators are the fixed scratch registers v[2:3] and v[4:5] (clobbers of the block: their low halves must be addressable
by name, which an asm operand of 64 bits is not).  Operand numbers: outputs r[0..8] = %0..%8 (early-clobber: they hold h[]
while the inputs are still being read), a[0..8] = %9..%17, then b[0..8] (multiplication) or d[1..8] = 2 a[1..8] (squaring),
then FOLD, then K.

Bounds: tools/fe29_bounds.py replays exactly this order on intervals (every accumulator inside int64, every multiplicand
inside int32, the N / L1 contracts fixed points of the six point formulas).

usage: python tools/gen_f29_asm.py  (rewrites the .inc; tests/test_fe29_bounds.py checks that the committed file is current)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "bee2_amd", "csrc", "bign_fe29_asm.inc")
L, B = 9, 29
MASK = "0x1fffffff"
A, AL, AH = "v[2:3]", "v2", "v3"
Bq, BL, BH = "v[4:5]", "v4", "v5"


def mad(acc, x, y):
    return f"v_mad_i64_i32 {acc}, vcc, {x}, {y}, {acc}"


def gen(square, scaled):
    """-> (lines of the asm template, number of operands)"""
    r = [f"%{i}" for i in range(L)]
    a = [f"%{L + i}" for i in range(L)]
    if square:
        d = [None] + [f"%{2 * L + i - 1}" for i in range(1, L)]          # d[j] = 2 a[j], j >= 1
        nxt = 2 * L + (L - 1)
    else:
        b = [f"%{2 * L + i}" for i in range(L)]
        nxt = 3 * L
    fold = f"%{nxt}"
    kreg = f"%{nxt + 1}" if scaled else None
    nops = nxt + (2 if scaled else 1)

    def products(k):
        """the products of column k as (x, y) register names"""
        out = []
        for i in range(max(0, k - (L - 1)), min(k, L - 1) + 1):
            j = k - i
            if square:
                if i < j:
                    out.append((a[i], d[j]))
                elif i == j:
                    out.append((a[i], a[i]))
            else:
                out.append((a[i], b[j]))
        return out

    # phase A: columns L .. 2L-2 on accumulator A; h[j] parked in r[j]
    colA = []
    for k in range(L, 2 * L - 1):
        ins = [mad(A, x, y) for x, y in products(k)]
        ins.append(f"v_and_b32 {r[k - L]}, {MASK}, {AL}")
        ins.append(f"v_ashrrev_i64 {A}, {B}, {A}")
        colA.append(ins)
    # (after the last column AL holds h[L-1], signed)
    # phase B: columns 0 .. L-1 on accumulator B
    colB = []
    for k in range(L):
        ins = [mad(Bq, x, y) for x, y in products(k)]
        ins.append(mad(Bq, r[k] if k < L - 1 else AL, fold))
        ins.append(f"v_and_b32 {r[k]}, {MASK}, {BL}")
        ins.append(f"v_ashrrev_i64 {Bq}, {B}, {Bq}")
        colB.append(ins)
    # interleave: A runs two columns ahead of B (B_k needs h[k] = the result of A's column L + k)
    # (each chain's first multiply-add takes the constant 0 as its addend: no initialisation)
    colA[0][0] = colA[0][0].rsplit(", ", 1)[0] + ", 0"
    colB[0][0] = colB[0][0].rsplit(", ", 1)[0] + ", 0"
    seq = list(colA[0])
    ia, ib = 1, 0
    while ia < len(colA) or ib < len(colB):
        ca = colA[ia] if ia < len(colA) else []
        cb = colB[ib] if ib < len(colB) and (ib + 1 < ia or ia >= len(colA)) else []
        if not ca and not cb:                               # (cannot happen: B is always at least one column behind)
            raise AssertionError
        n = max(len(ca), len(cb))
        for t in range(n):                                  # round-robin, the longer column's tail on its own
            if t < len(ca):
                seq.append(ca[t])
            if t < len(cb):
                seq.append(cb[t])
        if ca:
            ia += 1
        if cb:
            ib += 1
    # final: c = carry out of column L-1 (accumulator B, < 2^34 in magnitude, weight 2^(B L)): c FOLD into r[0], r[1].
    # c = cl + 2^B ch; cl FOLD < 2^42 -> its low B bits to r[0] (which becomes < 2 u: carried on into r[1]), the rest and ch FOLD to r[1]
    seq += [
        f"v_and_b32 {AL}, {MASK}, {BL}",                    # cl
        f"v_ashrrev_i64 {Bq}, {B}, {Bq}",                   # ch in BL (|ch| <= 2^5)
        f"v_mad_i64_i32 {A}, vcc, {AL}, {fold}, 0",         # cl FOLD
        f"v_mul_i32_i24 {BL}, {BL}, {fold}",                # ch FOLD (24-bit operands: |ch| small, FOLD < 2^23)
        f"v_and_b32 {BH}, {MASK}, {AL}",
        f"v_ashrrev_i64 {A}, {B}, {A}",                     # (cl FOLD) >> B < 2^13
        f"v_add_u32 {r[0]}, {r[0]}, {BH}",                  # < 2 u
        f"v_add3_u32 {r[1]}, {r[1]}, {AL}, {BL}",
        f"v_lshrrev_b32 {BH}, {B}, {r[0]}",                 # 0 / 1
        f"v_and_b32 {r[0]}, {MASK}, {r[0]}",
        f"v_add_u32 {r[1]}, {r[1]}, {BH}",
    ]
    if scaled:
        # r <- K r: t = r[k] K + cy on accumulator A; the carry out of the top limb (weight 2^(B L)) times FOLD into r[0], r[1]
        for k in range(L):
            seq += [mad(A, r[k], kreg) if k else f"v_mad_i64_i32 {A}, vcc, {r[k]}, {kreg}, 0", f"v_and_b32 {r[k]}, {MASK}, {AL}", f"v_ashrrev_i64 {A}, {B}, {A}"]
        # AL = carry (< 2^5 in magnitude: r[L-1] < u, K <= 8 ... r[1] a little over u): (carry FOLD) < 2^24 straight into r[0], then one carry step
        seq += [
            f"v_mad_i32_i24 {r[0]}, {AL}, {fold}, {r[0]}",
            f"v_ashrrev_i32 {BH}, {B}, {r[0]}",
            f"v_and_b32 {r[0]}, {MASK}, {r[0]}",
            f"v_add_u32 {r[1]}, {r[1]}, {BH}",
        ]
    return seq, nops


def emit_fn(name, square, scaled):
    seq, nops = gen(square, scaled)
    outs = ", ".join(f'"=&v"(r.l[{i}])' for i in range(L))
    ins = [f'"v"(a.l[{i}])' for i in range(L)]
    if square:
        ins += [f'"v"(d[{i}])' for i in range(1, L)]
    else:
        ins += [f'"v"(b.l[{i}])' for i in range(L)]
    ins.append('"v"(fold)')
    if scaled:
        ins.append('"v"(K)')
    assert L + len(ins) == nops <= 30, (name, nops)
    args = "lzT<8> &r, const lzT<8> &a" + ("" if square else ", const lzT<8> &b") + (", int32_t K" if scaled else "")
    body = []
    body.append(f"// {len(seq)} instructions: {sum(1 for s in seq if s.startswith('v_mad_i64'))} v_mad_i64_i32")
    body.append(f"__device__ __forceinline__ void {name}({args})")
    body.append("{")
    body.append("    const int32_t fold = LZ<8>::FOLD;")
    if square:
        body.append("    int32_t d[9];")
        body.append("#pragma unroll")
        body.append("    for (int i = 1; i < 9; ++i) d[i] = a.l[i] * 2;")
    body.append("    asm(")
    for s in seq:
        body.append(f'        "{s}\\n\\t"')
    body.append(f"        : {outs}")
    body.append(f"        : {', '.join(ins)}")
    body.append('        : "vcc", "v2", "v3", "v4", "v5");')
    body.append("}")
    return "\n".join(body), len(seq)


def text():
    parts = ["// bign_fe29_asm.inc -- GENERATED by tools/gen_f29_asm.py (which explains the schedule); do not edit.",
             "// GF(2^256 - 189) on nine signed 29-bit limbs: one inline-asm block per multiplication / squaring, included by bign_fe29.hpp.",
             "// Contract: operand limbs as f29_mul documents (A * B <= 3 u^2 per column sum); result N: l[0], l[2..8] in [0, u), l[1] in [0, u + 2^18).",
             ""]
    for name, sq, sc in (("f29_mul9_asm", False, False), ("f29_mul9k_asm", False, True), ("f29_sqr9_asm", True, False), ("f29_sqr9k_asm", True, True)):
        t, _ = emit_fn(name, sq, sc)
        parts += [t, ""]
    return "\n".join(parts)


if __name__ == "__main__":
    t = text()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == t else 1)
    open(OUT, "w").write(t)
    for name, sq, sc in (("mul", False, False), ("mul_k", False, True), ("sqr", True, False), ("sqr_k", True, True)):
        print(name, len(gen(sq, sc)[0]), "instructions,", gen(sq, sc)[1], "operands")
