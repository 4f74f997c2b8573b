"""Interval check of the carry-free 29-bit-limb arithmetic (bee2_amd/csrc/bign_fe29.hpp, bign_quad29.hpp).

Every limb is tracked as an interval [lo, hi]; the script replays f29_mul / f29_sqr / f29_fold / f29_carry and the
limb-wise additions exactly as the device code orders them, for the point formulas jac29_dbl, jac29_madd, quad29_dbl
and quad29_add, starting from the documented input contracts (N / L1), and asserts
  * every 64-bit column accumulator stays inside int64,
  * every 32-bit intermediate stays inside int32,
  * the outputs satisfy the contracts the next operation assumes (a fixed point: the outputs are fed back in).
Run: python tools/fe29_bounds.py   (pure Python, no GPU; also run by tests/test_fe29_bounds.py)"""
I64 = 1 << 63
I32 = 1 << 31
LAYOUT = {8: (9, 29, 189 << 5), 12: (14, 28, 317 << 8), 16: (19, 27, 569 << 1)}      # LZ<N>: L, B, FOLD
L, B, U, M, FOLD = 9, 29, 1 << 29, (1 << 29) - 1, 189 << 5


def configure(n):
    """limb layout of the curve with N = n 32-bit words (bign_fe29.hpp, LZ<N>)"""
    global L, B, U, M, FOLD
    L, B, FOLD = LAYOUT[n]
    U = 1 << B
    M = U - 1


class Fe:
    """nine limb intervals"""

    def __init__(self, lo, hi):
        self.lo, self.hi = list(lo), list(hi)
        assert len(self.lo) == L and all(a <= b for a, b in zip(self.lo, self.hi))
        assert all(-I32 <= a and b < I32 for a, b in zip(self.lo, self.hi)), "int32 overflow"

    def __repr__(self):
        return " ".join(f"[{a / U:+.4f},{b / U:+.4f}]" for a, b in zip(self.lo, self.hi))

    def within(self, other):
        return all(a >= c and b <= d for a, b, c, d in zip(self.lo, self.hi, other.lo, other.hi))


def lazy1():
    """L1: difference of two N values / negated N value"""
    n = norm()
    return sub(n, n)


def one():
    return Fe([1] + [0] * (L - 1), [1] + [0] * (L - 1))


def add(a, b):
    return Fe([x + y for x, y in zip(a.lo, b.lo)], [x + y for x, y in zip(a.hi, b.hi)])


def sub(a, b):
    return Fe([x - y for x, y in zip(a.lo, b.hi)], [x - y for x, y in zip(a.hi, b.lo)])


def neg(a):
    return Fe([-x for x in a.hi], [-x for x in a.lo])


def union(a, b):
    return Fe([min(x, y) for x, y in zip(a.lo, b.lo)], [max(x, y) for x, y in zip(a.hi, b.hi)])


def _prod(alo, ahi, blo, bhi):
    c = (alo * blo, alo * bhi, ahi * blo, ahi * bhi)
    return min(c), max(c)


def norm(slack0=1 << 20, slack1=1 << 21):
    """N: l[2..] in [0, u); l[0] within slack0 of [0, u) (f29_carry folds its last carry there), l[1] within slack1 (a
    multiplication folds the carry out of column L-1 there)"""
    return Fe([-slack0, -slack1] + [0] * (L - 2), [M + slack0, M + slack1] + [M] * (L - 2))


def _column(a, b, k, square, acc_lo, acc_hi, what):
    for i in range(max(0, k - (L - 1)), min(k, L - 1) + 1):
        j = k - i
        if square and i > j:
            continue
        if square and i < j:
            assert -I32 <= 2 * a.lo[j] and 2 * a.hi[j] < I32, f"{what}: a doubled limb leaves int32"
            p_lo, p_hi = _prod(a.lo[i], a.hi[i], 2 * a.lo[j], 2 * a.hi[j])
        elif square:
            p_hi = max(a.lo[i] ** 2, a.hi[i] ** 2)          # a_i^2 >= 0
            p_lo = 0 if a.lo[i] <= 0 <= a.hi[i] else min(a.lo[i] ** 2, a.hi[i] ** 2)
        else:
            p_lo, p_hi = _prod(a.lo[i], a.hi[i], b.lo[j], b.hi[j])
        acc_lo += p_lo
        acc_hi += p_hi
        assert -I64 <= acc_lo and acc_hi < I64, f"{what}: column {k} leaves int64 ({acc_lo / I64:.3f}, {acc_hi / I64:.3f})"
    return acc_lo, acc_hi


def mul(a, b, K=1, what="mul", square=False):
    """f29_mul2 / bign_fe29_asm.inc, in their order: phase A (columns L .. 2L-2 on a chain of their own), phase B (columns
    0 .. L-1 with h[k] FOLD), the 34-bit carry out of column L-1 folded into limbs 0 and 1, then the scaling chain (K != 1)"""
    acc_lo = acc_hi = 0
    h_lo, h_hi = [], []
    for k in range(L, 2 * L - 1):
        acc_lo, acc_hi = _column(a, b, k, square, acc_lo, acc_hi, what)
        h_lo.append(0)
        h_hi.append(M)
        acc_lo >>= B
        acc_hi >>= B
    assert -I32 <= acc_lo and acc_hi < I32, f"{what}: the top of the high half leaves int32"
    h_lo.append(acc_lo)
    h_hi.append(acc_hi)
    lo, hi = [0] * L, [M] * L
    acc_lo = acc_hi = 0
    for k in range(L):
        acc_lo, acc_hi = _column(a, b, k, square, acc_lo, acc_hi, what)
        acc_lo += min(h_lo[k] * FOLD, h_hi[k] * FOLD)
        acc_hi += max(h_lo[k] * FOLD, h_hi[k] * FOLD)
        assert -I64 <= acc_lo and acc_hi < I64, f"{what}: low column {k} with its fold term leaves int64"
        acc_lo >>= B
        acc_hi >>= B
    # c = cl + 2^B ch: cl in [0, u), ch = c >> B must be a 24-bit multiplicand (v_mul_i32_i24) and FOLD too
    ch_lo, ch_hi = acc_lo >> B, acc_hi >> B
    assert -(1 << 23) <= ch_lo and ch_hi < (1 << 23) and FOLD < (1 << 23), f"{what}: ch / FOLD leave 24 bits"
    t_hi = M * FOLD                                          # cl FOLD, int64 in the device code
    r0_hi = M + M                                            # r[0] + low B bits: below 2^31
    assert r0_hi < I32
    l1_lo = 0 + 0 + ch_lo * FOLD + 0
    l1_hi = M + (t_hi >> B) + ch_hi * FOLD + 1
    assert -I32 <= l1_lo and l1_hi < I32, what
    lo[1], hi[1] = l1_lo, l1_hi
    r = Fe(lo, hi)
    if K != 1:
        r = scale(r, K, what)
    return r


def scale(a, K, what="scale"):
    """f29_scale / the _k tail of the asm blocks: t = r[k] K + cy, the carry out of the top limb times FOLD into r[0], one carry step"""
    lo, hi = [0] * L, [M] * L
    c_lo = c_hi = 0
    for i in range(L):
        t_lo, t_hi = a.lo[i] * K + c_lo, a.hi[i] * K + c_hi
        assert -I64 <= t_lo and t_hi < I64
        c_lo, c_hi = t_lo >> B, t_hi >> B
    assert -(1 << 23) <= c_lo and c_hi < (1 << 23), f"{what}: the scaling carry leaves 24 bits"
    r0_lo, r0_hi = c_lo * FOLD, M + c_hi * FOLD
    assert -I32 <= r0_lo and r0_hi < I32, what
    lo[1], hi[1] = r0_lo >> B, M + (r0_hi >> B)
    return Fe(lo, hi)


def sqr(a, K=1, what="sqr"):
    return mul(a, a, K, what, square=True)


def carry(a, what="carry"):
    lo, hi = [0] * L, [0] * L
    c_lo = c_hi = 0
    for i in range(L):
        t_lo, t_hi = a.lo[i] + c_lo, a.hi[i] + c_hi
        assert -I32 <= t_lo and t_hi < I32, what
        lo[i], hi[i] = 0, M
        c_lo, c_hi = t_lo >> B, t_hi >> B
    lo[0], hi[0] = c_lo * FOLD, M + c_hi * FOLD
    return Fe(lo, hi)                                       # (l[0] within |c| FOLD of [0, u): checked against the contracts by the callers)


def to_words_ok(a, what):
    """f29_to_words accepts |l| < 4u"""
    assert all(-4 * U < x and y < 4 * U for x, y in zip(a.lo, a.hi)), what


# ---------------------------------------------------------------- one lane per point (bign_fe29.hpp)
def jac29_dbl(X, Y, Z):
    delta = sqr(Z, what="dbl Z^2")
    gamma = sqr(Y, what="dbl Y^2")
    beta4 = mul(X, gamma, 4, "dbl 4 X g")
    alpha = mul(sub(X, delta), add(X, delta), 3, "dbl alpha")
    Z3 = mul(Y, Z, 2, "dbl 2 Y Z")
    X3 = carry(sub(sqr(alpha, what="dbl alpha^2"), add(beta4, beta4)), "dbl X3 carry")
    t1 = sqr(gamma, 8, "dbl 8 g^2")
    Y3 = sub(mul(alpha, sub(beta4, X3), 1, "dbl alpha (4b - X3)"), t1)
    return X3, Y3, Z3


def jac29_madd(X, Y, Z, ex, ey):
    Z1Z1 = sqr(Z, what="madd Z^2")
    U2 = mul(ex, Z1Z1, 1, "madd U2")
    S2 = mul(ey, mul(Z, Z1Z1, 1, "madd Z^3"), 1, "madd S2")
    H = sub(U2, X)
    r = carry(sub(S2, Y), "madd r carry")
    HH = sqr(H, what="madd H^2")
    HHH = mul(H, HH, 1, "madd H^3")
    V = mul(X, HH, 1, "madd V")
    Z3 = mul(Z, H, 1, "madd Z3")
    X3 = carry(sub(sub(sub(sqr(r, what="madd r^2"), HHH), V), V), "madd X3 carry")
    Y3 = sub(mul(r, sub(V, X3), 1, "madd r (V - X3)"), mul(Y, HHH, 1, "madd Y1 H^3"))
    return X3, Y3, Z3


# ---------------------------------------------------------------- one point per quad (bign_quad29.hpp)
def quad29_dbl(X, Y, Z, D):
    gamma = sqr(Y, 1, "qdbl Y^2")
    Z3 = mul(Y, Z, 2, "qdbl 2 Y Z")
    alpha = sub(sqr(X, 3, "qdbl 3 X^2"), sqr(D, 3, "qdbl 3 D^2"))
    b4 = mul(X, gamma, 4, "qdbl 4 X g")
    b8 = mul(X, gamma, 8, "qdbl 8 X g")
    D3 = sqr(Z3, 1, "qdbl Z3^2")
    X3 = sub(sqr(alpha, 1, "qdbl alpha^2"), b8)
    t = sub(b4, X3)
    Y3 = sub(mul(alpha, t, 1, "qdbl alpha (4b - X3)"), sqr(gamma, 8, "qdbl 8 g^2"))
    return X3, Y3, Z3, D3


def quad29_add(X, Y, Z, D, ex, ey, ez, ezz):
    U1 = mul(X, ezz, 1, "qadd U1")
    U2 = mul(ex, D, 1, "qadd U2")
    t = mul(Z, D, 1, "qadd Z^3")
    w = mul(Y, ezz, 1, "qadd Y1 ZZ2")
    H = sub(U2, U1)
    S2 = mul(ey, t, 1, "qadd S2")
    HH = sqr(H, 1, "qadd H^2")
    ZZ = mul(Z, ez, 1, "qadd Z1 Z2")
    S1 = mul(w, ez, 1, "qadd S1")
    r = sub(S2, S1)
    H3 = mul(H, HH, 1, "qadd H^3")
    V = mul(U1, HH, 1, "qadd V")
    Z3 = mul(ZZ, H, 1, "qadd Z3")
    X3 = carry(sub(sub(sub(sqr(r, 1, "qadd r^2"), H3), V), V), "qadd X3 carry")
    Y3 = sub(mul(r, sub(V, X3), 1, "qadd r (V - X3)"), mul(S1, H3, 1, "qadd S1 H^3"))
    D3 = sqr(Z3, 1, "qadd Z3^2")
    return X3, Y3, Z3, D3


def pair29_dbl(X, Y, Z, D):
    gamma = sqr(Y, 1, "pdbl Y^2")
    Z3 = mul(Y, Z, 2, "pdbl 2 Y Z")
    alpha = sub(sqr(X, 3, "pdbl 3 X^2"), sqr(D, 3, "pdbl 3 D^2"))
    b4 = mul(X, gamma, 4, "pdbl 4 X g")
    b8 = mul(X, gamma, 8, "pdbl 8 X g")
    X3 = sub(sqr(alpha, 1, "pdbl alpha^2"), b8)
    D3 = sqr(Z3, 1, "pdbl Z3^2")
    Y3 = sub(mul(alpha, sub(b4, X3), 1, "pdbl alpha (4b - X3)"), mul(gamma, gamma, 8, "pdbl 8 g^2"))
    return X3, Y3, Z3, D3


def main():
    for n in (8, 12, 16):
        configure(n)
        N, L1 = norm(), lazy1()
        # one lane per point: contract X, Z: N; Y: L1; table / comb entries x: N, y: N or its negation (L1)
        X3, Y3, Z3 = jac29_dbl(N, L1, N)
        assert X3.within(N) and Z3.within(N) and Y3.within(L1), "jac29_dbl breaks its own contract"
        X3, Y3, Z3 = jac29_madd(N, L1, N, N, L1)
        assert X3.within(N) and Z3.within(N) and Y3.within(L1), "jac29_madd breaks its own contract"
        for v in (X3, Y3, Z3):
            to_words_ok(v, "to_words")
        # quads and pairs: contract X, Y: L1 (X: N after an addition, which lies within L1); Z, D: N; entries X, Y: L1, Z, ZZ: N
        for dbl in (quad29_dbl, pair29_dbl):
            X3, Y3, Z3, D3 = dbl(L1, L1, N, N)
            assert X3.within(L1) and Y3.within(L1) and Z3.within(N) and D3.within(N), "the doubling breaks its own contract"
        X3, Y3, Z3, D3 = quad29_add(L1, L1, N, N, L1, L1, N, N)          # pair29_add: the same products in another order
        assert X3.within(L1) and Y3.within(L1) and Z3.within(N) and D3.within(N), "quad29_add breaks its own contract"
        for v in (X3, Y3, Z3):
            to_words_ok(v, "to_words")
        if n == 8:
            # the debug ops of tests/test_gpu_bign.py (op 26: (a - 3b) carried, times -b scaled by 4; op 28: 2 (a-b)(a+b) - 3a)
            mul(carry(sub(sub(sub(N, N), N), N)), neg(N), 4, "debug op 26")
            to_words_ok(sub(sub(sub(mul(sub(N, N), add(N, N), 2, "debug op 28"), N), N), N), "debug op 28")
        print(f"N = {n}: {L} limbs of {B} bits, fold {FOLD}: every accumulator inside int64, "
              "the contracts (X, Z: N; Y: L1 / X, Y: L1; Z, D: N) are fixed points of all six formulas")
    configure(8)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
