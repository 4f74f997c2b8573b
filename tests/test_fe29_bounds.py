"""CPU: the interval proof that the carry-free 29-bit-limb formulas of bign_fe29.hpp / bign_quad29.hpp never
overflow their 64-bit column accumulators and keep their own input contracts (tools/fe29_bounds.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fe29_bounds  # noqa: E402


def test_fe29_formulas_stay_inside_their_bounds():
    assert fe29_bounds.main() == 0


def test_fe29_bound_checker_catches_an_overflow():
    """the checker is not vacuous: a product of two sums of two normalised values (4 u^2 per term) must trip it"""
    fe29_bounds.configure(8)                        # the tightest layout: 9 x 29 bits
    n = fe29_bounds.norm()
    s = fe29_bounds.add(n, n)
    try:
        fe29_bounds.mul(s, fe29_bounds.add(s, n), 1, "3 x 2")
    except AssertionError:
        return
    raise AssertionError("3u x 2u went through")


def _mul_exact(a, b, K, L, B, FOLD, square=False):
    """bign_fe29.hpp f29_mul2 + f29_scale / bign_fe29_asm.inc on Python ints, step by step (floor shifts, masks)"""
    M = (1 << B) - 1
    I64, I32 = 1 << 63, 1 << 31
    h, acc = [], 0
    for k in range(L, 2 * L - 1):
        for i in range(k - (L - 1), L):
            acc += a[i] * b[k - i]
            assert -I64 <= acc < I64
        h.append(acc & M)
        acc >>= B
    assert -I32 <= acc < I32
    h.append(acc)
    o, acc = [], 0
    for k in range(L):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
        acc += h[k] * FOLD
        assert -I64 <= acc < I64
        o.append(acc & M)
        acc >>= B
    cl, ch = acc & M, acc >> B
    assert -(1 << 23) <= ch < (1 << 23)
    t = cl * FOLD
    r0 = o[0] + (t & M)
    o[1] += (t >> B) + ch * FOLD + (r0 >> B)
    o[0] = r0 & M
    if K != 1:
        cy = 0
        for i in range(L):
            cy += o[i] * K
            o[i] = cy & M
            cy >>= B
        r0 = o[0] + cy * FOLD
        o[1] += r0 >> B
        o[0] = r0 & M
    assert all(-I32 <= x < I32 for x in o)
    return o


def test_two_chain_multiplication_is_exact_on_python_ints():
    """the arithmetic of round 6's multiplication (high half on its own carry chain, the fold inside the low chain, the 34-bit
    carry out of column L-1, the scaling chain) against big integers: random lazy operands and the corners of the contracts"""
    import random
    rnd = random.Random(29)
    for n, c in ((8, 189), (12, 317), (16, 569)):
        L, B, FOLD = fe29_bounds.LAYOUT[n]
        p, u = (1 << (32 * n)) - c, 1 << B
        assert pow(2, B * L, p) == FOLD
        val = lambda l: sum(x << (B * i) for i, x in enumerate(l))  # noqa: E731
        corners = [[u + (1 << 16)] * L, [-(u + (1 << 16))] * L, [u - 1] * L, [0] * L, [1] + [0] * (L - 1)]
        for it in range(400):
            if it < 25:
                a, b = corners[it // 5], corners[it % 5]
            else:
                a = [rnd.randrange(-(u + (1 << 16)), u + (1 << 16) + 1) for _ in range(L)]
                b = [rnd.randrange(-(u + (1 << 16)), u + (1 << 16) + 1) for _ in range(L)]
            for K in (1, 2, 3, 4, 8):
                r = _mul_exact(a, b, K, L, B, FOLD)
                assert (val(r) - K * val(a) * val(b)) % p == 0, (n, it, K)
                assert all(0 <= x < u for i, x in enumerate(r) if i != 1) and -(1 << 21) <= r[1] < u + (1 << 21), (n, it, K, r)


def test_generated_asm_multiplication_is_current():
    """bee2_amd/csrc/bign_fe29_asm.inc is what tools/gen_f29_asm.py writes today"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.call([sys.executable, os.path.join(root, "tools", "gen_f29_asm.py"), "--check"]) == 0
