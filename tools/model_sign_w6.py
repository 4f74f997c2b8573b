"""Model of the one-lane k G of the signing side (bee2_amd/csrc/bign_sign_kernels.hip mul_base_ct6<N, true>): signed 6-bit
windows recoded low to high with a carry, table entries |d| 64^w G, accumulator in Jacobian coordinates with the INCOMPLETE
8M + 3S mixed addition.  Checks, on the three standard curves and for random and adversarial scalars:
  * the digits rebuild k, |d| <= 32, the last window is non-negative and no carry leaves it;
  * the schedule argument of jac_madd_ct: before window w the accumulator is A G with 0 < |A| < |d_w| 64^w, and
    A = +-d_w 64^w (mod q) happens in the last window only and only for k = 0 (mod q);
  * the incomplete formula, run exactly as the kernel runs it (masks replaced by ifs), returns k G -- and Z = 0 for k = q.
python tools/model_sign_w6.py"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import orc_generic as OG
from model_rcb_a3_full import curve


def recode(k, nbits):
    W = (nbits + 1 + 5) // 6
    out, carry = [], 0
    for w in range(W):
        t = (k & 63) + carry
        k >>= 6
        carry = (t + 32) >> 6
        out.append(t - 64 * carry)
    assert carry == 0 and k == 0
    return out


def madd(T, E, p):
    X1, Y1, Z1 = T
    x2, y2 = E
    Z1Z1 = Z1 * Z1 % p; U2 = x2 * Z1Z1 % p; S2 = y2 * Z1 * Z1Z1 % p
    H = (U2 - X1) % p; r = (S2 - Y1) % p
    HH = H * H % p; HHH = H * HH % p; V = X1 * HH % p
    X3 = (r * r - HHH - 2 * V) % p
    return X3, (r * (V - X3) - Y1 * HHH) % p, Z1 * H % p


def mul_base(k, l, tab_cache={}):
    p, b, q, yG = curve(l)
    nbits = 2 * l
    ds = recode(k, nbits)
    assert sum(d << (6 * i) for i, d in enumerate(ds)) == k and all(-32 <= d <= 32 for d in ds) and ds[-1] >= 0
    if l not in tab_cache:
        base, t = [], (0, yG)
        for w in range(len(ds)):
            base.append(t)
            for _ in range(6):
                t = OG._add(t, t, p - 3, p)
        tab_cache[l] = base
    base = tab_cache[l]
    J, at_inf, A = (0, 1, 0), True, 0
    for w, d in enumerate(ds):
        if d == 0:
            continue
        T = d << (6 * w)
        if not at_inf:
            assert 0 < abs(A) < abs(T)
            coll = (A - T) % q == 0 or (A + T) % q == 0
            assert not coll or (w == len(ds) - 1 and k % q == 0), (hex(k), w)
        E = OG.mul(abs(d), base[w], p - 3, p)
        if d < 0:
            E = (E[0], (p - E[1]) % p)
        J = (E[0], E[1], 1) if at_inf else madd(J, E, p)
        at_inf = False
        A += T
    if at_inf or J[2] == 0:
        return None
    zi = pow(J[2], p - 2, p)
    return J[0] * zi * zi % p, J[1] * zi * zi * zi % p


def main():
    rnd = random.Random(11)
    for l in (128, 192, 256):
        p, b, q, yG = curve(l)
        G = (0, yG)
        nb = 2 * l
        rep = lambda pat: sum(v << (6 * i) for i, v in enumerate((pat * (nb // 6 + 2))[: nb // 6 + 1])) % (1 << nb)
        ks = [1, 2, 31, 32, 33, 63, 64, q - 1, q - 2, q, q + 1, (1 << nb) - 1, 1 << (nb - 1), rep([32]), rep([31]), rep([33]),
              rep([63]), rep([31, 32]), rep([32, 63, 0]), rep([0, 0, 32]), 0]
        ks += [rnd.randrange(1 << nb) for _ in range(12)] + [rnd.randrange(1, 1 << 40) for _ in range(4)]
        ks += [rnd.randrange(q, 1 << nb) for _ in range(6)]              # key generation multiplies any d below 2^(2l)
        for k in ks:
            want = OG.mul(k % q, G, p - 3, p) if k % q else None
            assert mul_base(k, l) == want, (l, hex(k))
        print(f"l = {l}: {len(ks)} scalars incl. 0, q, q +- 1, 2^(2l) - 1 and the carry-chain patterns: digits, schedule and k G ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
