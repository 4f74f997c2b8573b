"""Model of Renes-Costello-Batina (EUROCRYPT 2016) algorithm 4: complete addition of two PROJECTIVE points on
y^2 = x^3 - 3x + b (homogeneous coordinates, O = (0 : 1 : 0)), valid for every pair of inputs when the group has odd order.
Checked against textbook affine arithmetic on the three standard bign curves and a small curve where every pair of points
is tried, and in the schedule the device uses it in (bee2_amd/csrc/bign_sign_kernels.hip, the cooperative k G: one 4-bit
window per lane, butterfly sum over the wavefront) -- before that kernel was written.  python tools/model_rcb_a3_full.py"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests")]
import orc_generic as OG


def add(P, Q, b, p):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    t0 = X1 * X2 % p; t1 = Y1 * Y2 % p; t2 = Z1 * Z2 % p
    t3 = (X1 + Y1) % p; t4 = (X2 + Y2) % p; t3 = t3 * t4 % p
    t4 = (t0 + t1) % p; t3 = (t3 - t4) % p; t4 = (Y1 + Z1) % p
    X3 = (Y2 + Z2) % p; t4 = t4 * X3 % p; X3 = (t1 + t2) % p
    t4 = (t4 - X3) % p; X3 = (X1 + Z1) % p; Y3 = (X2 + Z2) % p
    X3 = X3 * Y3 % p; Y3 = (t0 + t2) % p; Y3 = (X3 - Y3) % p
    Z3 = b * t2 % p; X3 = (Y3 - Z3) % p; Z3 = (X3 + X3) % p
    X3 = (X3 + Z3) % p; Z3 = (t1 - X3) % p; X3 = (t1 + X3) % p
    Y3 = b * Y3 % p; t1 = (t2 + t2) % p; t2 = (t1 + t2) % p
    Y3 = (Y3 - t2) % p; Y3 = (Y3 - t0) % p; t1 = (Y3 + Y3) % p
    Y3 = (t1 + Y3) % p; t1 = (t0 + t0) % p; t0 = (t1 + t0) % p
    t0 = (t0 - t2) % p; t1 = t4 * Y3 % p; t2 = t0 * Y3 % p
    Y3 = X3 * Z3 % p; Y3 = (Y3 + t2) % p; X3 = t3 * X3 % p
    X3 = (X3 - t1) % p; Z3 = t4 * Z3 % p; t1 = t3 * t0 % p
    Z3 = (Z3 + t1) % p
    return X3, Y3, Z3


def affine(P, p):
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, p - 2, p)
    return X * zi % p, Y * zi % p


def proj(A):
    return (0, 1, 0) if A is None else (A[0], A[1], 1)


def curve(l):
    """(p, b, q, yG) of bign-curve{2l}v1 from the generated constants the device code is built with"""
    import re
    txt = open(os.path.join(ROOT, "bee2_amd", "csrc", "bign_curves.inc")).read()
    def limbs(name):
        body = re.search(r"#define BIGN%d_%s_LIMBS \{([^}]*)\}" % (l, name), txt).group(1)
        return sum(int(t.strip().rstrip("u"), 16) << (32 * i) for i, t in enumerate(body.split(",")))
    c = int(re.search(r"#define BIGN%d_CRANDALL_C (\d+)u" % l, txt).group(1))
    return (1 << (2 * l)) - c, limbs("B"), limbs("Q"), limbs("YG")


def small_curve():
    """an odd-order curve y^2 = x^3 - 3x + b over a small prime, all of its points"""
    for p in (1009, 1013, 1019, 1021):
        for b in range(1, p):
            pts = [None] + [(x, y) for x in range(p) for y in range(p) if (y * y - (x * x * x - 3 * x + b)) % p == 0]
            if len(pts) % 2 == 1 and (4 * (-3) ** 3 + 27 * b * b) % p:
                return p, b, pts
    raise SystemExit("no odd-order curve found")


def main():
    rnd = random.Random(7)
    # every pair of points of a small odd-order curve, with random projective scalings
    p, b, pts = small_curve()
    sub = pts[:1] + rnd.sample(pts[1:], 120)
    for A in sub:
        for B in sub:
            la, lb = rnd.randrange(1, p), rnd.randrange(1, p)
            PA = tuple(c * la % p for c in proj(A)); PB = tuple(c * lb % p for c in proj(B))
            assert affine(add(PA, PB, b, p), p) == OG._add(A, B, p - 3, p), (A, B)
    print(f"small curve p = {p}, b = {b}, order {len(pts)}: {len(sub) ** 2} pairs incl. O, P = Q, P = -Q: ok")

    # the standard curves: random multiples, doubling, inverse, O; then the device schedule
    for l in (128, 192, 256):
        p, b, q, yG = curve(l)
        Gp = (0, yG)
        mul = lambda k: OG.mul(k, Gp, p - 3, p)
        for _ in range(6):
            k1, k2 = rnd.randrange(1, q), rnd.randrange(1, q)
            A, B = mul(k1), mul(k2)
            for (U, V) in ((A, B), (A, A), (A, (A[0], (p - A[1]) % p)), (A, None), (None, B), (None, None)):
                assert affine(add(proj(U), proj(V), b, p), p) == OG._add(U, V, p - 3, p)
        # one window per lane: lane w holds digit_w * 16^w G (O for digit 0), then a butterfly of complete additions
        W = 2 * l // 4
        base = [Gp]
        for w in range(1, W):
            t = base[-1]
            for _ in range(4):
                t = OG._add(t, t, p - 3, p)
            base.append(t)
        for k in (rnd.randrange(1, q), q - 1, 1, 16, (1 << (2 * l)) - 1, q):
            lanes = []
            for w in range(W):
                d = (k >> (4 * w)) & 15
                lanes.append(proj(OG.mul(d, base[w], p - 3, p) if d else None))
            acc = [(0, 1, 0)] * 64
            for w in range(W):                              # lanes take windows w, w + 64, ... one after the other
                acc[w % 64] = add(acc[w % 64], lanes[w], b, p)
            s = 1
            while s < 64:
                acc = [add(acc[i], acc[i ^ s], b, p) for i in range(64)]
                s *= 2
            want = mul(k % q) if k % q else None
            assert all(affine(a, p) == want for a in acc), (l, hex(k))
        print(f"l = {l}: pairs and the cooperative schedule (6 scalars incl. q - 1, 2^(2l) - 1, q): ok")


if __name__ == "__main__":
    main()
