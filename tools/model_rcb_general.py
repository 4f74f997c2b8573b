"""Model of the complete addition of Renes-Costello-Batina (EUROCRYPT 2016), algorithm 1: homogeneous projective
coordinates, y^2 = x^3 + a x + b with ANY a, valid for every pair of inputs (O = (0 : 1 : 0), P = Q, P = -Q) when the
group has odd order.  Checked here against textbook affine arithmetic (tests/orc_generic.py) on random curves before
it is written for the device (bee2_amd/csrc/bign_generic_kernels.hip gp_add_complete).  python tools/model_rcb_general.py"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests")]
import orc_generic as OG


def add(P, Q, a, b3, p):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    t0 = X1 * X2 % p; t1 = Y1 * Y2 % p; t2 = Z1 * Z2 % p
    t3 = (X1 + Y1) % p; t4 = (X2 + Y2) % p; t3 = t3 * t4 % p
    t4 = (t0 + t1) % p; t3 = (t3 - t4) % p; t4 = (X1 + Z1) % p
    t5 = (X2 + Z2) % p; t4 = t4 * t5 % p; t5 = (t0 + t2) % p
    t4 = (t4 - t5) % p; t5 = (Y1 + Z1) % p; X3 = (Y2 + Z2) % p
    t5 = t5 * X3 % p; X3 = (t1 + t2) % p; t5 = (t5 - X3) % p
    Z3 = a * t4 % p; X3 = b3 * t2 % p; Z3 = (X3 + Z3) % p
    X3 = (t1 - Z3) % p; Z3 = (t1 + Z3) % p; Y3 = X3 * Z3 % p
    t1 = (t0 + t0) % p; t1 = (t1 + t0) % p; t2 = a * t2 % p
    t4 = b3 * t4 % p; t1 = (t1 + t2) % p; t2 = (t0 - t2) % p
    t2 = a * t2 % p; t4 = (t4 + t2) % p; t0 = t1 * t4 % p
    Y3 = (Y3 + t0) % p; t0 = t5 * t4 % p; X3 = t3 * X3 % p
    X3 = (X3 - t0) % p; t0 = t3 * t1 % p; Z3 = t5 * Z3 % p
    Z3 = (Z3 + t0) % p
    return X3, Y3, Z3


def affine(P, p):
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, p - 2, p)
    return X * zi % p, Y * zi % p


def ladder(k, G, a, b3, p, bits):
    """the device schedule: double-and-add-always from O, selection by the scalar bit"""
    T = (0, 1, 0)
    for i in reversed(range(bits)):
        T = add(T, T, a, b3, p)
        U = add(T, G, a, b3, p)
        if (k >> i) & 1:
            T = U
    return T


def main():
    rnd = random.Random(5)
    # the 256-bit bign prime and an isomorphic image of the standard curve (a != -3), plus small primes with every point
    p = 2 ** 256 - 189
    b = 0x77CE6C1515F3A8EDD2C13AABE4D8FBBE4CF55069978B9253B22E7D6BD69C03F1
    yG = 0x6BF7FC3CFB16D69F5CE4C9A351D6835D78913966C408F6521E29CF1804516A93
    u = rnd.randrange(2, p)
    a2, b2, y2 = (p - 3) * pow(u, 4, p) % p, b * pow(u, 6, p) % p, yG * pow(u, 3, p) % p
    for (aa, bb, G) in ((p - 3, b, (0, yG)), (a2, b2, (0, y2))):
        assert (G[1] * G[1] - bb) % p == 0
        for _ in range(20):
            k = rnd.getrandbits(256)
            got = affine(ladder(k, (G[0], G[1], 1), aa, 3 * bb % p, p, 256), p)
            assert got == OG.mul(k, G, aa, p), "ladder"
        # the exceptional pairs
        P1 = OG.mul(12345, G, aa, p)
        for (A, B) in ((None, P1), (P1, None), (P1, P1), (P1, (P1[0], (-P1[1]) % p)), (None, None)):
            pa = (0, 1, 0) if A is None else (A[0] * 7 % p, A[1] * 7 % p, 7)
            pb = (0, 1, 0) if B is None else (B[0] * 11 % p, B[1] * 11 % p, 11)
            assert affine(add(pa, pb, aa, 3 * bb % p, p), p) == OG._add(A, B, aa, p), (A is None, B is None)
    print("RCB algorithm 1 (general a): ladder and exceptional cases agree with affine arithmetic")


if __name__ == "__main__":
    main()
