"""CPU: the operation sequences of the complete additions the signing kernels use (Renes-Costello-Batina algorithm 1 for
any a, bign_generic_kernels.hip gp_add_complete; algorithm 4 for a = -3 on projective operands, bign_sign_kernels.hip
proj_add_complete) against textbook affine arithmetic, including the one-window-per-lane + butterfly schedule of
bign_mulbase_coop_kernel (tools/model_rcb_general.py, tools/model_rcb_a3_full.py)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import model_rcb_a3_full as A3  # noqa: E402
import model_rcb_general as GEN  # noqa: E402
import model_sign_w6 as W6  # noqa: E402
import orc_generic as OG  # noqa: E402


def test_rcb_algorithm_4_and_the_wavefront_schedule():
    A3.main()


def test_rcb_algorithm_1_general_a():
    GEN.main()


def test_the_models_are_not_vacuous():
    """a wrong constant must be caught: algorithm 4 with b + 1, on a small curve"""
    p, b, pts = A3.small_curve()
    rnd = random.Random(1)
    bad = 0
    for _ in range(50):
        A, B = rnd.choice(pts[1:]), rnd.choice(pts[1:])
        got = A3.affine(A3.add(A3.proj(A), A3.proj(B), (b + 1) % p, p), p)
        bad += got != OG._add(A, B, p - 3, p)
    assert bad > 40


def test_signed_window_schedule_keeps_the_incomplete_addition_safe():
    """mul_base_ct6<N, true>: digits rebuild k, the accumulator never meets +-(the point it is about to add) except for
    k = 0 (mod q) in the last window, and the formula as the kernel runs it returns k G (tools/model_sign_w6.py)"""
    assert W6.main() == 0


def test_signed_window_model_catches_a_collision():
    """not vacuous: adding a point to itself through the incomplete formula gives Z3 = 0, not 2P"""
    p, b, q, yG = A3.curve(128)
    P = OG.mul(5, (0, yG), p - 3, p)
    assert W6.madd((P[0], P[1], 1), P, p)[2] == 0
