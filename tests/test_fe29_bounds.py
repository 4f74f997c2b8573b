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
