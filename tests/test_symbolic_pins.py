"""A THIRD, symbolic witness for the interpolators (CPU only).

The reference ships no golden vectors for 'cubic' / 'lanczos3' (SURVEY.md section 8c), so every parity test ultimately leans on a
hand restatement of reference src/interpd.cu:87-150 -- twice (oracle/das_oracle.py and oracle/das_ref_body.inc).  Here the source
lines are typed once more as sympy expressions and everything downstream is DERIVED by computer algebra in exact rational (or
50-digit) arithmetic:
  * the Horner lines the device code executes (src/interpd.cu:103-106) are expanded symbolically -- they are NOT the Catmull-Rom
    polynomials of the comment beside them (:108-111); Catmull-Rom reproduces quadratics exactly, the executed lines only linears
    (which is why the reference's own test needs a 1e8 x tolerance for 'cubic', test/interpTest.m:127-133);
  * the oracle's weights ('cubic' = Catmull-Rom = MATLAB interp1 'cubic', 'cubic_dev' = the executed lines, 'lanczos3' = the a = 2
    kernel of :116-127,141-149) agree with the symbolic values at rational offsets to 1e-15;
  * the even / odd minimax polynomials the tiled HIP kernel evaluates (qups_amd/csrc/lanczos_poly.h, tile_util.h weights2) stay
    within 3e-6 of the symbolic Lanczos kernel and reproduce Catmull-Rom exactly.
Parity stays "partial" by the rules (no vector of the reference itself exists); this removes "two hand restatements" as the only
witnesses."""
import os
import re

import numpy as np
import pytest

sp = pytest.importorskip("sympy")
from oracle import das_oracle as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u = sp.Symbol("u")
half = sp.Rational(1, 2)
# reference src/interpd.cu:103-106, as executed (0.5 applied at :112)
EXECUTED = [half * (0 + u * (-1 + u * (+2 * u - 1))), half * (2 + u * (+0 + u * (-5 * u + 3))),
            half * (0 + u * (+1 + u * (+4 * u - 3))), half * (0 + u * (+0 + u * (-1 * u + 1)))]
# reference src/interpd.cu:108-111, the "naive" lines in the comment (Catmull-Rom), same 0.5
COMMENT = [half * (-1 * u**3 + 2 * u**2 - 1 * u + 0), half * (+3 * u**3 - 5 * u**2 + 0 * u + 2),
           half * (-3 * u**3 + 4 * u**2 + 1 * u + 0), half * (+1 * u**3 - 1 * u**2 + 0 * u + 0)]
# reference src/interpd.cu:116-127 with a = 2 (:134): L(v) = 2 sin(pi v) sin(pi v / a) / (pi^2 v^2), L(0) = 1; taps v = u+1, u, u-1, u-2 (:145-148)
v = sp.Symbol("v")
LANCZOS = 2 * sp.sin(sp.pi * v) * sp.sin(sp.pi * v / 2) / (sp.pi**2 * v**2)
RATIONALS = [sp.Rational(p, q) for p, q in ((0, 1), (1, 7), (3, 8), (1, 2), (5, 9), (11, 13), (99, 100))]


def test_executed_cubic_lines_are_not_the_commented_catmull_rom():
    ex = [sp.expand(e) for e in EXECUTED]
    co = [sp.expand(e) for e in COMMENT]
    assert ex == [sp.expand(half * p) for p in (2 * u**3 - u**2 - u, -5 * u**3 + 3 * u**2 + 2, 4 * u**3 - 3 * u**2 + u, -u**3 + u**2)]
    assert ex != co
    taps = [-1, 0, 1, 2]                                            # sample positions of the four taps relative to floor(tau)
    for name, W in (("comment", co), ("executed", ex)):
        assert sp.simplify(sum(W) - 1) == 0, name                   # both interpolate constants
        assert sp.simplify(sum(w * k for w, k in zip(W, taps)) - u) == 0, name          # ... and linears
    assert sp.simplify(sum(w * k**2 for w, k in zip(co, taps)) - u**2) == 0            # Catmull-Rom: quadratics too
    assert sp.simplify(sum(w * k**2 for w, k in zip(ex, taps)) - u**2) != 0            # the executed lines do not
    # both pass through the samples: weights at u = 0 are (0, 1, 0, 0)
    assert [w.subs(u, 0) for w in ex] == [0, 1, 0, 0] and [w.subs(u, 0) for w in co] == [0, 1, 0, 0]


@pytest.mark.parametrize("name,sym", [("cubic", COMMENT), ("cubic_dev", EXECUTED)])
def test_oracle_cubic_weights_equal_the_symbolic_polynomials(name, sym):
    for r in RATIONALS:
        off, w = O.interp_weights(np.array([float(r)]), name)
        assert off == -1
        exact = [float(e.subs(u, r)) for e in sym]                  # exact rational -> nearest double
        assert np.abs(w[:, 0] - exact).max() <= 1e-15, (name, r)


def test_oracle_lanczos_weights_equal_the_symbolic_kernel():
    assert sp.limit(LANCZOS, v, 0) == 1                            # L(0) = 1 is the continuous extension (:121-122)
    for r in RATIONALS:
        off, w = O.interp_weights(np.array([float(r)]), "lanczos3")
        assert off == -1
        exact = []
        for k in (r + 1, r, r - 1, r - 2):
            exact.append(1.0 if k == 0 else float(sp.N(LANCZOS.subs(v, k), 50)))
        assert np.abs(w[:, 0] - exact).max() <= 1e-15, r
    # window a = 2 ("lanczos3" notwithstanding): the kernel vanishes at |v| = 2 and is not normalised
    assert sp.simplify(LANCZOS.subs(v, 2)) == 0
    s = sum(float(sp.N(LANCZOS.subs(v, k), 30)) for k in (sp.Rational(3, 2), half, -half, -sp.Rational(3, 2)))
    assert abs(s - 1) > 1e-3


def _poly_coeffs():
    txt = open(os.path.join(ROOT, "qups_amd", "csrc", "lanczos_poly.h")).read()
    get = lambda nm: [float(x.strip().rstrip("f")) for x in re.search(r"#define QDAS_LANCZOS_" + nm + r" \{([^}]*)\}", txt).group(1).split(",")]
    return get("EI"), get("OI"), get("EO"), get("OO")


def test_tiled_kernel_polynomials_track_the_symbolic_kernels():
    """tile_util.h weights2<3>: w1,w2 = EI(q) +- s OI(q), w0,w3 = EO(q) +- s OO(q), s = u - 1/2, q = s^2; weights2<2>: exact"""
    EI, OI, EO, OO = _poly_coeffs()
    hor = lambda c, q: sum(ck * q**k for k, ck in enumerate(c))
    Lf = sp.lambdify(v, LANCZOS, "mpmath")
    worst = 0.0
    for k in range(0, 1001):
        uu = k / 1000.0
        s = uu - 0.5
        q = s * s
        w = [hor(EO, q) + s * hor(OO, q), hor(EI, q) + s * hor(OI, q), hor(EI, q) - s * hor(OI, q), hor(EO, q) - s * hor(OO, q)]
        ex = [1.0 if abs(t) < 1e-300 else float(Lf(t)) for t in (uu + 1, uu, uu - 1, uu - 2)]
        worst = max(worst, max(abs(a - b) for a, b in zip(w, ex)))
    assert worst <= 3.2e-6, worst
    # Catmull-Rom in the kernel's even/odd form (weights2<2>) is the commented polynomial EXACTLY
    s_ = sp.Symbol("s")
    q_ = s_**2
    ei, oi = sp.Rational(9, 16) - q_ / 4, -sp.Rational(11, 8) + sp.Rational(3, 2) * q_
    eo, oo = -sp.Rational(1, 16) + q_ / 4, sp.Rational(1, 8) - q_ / 2
    kern = [eo + s_ * oo, ei + s_ * oi, ei - s_ * oi, eo - s_ * oo]
    for a, b in zip(kern, COMMENT):
        assert sp.expand(a.subs(s_, u - half) - b) == 0
