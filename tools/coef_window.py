#!/usr/bin/env python3
"""Coefficient-window form of the pair loop ("next round" item 1) -- what it needs, interpolator by interpolator.

A 4-tap sample is  sum_k w_k(u) x[t-1+k].  If every w_k is a polynomial of degree D in the fraction u, then
    sample = sum_d u^d C_d[t],   C_d[t] = sum_k a_kd x[t-1+k]            (D + 1 complex coefficients per sample interval, independent of the pixel)
and a product costs D packed FMAs + 1 packed add (Horner on the complex coefficients, the real fraction broadcast) instead of
4 multiply-accumulates + the tap weights.  DESIGN.md section 9 (round 4) took D = 3 for granted.  This tool measures D:

  * Catmull-Rom `cubic` and the reference's executed `cubic_dev` lines (src/interpd.cu:89-113): EXACT at D = 3.
  * `lanczos3` (src/interpd.cu:116-150: L(v) = sinc(v) sinc(v/2) at v = u+1, u, u-1, u-2): NOT a cubic.  Chebyshev (near-minimax) fits of the four
    weights over u in [0, 1], max |error| by degree -- the kernel's own weights are accurate to 3e-6 (lanczos_poly.h), the parity bar is 5e-5 of the image.

It then prints the per-product accounting of the three forms (packed VALU instructions + LDS bytes gathered per product), which
`qdas_debug_issue_rate` mixes 3 / 4 / 5 (csrc/probe.hip) measure on the device: profiles/r05/coef_window.txt holds both.
"""
import numpy as np
from numpy.polynomial import chebyshev as Ch

L = lambda v: np.sinc(v) * np.sinc(v / 2)
u = (np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) + 1) / 2          # Chebyshev nodes on [0, 1]
ug = np.linspace(0, 1, 20001)


def fit_err(f, deg):
    c = Ch.chebfit(2 * u - 1, f(u), deg)
    return float(np.abs(Ch.chebval(2 * ug - 1, c) - f(ug)).max())


print("max |error| of a degree-D polynomial fit of the four lanczos3 tap weights over the fraction u in [0, 1]:")
print("   D    w(u+1)     w(u)      w(u-1)     w(u-2)     worst")
need = None
for D in range(2, 10):
    e = [fit_err(lambda t, j=j: L(t - j), D) for j in (-1, 0, 1, 2)]
    print(f"  {D:2d}  " + "  ".join(f"{v:9.2e}" for v in e) + f"  {max(e):9.2e}" + ("   <- first degree within the kernel's own 3e-6" if need is None and max(e) <= 3e-6 else ""))
    if need is None and max(e) <= 3e-6:
        need = D
cr = [lambda t: 0.5 * (-t ** 3 + 2 * t ** 2 - t), lambda t: 0.5 * (3 * t ** 3 - 5 * t ** 2 + 2), lambda t: 0.5 * (-3 * t ** 3 + 4 * t ** 2 + t), lambda t: 0.5 * (t ** 3 - t ** 2)]
print(f"Catmull-Rom at D = 3: worst {max(fit_err(f, 3) for f in cr):.1e} (exact); lanczos3 needs D = {need}")
print()
print("per product of one (pixel, trace) -- packed VALU instructions / LDS bytes gathered -- `share` = traces that share one delay: 1 = general mode and the")
print("fold alone (the fold halves the TRACES, it shares nothing), 2 = lateral-mirror or unfolded reciprocal mode and the HEADLINE (fold + mirror), 4 = two frames")
print("per launch in mirror mode (streams).  Index and tap weights are computed once per `share` traces, two transmits per packed instruction:")
print("  form                          share=1            share=2 (headline)  share=4 (streams)   LDS image of a window")
for name, mac, wts, lds, img in (("taps + weights (today), 4 taps", 4, 8.0, 32, "1x"),
                                 ("coefficients, D = 3 (cubic)  ", 4, 1.0, 32, "4x (+ a conversion pass per stage or 4x the staging traffic)"),
                                 (f"coefficients, D = {need} (lanczos3)", need + 1, 1.0, 8 * (need + 1), f"{need + 1}x")):
    cols = []
    for share in (1, 2, 4):
        valu = mac + (wts + 2.5) / share
        cols.append(f"{valu:5.2f} VALU {lds:3d} B")
    print(f"  {name}  " + "   ".join(cols) + f"   {img}")
print()
print("Reading: for lanczos3 the coefficient form costs MORE instructions than the tap form as soon as two traces share a delay (9.75 vs 9.25 at share 2 -- the")
print("headline --, 8.9 vs 6.6 in streams) and twice the LDS bytes per product; only the share-1 loops (general mode 42.6 ms, fold alone 21.6 ms) would gain on paper,")
print("14.5 -> 11.5 (-21 %), for 8x the LDS image of a window: 32-transmit stages become 4-transmit stages and the per-stage code -- 7.5 % of the headline today --")
print("is paid eight times as often, plus a conversion of every staged sample (8 coefficient vectors x 4 taps).  For `cubic` (BASELINE C2 / C5) the form is exact and")
print("the loop drops from 9.25 to 5.75 at share 2 -- before the conversion (~14 instructions per staged sample against ~8 uses of it on a lambda/4 grid = +1.75")
print("per product) and 4x the LDS per window (16-transmit mirror stages become 4-transmit stages).  csrc/probe.hip mixes 3 / 4 / 5 time the three loops on the device.")
