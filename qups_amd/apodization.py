"""Receive-apodization generators of the reference (``UltrasoundSystem.ap*``), host side.

Two forms of the same three pixel x receiver rules:

* ``ap_acceptance_angle / ap_cosine_angle / ap_aperture_growth`` return the MATERIALISED array
  ``I1 x I2 x I3 x N x 1`` exactly like the reference (``src/UltrasoundSystem.m:5165-5267, 5303-5429``) --
  pass it as ``'apod', A``;
* ``rx_apod_spec(kind, ...)`` returns the few numbers the kernels need to GENERATE the same weights from
  the geometry they already hold (``'rx-apod', spec`` option of :func:`qups_amd.das_spec`; C ABI
  ``qdas_desc.rx_apod_kind / rx_apod_p / rx_normals``): no I x N array is built, stored or streamed
  (C5: 268 MB; a C3-sized image: 2.1 GB).

Plus the transmit-side rules of focused sequences (``ap_scanline``, ``ap_multiline``, ``ap_translating_aperture``, ``:4892-5163``) in the
reference's broadcast shapes.

All arithmetic here is float64 numpy on the host (this is input marshalling, not the hot path).
"""
from __future__ import annotations

import numpy as np

from . import _lib

KINDS = {"acceptance": _lib.RXAPOD_ACCEPTANCE, "cosine": _lib.RXAPOD_COSINE,
         "fnumber-planar": _lib.RXAPOD_FNUMBER_PLANAR, "fnumber-oriented": _lib.RXAPOD_FNUMBER_ORIENTED}


def cosd(deg: float) -> float:
    """MATLAB ``cosd``: exact at multiples of 90 degrees (and 60 -> 0.5)."""
    d = float(deg) % 360.0
    exact = {0.0: 1.0, 60.0: 0.5, 90.0: 0.0, 120.0: -0.5, 180.0: -1.0, 240.0: -0.5, 270.0: 0.0, 300.0: 0.5}
    return exact.get(d, float(np.cos(np.deg2rad(d))))


def _geom(Pi, Pr, normals):
    Pi = np.asarray(Pi, dtype=np.float64)
    Pr = np.asarray(Pr, dtype=np.float64)
    Pi = Pi.reshape(3, *Pi.shape[1:], *([1] * (4 - Pi.ndim)))                 # 3 x I1 x I2 x I3
    r = Pi[..., None] - Pr.reshape(3, 1, 1, 1, -1)                            # 3 x I1 x I2 x I3 x N
    n = None if normals is None else np.asarray(normals, dtype=np.float64).reshape(3, 1, 1, 1, -1)
    return Pi, Pr, r, n


def _cosang(r, n):
    with np.errstate(invalid="ignore", divide="ignore"):
        return (r * n).sum(0) / np.sqrt((r * r).sum(0))                       # I1 x I2 x I3 x N


def ap_acceptance_angle(Pi, Pr, normals, theta: float = 45.0) -> np.ndarray:
    """``apod = r.nhat/|r| >= cosd(theta)`` (reference ``src/UltrasoundSystem.m:5355-5373``); ``I1 x I2 x I3 x N x 1``."""
    _, _, r, n = _geom(Pi, Pr, normals)
    with np.errstate(invalid="ignore"):
        return (_cosang(r, n) >= cosd(theta)).astype(np.float64)[..., None]


def ap_cosine_angle(Pi, Pr, normals, theta: float = 45.0) -> np.ndarray:
    """``cosd(min(90, (90/theta) * acosd(r)))`` with ``r`` clipped to [-1, 1] (reference ``:5414-5428``)."""
    _, _, r, n = _geom(Pi, Pr, normals)
    c = _cosang(r, n)
    c = np.where(np.isnan(c), 1.0, np.clip(c, -1.0, 1.0))                     # MATLAB max/min ignore NaN -> 1
    return np.cos(np.minimum(np.pi / 2, (90.0 / float(theta)) * np.arccos(c)))[..., None]


def ap_aperture_growth(Pi, Pr, normals=None, f: float = 1.5, Dmax: float = np.inf) -> np.ndarray:
    """f-number limited aperture (reference ``:5227-5262``): planar arrays ``z_i > f |2 (x_n - x_i)|``; arrays whose
    elements are rotated (non-zero orientation) use the equivalent width / depth in each element's frame."""
    Pi_, Pr_, r, n = _geom(Pi, Pr, normals)
    if rx_apod_spec("fnumber", normals=normals)["kind"] == _lib.RXAPOD_FNUMBER_PLANAR:
        d2 = np.abs(2.0 * (Pr_[0].reshape(1, 1, 1, -1) - Pi_[0][..., None]))
        z = Pi_[2][..., None] + 0.0 * d2
    else:
        d2 = np.abs(2.0 * (r[0] * n[2] - r[2] * n[0]))
        z = np.abs(r[0] * n[0] + r[2] * n[2])
    return ((z > f * d2) & (d2 < Dmax)).astype(np.float64)[..., None]


# ------------------------------------------------------------------------------------------
# transmit-side rules of focused sequences: weights per (lateral pixel coordinate, transmit).  The reference returns them in
# broadcast form -- singleton over depth: ``1 x I2 x 1 x 1 x M`` for a Cartesian scan with the lateral axis second -- and so do these;
# the plan multiplies such arrays out to ``I x M`` once (csrc/qdas_api.hip apod_fold_kernel) and runs the fused kernel with the transmit
# as the stage element.
# ------------------------------------------------------------------------------------------
def _lateral(xi, xdim: int) -> np.ndarray:
    shape = [1] * 5
    xi = np.asarray(xi, dtype=np.float64).reshape(-1)
    shape[xdim] = xi.size
    return xi.reshape(shape)


def ap_scanline(xi, xv, tol: float | None = None, xdim: int = 1) -> np.ndarray:
    """Scanline mask (reference ``src/UltrasoundSystem.m:4892-4968``): pixel column ``xi`` (lateral position, or angle of a polar scan) takes
    the transmits whose focus (or steering angle) ``xv`` lies within ``tol`` of it -- default: the mean spacing of ``xi`` (the reference's
    ``scan.dx``).  ``xdim`` = axis (0-based) of the lateral pixel dimension.  Shape: singleton but for ``xdim`` and the transmit axis."""
    xi = np.asarray(xi, dtype=np.float64).reshape(-1)
    xv = np.asarray(xv, dtype=np.float64).reshape(1, 1, 1, 1, -1)
    if tol is None:
        tol = float(np.mean(np.diff(xi))) if xi.size > 1 else np.inf
    return (np.abs(_lateral(xi, xdim) - xv) < tol).astype(np.float64)


def ap_multiline(xi, xv, xdim: int = 1) -> np.ndarray:
    """Multiline weights (reference ``:4970-5072``): every pixel column is interpolated linearly between the nearest transmit on its left
    (the last ``xv <= xi``) and on its right (the first ``xv >= xi``) in transmit order; columns without a transmit on both sides get nothing;
    where the two coincide the left one takes weight 1."""
    xi = np.asarray(xi, dtype=np.float64).reshape(-1)
    xv = np.asarray(xv, dtype=np.float64).reshape(-1)
    out = np.zeros((xi.size, xv.size))
    for k, x in enumerate(xi):                           # (host marshalling: a few hundred columns)
        left = np.nonzero(x - xv >= 0)[0]
        right = np.nonzero(x - xv <= 0)[0]
        if left.size == 0 or right.size == 0:
            continue
        l, r = left[-1], right[0]
        d = abs(xv[l] - xv[r])
        if d == 0:
            out[k, l] += 1.0
        else:
            out[k, l] += 1.0 - abs(xv[l] - x) / d
            out[k, r] += 1.0 - abs(xv[r] - x) / d
    shape = [1] * 5
    shape[xdim] = xi.size
    shape[4] = xv.size
    return out.reshape(shape)


def ap_translating_aperture(xi, xv, xn, tol, xdim: int = 1) -> np.ndarray:
    """Translating-aperture mask (reference ``:5074-5163``): ``|xi - xv| <= tol[0]`` and ``|xi - xn| <= tol[-1]`` -- depends on the pixel column,
    the receiver AND the transmit (``1 x I2 x 1 x N x M``): such plans run the generic kernel."""
    tol = np.atleast_1d(np.asarray(tol, dtype=np.float64))
    xn = np.asarray(xn, dtype=np.float64).reshape(1, 1, 1, -1, 1)
    xv = np.asarray(xv, dtype=np.float64).reshape(1, 1, 1, 1, -1)
    X = _lateral(xi, xdim)
    return ((np.abs(X - xv) <= tol[0]) & (np.abs(X - xn) <= tol[-1])).astype(np.float64)


def rx_apod_spec(kind: str, theta: float | None = None, f: float = 1.5, Dmax: float = np.inf, normals=None) -> dict:
    """Parameters of an in-kernel generated receive apodization: ``{'kind', 'p': (p0, p1), 'normals': 3 x N | None}``.

    ``kind``: ``'acceptance'`` (theta, default 45), ``'cosine'`` (theta, default 45), ``'fnumber'`` (f, Dmax; planar
    when every normal is +z or no normals are given, like the reference's ``any(us.rx.orientations)`` test)."""
    nrm = None if normals is None else np.asarray(normals, dtype=np.float64).reshape(3, -1)
    if kind == "acceptance":
        if nrm is None:
            raise ValueError("'acceptance' needs the element normals")
        return dict(kind=_lib.RXAPOD_ACCEPTANCE, p=(cosd(45.0 if theta is None else theta), 0.0), normals=nrm)
    if kind == "cosine":
        if nrm is None:
            raise ValueError("'cosine' needs the element normals")
        return dict(kind=_lib.RXAPOD_COSINE, p=(90.0 / float(45.0 if theta is None else theta), 0.0), normals=nrm)
    if kind == "fnumber":
        planar = nrm is None or bool(np.all(nrm[0] == 0) and np.all(nrm[1] == 0))
        return dict(kind=_lib.RXAPOD_FNUMBER_PLANAR if planar else _lib.RXAPOD_FNUMBER_ORIENTED,
                    p=(float(f), float(Dmax)), normals=None if planar else nrm)
    raise ValueError(f"unknown generated apodization {kind!r}: 'acceptance' | 'cosine' | 'fnumber'")
