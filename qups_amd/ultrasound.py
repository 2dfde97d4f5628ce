"""The caller's side of the path: a Python mirror of ``UltrasoundSystem.DAS`` / ``bfDAS`` / ``bfDASLUT``.

Only what those three methods need is modelled: small value classes carrying the
arrays the reference's definition classes emit -- element positions / normals, the pixel grid, the sequence type
and foci, the channel data with its time axis -- and the argument marshalling of
``UltrasoundSystem.DAS`` (reference ``src/UltrasoundSystem.m:3297-3371``), ``bfDAS`` (``:4429-4473``) and
``bfDASLUT`` (``:4476-4673``).  Simulators, other beamformers, interop, plotting etc. are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import geometry as G
from .das_spec import DasError, das_spec
from .interpd import sample2sep


@dataclass
class Transducer:
    """Element positions ``3 x N`` and normals ``3 x N`` (what ``positions()`` / ``orientations()`` return)."""
    pos: np.ndarray
    normals: np.ndarray
    fc: float = 5e6
    offset: np.ndarray = field(default_factory=lambda: np.zeros(3))

    @property
    def numel(self):
        return self.pos.shape[1]

    def positions(self):
        return self.pos

    @staticmethod
    def linear(numel, pitch, fc=5e6, offset=(0.0, 0.0, 0.0)):          # reference src/TransducerArray.m:95-109
        p, n = G.linear_array(numel, pitch, offset)
        return Transducer(p, n, fc, np.asarray(offset, float))

    @staticmethod
    def convex(numel, radius, angular_pitch_deg, fc=3.7e6, offset=(0.0, 0.0, 0.0)):   # reference src/TransducerConvex.m:85-102
        p, n = G.convex_array(numel, radius, angular_pitch_deg, offset)
        return Transducer(p, n, fc, np.asarray(offset, float))


@dataclass
class Sequence:
    """``type`` in ``{'FSA','PW','FC','VS','DV'}`` (reference ``src/Sequence.m:62``); ``focus`` is ``3 x M``: foci for
    FC/VS/DV, unit normal vectors for PW (reference ``src/UltrasoundSystem.m:3346``)."""
    type: str = "FSA"
    focus: np.ndarray | None = None
    c0: float = 1540.0
    numPulse: int | None = None


@dataclass
class Scan:
    """Pixel grid: ``positions()`` is ``3 x I1 x I2 x I3`` (reference ``src/Scan.m:194``)."""
    pos: np.ndarray

    @property
    def size(self):
        return tuple(self.pos.shape[1:4])

    def positions(self):
        return self.pos

    @staticmethod
    def cartesian(x, z, y=(0.0,)):                   # order 'ZXY' (reference src/ScanCartesian.m:11,126-143)
        return Scan(G.scan_cartesian(x, z, y))

    @staticmethod
    def polar(r, a_deg, origin=(0.0, 0.0, 0.0)):     # order 'RAY' (reference src/ScanPolar.m:11,99-115)
        return Scan(G.scan_polar(r, a_deg, origin))


@dataclass
class ChannelData:
    """``data`` is ``T x N x M x F...`` in ``order='TNM'`` (or ``T x M x N`` for ``'TMN'``); ``t0`` scalar or one value per
    transmit; ``fs`` scalar (reference ``src/ChannelData.m``; ``rectifyDims`` / ``rectifyt0`` are the caller's job here)."""
    data: object
    t0: object = 0.0
    fs: float = 1.0
    order: str = "TNM"


class UltrasoundSystem:
    def __init__(self, xdc: Transducer, seq: Sequence, scan: Scan, fs=None, rx: Transducer | None = None):
        self.tx, self.rx, self.seq, self.scan, self.fs = xdc, rx or xdc, seq, scan, fs
        self.xdc = xdc

    # ------------------------------------------------------------------------------------
    def _tx_geometry(self):
        """(Pv, Nv, options) by sequence type (reference src/UltrasoundSystem.m:3340-3352)"""
        t = self.seq.type
        if t == "FSA":
            return G.sequence_args("FSA", tx_pos=self.tx.positions(), tx_normals=self.tx.normals)
        if t == "PW":
            return G.sequence_args("PW", focus=self.seq.focus)
        if t in ("FC", "VS", "DV"):
            return G.sequence_args(t, focus=self.seq.focus, tx_offset=self.tx.offset)
        raise DasError(f"Unknown sequence type {t!r}.")

    # ---- receive-apodization generators (reference src/UltrasoundSystem.m:5165-5267, 5303-5429)
    def apAcceptanceAngle(self, theta=45.0):
        """materialised ``I1 x I2 x I3 x N x 1`` mask, reference ``:5303-5374``"""
        from . import apodization as A
        return A.ap_acceptance_angle(self.scan.positions(), self.rx.positions(), self.rx.normals, theta)

    def apCosineAngle(self, theta=45.0):
        """reference ``:5377-5429``"""
        from . import apodization as A
        return A.ap_cosine_angle(self.scan.positions(), self.rx.positions(), self.rx.normals, theta)

    def apApertureGrowth(self, f=1.5, Dmax=np.inf):
        """reference ``:5165-5267``"""
        from . import apodization as A
        return A.ap_aperture_growth(self.scan.positions(), self.rx.positions(), self.rx.normals, f, Dmax)

    def rx_apod(self, kind, **kw):
        """the same rules, GENERATED inside the beamforming kernel: ``us.DAS(chd, rx_apod=us.rx_apod('acceptance', theta=30))``"""
        from . import apodization as A
        return A.rx_apod_spec(kind, normals=self.rx.normals, **kw)

    def DAS(self, chd: ChannelData, *apods, c0=None, fmod=0.0, prec="single", device=-1, apod=1, interp="cubic",
            keep_tx=False, keep_rx=False, return_plan=False, kernel=0, rx_apod=None):
        """``b = DAS(us, chd, A1, ..., 'c0', c0, 'fmod', fc, 'interp', method, 'prec', type, 'keep_tx', tf, 'keep_rx', tf)``

        reference ``src/UltrasoundSystem.m:3172-3372``.  Output ``I1 x I2 x I3 x F... x [N] x [M]`` (``:3361``).
        """
        if prec not in ("single", "double", "halfT"):
            raise DasError("prec must be one of {'double', 'single', 'halfT'}")
        c0 = self.seq.c0 if c0 is None else c0
        apods = list(apods) + ([] if (np.isscalar(apod) and apod == 1) else [apod])
        fun = {(True, True): "DAS", (True, False): "SYN", (False, True): "MUL", (False, False): "BF"}[(not keep_tx, not keep_rx)]   # :3318-3322
        if chd.order not in ("TNM", "TMN"):
            raise DasError("ChannelData must be ordered T x perm(N x M) x ... (use rectifyDims).")
        Pv, Nv, opt = self._tx_geometry()
        ext = ["device", device, "input-precision", prec, "transpose", chd.order == "TMN", "interp", interp, "modulation", fmod]   # :3336-3338
        for a in apods:
            ext += ["apod", a]
        if rx_apod is not None:
            ext += ["rx-apod", rx_apod]
        out = das_spec(fun, self.scan.positions(), self.rx.positions(), Pv, Nv, chd.data, chd.t0, chd.fs, c0, *ext, *opt,
                       return_plan=return_plan, kernel=kernel)
        b, plan = out if return_plan else (out, None)
        nd = b.ndim
        b = b.permute(0, 1, 2, *range(5, nd), 3, 4)          # I1 x I2 x I3 x F... x [N] x [M]   (:3361)
        return (b, plan) if return_plan else b

    # ------------------------------------------------------------------------------------
    def delay_tables(self, c0=None, device=None):
        """``tau_rx (I1 x I2 x I3 x N)``, ``tau_tx (I1 x I2 x I3 x M)`` as ``bfDAS`` computes them
        (reference ``src/UltrasoundSystem.m:4429-4463``): ``dr/c0`` and ``dv/c0`` with the per-type sign rule.  numpy arrays, or --
        with ``device`` -- float64 torch tensors computed there (the tables of a 1024 x 1024 image over 256 elements are 2 x 2 GB)."""
        c0 = self.seq.c0 if c0 is None else c0
        Pi, Pr = self.scan.positions(), self.rx.positions()
        Pv, Nv, _ = self._tx_geometry()
        M = max(Pv.shape[1], Nv.shape[1])
        Pv = np.broadcast_to(Pv, (3, M)) if Pv.shape[1] == 1 else Pv
        Nv = np.broadcast_to(Nv, (3, M)) if Nv.shape[1] == 1 else Nv
        if device is not None:
            import torch
            tt = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=device)
            Pi_t, Pr_t, Pv_t, Nv_t = tt(Pi), tt(Pr), tt(Pv), tt(Nv)
            e = lambda P: P[:, None, None, None, :]
            ct = tt(np.asarray(c0, float))
            ct = ct.reshape(tuple(ct.shape) + (1,) * (4 - ct.ndim)) if ct.ndim else ct
            tau_rx = torch.stack([torch.linalg.norm(Pi_t - Pr_t[:, n, None, None, None], dim=0) for n in range(Pr_t.shape[1])], -1) / ct
            cols = []
            for m in range(M):                                     # one transmit at a time: no 3 x I x M temporary
                rv = Pi_t - Pv_t[:, m, None, None, None]
                if self.seq.type in ("DV", "FSA"):
                    cols.append(torch.linalg.norm(rv, dim=0))
                elif self.seq.type in ("VS", "FC"):
                    cols.append(torch.linalg.norm(rv, dim=0) * torch.sign((rv * Nv_t[:, m, None, None, None]).sum(0)))
                else:
                    cols.append((rv * Nv_t[:, m, None, None, None]).sum(0))
            return tau_rx, torch.stack(cols, -1) / ct
        dr = np.linalg.norm(Pi[..., None] - Pr[:, None, None, None, :], axis=0)
        rv = Pi[..., None] - Pv[:, None, None, None, :]
        t = self.seq.type
        if t in ("DV", "FSA"):
            dv = np.linalg.norm(rv, axis=0)
        elif t in ("VS", "FC"):
            dv = np.linalg.norm(rv, axis=0) * np.sign((rv * Nv[:, None, None, None, :]).sum(0))
        else:
            dv = (rv * Nv[:, None, None, None, :]).sum(0)
        c = np.asarray(c0, float)
        c = c.reshape(c.shape + (1,) * (4 - c.ndim)) if c.ndim else c
        return dr / c, dv / c

    def bfDAS(self, chd: ChannelData, *apods, c0=None, apod=1, fmod=0.0, interp="cubic", keep_tx=False, keep_rx=False,
              prec=None):
        """``b = bfDAS(us, chd, ...)`` (reference ``src/UltrasoundSystem.m:4334-4474``): delay tables + ``bfDASLUT``."""
        import torch
        dev = (chd.data.device if hasattr(chd.data, "is_cuda") and chd.data.is_cuda else "cuda") if torch.cuda.is_available() else None
        tau_rx, tau_tx = self.delay_tables(c0, device=dev)
        return self.bfDASLUT(chd, tau_rx, tau_tx, *apods, apod=apod, fmod=fmod, interp=interp, keep_tx=keep_tx, keep_rx=keep_rx, prec=prec)

    def bfDASLUT(self, chd: ChannelData, tau_rx, tau_tx, *apods, apod=1, fmod=0.0, interp="cubic", keep_tx=False,
                 keep_rx=False, prec=None):
        """``b = bfDASLUT(us, chd, tau_rx, tau_tx, ...)`` (reference ``src/UltrasoundSystem.m:4476-4673``):
        ``tau_rx`` is ``I1 x I2 x I3 x N``, ``tau_tx`` is ``I1 x I2 x I3 x M`` (times).  Output
        ``I1 x I2 x I3 x F... x [N] x [M]``: the aperture dimensions are moved behind the frame dimensions exactly as the
        reference does (``:4663-4664``) -- the same layout as ``DAS`` (``:3361``)."""
        if chd.order != "TNM":
            raise DasError("bfDASLUT needs data ordered T x N x M (use rectifyDims).")
        Isz = self.scan.size
        tr, tt = np.asarray(tau_rx) if not hasattr(tau_rx, "shape") else tau_rx, tau_tx
        N, M = self.rx.numel, (self.seq.numPulse or tt.shape[-1])
        if tuple(tr.shape) != Isz + (N,):
            raise DasError(f"Expected a receive delay table of size {Isz + (N,)}, got {tuple(tr.shape)}.",
                           "QUPS:UltrasoundSystem:bfDASLUT:incompatibleReceiveDelayTable")
        if tuple(tt.shape) != Isz + (M,):
            raise DasError(f"Expected a transmit delay table of size {Isz + (M,)}, got {tuple(tt.shape)}.",
                           "QUPS:UltrasoundSystem:bfDASLUT:incompatibleTransmitDelayTable")
        ws = list(apods) + ([] if (np.isscalar(apod) and apod == 1) else [apod])
        w = None
        for a in ws:                                       # separable apodizations multiply (reference :4644)
            a = np.asarray(a)
            a = a.reshape(a.shape + (1,) * (5 - a.ndim))
            w = a if w is None else w * a
        sdim = set() if keep_rx else {"rx"}
        sdim |= set() if keep_tx else {"tx"}
        b = sample2sep(chd.data, chd.t0, chd.fs, tr, tt, interp=interp, w=w, sdim=sdim, fmod=fmod,
                       **({"prec": prec} if prec else {}))           # I1 x I2 x I3 x [N] x [M] x F...
        return b.permute(0, 1, 2, *range(5, b.ndim), 3, 4)           # move the aperture dimensions to the end (:4663-4664)
