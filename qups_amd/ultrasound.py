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
from .interpd import sample2sep, wsinterpd


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

    def delays(self, tx: "Transducer"):
        """``tau = delays(seq, tx)``: ``N x S`` transmit delays (reference ``src/Sequence.m:888-931``): FC ``+|focus - p|/c0``,
        DV the negative, VS signed by whether every element lies in front of the focus, PW ``-(n . p)/c0``, FSA zeros."""
        p = np.asarray(tx.positions(), float)
        N = p.shape[1]
        if self.type == "FSA":
            return np.zeros((N, N))
        f = np.asarray(self.focus, float)
        if self.type == "PW":
            return -(f[:, None, :] * p[:, :, None]).sum(0) / self.c0
        v = f[:, None, :] - p[:, :, None]
        tau = np.sqrt((v ** 2).sum(0)) / self.c0
        if self.type == "FC":
            return tau
        if self.type == "DV":
            return -tau
        if self.type == "VS":
            return tau * np.where(np.all(f[2][None, :] > p[2][:, None], axis=0), 1.0, -1.0)[None, :]
        raise DasError(f"Unknown sequence type {self.type!r}.")

    def apodization(self, tx: "Transducer"):
        """``N x S`` transmit apodization (reference ``src/Sequence.m:953-975``): identity for FSA, ones otherwise"""
        N = tx.numel
        if self.type == "FSA":
            return np.eye(N)
        S = self.numPulse or (np.asarray(self.focus).shape[1] if self.focus is not None else N)
        return np.ones((N, S))


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
    """``data`` is ``T x N x M x F...`` in ``order='TNM'`` (any permutation of the three letters is accepted and put in order by
    :meth:`rectifyDims`); ``t0`` scalar or one value per transmit (``1 x 1 x M``); ``fs`` scalar (reference ``src/ChannelData.m``)."""
    data: object
    t0: object = 0.0
    fs: float = 1.0
    order: str = "TNM"

    # -- the pieces of ChannelData the DAS path calls
    def _torch_data(self):
        import torch
        return self.data if hasattr(self.data, "is_cuda") else torch.from_numpy(np.asarray(self.data))

    @property
    def T(self):
        return int(self.data.shape[self.order.index("T")])

    def rectifyDims(self):
        """``T x N x M x ...`` order (reference ``src/ChannelData.m:1895-1913``): permutes ``data`` (and a non-scalar ``t0``)"""
        if self.order[:3] == "TNM":
            return self
        d = self._torch_data()
        lead = [self.order.index(c) for c in "TNM"]
        perm = lead + [k for k in range(max(d.ndim, 3)) if k not in lead]
        d = d.reshape(tuple(d.shape) + (1,) * (len(perm) - d.ndim)).permute(*perm)
        t0 = self.t0
        if np.ndim(t0) > 0 and np.size(t0) > 1:
            t0a = np.asarray(t0)
            t0a = t0a.reshape(t0a.shape + (1,) * (len(perm) - t0a.ndim)).transpose(perm)
            t0 = t0a
        return ChannelData(d, t0, self.fs, "TNM")

    def zeropad(self, B=0, A=0):
        """``B`` zeros in front (``t0`` moves back by ``B/fs``), ``A`` behind, along time (reference ``src/ChannelData.m:1153-1183``)"""
        import torch
        if A < 0 or B < 0:
            raise DasError("Data append or prepend size must be positive.")
        if A == 0 and B == 0:
            return self
        d = self._torch_data()
        ax = self.order.index("T")
        z = lambda n: torch.zeros(tuple(d.shape[:ax]) + (n,) + tuple(d.shape[ax + 1:]), dtype=d.dtype, device=d.device)
        return ChannelData(torch.cat([z(B), d, z(A)], ax), np.asarray(self.t0, float) - B / self.fs if np.ndim(self.t0) else float(self.t0) - B / self.fs,
                           self.fs, self.order)

    def filter(self, b, dim=None, a=None, sos=None, gain=1.0):
        """Filter the data along time -- what ``filter(chd, D)`` does with a ``digitalFilter`` ``D`` (reference ``src/ChannelData.m:857-888``: ``filter(D, x)``
        along the time dimension, then ``t0 -= L / fs`` with ``L = filtord(D) / 2`` for an FIR and ``L = filtord(D)`` for an IIR filter, ``:876-879``).
        FIR (``b`` alone): the causal convolution ``y[t] = sum_k b[k] x[t - k]`` on the device (``qdas_convd`` with the ``'causal'`` window).
        IIR: second-order sections ``sos`` (``n x 6``, MATLAB's ``D.Coefficients``) with ``gain``, or a transfer function ``(b, a)`` -- converted to sections
        (``scipy.signal.tf2sos``: MATLAB's own ``filter(b, a, x)`` runs the direct form; the two agree to rounding) -- on the device (``qdas_iir``)."""
        import torch
        ax = self.order.index("T") if dim is None else int(dim) - 1
        d = self._torch_data()
        if sos is not None or a is not None:
            from .convd import sosfilt
            if sos is None:
                from scipy.signal import tf2sos
                bb, aa = np.atleast_1d(np.asarray(b, float)), np.atleast_1d(np.asarray(a, float))
                order = max(len(bb), len(aa)) - 1
                sos = tf2sos(bb, aa) if order > 0 else np.array([[bb[0], 0, 0, aa[0], 0, 0]], float)
            else:
                sos = np.asarray(sos, float).reshape(-1, 6)
                deg = lambda c: 2 if c[2] != 0 else (1 if c[1] != 0 else 0)
                order = int(max(sum(deg(r[0:3]) for r in sos), sum(deg(r[3:6]) for r in sos)))      # filtord of the cascade: the degree of its transfer function
            if d.dtype in (torch.float16, torch.complex32):
                d = d.to(torch.complex64 if d.is_complex() else torch.float32)
            y = sosfilt(d, sos, ax + 1, gain)
            t0 = self.t0
            if ax == self.order.index("T"):
                t0 = np.asarray(t0, float) - order / self.fs if np.ndim(t0) else float(t0) - order / self.fs
            return ChannelData(y, t0, self.fs, self.order)
        from .convd import convd
        bt = b if hasattr(b, "is_cuda") else torch.from_numpy(np.asarray(b))
        bt = bt.reshape(-1)
        if not (bt.is_floating_point() or bt.is_complex()):
            bt = bt.to(torch.float64)
        if d.dtype in (torch.float32, torch.complex64, torch.float16, torch.complex32) and bt.dtype in (torch.float64, torch.complex128):
            bt = bt.to(torch.complex64 if bt.is_complex() else torch.float32)      # (coefficients follow the data's precision, as MATLAB's filter does for single data)
        y = convd(d, bt.reshape((1,) * ax + (-1,) + (1,) * (d.ndim - ax - 1)), ax + 1, "causal")
        t0 = self.t0
        if ax == self.order.index("T"):
            L = (bt.numel() - 1) / 2.0                           # filtord / 2
            t0 = np.asarray(t0, float) - L / self.fs if np.ndim(t0) else float(t0) - L / self.fs
        return ChannelData(y, t0, self.fs, self.order)

    def hilbert(self, N=None):
        """analytic signal along time (reference ``src/ChannelData.m:935-966``: ``hilbert(chd, N)``, zero-padded / truncated to ``N`` points)
        on the device: real fp32 / int16 traces go through the one-pass kernel of ``qdas_pre_*`` (``qups_amd/csrc/pre.hip``)"""
        from .preproc import hilbert
        d = self._torch_data()
        ax = self.order.index("T")
        if d.is_complex():
            raise DasError("hilbert expects real data")
        y = hilbert(d.movedim(ax, 0), N)
        return ChannelData(y.movedim(0, ax), self.t0, self.fs, self.order)

    def downmix(self, fc: float):
        """``chd.data .* exp(-2i*pi*fc*time)`` (reference ``src/ChannelData.m:757-766``); for real traces the Hilbert transform and the mixing are
        ONE pass over the data (``hilbert`` with ``fdown``), complex data are multiplied on the device"""
        import torch
        d = self._torch_data()
        ax = self.order.index("T")
        t0 = np.asarray(self.t0, float)
        if not d.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError("qups_amd: no HIP device visible -- downmix has no CPU fallback")
            d = d.cuda()
        n = torch.arange(d.shape[ax], device=d.device, dtype=torch.float64).reshape((1,) * ax + (-1,) + (1,) * (d.ndim - ax - 1))
        if t0.size == 1:
            t0t = float(t0.reshape(-1)[0])
        else:                                                        # one start time per transmit: an array that broadcasts against the data
            if t0.ndim > d.ndim or any(a not in (1, b) for a, b in zip(t0.shape, d.shape)):
                raise DasError("downmix: t0 must be a scalar or broadcast against the data")
            t0t = torch.from_numpy(t0.reshape(t0.shape + (1,) * (d.ndim - t0.ndim))).to(d.device)
        cyc = fc * (t0t + n / self.fs)
        ph = (-2.0 * np.pi) * (cyc - torch.floor(cyc))
        cdt = torch.complex128 if d.dtype in (torch.float64, torch.complex128) else torch.complex64
        return ChannelData(d.to(cdt) * torch.polar(torch.ones_like(ph), ph).to(cdt), self.t0, self.fs, self.order)

    def downsample(self, ratio: int):
        """every ``ratio``-th sample along time (reference ``src/ChannelData.m:1042-1058``: ``subD(chd, 1:ratio:chd.T, chd.tdim)``); ``fs`` follows"""
        ratio = int(ratio)
        if ratio < 1:
            raise DasError("ratio must be a positive integer")
        d = self._torch_data()
        ax = self.order.index("T")
        idx = [slice(None)] * d.ndim
        idx[ax] = slice(0, None, ratio)
        return ChannelData(d[tuple(idx)].contiguous(), self.t0, self.fs / ratio, self.order)

    def sample(self, tau, interp="linear", w=1, sdim=None, fmod=0.0, **kw):
        """``y = sample(chd, tau, interp, w, sdim, fmod)`` (reference ``src/ChannelData.m:1230-1336``): ``tau`` holds TIMES and
        broadcasts against ``T x N x M x F...`` in every dimension but the first; sample indices ``(tau - t0) * fs`` (``:1317``),
        upmix ``exp(2i pi fmod/fs * ntau)`` (``:1320``), then ``wsinterpd(data, ntau, 1, w, sdim, interp, 0, omega)`` (``:1327``)."""
        import torch
        chd = self.rectifyDims()
        as_t = lambda a: a if hasattr(a, "is_cuda") else torch.from_numpy(np.asarray(a, dtype=np.float64))
        tt = as_t(tau).to(torch.float64)
        t0 = as_t(np.asarray(chd.t0, dtype=np.float64)).to(tt.device)
        ntau = (tt - t0) * chd.fs
        return wsinterpd(chd.data, ntau, 1, w, sdim, interp, 0.0, 2j * np.pi * fmod / chd.fs, **kw)

    def rectifyt0(self, interp="cubic"):
        """one start time for all transmits (reference ``src/ChannelData.m:1205-1228``): traces are resampled onto the earliest
        ``t0``; the time axis grows by ``ceil(max(t0 - min t0) * fs)`` samples at zeropad and again in the sampling grid, as in
        the reference (``:1220-1222``)."""
        if np.size(self.t0) == 1:
            return self
        chd = self.rectifyDims()
        t0 = np.asarray(chd.t0, float)
        t0_ = float(t0.min())
        npad = int(np.ceil((t0 - t0_).max() * chd.fs))
        chd = chd.zeropad(0, npad)
        nd = max(chd._torch_data().ndim, t0.ndim)
        tau = t0_ + np.arange(chd.T + npad).reshape((-1,) + (1,) * (nd - 1)) / chd.fs
        return ChannelData(chd.sample(tau, interp), t0_, chd.fs, "TNM")


def _cat_tx(parts):
    import torch
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=4)


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

    @staticmethod
    def _chd_array(chd):
        """``(list of ChannelData, concatenation dimension | None)`` of an ND-array of ChannelData (reference ``src/UltrasoundSystem.m:3301-3305``,
        ``:4670-4672``): a single object, a numpy object array with at most one non-singleton dimension, or a list / tuple -- MATLAB's
        ``[chd1, chd2, ...]``, a 1 x K row.  The dimension is 0-based here (MATLAB's ``chddim - 1``): the images are concatenated along it."""
        if isinstance(chd, ChannelData):
            return [chd], None
        if isinstance(chd, (list, tuple)):
            arr = np.empty((1, len(chd)), dtype=object)
            for k, c in enumerate(chd):
                arr[0, k] = c
        else:
            arr = np.asarray(chd, dtype=object)
            if arr.ndim < 2:
                arr = arr.reshape(1, -1)                       # (MATLAB has no 1-D arrays: a vector is a row)
        items = list(arr.reshape(-1, order="F"))
        if not items or not all(isinstance(c, ChannelData) for c in items):
            raise DasError("chd must be a ChannelData or an array of ChannelData")
        dims = [d for d, n in enumerate(arr.shape) if n > 1]
        if len(dims) > 1:
            raise DasError("ChannelData array can only contain up to one non-scalar dimension.")     # (:3304)
        return items, (dims[0] if dims else None)

    def DAS(self, chd: ChannelData, *apods, c0=None, fmod=0.0, prec="single", device=-1, apod=1, interp="cubic",
            keep_tx=False, keep_rx=False, return_plan=False, kernel=0, rx_apod=None):
        """``b = DAS(us, chd, A1, ..., 'c0', c0, 'fmod', fc, 'interp', method, 'prec', type, 'keep_tx', tf, 'keep_rx', tf)``

        reference ``src/UltrasoundSystem.m:3172-3372``.  Output ``I1 x I2 x I3 x F... x [N] x [M]`` (``:3361``).  ``chd`` may be an array of
        ChannelData with one non-scalar dimension (``:3301-3305``): each is beamformed on its own (``:3325``) and the images are
        concatenated along that dimension (``:3368``); with ``return_plan`` the plan of the first element comes back (the reference's
        loop runs from the last element to the first and keeps the last call's kernel handle).
        """
        if prec not in ("single", "double", "halfT"):
            raise DasError("prec must be one of {'double', 'single', 'halfT'}")
        chds, chddim = self._chd_array(chd)
        if len(chds) > 1 or chddim is not None:
            import torch
            kw = dict(c0=c0, fmod=fmod, prec=prec, device=device, apod=apod, interp=interp, keep_tx=keep_tx, keep_rx=keep_rx, kernel=kernel, rx_apod=rx_apod)
            D = max([c.data.ndim for c in chds] + [0 if chddim is None else chddim + 1])        # max dimension of data (:3305)
            outs, plan = [], None
            for k, c in enumerate(chds):
                o = self.DAS(c, *apods, return_plan=(return_plan and k == 0), **kw)
                if return_plan and k == 0:
                    o, plan = o
                outs.append(o)
            nd = max(max(o.ndim for o in outs), (chddim or 0) + 1, D)
            outs = [o.reshape(tuple(o.shape) + (1,) * (nd - o.ndim)) for o in outs]
            b = outs[0] if chddim is None else torch.cat(outs, dim=chddim)
            return (b, plan) if return_plan else b
        chd = chds[0]
        c0 = self.seq.c0 if c0 is None else c0
        apods = list(apods) + ([] if (np.isscalar(apod) and apod == 1) else [apod])
        fun = {(True, True): "DAS", (True, False): "SYN", (False, True): "MUL", (False, False): "BF"}[(not keep_tx, not keep_rx)]   # :3318-3322
        if np.size(chd.t0) > 1 and np.asarray(chd.t0).reshape(-1).size != self._num_tx(chd):
            chd = chd.rectifyt0()                              # t0 varies outside the transmit dimension   (:3330)
        if chd.order[0] != "T" or chd.order[:3] not in ("TNM", "TMN"):
            chd = chd.rectifyDims()                            # time is not the first dimension             (:3333)
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

    def greens(self, scat_pos, scat_amp, waveform, wv_t0, wv_fs, fs=None, R0=None, interp="cubic", focus=True):
        """``chd = greens(us, scat)`` (reference ``src/UltrasoundSystem.m:463-882``): full-synthetic-aperture channel data of point
        scatterers from the simulator kernel, then -- like the reference's last step (``:877``) -- ``focusTx`` synthesises this
        system's transmit sequence from it.  The transmit-receive waveform is an input (samples, start time, sampling frequency)."""
        import torch
        from .greens import greens as _greens
        fs = fs or self.fs
        y, t0 = _greens(self.rx.positions(), self.tx.positions(), scat_pos, scat_amp, self.seq.c0, waveform, wv_t0, wv_fs, fs,
                        R0=R0, interp=interp)
        chd = ChannelData(y.contiguous(), t0, fs, "TNM")
        return self.focusTx(chd, self.seq, interp=interp) if focus else chd

    def _num_tx(self, chd):
        return int(chd.data.shape[chd.order.index("M")])

    def focusTx(self, chd: ChannelData, seq: Sequence | None = None, interp="cubic", buffer=0):
        """``chd = focusTx(us, chd, seq)`` (reference ``src/UltrasoundSystem.m:3374-3503``): synthesise the transmits of ``seq`` from
        full-synthetic-aperture data by delaying and summing over the transmit elements,
        ``z[t', n, m'] = sum_m apd[m, m'] * x(time[t'] - tau[m, m'], n, m)`` with ``tau = -seq.delays(tx)`` shifted so that all delays
        are non-negative (``:3457-3470``; the reference samples with ``sample2sep`` at ``:3498``)."""
        import torch
        from .interpd import das_lut
        seq = seq or self.seq
        chd = chd.rectifyDims()
        tau = -np.asarray(seq.delays(self.tx), float)            # M x M'
        apd = np.broadcast_to(np.asarray(seq.apodization(self.tx), float), tau.shape)
        if seq.type == "FSA" and not np.count_nonzero(tau) and np.array_equal(apd, np.eye(self.tx.numel)):
            return chd                                             # already FSA (:3461)
        i = apd != 0
        nmin = int(np.floor(np.nanmin(tau[i]) * chd.fs))
        nmax = int(np.ceil(np.nanmax(tau[i]) * chd.fs))
        t0 = float(np.asarray(chd.t0).reshape(-1)[0]) + nmin / chd.fs
        tau = tau - nmin / chd.fs
        pad = (nmax - nmin) + int(buffer)
        d = chd._torch_data()
        if d.dtype in (torch.float32, torch.float64, torch.complex64, torch.complex128):
            # the positions are the record's own time grid plus ONE offset per (element, synthesised transmit): the shift-and-sum kernel
            # (qdas_shift_sum: per-pair tap offset and weights, LDS-staged windows, zero weights skipped; round 3).  The zeros the reference appends
            # (chd = zeropad(chd, 0, nmax - nmin + buffer), :3479) are not stored: `tpad` makes the kernel read them as in-range zeros (round 6)
            from .interpd import shift_sum
            dev = d.device if d.is_cuda else torch.device("cuda")
            z = shift_sum(d.to(dev), -tau * chd.fs, apd, interp, tpad=pad)
            return ChannelData(z, t0, chd.fs, "TNM")
        chd = ChannelData(chd.data, t0, chd.fs, "TNM").zeropad(0, pad)
        d = chd._torch_data()
        T2, N, M = d.shape[:3]
        dev = d.device if d.is_cuda else torch.device("cuda")
        Mp = tau.shape[1]
        # other data types: ONE launch of the general single-delay kernel over the index space (t', n, m, m', frames...): the sample index depends
        # on (t', m, m'), the data on (n, m, frames), the weight on (m, m'); the transmit elements m are summed
        ntau = torch.arange(T2, dtype=torch.float64, device=dev).reshape(T2, 1, 1, 1) - torch.from_numpy(tau * chd.fs).to(dev).reshape(1, 1, M, Mp)
        xd = d.to(dev).reshape((T2, N, M, 1) + tuple(d.shape[3:]))
        z = wsinterpd(xd, ntau, 1, apd.reshape(1, 1, M, Mp), [3], interp, 0.0)        # T' x N x 1 x M' x F...
        z = z.reshape((T2, N, Mp) + tuple(d.shape[3:]))
        return ChannelData(z, t0, chd.fs, "TNM")

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
            # blocks of elements whose 3 x I x block temporary stays below ~256 MB: a handful of launches instead of one per element
            I = int(np.prod(Pi_t.shape[1:]))
            nb = max(1, (1 << 25) // max(3 * I, 1))
            ex = lambda P, a, b: P[:, None, None, None, a:b]
            tau_rx = torch.cat([torch.linalg.norm(Pi_t[..., None] - ex(Pr_t, a, min(a + nb, Pr_t.shape[1])), dim=0)
                                for a in range(0, Pr_t.shape[1], nb)], -1) / ct
            cols = []
            for a in range(0, M, nb):
                rv = Pi_t[..., None] - ex(Pv_t, a, min(a + nb, M))
                if self.seq.type in ("DV", "FSA"):
                    cols.append(torch.linalg.norm(rv, dim=0))
                elif self.seq.type in ("VS", "FC"):
                    cols.append(torch.linalg.norm(rv, dim=0) * torch.sign((rv * ex(Nv_t, a, min(a + nb, M))).sum(0)))
                else:
                    cols.append((rv * ex(Nv_t, a, min(a + nb, M))).sum(0))
            return tau_rx, torch.cat(cols, -1) / ct
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
              prec=None, bsize=None):
        """``b = bfDAS(us, chd, ...)`` (reference ``src/UltrasoundSystem.m:4334-4474``): delay tables + ``bfDASLUT``."""
        import torch
        first = self._chd_array(chd)[0][0]                 # (an array of ChannelData: the tables serve every element, :4429-4463)
        dev = (first.data.device if hasattr(first.data, "is_cuda") and first.data.is_cuda else "cuda") if torch.cuda.is_available() else None
        tau_rx, tau_tx = self.delay_tables(c0, device=dev)
        return self.bfDASLUT(chd, tau_rx, tau_tx, *apods, apod=apod, fmod=fmod, interp=interp, keep_tx=keep_tx, keep_rx=keep_rx, prec=prec, bsize=bsize)

    def bfDASLUT(self, chd: ChannelData, tau_rx, tau_tx, *apods, apod=1, fmod=0.0, interp="cubic", keep_tx=False,
                 keep_rx=False, prec=None, bsize=None):
        """``b = bfDASLUT(us, chd, tau_rx, tau_tx, ...)`` (reference ``src/UltrasoundSystem.m:4476-4673``):
        ``tau_rx`` is ``I1 x I2 x I3 x N``, ``tau_tx`` is ``I1 x I2 x I3 x M`` (times).  Output
        ``I1 x I2 x I3 x F... x [N] x [M]``: the aperture dimensions are moved behind the frame dimensions exactly as the
        reference does (``:4663-4664``) -- the same layout as ``DAS`` (``:3361``).

        ``bsize``: transmits per block (reference ``:4573``: default from a 1 GB bound on the multiplied-out weights; ``:4641-4655``: the
        transmits are spliced, the apodization arrays with a transmit dimension are indexed per block, blocks are summed -- or, with
        ``keep_tx``, concatenated).  The multiplied-out ``I x N x M`` weight array never exists for more than one block."""
        chds, chddim = self._chd_array(chd)
        # (:4580-4587, :4604-4611) the tables serve every ChannelData of the array: one receiver count, one transmit count
        Ns = sorted({int(c.data.shape[c.order.index("N")]) for c in chds})
        if len(Ns) != 1:
            raise DasError("Expected a single receiver size, but instead they have sizes [" + ",".join(str(v) for v in Ns) + "].",
                           "QUPS:UltrasoundSystem:bfDASLUT:nonUniqueReceiverSize")
        Ms = sorted({int(c.data.shape[c.order.index("M")]) for c in chds})
        if len(Ms) != 1:
            raise DasError("Expected a single transmit size, but instead they have sizes [" + ",".join(str(v) for v in Ms) + "].",
                           "QUPS:UltrasoundSystem:bfDASLUT:nonUniqueTransmitSize")
        if len(chds) > 1 or chddim is not None:          # each ChannelData on its own (:4629), images concatenated along the array's dimension (:4670-4672)
            import torch
            outs = [self.bfDASLUT(c, tau_rx, tau_tx, *apods, apod=apod, fmod=fmod, interp=interp, keep_tx=keep_tx, keep_rx=keep_rx, prec=prec, bsize=bsize)
                    for c in chds]
            nd = max(max(o.ndim for o in outs), (chddim or 0) + 1)
            outs = [o.reshape(tuple(o.shape) + (1,) * (nd - o.ndim)) for o in outs]
            return outs[0] if chddim is None else torch.cat(outs, dim=chddim)
        chd = chds[0]
        if chd.order[:3] != "TNM":
            chd = chd.rectifyDims()
        Isz = self.scan.size
        tr, tt = np.asarray(tau_rx) if not hasattr(tau_rx, "shape") else tau_rx, tau_tx
        N, M = Ns[0], Ms[0]
        if tuple(tr.shape) != Isz + (N,):
            raise DasError(f"Expected a receive delay table of size {Isz + (N,)}, got {tuple(tr.shape)}.",
                           "QUPS:UltrasoundSystem:bfDASLUT:incompatibleReceiveDelayTable")
        if tuple(tt.shape) != Isz + (M,):
            raise DasError(f"Expected a transmit delay table of size {Isz + (M,)}, got {tuple(tt.shape)}.",
                           "QUPS:UltrasoundSystem:bfDASLUT:incompatibleTransmitDelayTable")
        ws = [np.asarray(a) for a in list(apods) + ([] if (np.isscalar(apod) and apod == 1) else [apod])]
        ws = [a.reshape(a.shape + (1,) * (5 - a.ndim)) for a in ws]
        if bsize is None:                                  # the reference's heuristic (:4573): blocks whose multiplied-out weights stay below 1 GB (8-byte entries)
            full = np.max([a.shape for a in ws], axis=0) if ws else np.ones(5, int)
            gb = 2.0 ** -30 * 8 * float(np.prod(full))
            bsize = max(1, min(M, int(np.floor(M / gb)) if gb > 0 else M))
        bsize = int(bsize)
        if bsize < 1:
            raise DasError("bsize must be a positive integer")
        sdim = set() if keep_rx else {"rx"}
        sdim |= set() if keep_tx else {"tx"}
        kw = {"prec": prec} if prec else {}
        w0 = None                                          # arrays without a transmit dimension: multiplied once (:4644)
        for a in ws:
            if a.shape[4] == 1:
                w0 = a if w0 is None else w0 * a
        t0a = np.asarray(chd.t0, dtype=np.float64).reshape(-1)
        parts, total = [], None
        for m0 in range(0, M, bsize):
            sl = slice(m0, min(M, m0 + bsize))
            w = w0
            for a in ws:                                   # per-block factors (:4647-4650)
                if a.shape[4] != 1:
                    w = a[..., sl] if w is None else w * a[..., sl]
            bm = sample2sep(chd.data[:, :, sl], t0a if t0a.size == 1 else t0a[sl], chd.fs, tr, tt[..., sl], interp=interp, w=w, sdim=sdim, fmod=fmod, **kw)
            if keep_tx:
                parts.append(bm)
            else:
                total = bm if total is None else total + bm
        b = _cat_tx(parts) if keep_tx else total          # I1 x I2 x I3 x [N] x [M] x F...
        return b.permute(0, 1, 2, *range(5, b.ndim), 3, 4)           # move the aperture dimensions to the end (:4663-4664)
