"""qups_amd -- MI355X-native delay-and-sum beamforming engine.

A from-scratch drop-in for ONE hot path of thorstone25/qups: ``UltrasoundSystem.DAS`` /
``bfDAS`` -> ``das_spec`` -> the ``DAS*`` device kernels.
The compute lives in ``libqdas.so`` (hand-written HIP for gfx950, C ABI in ``include/qdas.h``);
this package is the host-side mirror of the reference's interface for that path.
"""
from .das_spec import (DasError, DasPlan, DasProblem, MultiDevicePlan, build_problem, clear_plan_cache, das_spec, parse_options,
                       plan_cache_info, problem_key)  # noqa: F401

from .interpd import das_lut, sample2sep, wsinterpd2  # noqa: F401,E402
from . import apodization  # noqa: F401,E402
from . import preproc  # noqa: F401,E402
from .convd import convd, sosfilt  # noqa: F401,E402
from .ultrasound import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem  # noqa: F401,E402

__all__ = ["das_spec", "DasPlan", "MultiDevicePlan", "DasProblem", "DasError", "build_problem", "parse_options", "das_lut", "sample2sep",
           "wsinterpd2", "convd", "UltrasoundSystem", "Transducer", "Sequence", "Scan", "ChannelData"]
