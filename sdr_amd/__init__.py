"""sdr_amd -- MI355X (gfx950) implementation of RadioNode's baseband DSP hot path.

The product is the C-ABI shared library libsdrx.so (include/sdrx.h, built from sdr_amd/csrc
by `python -m sdr_amd.build`).  This package only loads it and mirrors the reference's class
interface (sdr_amd.host) and the device-resident entry points (sdr_amd.device) for tests and
bench.py.  There is no CPU fallback: without the library every call raises.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "host", "device"]
