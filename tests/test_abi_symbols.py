"""CPU-side checks of the drop-in boundary: libsdrx.so loads and exports every symbol that
include/sdrx.h declares (no compute calls: there is no GPU here), the ctypes mirror of the ABI
structs has the C layout, and host-only entry points (filter design, PSK state init) agree with
the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from sdr_amd import _lib, build
    if build.stale():
        build.build(verbose=False)
    return _lib.lib()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sdrx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith(("sdrx_", "am_gpu", "fm_gpu", "ones_zeros"))))


def test_every_declared_symbol_is_exported(lib):
    from sdr_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), f"libsdrx.so does not export {name}"
        assert name in _lib.SIGNATURES, f"{name} missing from the ctypes signature table"
    assert sorted(_lib.SIGNATURES) == declared


def test_legacy_symbols_are_plain_c(lib):
    """The reference binds these unmangled (ProfileTest/main.cpp:13-35, rtlsdr_receiver.cpp:82-107)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "sdr_amd", "libsdrx.so")]).decode()
    for sym in ("am_gpu_initialize", "am_gpu_free", "am_gpu_demodulation", "fm_gpu_initialize", "fm_gpu_free",
                "fm_gpu_set_samples_per_bit", "fm_gpu_demodulation", "ones_zeros_demodulation"):
        assert re.search(rf"\bT {sym}\b", out), sym


def test_struct_layouts(lib):
    from sdr_amd import _lib
    assert C.sizeof(_lib.PskState) == 24 * 8 + 6 * 4 + 4 * 64 * 8
    assert _lib.PskState.uw.offset == 24 * 8 + 24
    assert C.sizeof(_lib.FirJob) == 8 + 2 * 8 + 3 * 8 + 8 + 8 + 3 * 8 + 2 * 8 + 2 * 8 + 8 + 8 + 8 + 8 + 8 + 8  # ... stream, dtype, u8_flags, u8_bias, n_before, taps_f32, am_work
    assert _lib.FirJob.dtype.offset == 128 and _lib.FirJob.taps_f32.offset == 152 and _lib.FirJob.am_work.offset == 160
    assert lib.sdrx_fir_state_doubles(64) == 128


@pytest.mark.parametrize("args", [(O.LPF, 64, 2.0e6, 1.0e5, None), (O.HPF, 51, 44.1, 4.0, None), (O.BPF, 129, 2.4e6, 1e5, 3e5),
                                  (O.LPF, 1, 2.0, 0.5, None), (O.LPF, 1001, 2e6, 1e5, None), (O.BPF, 64, 2e6, 2e5, 1e5)])
def test_fir_design_matches_oracle(lib, args):
    """sdrx_fir_design is host arithmetic (filt.cpp:45-154): same taps and error codes as the oracle."""
    ft, T, Fs, Fx, Fu = args
    taps = np.zeros(max(T, 1))
    err = lib.sdrx_fir_design(ft, 0 if Fu is None else 1, T, Fs, Fx, 0.0 if Fu is None else Fu, 1000,
                              taps.ctypes.data_as(C.c_void_p))
    oerr, otaps = O.fir_design(ft, T, Fs, Fx, Fu)
    assert err == oerr
    if err == 0:
        assert np.array_equal(taps, otaps)


def test_psk_state_init_matches_oracle(lib):
    from sdr_amd import _lib
    st = _lib.PskState()
    lib.sdrx_psk_state_init_bpsk(C.byref(st), 0xDEADBEEF600DF00D, 64)
    o = O.psk_new(O.BPSK, 0xDEADBEEF600DF00D, 64)
    assert bytes(st)[:192] == bytes(o)[:192]
    assert np.array_equal(np.array(st.uw), np.array(o.uw))
    assert list(st.cc) == list(o.cc) == [5, 5, 5, 5]
    lib.sdrx_psk_state_init_qpsk(C.byref(st), None, 16)
    o = O.psk_new(O.QPSK)
    assert np.array_equal(np.array(st.uw), np.array(o.uw))


def test_missing_library_fails_loudly(monkeypatch):
    from sdr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsdrx.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
