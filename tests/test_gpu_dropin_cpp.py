"""The C++ drop-in headers (include/sdrx/*.h: Filter, AM<T>, FM<T>, BPSK, QPSK with the reference's
class names and public members) driven like the reference's own unit tests, on the GPU, and checked
against the CPU oracle.  tests/cpp/dropin_driver is built by __graft_entry__.build()."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "dropin_driver")
L = O.lib()


def test_cpp_dropin_classes(tmp_path, golden):
    assert os.path.exists(DRIVER), "tests/cpp/dropin_driver missing: run __graft_entry__.build()"
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    rng = np.random.default_rng(42)
    x = rng.integers(-127, 128, 5000).astype(np.float64)
    bits = np.array(golden["TEST_Modulator_Demodulator__BPSK__Full_ModDemod__input_stream"])
    sym = np.array(golden["TEST_QPSK__QPSK__Deconstructed_ModDemod__input_stream"])
    x.tofile(ind / "x.f64")
    bits.tofile(ind / "bits.u8")
    sym.tofile(ind / "sym.u8")
    F = np.load(os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz"))  # vectors the compiled reference produced
    for call in range(2):
        np.ascontiguousarray(F[f"bpsk2_iq{call}"]).tofile(ind / f"bpsk2_iq{call}.f64")
    r = subprocess.run([DRIVER, str(ind), str(outd)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rd = lambda name, dt=np.float64: np.fromfile(outd / name, dtype=dt)  # noqa: E731

    # Filter
    o = O.Fir(O.LPF, 64, 2.0e6, 1.0e5)
    want = np.concatenate([o.execute(x[:1000], fast=True), o.execute(x[1000:-1], fast=True), o.execute(x[-1:])])
    assert np.array_equal(rd("fir_lpf.f64"), want)
    assert np.array_equal(rd("fir_lpf_taps.f64"), o.taps)
    assert np.array_equal(rd("fir_hpf.f64"), O.Fir(O.HPF, 51, 2.0e6, 3.0e5).execute(x, fast=True))
    assert np.array_equal(rd("fir_bpf.f64"), O.Fir(O.BPF, 129, 2.0e6, 1.0e5, 4.0e5).execute(x, fast=True))
    assert list(rd("fir_err.f64")) == [-3.0, 0.0, 0.0]
    # AM (TEST_Modulator_Demodulator.cpp:123-162)
    mod = O.am_modulate(bits, 8000)
    assert np.array_equal(rd("am_mod.f64"), mod)
    assert np.array_equal(rd("am_bits.u8", np.uint8), O.am_demodulate(mod * 127.0, mod * 127.0, 8))
    assert np.array_equal(rd("am_bits.u8", np.uint8), bits)
    # FM
    fm_mod = O.fm_modulate(100.0, 20.0, 10.0, bits[:200], 100)
    assert np.array_equal(rd("fm_mod.f64"), fm_mod)
    assert np.array_equal(rd("fm_bits.u8", np.uint8), O.fm_demodulate(100.0, 20.0, 10.0, fm_mod, 100))
    # BPSK
    bm = np.zeros(8000)
    L.orc_bpsk_modulate(bits, 1000, bm, 8000)
    assert np.array_equal(rd("bpsk_mod.f64"), bm)
    st = O.psk_new(O.BPSK, 0x14EB, 16, O.MATH_SHARED)
    iq = np.repeat(bm, 2)
    want = np.concatenate([O.psk_demodulate(st, iq), O.psk_demodulate(st, iq)])
    got = rd("bpsk_dec.f64")
    assert np.array_equal(got, want)
    assert np.array_equal(got[:990], np.array(golden["TEST_Modulator_Demodulator__BPSK__Full_ModDemod__output_stream"])[:990])
    s = rd("bpsk_state.f64")
    assert list(s) == [st.nco_delay, st.reg_mu, st.dds_delay, st.lf_y, st.plf_y, float(st.cc[0])]
    # QPSK stages
    mr, mi = np.zeros(8000), np.zeros(8000)
    L.orc_qpsk_modulate(sym, 1000, mr, mi, 8000)
    assert np.array_equal(rd("qpsk_mod_re.f64"), mr) and np.array_equal(rd("qpsk_mod_im.f64"), mi)
    qiq = np.empty(16000)
    qiq[0::2], qiq[1::2] = mr, mi
    q = O.psk_new(O.QPSK, math_mode=O.MATH_SHARED)
    re, im = np.zeros(2000), np.zeros(2000)
    L.orc_psk_decode_pulse_shape(qiq, 8000, re, im)
    assert np.array_equal(rd("qpsk_corr_re.f64"), re) and np.array_equal(rd("qpsk_corr_im.f64"), im)
    L.orc_psk_timing(C.byref(q), re, im, 2000)
    assert np.array_equal(rd("qpsk_time_re.f64"), re) and np.array_equal(rd("qpsk_time_im.f64"), im)
    m = L.orc_psk_phase_correction(C.byref(q), re, im, 2000)
    assert np.array_equal(rd("qpsk_phase.f64"), re[:m])
    L.orc_psk_ambiguity(C.byref(q), re, m, None, 0)
    assert np.array_equal(rd("qpsk_amb.f64"), re[:m])
    amb = np.array(golden["TEST_QPSK__QPSK__Deconstructed_ModDemod__real_phaseambcorrect"])
    assert np.array_equal(rd("qpsk_amb.f64")[81:], amb[:919])
    assert np.array_equal(rd("qpsk_dec.f64"), O.psk_demodulate(O.psk_new(O.QPSK, math_mode=O.MATH_SHARED), qiq))

    # ---- round 2 members
    st = rd("statics.f64")
    rrc = np.array((C.c_double * 129).in_dll(L, "ORC_RRC"))
    assert list(st[:4]) == [1, -1, 1, 0]                                   # BPSK.cpp:14-15
    assert np.array_equal(st[4:133], rrc) and np.array_equal(st[133:262], rrc)  # BPSK::RRC_COEFF, QPSK::RRC_COEFF
    assert list(st[262:286]) == [1, -1, 1, -1, 1, 1, -1, -1, 0, 1, 2, 3, 2, 0, 3, 1, 3, 2, 1, 0, 1, 3, 0, 2]  # QPSK.cpp:14-19
    assert list(st[286:350]) == [0] * 48 + [3, 0, 2, 1, 0, 3, 2, 1, 2, 0, 1, 2, 3, 2, 2, 3] and st[350] == 129  # QPSK.cpp:20-28
    for name, fn, (n_in, n_out, ratio) in (("rs_raw.f64", L.orc_raw_upsample, (10, 80, 8.0)),
                                           ("rs_zeros.f64", L.orc_upsample_fill_w_zeros, (10, 80, 8.0)),
                                           ("rs_ramp.f64", L.orc_upsample, (10, 75, 7.5))):
        want = np.full(n_out + 4, 7.25)
        fn(x.copy(), n_in, want, n_out, ratio)
        assert np.array_equal(rd(name), want), name
    d = x[:1001].copy()
    r = L.orc_decimate(d, 1001, 4)
    assert np.array_equal(rd("rs_decimate.f64"), np.concatenate([d, [r]]))
    z = x[:100].copy()
    r = L.orc_decimate_block(z, 50, 3)
    assert np.array_equal(rd("rs_decimate_block.f64"), np.concatenate([z, [r]]))
    acc = np.full(10, 0.5)
    L.orc_fm_convolve(x, 1003, np.ascontiguousarray(x[2000:2150]), 150, acc, 100)
    assert np.array_equal(rd("fm_convolve.f64"), acc)
    for call in range(2):  # BPSK::demodulate overload 2 against what the reference's own class produced
        assert np.array_equal(rd(f"bpsk2_bits{call}.u8", np.uint8), F[f"bpsk2_bits{call}"])
        k = F[f"bpsk2_plot_re{call}"].size
        assert np.array_equal(rd(f"bpsk2_plot_re{call}.f64")[:k], F[f"bpsk2_plot_re{call}"])
        assert np.array_equal(rd(f"bpsk2_plot_im{call}.f64")[:k], F[f"bpsk2_plot_im{call}"])
        assert np.array_equal(rd(f"bpsk2_after{call}.f64"), F[f"bpsk2_iq_after{call}"])
