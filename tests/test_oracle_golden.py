"""Pin the CPU restatement (oracle/) against every golden vector the reference's own unit
tests hold for the path.  Each test mirrors one reference TEST (file:line in the docstring)
with the reference's own tolerance.  CPU only.
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

L = O.lib()


def G(golden, key):
    return np.array(golden[key])


# ----------------------------------------------------------------------------- AM / Resample

def test_am_modulate(golden):
    """TEST(AM, modulate) TEST_AM.cpp:14-35: 1000 bits -> 8000 samples, exact."""
    bits = G(golden, "TEST_AM__AM__modulate__inputStream")
    want = G(golden, "TEST_AM__AM__modulate__modulatedStream")
    got = O.am_modulate(bits, 8000)
    assert np.array_equal(got, want)


def test_am_demodulate(golden):
    """TEST(AM, demodulate) TEST_AM.cpp:38-58: amplitude x127, ds=8, exact bits."""
    x = G(golden, "TEST_AM__AM__demodulate__modulatedStream") * 127.0
    want = G(golden, "TEST_AM__AM__demodulate__ouputStream")
    got = O.am_demodulate(x, x, 8)
    assert np.array_equal(got[:1000], want)


def test_am_moddemod(golden):
    """TEST(AM, ModDemod) TEST_Modulator_Demodulator.cpp:123-162."""
    bits = G(golden, "TEST_Modulator_Demodulator__AM__ModDemod__inputStream")
    mod = O.am_modulate(bits, 8000)
    assert np.array_equal(mod, G(golden, "TEST_Modulator_Demodulator__AM__ModDemod__modulatedStream"))
    mod *= 127.0
    got = O.am_demodulate(mod, mod, 8)
    assert np.array_equal(got[:1000], G(golden, "TEST_Modulator_Demodulator__AM__ModDemod__ouputStream"))


def test_upsample_fill_w_zeros(golden):
    """TEST(Resample, upsample_with) TEST_Resample.cpp:8-28."""
    msg = G(golden, "TEST_Resample__Resample__upsample_with__message")
    out = np.zeros(64)
    L.orc_upsample_fill_w_zeros(msg, 8, out, 64, 8.0)
    assert np.array_equal(out, G(golden, "TEST_Resample__Resample__upsample_with__final_message"))


# ----------------------------------------------------------------------------- FM

def test_fm_modulate(golden):
    """TEST(FM, modulate) TEST_FM.cpp:14-37: FM(100,10,20,10), spb 100, 1e-9."""
    bits = G(golden, "TEST_FM__FM__modulate__inputStream")
    want = G(golden, "TEST_FM__FM__modulate__modulatedStream")
    got = O.fm_modulate(100.0, 20.0, 10.0, bits, 100)
    assert np.allclose(got, want, rtol=0, atol=1e-9)


def test_fm_demodulate(golden):
    """TEST(FM, demodulate) TEST_FM.cpp:40-59: FM(100,100,20,10).demodulate(.., 1000, 100) -> 10 bits."""
    x = G(golden, "TEST_FM__FM__demodulate__modulatedRealStream")
    want = G(golden, "TEST_FM__FM__demodulate__outputStream")
    got = O.fm_demodulate(100.0, 20.0, 10.0, x, 100)
    assert got.size == 10
    assert np.array_equal(got, want)


def test_fm_moddemod_resampled(golden):
    """TEST(FM, ModDemodResampled) TEST_FM.cpp:61-99: FM(100,50,20,10), /2 decimated, ds=50."""
    bits = G(golden, "TEST_FM__FM__ModDemodResampled__inputStream")
    stream = G(golden, "TEST_FM__FM__ModDemodResampled__modulatedStream")
    mod = O.fm_modulate(100.0, 20.0, 10.0, bits, 100)
    assert np.allclose(mod, stream, rtol=0, atol=1e-9)
    dec = stream.copy()
    assert L.orc_decimate(dec, 1000, 2) == 500
    got = O.fm_demodulate(50.0, 20.0, 10.0, dec[:500].copy(), 50)
    assert got.size == 10
    assert np.array_equal(got, G(golden, "TEST_FM__FM__ModDemodResampled__outputStream"))
    assert np.array_equal(got, bits)


# ----------------------------------------------------------------------------- BPSK stages

def test_bpsk_encode_lut(golden):
    """TEST(BPSK, encode_LUT) TEST_BPSK.cpp:18-36."""
    data = G(golden, "TEST_BPSK______BPSK_DataToBeDifferentiallyEncoded")[:1250]
    want = G(golden, "TEST_BPSK__BPSK__encode_LUT__encoded_data")[:1250]
    re = np.zeros(1250)
    L.orc_psk_encode_lut(O.BPSK, data, 1250, re, None)
    assert np.array_equal(re, want)


def test_bpsk_encode_pulse_shape(golden):
    """TEST(BPSK, encode_pulseShape) TEST_BPSK.cpp:38-83: 10 symbols -> 80 samples @1e-8."""
    sym = G(golden, "TEST_BPSK__BPSK__encode_pulseShape__Q_Channel")
    want = G(golden, "TEST_BPSK__BPSK__encode_pulseShape__pulseShaped")
    out = np.zeros(80)
    L.orc_psk_pulse_shape(sym, 10, out, 80 // 8)
    assert np.allclose(out, want, rtol=0, atol=1e-8)


def test_bpsk_decode_pulse_shape(golden):
    """TEST(BPSK, decode_pulseShape) TEST_BPSK.cpp:85-110: 800 -> 200 (198 checked) @1e-5."""
    ps = G(golden, "TEST_BPSK__BPSK__decode_pulseShape__pulseShaped")[:800]
    want = G(golden, "TEST_BPSK__BPSK__decode_pulseShape__RxCorrelated")
    iq = np.zeros(1600)
    iq[0::2] = ps
    re, im = np.zeros(200), np.zeros(200)
    assert L.orc_psk_decode_pulse_shape(iq, 800, re, im) == 200
    assert np.allclose(re[:198], want[:198], rtol=0, atol=1e-5)
    assert np.all(im == 0)


def _interp(st_interp, x, mu):
    """One InterpolationFilter step via the timing-loop internals is private; restate here (BPSK.cpp:319-344)."""
    K = -0.5
    d = st_interp
    V1 = d["d1"] - K * x
    V2 = V1 + d["d2"]
    V3 = V2 - d["d3"]
    V4 = K * x - d["d1"] + d["d4"]
    V5 = V4 + d["d2"]
    V6 = d["d3"] + V5
    V7 = V3 * mu + V6
    V8 = d["d5"] + mu * V7
    d["d3"], d["d2"], d["d1"], d["d5"], d["d4"] = d["d2"], d["d1"], K * x, d["d4"], x
    return V8


def test_bpsk_timing_loop(golden):
    """TEST(BPSK, symbol_timing_synchronization) TEST_BPSK.cpp:206-224: 1000 points @1e-8.
    Pins InterpolationFilter, GardnerDetector, LoopFilter, NcoControl and StrobeEnabledRegister
    together exactly as the reference's loop composes them."""
    re = G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxCorrelated_real")[:1000].copy()
    im = G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxCorrelated_imag")[:1000].copy()
    st = O.psk_new(O.BPSK)
    L.orc_psk_timing(C.byref(st), re, im, 1000)
    wr = G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxSymbolSynchronized_real")[:1000]
    wi = G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxSymbolSynchronized_imag")[:1000]
    assert np.allclose(re, wr, rtol=0, atol=1e-8)
    assert np.allclose(im, wi, rtol=0, atol=1e-8)


def test_bpsk_interpolation_filter_vector(golden):
    """TEST(BPSK, InterpolationFilter) TEST_BPSK.cpp:113-127: 200 points @1e-5 (stage formula)."""
    IN = G(golden, "TEST_BPSK__BPSK__InterpolationFilter__IN")
    MU = G(golden, "TEST_BPSK__BPSK__InterpolationFilter__MU")
    OUT = G(golden, "TEST_BPSK__BPSK__InterpolationFilter__OUT")
    d = dict(d1=0.0, d2=0.0, d3=0.0, d4=0.0, d5=0.0)
    for i in range(200):
        assert abs(OUT[i] - _interp(d, IN[i], MU[i])) < 1e-5


def test_bpsk_phase_correction(golden):
    """TEST(BPSK, rx_phase_correction) TEST_BPSK.cpp:367-383: 1000 -> 500 decisions (tol 0.1 == exact)."""
    re = G(golden, "TEST_BPSK__BPSK__rx_phase_correction__RxCorrelated_real")[:1000].copy()
    im = G(golden, "TEST_BPSK__BPSK__rx_phase_correction__RxCorrelated_imag")[:1000].copy()
    want = G(golden, "TEST_BPSK__BPSK__rx_phase_correction__RxSymbolSynchronized_real")
    for mode in (O.MATH_LIBM, O.MATH_SHARED):
        r, i = re.copy(), im.copy()
        st = O.psk_new(O.BPSK, math_mode=mode)
        assert L.orc_psk_phase_correction(C.byref(st), r, i, 1000) == 500
        assert np.array_equal(r[1:500], want[1:500])
        assert np.all(i[:500] == 0)


def test_bpsk_phase_ambiguity(golden):
    """TEST(BPSK, phase_ambiguity_correction) TEST_BPSK.cpp:386-399: default BPSK(0x4E,8), 1000 bits."""
    x = G(golden, "TEST_BPSK__BPSK__phase_ambiguity_correction__RX_PHASE_CORRECTED")[:1000].copy()
    want = G(golden, "TEST_BPSK__BPSK__phase_ambiguity_correction__RX_PHASE_AMBIGUITY_DETECTED")
    st = O.psk_new(O.BPSK)
    L.orc_psk_ambiguity(C.byref(st), x, 1000, None, 0)
    assert np.array_equal(x[529 + 8: 529 + 8 + 200], want[:200])


def test_bpsk_full_moddemod(golden):
    """TEST(BPSK, Full_ModDemod) TEST_Modulator_Demodulator.cpp:12-61: BPSK(0x14EB,16),
    modulate 1000 bits (800 samples @1e-6), demodulate -> 990 bits exact."""
    bits = G(golden, "TEST_Modulator_Demodulator__BPSK__Full_ModDemod__input_stream")
    want_mod = G(golden, "TEST_Modulator_Demodulator__BPSK__Full_ModDemod__modulated_stream")
    want_out = G(golden, "TEST_Modulator_Demodulator__BPSK__Full_ModDemod__output_stream")
    assert bits.size == 1000
    mod = np.zeros(8000)
    L.orc_bpsk_modulate(bits, 1000, mod, 8000)
    assert np.allclose(mod[:800], want_mod[:800], rtol=0, atol=1e-6)
    iq = np.repeat(mod, 2)  # real == imag (TEST_Modulator_Demodulator.cpp:41-46)
    for mode in (O.MATH_LIBM, O.MATH_SHARED):
        st = O.psk_new(O.BPSK, 0x14EB, 16, math_mode=mode)
        out = O.psk_demodulate(st, iq)
        assert out.size == 1000
        assert np.array_equal(out[:990], want_out[:990])


# ----------------------------------------------------------------------------- QPSK stages

def test_qpsk_encode_lut(golden):
    """TEST(QPSK, encode_LUT) TEST_QPSK.cpp:12-36."""
    data = G(golden, "TEST_QPSK__QPSK__encode_LUT__QPSK_DataToBeDifferentiallyEncoded")
    re, im = np.zeros(100), np.zeros(100)
    L.orc_psk_encode_lut(O.QPSK, data, 100, re, im.ctypes.data)
    assert np.array_equal(re, G(golden, "TEST_QPSK__QPSK__encode_LUT__encoded_real_data"))
    assert np.array_equal(im, G(golden, "TEST_QPSK__QPSK__encode_LUT__encoded_imag_data"))


def test_qpsk_encode_pulse_shape(golden):
    """TEST(QPSK, encode_pulseShape) TEST_QPSK.cpp:38-71: 1000 symbols -> 8000 @1e-8 on both rails."""
    for ch in ("Q", "I"):
        sym = G(golden, f"TEST_QPSK__QPSK__encode_pulseShape__{ch}_Channel")
        want = G(golden, f"TEST_QPSK__QPSK__encode_pulseShape__pulseShaped_{ch}_Channel")
        out = np.zeros(8000)
        L.orc_psk_pulse_shape(sym, 1000, out, 1000)
        assert np.allclose(out, want, rtol=0, atol=1e-8)


def test_qpsk_decode_pulse_shape(golden):
    """TEST(QPSK, decode_pulseShape) TEST_QPSK.cpp:74-108: 100 complex -> 25 (20 checked) @1e-8."""
    iq = np.zeros(200)
    iq[0::2] = G(golden, "TEST_QPSK__QPSK__decode_pulseShape__pulseShaped_Real")
    iq[1::2] = G(golden, "TEST_QPSK__QPSK__decode_pulseShape__pulseShaped_Imag")
    re, im = np.zeros(25), np.zeros(25)
    assert L.orc_psk_decode_pulse_shape(iq, 100, re, im) == 25
    assert np.allclose(re[:20], G(golden, "TEST_QPSK__QPSK__decode_pulseShape__RxCorrelated_Real"), rtol=0, atol=1e-8)
    assert np.allclose(im[:20], G(golden, "TEST_QPSK__QPSK__decode_pulseShape__RxCorrelated_Imag"), rtol=0, atol=1e-8)


def test_qpsk_timing_loop(golden):
    """TEST(QPSK, symbol_timing_synchronization) TEST_QPSK.cpp:207-241: 2000 points @1e-5."""
    re = G(golden, "TEST_QPSK__QPSK__symbol_timing_synchronization__RxCorrelated_real").copy()
    im = G(golden, "TEST_QPSK__QPSK__symbol_timing_synchronization__RxCorrelated_imag").copy()
    st = O.psk_new(O.QPSK)
    L.orc_psk_timing(C.byref(st), re, im, 2000)
    assert np.allclose(re, G(golden, "TEST_QPSK__QPSK__symbol_timing_synchronization__RxSymbolSynchronized_real"), rtol=0, atol=1e-5)
    assert np.allclose(im, G(golden, "TEST_QPSK__QPSK__symbol_timing_synchronization__RxSymbolSynchronized_imag"), rtol=0, atol=1e-5)


def test_qpsk_phase_correction(golden):
    """TEST(QPSK, rx_phase_correction) TEST_QPSK.cpp:341-374: 1000 -> 500 symbols @1e-8 (exact)."""
    re0 = G(golden, "TEST_QPSK__QPSK__rx_phase_correction__RxCorrelated_real")[:1000]
    im0 = G(golden, "TEST_QPSK__QPSK__rx_phase_correction__RxCorrelated_imag")[:1000]
    want = G(golden, "TEST_QPSK__QPSK__rx_phase_correction__RxSymbolSynchronized_real")
    for mode in (O.MATH_LIBM, O.MATH_SHARED):
        re, im = re0.copy(), im0.copy()
        st = O.psk_new(O.QPSK, math_mode=mode)
        assert L.orc_psk_phase_correction(C.byref(st), re, im, 1000) == 500
        assert np.array_equal(re[1:500], want[1:500])


def test_qpsk_phase_ambiguity(golden):
    """TEST(QPSK, phase_ambiguity_correction) TEST_QPSK.cpp:376-407: 400 symbols, data from 161."""
    x = G(golden, "TEST_QPSK__QPSK__phase_ambiguity_correction__RX_PHASE_CORRECTED")[:400].copy()
    want = G(golden, "TEST_QPSK__QPSK__phase_ambiguity_correction__RX_PHASE_AMBIGUITY_DETECTED")
    st = O.psk_new(O.QPSK)
    L.orc_psk_ambiguity(C.byref(st), x, 400, None, 0)
    assert np.array_equal(x[161:400], want[: 400 - 161])


@pytest.mark.parametrize("src", ["TEST_QPSK__QPSK__Deconstructed_ModDemod", "TEST_Modulator_Demodulator__QPSK__Full_ModDemod"])
def test_qpsk_deconstructed_and_full(golden, src):
    """TEST(QPSK, Deconstructed_ModDemod) TEST_QPSK.cpp:409-507 and TEST(QPSK, Full_ModDemod)
    TEST_Modulator_Demodulator.cpp:63-120: every stage pinned, then end to end."""
    sym = G(golden, f"{src}__input_stream")
    mr, mi = np.zeros(8000), np.zeros(8000)
    L.orc_qpsk_modulate(sym, 1000, mr, mi, 8000)
    assert np.allclose(mr, G(golden, f"{src}__real_modulated_stream"), rtol=0, atol=1e-8)
    assert np.allclose(mi, G(golden, f"{src}__imag_modulated_stream"), rtol=0, atol=1e-8)
    iq = np.empty(16000)
    iq[0::2], iq[1::2] = mr, mi
    for mode in (O.MATH_LIBM, O.MATH_SHARED):
        st = O.psk_new(O.QPSK, math_mode=mode)
        re, im = np.zeros(2000), np.zeros(2000)
        assert L.orc_psk_decode_pulse_shape(iq, 8000, re, im) == 2000
        assert np.allclose(re, G(golden, f"{src}__real_correlated"), rtol=0, atol=1e-7)
        assert np.allclose(im, G(golden, f"{src}__imag_correlated"), rtol=0, atol=1e-7)
        L.orc_psk_timing(C.byref(st), re, im, 2000)
        assert np.allclose(re, G(golden, f"{src}__real_timesync"), rtol=0, atol=1e-7)
        assert np.allclose(im, G(golden, f"{src}__imag_timesync"), rtol=0, atol=1e-7)
        assert L.orc_psk_phase_correction(C.byref(st), re, im, 2000) == 1000
        pc = G(golden, f"{src}__real_phasecorrect")
        assert np.array_equal(re[:1000], pc)
        L.orc_psk_ambiguity(C.byref(st), re, 1000, None, 0)
        amb = G(golden, f"{src}__real_phaseambcorrect")
        assert np.array_equal(re[81:1000], amb[: 1000 - 81])
        # end to end through demodulate()
        st2 = O.psk_new(O.QPSK, math_mode=mode)
        out = O.psk_demodulate(st2, iq)
        assert out.size == 1000
        assert np.array_equal(out[81:], amb[: 1000 - 81])


# ----------------------------------------------------------------------------- shared math vs libm

def test_shared_math_close_to_libm():
    """include/sdrx_math.h must sit within a few ulp of libm so that the GPU-side definition
    makes the same decisions as the reference's libm calls (measured, not assumed)."""
    rng = np.random.default_rng(7)
    re = rng.uniform(-100, 100, 200000)
    im = rng.uniform(-100, 100, 200000)
    a = O.fm_discriminator(re, im, math_mode=O.MATH_LIBM)
    b = O.fm_discriminator(re, im, math_mode=O.MATH_SHARED)
    assert np.max(np.abs(a - b)) <= 4 * np.finfo(np.float64).eps * np.pi
    want = np.angle((re[1:] + 1j * im[1:]) * np.conj(re[:-1] + 1j * im[:-1]))
    assert np.max(np.abs(a[1:] - want)) < 1e-14


@pytest.mark.parametrize("form", [0, 1])
def test_shared_sincos_forms_close_to_libm(form):
    """sdrx_sincos (form 0: FFT twiddles) and sdrx_sincos_tab (form 1: the PSK carrier loop, table + short polynomials):
    absolute distance from libm below 2.5e-16 (results are <= 1 in magnitude: about 1 ulp of 1) over the carrier loop's
    working range and up to the 1.5e6 limit of the straight-line path; exact values at 0; NaN for non-finite input."""
    rng = np.random.default_rng(11 + form)
    x = np.concatenate([rng.uniform(-7, 7, 300000), rng.uniform(-1000, 1000, 300000), rng.uniform(-1.5e6, 1.5e6, 300000),
                        np.arange(-64, 65) * (np.pi / 32), [0.0, -0.0]])
    sn, cs = np.zeros_like(x), np.zeros_like(x)
    L.orc_sincos(x, x.size, form, sn, cs)
    assert np.max(np.abs(sn - np.sin(x))) < 2.5e-16 and np.max(np.abs(cs - np.cos(x))) < 2.5e-16
    assert sn[-2] == 0.0 and cs[-2] == 1.0
    bad = np.array([np.inf, -np.inf, np.nan])
    sn, cs = np.zeros(3), np.zeros(3)
    with np.errstate(all="ignore"):
        L.orc_sincos(bad, 3, form, sn, cs)
    assert np.all(np.isnan(sn)) and np.all(np.isnan(cs))
    big = np.array([1.6e6, -3.3e7, 1e9])  # the rare path: coarse reduction first
    sn, cs = np.zeros(3), np.zeros(3)
    L.orc_sincos(big, 3, form, sn, cs)
    assert np.max(np.abs(sn - np.sin(big))) < 1e-6 and np.max(np.abs(cs - np.cos(big))) < 1e-6


def test_chain_streaming_equals_one_shot():
    """Filter state is carried across execute() calls (filt.cpp:256-259): chunked == one shot."""
    rng = np.random.default_rng(3)
    n = 4096
    re = rng.integers(-127, 128, n).astype(np.float64)
    im = rng.integers(-127, 128, n).astype(np.float64)
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    assert err == 0
    hI, hQ, pre = np.zeros(63), np.zeros(63), np.zeros(2)
    out = np.zeros(n // 10 + 2)
    m = L.orc_chain_fir_decim_disc(taps, 64, hI, hQ, pre, re, im, n, 10, out, O.MATH_SHARED)
    hI2, hQ2, pre2 = np.zeros(63), np.zeros(63), np.zeros(2)
    parts = []
    for k in range(0, n, 1000):
        r, i = re[k:k + 1000].copy(), im[k:k + 1000].copy()
        o = np.zeros(102)
        mm = L.orc_chain_fir_decim_disc(taps, 64, hI2, hQ2, pre2, r, i, r.size, 10, o, O.MATH_SHARED)
        parts.append(o[:mm])
    assert np.array_equal(np.concatenate(parts)[:m], out[:m])


# ---------------------------------------------------------------- oracle helpers without a reference KAT

def test_u8_ingest_restatement():
    """orc_u8_to_iq / orc_u8_rotate_90 follow rtlsdr_receiver.cpp:1587-1588 and Baseband_Stream.cpp:82-148.
    The reference holds no vector for them (parity unpinned): check the defining properties instead --
    byte - 127 on unsigned bytes, char semantics for load_buffer, and rotate_90 = multiplication of the
    sample sequence by 1, j, -1, -j when a byte's negation is 255 - byte."""
    raw = np.arange(512, dtype=np.uint8).reshape(-1, 2)  # every byte value (wraps)
    re, im = O.u8_to_iq(raw)
    assert np.array_equal(re, raw[:, 0].astype(np.float64) - 127.0)
    assert np.array_equal(im, raw[:, 1].astype(np.float64) - 127.0)
    re, im = O.u8_to_iq(raw, 127.4, signed_char=True)
    assert np.array_equal(re, raw[:, 0].astype(np.int8).astype(np.float64) - 127.4)
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, (64, 2), dtype=np.uint8)
    r0, i0 = O.u8_to_iq(raw, 127.5)
    r1, i1 = O.u8_to_iq(raw, 127.5, rotate90=True)  # with bias 127.5, 255 - b maps to exactly -(b - 127.5)
    z0, z1 = r0 + 1j * i0, r1 + 1j * i1
    assert np.array_equal(z1, z0 * np.tile([1, 1j, -1, -1j], 16))


def test_fused_fir_variant_is_close_to_the_reference_sum():
    """orc_fir_execute_fma defines the library's fast modes; it must stay within 1e-12 * sum|h||x| of Filter::execute."""
    rng = np.random.default_rng(1)
    x = rng.integers(-127, 128, 5000).astype(np.float64)
    for T in (1, 2, 64, 129, 1000):
        a, b = O.Fir(O.LPF, T, 2.0e6, 1.0e5), O.Fir(O.LPF, T, 2.0e6, 1.0e5)
        scale = np.convolve(np.abs(x), np.abs(a.taps))[: x.size] + 1e-300
        for lo, hi in ((0, 1234), (1234, 5000)):
            assert np.max(np.abs(a.execute(x[lo:hi]) - b.execute_fma(x[lo:hi])) / scale[lo:hi]) <= 1e-12


# ---------------------------------------------------------------- bit-domain back end

FRAMED_MESSAGE_64 = [1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     1, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 1]


def test_crc_check():
    """TEST(CRC, crc_check): TEST_crc.cpp:8-17"""
    assert O.crc8([0x83, 0x01]) == 0x17


def test_packetframer_framed_message():
    """TEST(PacketFramer, FramedMessage): TEST_PacketFramer.cpp:30-54 (crc disabled, 4 data bytes, zeros after)"""
    bits = O.frame([0x83, 0x01, 0x00, 0xAB], crc_enable=False)
    assert bits.size == (250 + 4) * 8
    assert list(bits[:64]) == FRAMED_MESSAGE_64 and not bits[64:80].any()


def _feed_one_by_one(first64, crc_enable):
    """The reference tests append one bit at a time to a ring buffer of exactly one packet and call
    deserialize after each: only the call on the full buffer (2032 bits) can return non-zero.  The batch
    form equals that last call once one more bit has arrived (the receiver's loop wants count > 2032)."""
    stream = np.zeros(2033, dtype=np.uint8)
    stream[:64] = first64
    for n in (1, 64, 1000, 2032):  # not enough bits: nothing happens, nothing is consumed
        r = O.deframe(stream[:n], crc_enable)
        assert r["events"] == 0 and r["consumed"] == 0
    return O.deframe(stream, crc_enable)


def test_packetframer_good_deframed_packet():
    """TEST(PacketFramer, GoodDeframedPacket): TEST_PacketFramer.cpp:57-125"""
    r = _feed_one_by_one(FRAMED_MESSAGE_64, crc_enable=False)
    assert r["events"] == 1 and r["kind"][0] == 1 and r["pos"][0] == 0 and r["len"][0] == 4
    assert list(r["payload"][0][:4]) == [0x83, 0x01, 0x00, 0xAB] and not r["payload"][0][4:].any()
    assert r["consumed"] == 2032


def test_packetframer_bad_crc():
    """TEST(PacketFramer, BadCRC): TEST_PacketFramer.cpp:128-195 -- return -1, header valid, one bad bit"""
    r = _feed_one_by_one(FRAMED_MESSAGE_64, crc_enable=True)
    assert r["events"] >= 1 and r["kind"][0] == 3 and r["pos"][0] == 0


def test_packetframer_bad_header():
    """TEST(PacketFramer, BadHeader): TEST_PacketFramer.cpp:198-265 -- no header: returns 0 after 500 removed bits"""
    bad = list(FRAMED_MESSAGE_64)
    bad[0] = bad[1] = 0
    r = _feed_one_by_one(bad, crc_enable=True)
    assert r["events"] == 0 and r["zero_returns"] == 1 and r["consumed"] == 500


def test_packetframer_bad_length():
    """TEST(PacketFramer, BadLength): TEST_PacketFramer.cpp:268-335 -- length 255 > 250: return -1"""
    bad = list(FRAMED_MESSAGE_64)
    bad[16:24] = [1] * 8
    r = _feed_one_by_one(bad, crc_enable=False)
    assert r["events"] >= 1 and r["kind"][0] == 2 and r["pos"][0] == 0 and r["len"][0] == 255



# ---- DSP primitives.  The reference holds no vector for them: the restatements are pinned to
# the compiled reference in the build container (test_oracle_vs_ref.py::test_dsp_primitives_vs_reference); here,
# their defining properties, which also run where /root/reference is absent.

def test_dsp_primitive_restatements_properties():
    rng = np.random.default_rng(77)
    x = rng.normal(0, 5, 400) + 9.0
    # DC blocker: y(0) = x(0); the difference equation holds sample for sample; a constant input decays like R^n
    y = O.dc_block(x, 0.995)
    assert y[0] == x[0]
    assert np.array_equal(y[1:], np.array([x[i] - x[i - 1] + 0.995 * y[i - 1] for i in range(1, 400)]))
    c = O.dc_block(np.full(50, 3.0), 0.5)
    assert np.allclose(c, 3.0 * 0.5 ** np.arange(50), rtol=1e-12, atol=0)
    # first-order low pass: step response 1 - (1 - ePow)^(n+1); state carried over a split call
    e = O.lib().orc_lowpass1_coef(3000.0, 1.0 / 48000.0)
    assert e == 1.0 - np.exp(-3000.0 / 48000.0)
    st = np.zeros(1)
    s1 = O.lowpass1(np.ones(100), e, st)
    assert np.allclose(s1, 1 - (1 - e) ** np.arange(1, 101), rtol=1e-12)
    st2 = np.zeros(1)
    parts = np.concatenate([O.lowpass1(np.ones(37), e, st2), O.lowpass1(np.ones(63), e, st2)])
    assert np.array_equal(parts, s1) and st2[0] == st[0]
    # notch: first two samples of a block untouched and not in the delay line; difference equation otherwise
    coef = O.notch_design(0.9, 0.05)
    st = np.zeros(4)
    y = O.notch(x, coef, st)
    assert np.array_equal(y[:2], x[:2])
    x1 = x2 = y1 = y2 = 0.0
    for i in range(2, 400):
        w = coef[0] * x[i] + coef[1] * x1 + coef[2] * x2 + coef[3] * y1 + coef[4] * y2
        assert y[i] == w
        x2, x1, y2, y1 = x1, x[i], y1, w
    assert np.array_equal(st, [x1, x2, y1, y2])
    # integrator: running sum scaled by the time step
    t = 1.0 / 48000.0
    y = O.integrate(x, t)
    assert y[0] == x[0] and np.allclose(y, x[0] + t * np.concatenate([[0.0], np.cumsum(x[1:])]), rtol=1e-12)
    # normalize: integer maximum, affine map
    y, m = O.normalize(x, 0.0)
    assert m == np.floor(np.abs(x).max()) and np.array_equal(y, (x / m) * 2.0 - 1.0)
    assert O.normalize(x, 1e4)[1] == 1e4
    # fifth-order decimator: steady state is the symmetric 1-5-10-10-5-1 window over every second sample, / 11
    st = np.zeros(6)
    y, r = O.fifth_order_decimate(x, 2, st)
    assert r == 200
    k = 10
    a, b, c_, d_, e_, f = x[2 * k - 6], x[2 * k - 4], x[2 * k - 4], x[2 * k - 2], x[2 * k - 2], x[2 * k]
    assert y[k] == ((1 * (a + f)) + ((b + e_) * 5) + ((c_ + d_) * 10)) / 11
    assert np.array_equal(y[200:], x[200:])  # samples beyond the last written one keep their values
    # fft: against numpy, both math modes; linearity
    z = rng.normal(0, 1, 256) + 1j * rng.normal(0, 1, 256)
    for mode in (O.MATH_LIBM, O.MATH_SHARED):
        assert np.allclose(O.fft(z, mode), np.fft.fft(z), rtol=0, atol=1e-10)
    assert np.max(np.abs(O.fft(z, O.MATH_SHARED) - O.fft(z, O.MATH_LIBM))) <= 1e-12 * 8 * np.abs(np.fft.fft(z)).max()


def test_rtlfm_full_demod_restatement_properties():
    """rtl_fm's full_demod restatement (parity unpinned: rtl_fm.c needs librtlsdr and the vendored copy does not
    compile): with every option off it is the plain chain; each stage's defining property on top of that."""
    rng = np.random.default_rng(3)
    n = 4000
    ph = np.cumsum(rng.normal(0.03, 0.05, n))
    raw = np.zeros((n, 2), dtype=np.uint8)
    raw[:, 0] = np.clip(np.round(60 * np.cos(ph) + 127), 0, 255)
    raw[:, 1] = np.clip(np.round(60 * np.sin(ph) + 127), 0, 255)

    def run(**kw):
        st, post, pcm = np.zeros(5, dtype=np.int32), np.zeros(4, dtype=np.int32), np.zeros(1000, dtype=np.int16)
        m = O.rtlfm_full(raw, 10, st, post, pcm, **kw)
        return m, st, post, pcm

    st0 = np.zeros(5, dtype=np.int32)
    plain = O.rtlfm_chain(raw, 10, st0)
    m, st, post, pcm = run()
    assert m == plain.size == 400 and np.array_equal(pcm[:m], plain) and np.array_equal(st, st0) and post[3] == 400
    # low_pass_simple: sums of 4 (int16 wrap), one duplicate written past the output, result_len = 100
    m4, _, _, p4 = run(post_downsample=4)
    assert m4 == 100
    assert np.array_equal(p4[:100], plain.astype(np.int32).reshape(100, 4).sum(axis=1).astype(np.int16))
    assert p4[101] == plain[100]  # signal2[i/step + 1] = signal2[i/step]: the untouched sample 100 copied to 101
    # squelch: the lowpassed rms of this signal is about 10 * 60: closed above that, open below; a closed call gives zeros
    m, st_s, post_s, p_s = run(squelch_level=1000)
    assert post_s[2] == 1 and not p_s[:m].any() and st_s[3] == 0 and st_s[4] == 0
    m, _, post_o, p_o = run(squelch_level=300)
    assert post_o[2] == 0 and np.array_equal(p_o[:m], plain)
    # de-emphasis: avg follows the input with the rounded division of rtl_fm.c:607-613
    m, _, post_d, p_d = run(deemph_a=3)
    avg = 0
    for i in range(m):
        d = int(plain[i]) - avg
        q = (d + 1) if d > 0 else (d - 1)
        avg += int(q / 3)  # C division truncates towards zero
        assert p_d[i] == np.int16(avg)
    assert post_d[0] == avg
    # dc block: the block's mean, averaged 1:9 with the carried one (0), is subtracted
    m, _, post_b, p_b = run(dc_block=True)
    mean = int(int(plain.astype(np.int64).sum()) / m)
    a = int((mean + 0) / 10)
    assert post_b[1] == a and np.array_equal(p_b[:m], (plain.astype(np.int32) - a).astype(np.int16))


def test_deemph_magic_division_is_exact_on_its_whole_domain():
    """The de-emphasis kernel (rtlfm_post_kernel) replaces rtl_fm's `(d +- a/2) / a` (C division, truncating) by a
    signed multiply-high with magic = floor(2^32 / a) + 1: trunc(x / a) = (x * magic) >> 32 for x >= 0 and that + 1 for
    x < 0.  Checked here for every divisor the kernel takes that path for (3 <= a < 2048) and every x it can meet
    there (avg and the sample both in the int16 range: |d| <= 65535, |x| <= 65535 + 1023)."""
    x = np.arange(-(65535 + 1023), 65535 + 1024, dtype=np.int64)
    neg = (x < 0).astype(np.int64)
    for a in range(3, 2048):
        magic = 0x100000000 // a + 1
        assert magic < 1 << 31
        got = ((x * magic) >> 32) + neg
        want = np.sign(x) * (np.abs(x) // a)  # truncation towards zero
        assert np.array_equal(got, want), a
