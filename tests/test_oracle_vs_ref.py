"""Validate the C restatement (oracle/sdr_oracle.c) bit-for-bit against the REFERENCE's own
classes compiled from /root/reference (oracle/_ref/libsdr_ref.so, `make -C oracle ref`).
Skipped where that library was not built (the reference sources do not travel); the
committed fixtures in tests/golden/ref_fixtures.npz carry the same comparison everywhere
(tests/test_ref_fixtures.py).  CPU only.
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from signals import psk_test_signal

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (reference absent)")

L = O.lib()


def _ref_fir(ftype, T, Fs, Fx, Fu=None):
    R = O.ref()
    return R.ref_fir_new(ftype, 0 if Fu is None else 1, T, Fs, Fx, 0.0 if Fu is None else Fu)


@pytest.mark.parametrize("ftype,Fx,Fu", [(O.LPF, 1.0e5, None), (O.HPF, 3.0e5, None), (O.BPF, 1.0e5, 4.0e5)])
@pytest.mark.parametrize("T", [1, 2, 51, 64, 129, 1000])
def test_fir_taps_and_streaming(ftype, Fx, Fu, T):
    """filt.cpp:45-166,236-273: taps identical; execute() identical across 3 ragged chunks."""
    R = O.ref()
    h = _ref_fir(ftype, T, 2.0e6, Fx, Fu)
    assert R.ref_fir_error(h) == 0
    mine = O.Fir(ftype, T, 2.0e6, Fx, Fu)
    assert mine.err == 0
    taps = np.zeros(T)
    R.ref_fir_get_taps(h, taps)
    assert np.array_equal(taps, mine.taps)
    rng = np.random.default_rng(T * 7 + ftype)
    x = rng.integers(-127, 128, 3000).astype(np.float64) + rng.uniform(-0.5, 0.5, 3000)
    mine_fast = O.Fir(ftype, T, 2.0e6, Fx, Fu)
    for lo, hi in ((0, 1), (1, 1300), (1300, 3000)):
        a = x[lo:hi].copy()
        R.ref_fir_execute(h, a, a.size)
        b = mine.execute(x[lo:hi])
        c = mine_fast.execute(x[lo:hi], fast=True)
        assert np.array_equal(a, b)
        assert np.array_equal(a, c)
    R.ref_fir_free(h)


@pytest.mark.parametrize("args,code", [
    ((O.LPF, 64, 0.0, 1.0, None), -1), ((O.LPF, 64, 2.0e6, 0.0, None), -2), ((O.LPF, 64, 2.0e6, 1.0e6, None), -2),
    ((O.LPF, 0, 2.0e6, 1.0e5, None), -3), ((O.LPF, 1001, 2.0e6, 1.0e5, None), -3), ((O.BPF, 64, 2.0e6, 1.0e5, None), -5),
    ((O.BPF, 64, 0.0, 1.0, 2.0), -10), ((O.BPF, 64, 2.0e6, 2.0e5, 1.0e5), -11), ((O.BPF, 64, 2.0e6, 0.0, 1.0e5), -12),
    ((O.BPF, 64, 2.0e6, 1.0e5, 1.0e6), -13), ((O.BPF, 1024, 2.0e6, 1.0e5, 2.0e5), -14), ((O.LPF, 64, 2.0e6, 1.0e5, 2.0e5), -16),
])
def test_fir_error_codes(args, code):
    """filt.cpp:54-67,85-99 / filt.h:91-102."""
    R = O.ref()
    ftype, T, Fs, Fx, Fu = args
    h = _ref_fir(ftype, T, Fs, Fx, Fu)
    assert R.ref_fir_error(h) == code
    assert O.Fir(ftype, T, Fs, Fx, Fu).err == code
    R.ref_fir_free(h)


def test_am_matches_reference():
    """AM.h:41-77 on the radio value range, ds in {1, 8, 7} (ragged), incl. exact-threshold points."""
    R = O.ref()
    rng = np.random.default_rng(11)
    n = 10007
    re = rng.integers(-127, 128, n).astype(np.float64)
    im = rng.integers(-127, 128, n).astype(np.float64)
    re[:4] = [80.0, 48.0, 79.99999999999999, 0.0]
    im[:4] = [0.0, 64.0, 0.0, 80.00000000000001]
    for ds in (1, 8, 7):
        a = np.zeros(n, dtype=np.uint8)
        na = R.ref_am_demodulate(re, im, a, n, ds)
        b = O.am_demodulate(re, im, ds)
        assert na == b.size
        assert np.array_equal(a[:na], b)
    bits = rng.integers(0, 2, 1000).astype(np.uint8)
    for len_mod in (8000, 8003, 2500):
        a = np.full(len_mod, -7.0)
        b = np.full(len_mod, -7.0)
        R.ref_am_modulate(bits, 1000, a, len_mod)
        O.am_modulate(bits, len_mod, out=b)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", [O.BPSK, O.QPSK])
@pytest.mark.parametrize("phase_deg", [0.0, 90.0, 180.0, 270.0, 33.0])
def test_psk_chain_matches_reference(kind, phase_deg):
    """BPSK.cpp / QPSK.cpp end to end over 3 consecutive blocks (state carried): decisions,
    every loop state double and the confidence counters must be IDENTICAL to the reference
    (libm mode), and the shared-math mode must make the same decisions."""
    R = O.ref()
    uw = 0xDEADBEEF600DF00D
    if kind == O.BPSK:
        h = R.ref_bpsk_new(uw, 64)
        st = O.psk_new(O.BPSK, uw, 64, O.MATH_LIBM)
        st2 = O.psk_new(O.BPSK, uw, 64, O.MATH_SHARED)
    else:
        h = R.ref_qpsk_new(None, 16)
        st = O.psk_new(O.QPSK, None, 16, O.MATH_LIBM)
        st2 = O.psk_new(O.QPSK, None, 16, O.MATH_SHARED)
    for blk in range(3):
        iq = psk_test_signal(kind, nsym=1024, seed=100 * blk + int(phase_deg), phase_deg=phase_deg, snr_db=20.0,
                             timing_offset=3, uw_bpsk=uw)
        n = iq.size // 2
        a = np.zeros(n)
        na = (R.ref_bpsk_demodulate if kind == O.BPSK else R.ref_qpsk_demodulate)(h, iq, n, a, None)
        b = O.psk_demodulate(st, iq)
        b2 = O.psk_demodulate(st2, iq)
        assert na == b.size == n // 8
        assert np.array_equal(a[:na], b)
        assert np.array_equal(b, b2)
        s = np.zeros(24)
        cc = np.zeros(4, dtype=np.int32)
        (R.ref_bpsk_state if kind == O.BPSK else R.ref_qpsk_state)(h, s, cc)
        assert np.array_equal(s, st.loop_state())
        ncc = 1 if kind == O.BPSK else 4
        assert list(cc[:ncc]) == list(st.cc)[:ncc]
        assert np.allclose(st.loop_state(), st2.loop_state(), rtol=0, atol=1e-9)
    (R.ref_bpsk_free if kind == O.BPSK else R.ref_qpsk_free)(h)


def test_psk_modulators_match_reference():
    """BPSK.cpp:96-146 / QPSK.cpp:127-188 including the byte-count memset: outputs beyond the first
    len_mod/8 doubles accumulate onto what the caller left there."""
    R = O.ref()
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, 500).astype(np.uint8)
    sym = rng.integers(0, 4, 500).astype(np.uint8)
    prior = rng.uniform(-1, 1, 4000)
    h = R.ref_bpsk_new(0x4E, 8)
    a, b = prior.copy(), prior.copy()
    R.ref_bpsk_modulate(h, bits, 500, a, 4000)
    L.orc_bpsk_modulate(bits, 500, b, 4000)
    assert np.array_equal(a, b)
    R.ref_bpsk_free(h)
    q = R.ref_qpsk_new(None, 16)
    ar, ai, br, bi = prior.copy(), prior[::-1].copy(), prior.copy(), prior[::-1].copy()
    R.ref_qpsk_modulate(q, sym, 500, ar, ai, 4000)
    L.orc_qpsk_modulate(sym, 500, br, bi, 4000)
    assert np.array_equal(ar, br) and np.array_equal(ai, bi)
    R.ref_qpsk_free(q)


def test_resample_helpers_match_reference():
    """Resample.h:33-52 (incl. the tail-clear quirk) and Resample.cpp:24-36."""
    R = O.ref()
    rng = np.random.default_rng(9)
    for n, by in ((1000, 2), (1001, 4), (17, 5), (8, 8), (3, 10)):
        x = rng.uniform(-1, 1, n)
        a, b = x.copy(), x.copy()
        assert R.ref_decimate(a, n, by) == L.orc_decimate(b, n, by)
        assert np.array_equal(a, b)


def test_every_resample_function_matches_reference():
    """Resample.h:12-54 / Resample.cpp:10-70, all six: upsample (ramp), raw_upsample, upsample_fill_w_zeros,
    decimate<T>, decimate(BBP_Block*) (drops the first point, n_points = kept + 1), decimateComplexArray (tail quirk)."""
    R = O.ref()
    rng = np.random.default_rng(10)
    for n_in, n_out, ratio in ((10, 80, 8.0), (10, 75, 7.5), (7, 7, 1.0), (5, 23, 4.6), (3, 2, 0.9), (6, 13, 2.2)):
        x = rng.uniform(-3, 3, n_in + 64)  # upsample() ignores in_length: give it room to read on
        for fn_r, fn_o in ((R.ref_raw_upsample, L.orc_raw_upsample), (R.ref_upsample_fill_w_zeros, L.orc_upsample_fill_w_zeros),
                           (R.ref_upsample, L.orc_upsample)):
            a, b = np.full(n_out + 4, 7.25), np.full(n_out + 4, 7.25)  # untouched outputs must stay untouched
            assert fn_r(x.copy(), n_in, a, n_out, ratio) == fn_o(x.copy(), n_in, b, n_out, ratio)
            assert np.array_equal(a, b), (fn_o, n_in, n_out, ratio)
    for n, by in ((1000, 2), (1001, 4), (17, 5), (8, 8), (3, 10), (1, 1), (12, 3), (13, 3)):
        z = rng.uniform(-1, 1, 2 * n)
        a, b = z.copy(), z.copy()
        assert R.ref_decimate_complex(a, n, by) == L.orc_decimate_complex(b, n, by)
        assert np.array_equal(a, b), ("decimateComplexArray", n, by)
        a, b = z.copy(), z.copy()
        assert R.ref_decimate_block(a, n, by) == L.orc_decimate_block(b, n, by)
        assert np.array_equal(a, b), ("decimate(BBP_Block*)", n, by)


@pytest.mark.parametrize("rotate90", [False, True])
def test_u8_ingest_matches_baseband_stream(rotate90):
    """orc_u8_to_iq / orc_u8_rotate_90 against Baseband_Stream::load_buffer / load_buffer_and_rotate90
    (Baseband_Stream.cpp:82-148) driven through the stream's own thread and post callback: `char` bytes (signed on
    x86), bias 127.4, rotate_90 on groups of four samples.  Every byte value on both rails, plus random blocks."""
    R = O.ref()
    rng = np.random.default_rng(11 + int(rotate90))
    allb = np.arange(256, dtype=np.uint8)
    raws = [np.stack([allb, allb[::-1]], axis=1), np.stack([np.roll(allb, 77), allb], axis=1),
            rng.integers(0, 256, (4096, 2)).astype(np.uint8)]
    for raw in raws:
        n = raw.shape[0]
        re, im = np.zeros(n), np.zeros(n)
        assert R.ref_baseband_load(np.ascontiguousarray(raw).reshape(-1), 2 * n, int(rotate90), re, im) == n
        ore, oim = O.u8_to_iq(raw, 127.4, signed_char=True, rotate90=rotate90)
        assert np.array_equal(re, ore) and np.array_equal(im, oim)


def test_bpsk_demodulate_overload_2_matches_reference():
    """BPSK::demodulate(Complex_Array&, uint8_t*, n, downsample, plot_re, plot_im) (BPSK.cpp:178-211) = decimateComplexArray
    + the stages, post-timing samples to the plot buffers, bits = real > 0.5.  Two calls on one object (state carried)."""
    R = O.ref()
    h = R.ref_bpsk_new(0x14EB, 16)
    st = O.psk_new(O.BPSK, 0x14EB, 16)
    for call, ds in enumerate((2, 3)):
        base = np.asarray(psk_test_signal(O.BPSK, 300, seed=70 + call, phase_deg=20.0, snr_db=25.0, uw_bpsk=0x14EB, uw_len=16))
        z = np.repeat(base.reshape(-1, 2), ds, axis=0).reshape(-1).copy()  # the transmitter's raw_upsample
        n = z.size // 2
        a = z.copy()
        bits, pr, pi = np.zeros(n, dtype=np.uint8), np.zeros(n), np.zeros(n)
        m = R.ref_bpsk_demodulate2(h, a, n, ds, bits, pr, pi)
        b = z.copy()
        points = L.orc_decimate_complex(b, n, ds)
        assert np.array_equal(a, b)  # the caller's array after decimateComplexArray
        out, tr, ti = np.zeros(n), np.zeros(n), np.zeros(n)
        m2 = L.orc_psk_demodulate(C.byref(st), b[:2 * points].copy(), points, out, tr.ctypes.data, ti.ctypes.data)
        assert m == m2
        assert np.array_equal(bits[:m], (out[:m] > 0.5).astype(np.uint8))
        assert np.array_equal(pr[:points // 4], tr[:points // 4]) and np.array_equal(pi[:points // 4], ti[:points // 4])
    R.ref_bpsk_free(h)


# ---------------------------------------------------------------- crc8 / PacketFramer / RingBuffer

def test_crc8_and_frame_vs_reference():
    rng = np.random.default_rng(4)
    R = O.ref()
    for ln in (0, 1, 2, 17, 249, 250):
        data = rng.integers(0, 256, max(ln, 1)).astype(np.uint8)[:ln]
        d = np.ascontiguousarray(data if ln else np.zeros(1, dtype=np.uint8))
        assert O.lib().orc_crc8(d, ln) == R.ref_crc8(d, ln)
        for crc_enable in (0, 1):
            a, b = np.zeros(2032, dtype=np.uint8), np.zeros(2032, dtype=np.uint8)
            assert O.lib().orc_frame(d, ln, crc_enable, a) == R.ref_frame(d, ln, crc_enable, b) == 2032
            assert np.array_equal(a, b)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("crc_enable", [True, False])
def test_deframe_vs_reference(seed, crc_enable):
    """The literal restatement of deserialize + the receiver's loop against the reference's own PacketFramer /
    RingBuffer / crc8 on random streams with good packets, bad CRCs, bad lengths and bare unique words:
    same events, same payloads, same number of zero returns, same bits consumed -- for several stream
    lengths, so that the end-of-stream behaviour (partial passes, the 500-bit early return) is exercised."""
    from signals import framed_bit_stream
    full = framed_bit_stream(60000, 100 + seed, packet_every=2500 if seed % 2 else 400, crc_enable=crc_enable)
    for n in (60000, 59999, 50011, 31000, 2033, 2032, 2600, 4063, 4064, 4065, 4600):
        a, b = O.deframe(full[:n], crc_enable), O.ref_deframe(full[:n], crc_enable)
        for k in ("events", "consumed", "zero_returns"):
            assert a[k] == b[k], (n, k, a[k], b[k])
        for k in ("pos", "kind", "len", "crc", "payload"):
            assert np.array_equal(a[k], b[k]), (n, k)
    assert O.deframe(full, crc_enable)["events"] > 5



# ---- DSP primitives: DigitalSignalProcessing/*.cpp compiled from the reference ----

def test_dsp_primitives_vs_reference():
    R = O.ref()
    rng = np.random.default_rng(31)
    for n in (0, 1, 2, 3, 17, 1000):
        x = rng.normal(0, 40, n) + 3.0
        # DC_Filter
        y_ref = np.zeros(n)
        R.ref_dc_block(x.copy(), y_ref, n, 0.995)
        assert np.array_equal(O.dc_block(x, 0.995), y_ref)
        # FirstOrderDigitalIntegration (tStep = 1.0f / sampleRate)
        a = x.copy()
        R.ref_integrate(48000.0, a, n)
        assert np.array_equal(O.integrate(x, 1.0 / 48000.0), a)
    # LowPassFilter: one object, several blocks, state carried
    h = R.ref_lowpass1_new(3000.0, 1.0 / 48000.0)
    epow = O.lib().orc_lowpass1_coef(3000.0, 1.0 / 48000.0)
    st = np.zeros(1)
    for n in (5, 0, 1, 400):
        x = rng.normal(0, 10, n)
        y_ref = np.zeros(n)
        R.ref_lowpass1(h, x, y_ref, n)
        assert np.array_equal(O.lowpass1(x, epow, st), y_ref)
    R.ref_lowpass1_free(h)
    # NotchFilter: the reference's delay line is function-static (one per process): this is the
    # only place the checker calls it, blocks in sequence, the oracle state in lockstep.
    coef = O.notch_design(0.9, 0.05)
    st = np.zeros(4)
    for n in (7, 2, 1, 0, 300, 64):
        x = rng.normal(0, 10, n)
        b = x.copy()
        R.ref_notch(0.9, 0.05, b, n)
        assert np.array_equal(O.notch(x, coef, st), b)
    # normalize: size_t running maximum
    for n, scale, prev in ((100, 50.0, 0.0), (100, 50.0, 1e4), (33, 0.2, 0.0), (1, 7.9, 3.5), (64, 1e6, 0.0)):
        x = rng.normal(0, scale, n)
        b = x.copy()
        with np.errstate(all="ignore"):
            m_ref = R.ref_normalize(b, n, prev)
        y, m = O.normalize(x, prev)
        assert m == m_ref and np.array_equal(y, b, equal_nan=True)
    # fifth_order_filter::decimate(double*): state carried over blocks, 1..3 passes
    h = R.ref_fifth_new()
    st = np.zeros(6)
    for n, by in ((64, 2), (1, 2), (2, 2), (3, 2), (129, 4), (1000, 8), (5, 2), (40, 1)):
        x = rng.normal(0, 10, n)
        b = x.copy()
        r_ref = R.ref_fifth_decimate(h, b, n, by)
        y, r = O.fifth_order_decimate(x, by, st)
        assert r == r_ref and np.array_equal(y, b)
    R.ref_fifth_free(h)
    # fft: libm mode is the reference's arithmetic bit for bit; the shared-math mode (the device's) within 1e-12
    for n in (1, 2, 4, 64, 1024):
        z = rng.normal(0, 1, n) + 1j * rng.normal(0, 1, n)
        b = z.copy()
        R.ref_fft(b.view(np.float64), n)
        assert np.array_equal(O.fft(z, O.MATH_LIBM), b)
        assert np.max(np.abs(O.fft(z, O.MATH_SHARED) - b), initial=0.0) <= 1e-12 * max(1.0, np.abs(b).max(initial=0.0)) * np.log2(max(n, 2))
        if n >= 2:
            assert np.allclose(b, np.fft.fft(z), rtol=0, atol=1e-9)


def test_ones_zeros_confidence_vs_reference():
    """The receiver's ones/zeros CPU branch over the reference's own confidence_counter (increment(1) / decrement(1)
    per sample, clamped at both rails) against the restatement: small maxima so that both rails are hit many times."""
    R = O.ref()
    rng = np.random.default_rng(99)
    for n, cmax, start, p_one in ((0, 10, 5, 0.5), (1000, 10, 5, 0.5), (5000, 37, 0, 0.7), (5000, 37, 37, 0.3),
                                  (20000, 500000, 250000, 0.9), (3000, 1, 0, 0.5)):
        mag = np.where(rng.random(n) < p_one, 100.0, 10.0)
        ph = rng.uniform(0, 2 * np.pi, n)
        re, im = np.ascontiguousarray(mag * np.cos(ph)), np.ascontiguousarray(mag * np.sin(ph))
        want = R.ref_ones_zeros_confidence(re, im, n, cmax, 0, cmax, start)
        assert O.lib().orc_ones_zeros_confidence(re, im, n, cmax, start) == want


def test_fifth_order_block_overload_vs_reference():
    """fifth_order_filter::decimate(BBP_Block*): both rails, n_points shrinking pass by pass (3 after a pass without
    an iteration), state carried over several blocks: bit for bit against the reference's class."""
    R = O.ref()
    rng = np.random.default_rng(5)
    h = R.ref_fifth_new()
    st = np.zeros(12)
    cap = 1200
    for n, by in ((64, 2), (1, 2), (2, 4), (3, 2), (129, 4), (1000, 8), (5, 2), (40, 1), (7, 16)):
        z = rng.normal(0, 10, cap) + 1j * rng.normal(0, 10, cap)
        b = z.copy()
        r_ref = R.ref_fifth_decimate_block(h, b.view(np.float64), cap, n, by)
        y, r = O.fifth_order_decimate_block(z, n, by, st)
        assert r == r_ref and np.array_equal(y, b), (n, by)
    R.ref_fifth_free(h)
