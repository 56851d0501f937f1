"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Bar: bit-exact for bits / symbol decisions AND for filtered samples, loop state
and discriminator output (the kernels keep the reference's operation order without FMA
contraction; oracle in ORC_MATH_SHARED mode for the two transcendental calls).  Only the
`fast_fma` variant is compared with a tolerance (1e-12 relative to sum|h||x|, stated in its test).
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from signals import fm_iq, framed_bit_stream, ook_iq, psk_symbols, psk_test_signal, radio_iq

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from sdr_amd import _lib, device, host  # noqa: E402

L = O.lib()


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    device.init(0)
    yield
    torch.cuda.synchronize()


def G(golden, key):
    return np.array(golden[key])


def _rows_dev(a):
    return torch.tensor(np.ascontiguousarray(a), device="cuda")


# =============================================================================== Filter

@pytest.mark.parametrize("ftype,Fx,Fu", [(O.LPF, 1.0e5, None), (O.HPF, 3.0e5, None), (O.BPF, 1.0e5, 4.0e5)])
@pytest.mark.parametrize("T", [1, 2, 51, 64, 129, 1000])
def test_filter_execute_streaming(ftype, Fx, Fu, T):
    """Filter::execute (filt.cpp:267-273): taps and outputs bit-identical over ragged chunks."""
    f = host.Filter(ftype, T, 2.0e6, Fx, Fu)
    o = O.Fir(ftype, T, 2.0e6, Fx, Fu)
    assert f.get_error_flag() == 0 == o.err
    assert np.array_equal(f.get_taps(), o.taps)
    rng = np.random.default_rng(T * 7 + ftype)
    x = rng.integers(-127, 128, 20000).astype(np.float64) + rng.uniform(-0.5, 0.5, 20000)
    for lo, hi in ((0, 1), (1, 2), (2, 1300), (1300, 1301 + T), (1301 + T, 20000)):
        a = x[lo:hi].copy()
        f.execute(a)
        b = o.execute(x[lo:hi], fast=True)
        assert np.array_equal(a, b), (lo, hi)
    f.init()  # Filter::init clears the shift register
    o2 = O.Fir(ftype, T, 2.0e6, Fx, Fu)
    a = x[:500].copy()
    f.execute(a)
    assert np.array_equal(a, o2.execute(x[:500], fast=True))


def test_filter_do_sample_and_errors():
    """do_sample (filt.cpp:248-265) and the constructor's error codes (filt.cpp:54-67,85-99)."""
    f = host.Filter(O.LPF, 51, 44.1, 4.0)
    o = O.Fir(O.LPF, 51, 44.1, 4.0)
    xs = np.linspace(-3, 5, 40)
    got = np.array([f.do_sample(v) for v in xs])
    assert np.array_equal(got, o.execute(xs))
    cases = [((O.LPF, 64, 0.0, 1.0, None), -1), ((O.LPF, 64, 2e6, 0.0, None), -2), ((O.LPF, 64, 2e6, 1e6, None), -2),
             ((O.LPF, 0, 2e6, 1e5, None), -3), ((O.LPF, 1001, 2e6, 1e5, None), -3), ((O.BPF, 64, 2e6, 1e5, None), -5),
             ((O.BPF, 64, 0.0, 1.0, 2.0), -10), ((O.BPF, 64, 2e6, 2e5, 1e5), -11), ((O.BPF, 64, 2e6, 0.0, 1e5), -12),
             ((O.BPF, 64, 2e6, 1e5, 1e6), -13), ((O.BPF, 1024, 2e6, 1e5, 2e5), -14), ((O.LPF, 64, 2e6, 1e5, 2e5), -16)]
    for (ft, T, Fs, Fx, Fu), code in cases:
        g = host.Filter(ft, T, Fs, Fx, Fu)
        assert g.get_error_flag() == code
        buf = np.ones(16)
        g.execute(buf)  # do_sample returns 0 once the flag is set (filt.cpp:254)
        assert np.all(buf == 0)
        assert g.do_sample(1.0) == 0.0


def test_filter_1024_taps_cap_lifted():
    """BASELINE config 5 needs T = 1024; the reference rejects it (-3).  With the cap lifted the
    restatement of filt.cpp:116-121,248-265 is the checker."""
    assert host.Filter(O.LPF, 1024, 2.0e6, 1.0e5).get_error_flag() == -3
    f = host.Filter(O.LPF, 1024, 2.0e6, 1.0e5, max_taps=4096)
    o = O.Fir(O.LPF, 1024, 2.0e6, 1.0e5, max_taps=4096)
    assert f.get_error_flag() == 0 == o.err
    rng = np.random.default_rng(5)
    x = rng.integers(-127, 128, 50000).astype(np.float64)
    a = x.copy()
    f.execute(a)
    assert np.array_equal(a, o.execute(x, fast=True))


# =============================================================================== AM

def test_am_golden(golden):
    """TEST(AM, modulate / demodulate / ModDemod): TEST_AM.cpp:14-58, TEST_Modulator_Demodulator.cpp:123-162."""
    bits = G(golden, "TEST_AM__AM__modulate__inputStream")
    am = host.AM()
    mod = np.zeros(8000)
    am.modulate(bits, 1000, mod, None, 8000)
    assert np.array_equal(mod, G(golden, "TEST_AM__AM__modulate__modulatedStream"))
    x = G(golden, "TEST_AM__AM__demodulate__modulatedStream") * 127.0
    out = np.zeros(1000, dtype=np.uint8)
    assert am.demodulate(x, x, out, 8000, 8) == 1000
    assert np.array_equal(out, G(golden, "TEST_AM__AM__demodulate__ouputStream"))


@pytest.mark.parametrize("ds", [1, 8, 7, 100])
def test_am_demod_vs_oracle(ds):
    n = 100003
    re, im = radio_iq(n, 11)
    # values on and around the decision boundary sqrt(re^2+im^2) == 80 (AM.h:73)
    re[:6] = [80.0, 48.0, np.nextafter(80.0, 0), 0.0, 79.99999999999997, 56.568542494923804]
    im[:6] = [0.0, 64.0, 0.0, np.nextafter(80.0, 100), 0.0, 56.568542494923804]
    out = np.zeros(n, dtype=np.uint8)
    m = host.AM().demodulate(re, im, out, n, ds)
    want = O.am_demodulate(re, im, ds)
    assert m == want.size
    assert np.array_equal(out[:m], want)


def test_am_threshold_boundary_exhaustive():
    """The kernels compare re^2+im^2 with the smallest double whose rounded sqrt is >= 80; sweep the
    500 doubles either side of 6400 to show this equals sqrt(...) >= 80 bit for bit."""
    s = 6400.0
    vals = [s]
    lo = hi = s
    for _ in range(500):
        lo = np.nextafter(lo, 0)
        hi = np.nextafter(hi, 1e9)
        vals += [lo, hi]
    v = np.array(vals)
    re = np.sqrt(v)  # re^2 lands near 6400 in many different roundings
    im = np.zeros_like(re)
    out = np.zeros(v.size, dtype=np.uint8)
    host.AM().demodulate(re, im, out, v.size, 1)
    assert np.array_equal(out, O.am_demodulate(re, im, 1))


def test_ones_zeros_and_legacy_am():
    """ones_zeros_demodulation (ones_zeros.cu:28-52 / rtlsdr_receiver.cpp:2063-2086) and the
    am_gpu_* legacy symbols (am.cu:94-152) incl. ProfileTest's ramp input (ProfileTest/main.cpp:124-135)."""
    Lx = _lib.lib()
    n = 512000
    buf = (np.arange(n) % 255).astype(np.float64)
    Lx.am_gpu_initialize(n, 2)
    out = np.zeros(n, dtype=np.uint8)
    for radio in (0, 1):
        t = Lx.am_gpu_demodulation(radio, buf.ctypes.data, buf.ctypes.data, out.ctypes.data, n, 1)
        assert t > 0
        assert np.array_equal(out, O.am_demodulate(buf, buf, 1))
    Lx.am_gpu_free()
    re, im = radio_iq(300001, 3)
    assert Lx.ones_zeros_demodulation(re.ctypes.data, im.ctypes.data, re.size) == L.orc_ones_zeros(re, im, re.size)


# =============================================================================== FM (FSK)

def test_fm_golden(golden):
    """TEST(FM, modulate / demodulate / ModDemodResampled): TEST_FM.cpp:14-99."""
    bits = G(golden, "TEST_FM__FM__modulate__inputStream")
    fm = host.FM(100, 10, 20, 10)
    mod = np.zeros(1000)
    fm.modulate(bits, 10, mod, None, 1000, 100)
    assert np.allclose(mod, G(golden, "TEST_FM__FM__modulate__modulatedStream"), rtol=0, atol=1e-9)
    x = G(golden, "TEST_FM__FM__demodulate__modulatedRealStream")
    out = np.zeros(10, dtype=np.uint8)
    assert host.FM(100, 100, 20, 10).demodulate(x, np.zeros(1000), out, 1000, 100) == 10
    assert np.array_equal(out, G(golden, "TEST_FM__FM__demodulate__outputStream"))
    stream = G(golden, "TEST_FM__FM__ModDemodResampled__modulatedStream").copy()
    L.orc_decimate(stream, 1000, 2)
    half = stream[:500].copy()
    out2 = np.zeros(10, dtype=np.uint8)
    assert host.FM(100, 50, 20, 10).demodulate(half, half, out2, 500, 50) == 10
    assert np.array_equal(out2, G(golden, "TEST_FM__FM__ModDemodResampled__outputStream"))


@pytest.mark.parametrize("ds", [1, 2, 15, 16, 17, 31, 32, 33, 100, 333, 2000])
def test_fm_demod_vs_oracle(ds):
    """FM<T>::demodulate (FM.h:83-148): bits identical on noisy FSK and on pure noise (ties matter)."""
    rng = np.random.default_rng(ds)
    nbits = 700
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = O.fm_modulate(2.0e6, 19000.0, 20000.0, bits, ds) * 40 + rng.normal(0, 1.0, nbits * ds)
    x = np.concatenate([x, rng.normal(0, 1, ds // 2)])  # ragged tail: n not a multiple of ds
    out = np.zeros(nbits + 4, dtype=np.uint8)
    m = host.FM(10e6, 2.0e6).demodulate(x, x, out, x.size, ds)
    want = O.fm_demodulate(2.0e6, 19000.0, 20000.0, x, ds)
    assert m == want.size == nbits
    assert np.array_equal(out[:m], want)


def test_legacy_fm_symbols():
    """fm_gpu_* (fm.cu:48-167) with ProfileTest's arguments (ProfileTest/main.cpp:160,184)."""
    Lx = _lib.lib()
    n = 512000
    buf = (np.arange(n) % 255).astype(np.float64)
    Lx.fm_gpu_initialize(1, n, 2000000, 100, 10.0, 20.0)
    out = np.zeros(n // 100, dtype=np.uint8)
    Lx.fm_gpu_demodulation(0, buf.ctypes.data, buf.ctypes.data, out.ctypes.data, n, 100)
    assert np.array_equal(out, O.fm_demodulate(2000000.0, 10.0, 20.0, buf, 100))
    Lx.fm_gpu_set_samples_per_bit(50, 0)  # the receiver's one-argument call: rate falls back to init
    out2 = np.zeros(n // 50, dtype=np.uint8)
    Lx.fm_gpu_demodulation(0, buf.ctypes.data, buf.ctypes.data, out2.ctypes.data, n, 50)
    assert np.array_equal(out2, O.fm_demodulate(2000000.0, 10.0, 20.0, buf, 50))
    Lx.fm_gpu_free()


# =============================================================================== chains

def _oracle_chain_am(taps, re, im, ds, hI, hQ):
    bits = np.zeros((re.size + ds - 1) // ds + 1, dtype=np.uint8)
    m = L.orc_chain_fir_am(taps, taps.size, hI, hQ, re, im, re.size, ds, bits)
    return bits[:m]


def _oracle_chain_disc(taps, re, im, D, hI, hQ, pre):
    out = np.zeros((re.size + D - 1) // D + 1)
    m = L.orc_chain_fir_decim_disc(taps, taps.size, hI, hQ, pre, re, im, re.size, D, out, O.MATH_SHARED)
    return out[:m]


@pytest.mark.parametrize("ds", [1, 5, 6, 8, 10, 12, 16, 20, 25, 100])
def test_chain_fir_am_host(ds):
    """BASELINE config 1: Filter(LPF,64,2e6,1e5) on I and Q, AM::demodulate(ds); streamed in chunks."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    n = 1 << 17
    re, im, _ = ook_iq(n, 1, ds=8)
    ch = host.Chain(taps, ds)
    hI, hQ = np.zeros(63), np.zeros(63)
    step = 48000  # multiple of ds for every ds tested
    for k in range(0, n, step):
        r, i = re[k:k + step].copy(), im[k:k + step].copy()
        got = ch.fir_am(r, i)
        want = _oracle_chain_am(taps, r, i, ds, hI, hQ)
        assert np.array_equal(got, want), k


@pytest.mark.parametrize("D,T", [(10, 64), (1, 64), (2, 33), (3, 64), (4, 129), (7, 20), (16, 64), (64, 64), (100, 64), (10, 1), (10, 2),
                                 (5, 33), (5, 61), (5, 64), (8, 7), (8, 64), (8, 97), (12, 64), (12, 100), (12, 144), (12, 145), (10, 120), (10, 121),
                                 (5, 80), (5, 81), (6, 64), (6, 97), (16, 64), (16, 3), (20, 64), (20, 200), (25, 64), (25, 129)])
def test_chain_fir_fmdisc_host(D, T):
    """BASELINE config 2: LPF -> decimate by D -> phase-difference discriminator; every FIR register
    blocking (R = 8, 4, 2, 1), gather tiles (D >= T), the systolic kernel (D in 5, 6, 8, 10, 12, 16, 20, 25: one stage,
    whole and partial last stages, and filters long enough to fall back to workgroup tiles) and the
    state carry across ragged chunks."""
    err, taps = O.fir_design(O.LPF, T, 2.4e6, 1.0e5)
    assert err == 0
    n = 200000
    re, im = fm_iq(n, seed=2)
    ch = host.Chain(taps, D)
    hI, hQ, pre = np.zeros(max(T - 1, 1)), np.zeros(max(T - 1, 1)), np.zeros(2)
    cuts = [0, D * 3, D * 3 + D * 1000, D * 3 + D * 1000 + D * 7919, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        r, i = re[lo:hi].copy(), im[lo:hi].copy()
        got = ch.fir_fmdisc(r, i)
        want = _oracle_chain_disc(taps, r, i, D, hI, hQ, pre)
        assert got.size == want.size
        assert np.array_equal(got, want), (lo, hi, np.flatnonzero(got != want)[:5])


def test_chain_device_layouts_multichannel():
    """Device layer: 5 channels, split and interleaved layouts give the same bits/angles as the oracle."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    C_, n, D = 5, 70001, 10
    res, ims = zip(*[fm_iq(n, seed=40 + c) for c in range(C_)])
    re = torch.tensor(np.stack(res), device="cuda")
    im = torch.tensor(np.stack(ims), device="cuda")
    iq = torch.stack([re, im], dim=-1).contiguous()
    n_out = (n + D - 1) // D
    for x in ((re, im), iq):
        ch = device.FirChain(taps, nchan=C_, decim=D)
        ang = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
        ch.fir_fmdisc(x, ang)
        # second call continues the streams
        ang2 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
        ch.fir_fmdisc(x, ang2)
        torch.cuda.synchronize()
        for c in range(C_):
            hI, hQ, pre = np.zeros(63), np.zeros(63), np.zeros(2)
            w1 = _oracle_chain_disc(taps, res[c], ims[c], D, hI, hQ, pre)
            w2 = _oracle_chain_disc(taps, res[c], ims[c], D, hI, hQ, pre)
            assert np.array_equal(ang[c].cpu().numpy(), w1)
            assert np.array_equal(ang2[c].cpu().numpy(), w2)
    ch = device.FirChain(taps, nchan=C_, decim=1)
    bits = torch.zeros(C_, n, dtype=torch.uint8, device="cuda")
    ch.fir_am(iq * 1.0, bits)
    torch.cuda.synchronize()
    for c in range(C_):
        assert np.array_equal(bits[c].cpu().numpy(), _oracle_chain_am(taps, res[c], ims[c], 1, np.zeros(63), np.zeros(63)))


@pytest.mark.parametrize("D", [5, 6, 8, 10, 12, 16, 20, 25])
def test_decimating_fir_store_all_layouts(D):
    """sdrx_fir_dev with decimation (Filter on each rail, then decimate<T>, Resample.h:33-52): real,
    split and interleaved inputs through the systolic kernel, 3 channels, two calls with carried state;
    samples bit-identical to the oracle's Filter + decimate."""
    T = 64
    err, taps = O.fir_design(O.BPF, T, 2.4e6, 1.0e5, 3.0e5)
    assert err == 0
    C_, n = 3, 30011
    res, ims = zip(*[radio_iq(n, 70 + c) for c in range(C_)])
    re = torch.tensor(np.stack(res), device="cuda")
    im = torch.tensor(np.stack(ims), device="cuda")
    iq = torch.stack([re, im], dim=-1).contiguous()
    n_out = (n + D - 1) // D

    for x, rails in ((re, 1), ((re, im), 2), (iq, 2)):
        ch = device.FirChain(taps, nchan=C_, decim=D)
        outs = []
        for _ in range(2):
            o0 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
            o1 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda") if rails == 2 else None
            ch.fir(x, o0, o1)
            outs.append((o0, o1))
        torch.cuda.synchronize()
        for c in range(C_):
            fi, fq = O.Fir(O.BPF, T, 2.4e6, 1.0e5, 3.0e5), O.Fir(O.BPF, T, 2.4e6, 1.0e5, 3.0e5)
            for o0, o1 in outs:
                # the second call starts at sample n of the doubled stream: its decimation phase restarts at 0
                yi = fi.execute(res[c], fast=True)[::D]
                assert np.array_equal(o0[c].cpu().numpy(), yi), (D, rails, c)
                if rails == 2:
                    yq = fq.execute(ims[c], fast=True)[::D]
                    assert np.array_equal(o1[c].cpu().numpy(), yq), (D, rails, c)


@pytest.mark.parametrize("T", [32, 64, 129, 1024])
@pytest.mark.parametrize("mode", [1, 2])
def test_fast_modes_are_the_fused_chain_bit_for_bit(mode, T):
    """fast_fma = 1 (VALU fma chain) and 2 (fp64 MFMA Toeplitz kernel) both compute
    acc = fma(x[n-k], h[k], acc), k ascending: checked bit for bit against the oracle's fused variant,
    real / split / interleaved layouts, 3 channels, ragged length, two calls with carried history."""
    err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5, max_taps=4096)
    assert err == 0
    C_, n = 3, 5000 + 7
    res, ims = zip(*[radio_iq(n, 90 + c) for c in range(C_)])
    re = torch.tensor(np.stack(res), device="cuda")
    im = torch.tensor(np.stack(ims), device="cuda")
    iq = torch.stack([re, im], dim=-1).contiguous()
    for x, rails, inter_out in ((re, 1, False), ((re, im), 2, False), (iq, 2, False), (iq, 2, True)):
        ch = device.FirChain(taps, nchan=C_, decim=1)
        outs = []
        for _ in range(2):
            if inter_out:
                o0 = torch.zeros(C_, n, 2, dtype=torch.float64, device="cuda")
                ch.fir(x, o0, None, fast_fma=mode)
                outs.append((o0[..., 0], o0[..., 1]))
            else:
                o0 = torch.zeros(C_, n, dtype=torch.float64, device="cuda")
                o1 = torch.zeros(C_, n, dtype=torch.float64, device="cuda") if rails == 2 else None
                ch.fir(x, o0, o1, fast_fma=mode)
                outs.append((o0, o1))
        torch.cuda.synchronize()
        for c in range(C_):
            fi = O.Fir(O.LPF, T, 2.0e6, 1.0e5, max_taps=4096)
            fq = O.Fir(O.LPF, T, 2.0e6, 1.0e5, max_taps=4096)
            for o0, o1 in outs:
                assert np.array_equal(o0[c].cpu().numpy(), fi.execute_fma(res[c])), (mode, T, rails, c)
                if rails == 2:
                    assert np.array_equal(o1[c].cpu().numpy(), fq.execute_fma(ims[c])), (mode, T, rails, c)


# =============================================================================== u8 I/Q ingest

@pytest.mark.parametrize("bias,signed_char,rot", [(127.0, False, False), (127.4, True, False), (127.4, True, True), (127.0, False, True)])
@pytest.mark.parametrize("D,T", [(10, 64), (1, 64), (4, 129), (100, 64)])
def test_u8_ingest_chain(D, T, bias, signed_char, rot):
    """SDRX_IQ_U8: the kernels convert raw radio bytes on load.  Bit-exact against the oracle's
    restatement of the three reference conversions (rtlsdr_receiver.cpp:1587-1588; Baseband_Stream
    load_buffer / load_buffer_and_rotate90, Baseband_Stream.cpp:82-148) followed by the fp64 chain;
    2 channels, two calls with carried state (rotate_90 restarts at every call, like the reference)."""
    err, taps = O.fir_design(O.LPF, T, 2.4e6, 1.0e5)
    C_, n = 2, 40000
    rng = np.random.default_rng(D * 1000 + T)
    raw = rng.integers(0, 256, (C_, n, 2), dtype=np.uint8)
    # an FM-ish signal in bytes so that AM bits and angles are not all noise
    t = np.arange(n)
    ph = 30 * np.sin(2 * np.pi * t / 2400.0)
    raw[0, :, 0] = np.clip(np.round(127.4 + 100 * np.cos(ph)), 0, 255).astype(np.uint8)
    raw[0, :, 1] = np.clip(np.round(127.4 + 100 * np.sin(ph)), 0, 255).astype(np.uint8)
    x = torch.tensor(raw, device="cuda")
    n_out = (n + D - 1) // D
    cd = device.FirChain(taps, nchan=C_, decim=D)
    ca = device.FirChain(taps, nchan=C_, decim=D)
    for ch in (cd, ca):
        ch.set_u8_conversion(bias, signed_char, rot)
    got = []
    for _ in range(2):
        ang = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
        bits = torch.zeros(C_, n_out, dtype=torch.uint8, device="cuda")
        cd.fir_fmdisc(x, ang)
        ca.fir_am(x, bits)
        got.append((ang.cpu().numpy(), bits.cpu().numpy()))
    for c in range(C_):
        re, im = O.u8_to_iq(raw[c], bias, signed_char, rot)
        H = max(T - 1, 1)
        hI, hQ, pre = np.zeros(H), np.zeros(H), np.zeros(2)
        aI, aQ = np.zeros(H), np.zeros(H)
        for ang, bits in got:
            assert np.array_equal(ang[c], _oracle_chain_disc(taps, re, im, D, hI, hQ, pre)), (c,)
            assert np.array_equal(bits[c], _oracle_chain_am(taps, re, im, D, aI, aQ)), (c,)


def test_u8_ingest_f32():
    """u8 bytes straight into the fp32 twin: within 1e-5 of the fp64 path on the same bytes."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    n, D = 50000, 10
    raw = np.random.default_rng(3).integers(0, 256, (1, n, 2), dtype=np.uint8)
    x = torch.tensor(raw, device="cuda")
    ch = device.FirChain(taps, nchan=1, decim=D, dtype=torch.float32)
    o0 = torch.zeros(1, n // D, dtype=torch.float32, device="cuda")
    o1 = torch.zeros(1, n // D, dtype=torch.float32, device="cuda")
    ch.fir(x, o0, o1)
    re, im = O.u8_to_iq(raw[0])
    yi = O.Fir(O.LPF, 64, 2.4e6, 1.0e5).execute(re, fast=True)[::D]
    yq = O.Fir(O.LPF, 64, 2.4e6, 1.0e5).execute(im, fast=True)[::D]
    si = (np.convolve(np.abs(re), np.abs(taps))[:n] + 1e-30)[::D]
    sq = (np.convolve(np.abs(im), np.abs(taps))[:n] + 1e-30)[::D]
    assert np.max(np.abs(o0[0].cpu().numpy() - yi) / si) <= 1e-5
    assert np.max(np.abs(o1[0].cpu().numpy() - yq) / sq) <= 1e-5


@pytest.mark.parametrize("ds,rot", [(10, False), (10, True), (1, False), (7, True), (256, False)])
def test_rtlfm_int16_chain(ds, rot):
    """sdrx_rtlfm_dev = rtl_fm's callback conversion + low_pass + fm_demod (rtl_fm.c:303-323, 422-435, 518-546,
    808-822) on raw bytes: int16 pcm identical to the oracle's restatement, 3 channels, ragged chunks whose
    lengths are not multiples of the downsample factor (partial boxcar sums and pre_r/pre_j carried)."""
    C_, n = 3, 100000
    rng = np.random.default_rng(ds)
    raw = rng.integers(0, 256, (C_, n, 2), dtype=np.uint8)
    t = np.arange(n)
    ph = 40 * np.sin(2 * np.pi * t / 4800.0)
    raw[0, :, 0] = np.clip(np.round(127.4 + 110 * np.cos(ph)), 0, 255).astype(np.uint8)
    raw[0, :, 1] = np.clip(np.round(127.4 + 110 * np.sin(ph)), 0, 255).astype(np.uint8)
    fm = device.RtlFm(C_, ds, rot)
    sts = [np.zeros(5, dtype=np.int32) for _ in range(C_)]
    cuts = [0, 4 * 3, 4 * 3 + 4 * 2501, 4 * 3 + 4 * 2501 + 4 * 9973, n]  # multiples of 4 (rotate_90 works on 8-byte groups)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        x = torch.tensor(raw[:, lo:hi].copy(), device="cuda")
        pcm = torch.zeros(C_, (hi - lo) // ds + 2, dtype=torch.int16, device="cuda")
        m = fm.demod(x, pcm)
        got = pcm.cpu().numpy()
        for c in range(C_):
            want = O.rtlfm_chain(raw[c, lo:hi], ds, sts[c], rot)
            assert want.size == m
            assert np.array_equal(got[c, :m], want), (c, lo, hi)
        st_dev = fm.state[fm.cur].cpu().numpy()
        for c in range(C_):
            assert np.array_equal(st_dev[c], sts[c]), (c, lo, hi)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4095, 4096, 4097, 65536, 65537, 200001])
def test_ones_zeros_confidence_walk(n):
    """sdrx_ones_zeros_confidence_dev: the receiver's CPU branch walks a confidence_counter one sample at a time,
    clamped at 0 and max after every step; the device composes the clamp maps in order.  Six channels with maxima
    small enough that the rails are hit all the time and one like the receiver's (500000), starts at either rail
    and in the middle, envelopes on both sides of and exactly at the threshold: counters equal the literal loop's."""
    rng = np.random.default_rng(7 + n)
    cfg = [(10, 5, 0.5), (37, 0, 0.7), (37, 37, 0.3), (500000, 250000, 0.9), (1, 0, 0.5), (3, 3, 0.02)]
    nchan = len(cfg)
    re, im = np.zeros((nchan, n)), np.zeros((nchan, n))
    for c, (_, _, p_one) in enumerate(cfg):
        mag = np.where(rng.random(n) < p_one, 100.0, 10.0)
        mag[rng.random(n) < 0.05] = 80.0  # exactly at the threshold: counts as a one
        ph = rng.integers(0, 4, n) * (np.pi / 2)
        re[c], im[c] = np.round(mag * np.cos(ph)), np.round(mag * np.sin(ph))
    for cmax_all in (None, 7):  # one launch takes one max: per-channel maxima need their own launches
        for c, (cmax, start, _) in enumerate(cfg):
            cm = cmax if cmax_all is None else cmax_all
            st = min(start, cm)
            cnt = torch.tensor([st], dtype=torch.int32, device="cuda")
            device.ones_zeros_confidence(_rows_dev(re[c:c + 1]), _rows_dev(im[c:c + 1]), cm, cnt)
            assert int(cnt.item()) == L.orc_ones_zeros_confidence(np.ascontiguousarray(re[c]), np.ascontiguousarray(im[c]), n, cm, st)
    # all channels in one launch
    cnt = torch.tensor([min(s, 7) for _, s, _ in cfg], dtype=torch.int32, device="cuda")
    device.ones_zeros_confidence(_rows_dev(re), _rows_dev(im), 7, cnt)
    want = [L.orc_ones_zeros_confidence(np.ascontiguousarray(re[c]), np.ascontiguousarray(im[c]), n, 7, min(cfg[c][1], 7))
            for c in range(nchan)]
    assert cnt.cpu().tolist() == want


@pytest.mark.parametrize("nchan,n", [(1, 262144), (1, 262145), (1, 1200000), (1, 9000001), (3, 300001), (20, 524288 + 77),
                                     (255, 262144 + 8191), (256, 262144)])
def test_ones_zeros_confidence_walk_in_segments(nchan, n):
    """A few long streams (one radio's block): the walk is cut into segments, one workgroup each, whose maps a second
    kernel applies in order.  Small maxima (the rails are hit inside every segment), one like the receiver's, and two
    calls in a row on two streams (they share the per-GPU array of maps): counters equal the literal loop's."""
    rng = np.random.default_rng(nchan * 31 + n)
    mag = np.where(rng.random((nchan, n)) < rng.random((nchan, 1)), 100.0, 10.0)
    ph = rng.integers(0, 4, (nchan, n)) * (np.pi / 2)
    re, im = np.round(mag * np.cos(ph)), np.round(mag * np.sin(ph))
    re_d, im_d = _rows_dev(re), _rows_dev(im)
    side = torch.cuda.Stream()
    for cmax in (5, 500000):
        start = rng.integers(0, cmax + 1, nchan).astype(np.int32)
        cnt = torch.tensor(start, device="cuda")
        cnt2 = torch.tensor(start[::-1].copy(), device="cuda")
        torch.cuda.synchronize()
        device.ones_zeros_confidence(re_d, im_d, cmax, cnt)
        with torch.cuda.stream(side):
            device.ones_zeros_confidence(re_d, im_d, cmax, cnt2)
        torch.cuda.synchronize()
        want = [L.orc_ones_zeros_confidence(np.ascontiguousarray(re[c]), np.ascontiguousarray(im[c]), n, cmax, int(start[c]))
                for c in range(nchan)]
        want2 = [L.orc_ones_zeros_confidence(np.ascontiguousarray(re[c]), np.ascontiguousarray(im[c]), n, cmax, int(start[::-1][c]))
                 for c in range(nchan)]
        assert cnt.cpu().tolist() == want
        assert cnt2.cpu().tolist() == want2


def test_ones_zeros_confidence_walk_under_stream_capture():
    """A long stream while its stream is being captured into a HIP graph: the segmented form orders its calls with an
    event, which a capture must not see, so the one-workgroup form is captured instead; the replayed graph walks the
    counter like the literal loop, twice in a row (the second replay starts from the first one's counter)."""
    rng = np.random.default_rng(99)
    n = 700001
    mag = np.where(rng.random((1, n)) < 0.52, 100.0, 10.0)
    re, im = mag, np.zeros((1, n))
    re_d, im_d = _rows_dev(re), _rows_dev(im)
    cnt = torch.tensor([40], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        device.ones_zeros_confidence(re_d, im_d, 300, cnt)
    cnt.fill_(40)
    g.replay()
    torch.cuda.synchronize()
    first = L.orc_ones_zeros_confidence(np.ascontiguousarray(re[0]), np.ascontiguousarray(im[0]), n, 300, 40)
    assert int(cnt.item()) == first
    g.replay()
    torch.cuda.synchronize()
    assert int(cnt.item()) == L.orc_ones_zeros_confidence(np.ascontiguousarray(re[0]), np.ascontiguousarray(im[0]), n, 300, first)
    # and outside a capture the same call takes the segmented form and agrees
    cnt2 = torch.tensor([40], dtype=torch.int32, device="cuda")
    device.ones_zeros_confidence(re_d, im_d, 300, cnt2)
    assert int(cnt2.item()) == first


# =============================================================================== DSP primitives

@pytest.mark.parametrize("cw", [4, 16, 64, -4, -16])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 15, 16, 17, 18, 33, 63, 64, 65, 128, 129, 200, 1000, 4099])
def test_dsp_recurrences_bit_exact(n, cw, tune):
    """(cw = rows per wave: the launcher's three kernel shapes.)  DC_Filter, LowPassFilter, NotchFilter, FirstOrderDigitalIntegration: 70 channels (more than one wave, ragged
    last wave), three calls in a row so that the carried state (lowpass, notch) and the restart (dc block,
    integrate) are both seen; every row equals the oracle's bit for bit."""
    tune("SDRX_IIR_CW", abs(cw))
    tune("SDRX_IIR_PAIR", 1 if cw < 0 else 0)  # cw < 0: the two-wave form with |cw| rows per workgroup
    rng = np.random.default_rng(1000 + n)
    nchan = 70
    epow = device.lowpass1_coef(3000.0, 1.0 / 48000.0)
    assert epow == L.orc_lowpass1_coef(3000.0, 1.0 / 48000.0)
    coef = device.notch_design(0.9, 0.05)
    assert np.array_equal(coef, O.notch_design(0.9, 0.05))
    st_lp, st_no = np.zeros((nchan, 1)), np.zeros((nchan, 4))
    d_lp = torch.zeros(nchan, 4, dtype=torch.float64, device="cuda")
    d_no = torch.zeros(nchan, 4, dtype=torch.float64, device="cuda")
    for call in range(3):
        x = rng.normal(0, 30, (nchan, n)) + 2.5
        xd = _rows_dev(x)
        # x -> y: the recurrence as written; in place: the reference's ALIASED call, which reads input[i-1] after
        # output[i-1] overwrote it (DC_Filter.cpp:18) -- a different recurrence, pinned by a reference fixture
        y = torch.empty_like(xd)
        device.iir(device.IIR_DC_BLOCK, 0.995, xd, y)
        assert np.array_equal(y.cpu().numpy(), np.stack([O.dc_block(r, 0.995) for r in x]).reshape(nchan, n))
        z = xd.clone()
        device.iir(device.IIR_DC_BLOCK, 0.995, z)
        assert np.array_equal(z.cpu().numpy(), np.stack([O.dc_block_inplace(r, 0.995) for r in x]).reshape(nchan, n))
        z = xd.clone()
        device.iir(device.IIR_INTEGRATE, 1.0 / 48000.0, z)
        assert np.array_equal(z.cpu().numpy(), np.stack([O.integrate(r, 1.0 / 48000.0) for r in x]).reshape(nchan, n))
        device.iir(device.IIR_LOWPASS, epow, xd, y, state=d_lp)
        want = np.stack([O.lowpass1(x[c], epow, st_lp[c]) for c in range(nchan)]).reshape(nchan, n)
        assert np.array_equal(y.cpu().numpy(), want)
        assert np.array_equal(d_lp.cpu().numpy()[:, 0], st_lp[:, 0])
        z = xd.clone()
        device.iir(device.IIR_NOTCH, coef, z, state=d_no)
        want = np.stack([O.notch(x[c], coef, st_no[c]) for c in range(nchan)]).reshape(nchan, n)
        assert np.array_equal(z.cpu().numpy(), want)
        assert np.array_equal(d_no.cpu().numpy(), st_no)


@pytest.mark.parametrize("n,scale,prev", [(0, 1.0, 3.7), (1, 7.9, 3.5), (100, 50.0, 0.0), (100, 50.0, 1e4), (33, 0.2, 0.0),
                                           (5000, 1e6, 0.0), (2049, 300.0, 12.0)])
def test_dsp_normalize_bit_exact(n, scale, prev):
    """normalize(): the size_t running maximum as a reduction (also rows whose maximum truncates to 0: division by
    zero like the reference), returned maximum and scaled samples bit-identical."""
    rng = np.random.default_rng(n + int(scale))
    nchan = 9
    x = rng.normal(0, scale, (nchan, n))
    xd = _rows_dev(x)
    mx = torch.full((nchan,), prev, dtype=torch.float64, device="cuda")
    device.normalize(xd, mx)
    got, got_m = xd.cpu().numpy(), mx.cpu().numpy()
    for c in range(nchan):
        want, m = O.normalize(x[c], prev)
        assert got_m[c] == m
        assert np.array_equal(got[c], want, equal_nan=True)


@pytest.mark.parametrize("n,by", [(1, 2), (2, 2), (3, 2), (4, 2), (5, 2), (64, 2), (513, 2), (514, 2), (515, 4), (1000, 8),
                                  (5000, 4), (40, 1), (2054, 2), (2055, 2), (8192, 2), (10001, 2), (20000, 8)])
def test_dsp_fifth_order_decimate_bit_exact(n, by):
    """fifth_order_filter::decimate(double*): in place, log2(by) passes over the same n, state carried across two
    calls; whole row (written part and untouched tail) and the six state doubles bit-identical."""
    rng = np.random.default_rng(n * 16 + by)
    nchan = 5
    st = np.zeros((nchan, 6))
    d_st = torch.zeros(nchan, 6, dtype=torch.float64, device="cuda")
    for call in range(2):
        x = rng.normal(0, 10, (nchan, n))
        xd = _rows_dev(x)
        r = device.fifth_order_decimate(xd, by, d_st)
        got = xd.cpu().numpy()
        for c in range(nchan):
            want, rw = O.fifth_order_decimate(x[c], by, st[c])
            assert r == rw
            assert np.array_equal(got[c], want), (call, c)
        assert np.array_equal(d_st.cpu().numpy(), st)


@pytest.mark.parametrize("n,by", [(1, 2), (2, 4), (3, 2), (5, 2), (7, 16), (64, 2), (513, 2), (515, 4), (1000, 8), (5000, 4),
                                  (40, 1), (2055, 2), (8192, 2), (10001, 4)])
def test_dsp_fifth_order_block_overload_bit_exact(n, by):
    """fifth_order_filter::decimate(BBP_Block*): complex points, both rails, n_points shrinking pass by pass, state
    carried across two calls; whole rows, the twelve state doubles and the returned n_points equal the oracle's."""
    rng = np.random.default_rng(n * 32 + by)
    nchan, cap = 5, max(n, 3) + 11
    st = np.zeros((nchan, 12))
    d_st = torch.zeros(nchan, 12, dtype=torch.float64, device="cuda")
    for call in range(2):
        z = rng.normal(0, 10, (nchan, cap)) + 1j * rng.normal(0, 10, (nchan, cap))
        zd = torch.tensor(z, device="cuda")
        r = device.fifth_order_decimate_block(zd, n, by, d_st)
        got = zd.cpu().numpy()
        for c in range(nchan):
            want, rw = O.fifth_order_decimate_block(z[c], n, by, st[c])
            assert r == rw, (r, rw)
            assert np.array_equal(got[c], want), (call, c)
        assert np.array_equal(d_st.cpu().numpy(), st)


@pytest.mark.parametrize("n", [1, 2, 4, 8, 64, 512, 1024, 4096])
def test_dsp_fft(n):
    """fft(): bit-identical to the oracle in shared-math mode (same butterflies, same twiddle arithmetic), within
    1e-12 relative of the libm build (the reference's arithmetic, pinned to the compiled reference on the CPU)."""
    rng = np.random.default_rng(n)
    nchan = 7
    z = rng.normal(0, 1, (nchan, n)) + 1j * rng.normal(0, 1, (nchan, n))
    zd = torch.tensor(z, device="cuda")
    device.fft(zd)
    got = zd.cpu().numpy()
    for c in range(nchan):
        assert np.array_equal(got[c], O.fft(z[c], O.MATH_SHARED))
        ref = O.fft(z[c], O.MATH_LIBM)
        assert np.max(np.abs(got[c] - ref)) <= 1e-12 * max(1.0, np.abs(ref).max()) * np.log2(max(n, 2))


@pytest.mark.parametrize("n,nchan", [(16, 150001), (64, 40001), (128, 3), (256, 9001), (512, 5001), (1024, 2100), (2048, 1100), (4096, 301)])
def test_dsp_fft_many_rows(n, nchan, tune):
    """More rows than one pass of the persistent workgroups holds (each walks several tiles; a tile is several short
    transforms side by side, the last one ragged): rows repeat with period 97, so every row is checked, 97 against the
    oracle and the rest against their copies; under every workgroup size the library has."""
    rng = np.random.default_rng(n + nchan)
    base = rng.normal(0, 1, (97, n)) + 1j * rng.normal(0, 1, (97, n))
    want = np.stack([O.fft(base[c], O.MATH_SHARED) for c in range(97)])
    zb = torch.tensor(base, device="cuda")
    idx = torch.arange(nchan, device="cuda") % 97
    for nt, ct in ((0, 1), (0, 0), (256, 0), (512, 0), (1024, 0)):  # ct: compile-time-length instances on / off
        tune("SDRX_FFT_NT", nt)
        tune("SDRX_FFT_CT", ct)
        zd = zb[idx].contiguous()
        device.fft(zd)
        m = min(97, nchan)
        assert np.array_equal(zd[:m].cpu().numpy(), want[:m]), (nt, ct)
        assert torch.equal(torch.view_as_real(zd), torch.view_as_real(zd[:m][idx].contiguous())), (nt, ct)


@pytest.mark.parametrize("log2n,nchan", [(13, 3), (14, 2), (15, 1), (16, 1), (17, 2), (18, 1), (20, 1)])
def test_dsp_fft_longer_than_lds(log2n, nchan):
    """fft() above 4096 points (FFT.cpp:9-29 recurses for any power of two): permutation + 4096-point blocks in LDS + the
    upper stages (1, 2 or 3 at a time) from global memory; every row bit-identical to the oracle's recursion in shared-math
    mode, rows at a pitch larger than n untouched between them, and within 1e-12 x log2 N of numpy's transform."""
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    z = (rng.normal(0, 1, (nchan, n)) + 1j * rng.normal(0, 1, (nchan, n))).astype(np.complex128)
    pitch = n + 16
    buf = torch.full((nchan, pitch), 7.0 - 3.0j, dtype=torch.complex128, device="cuda")
    buf[:, :n] = torch.tensor(z, device="cuda")
    _lib.check(_lib.lib().sdrx_fft_dev(C.c_void_p(buf.data_ptr()), n, nchan, pitch, None), "sdrx_fft_dev")
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.all(got[:, n:] == 7.0 - 3.0j)
    for c in range(nchan):
        want = O.fft(z[c], O.MATH_SHARED)
        assert np.array_equal(got[c, :n], want), (log2n, c)
        assert np.max(np.abs(want - np.fft.fft(z[c]))) <= 1e-12 * log2n * np.abs(want).max()


def test_dsp_unsupported_and_empty():
    z = torch.zeros(2, 12, dtype=torch.complex128, device="cuda")
    with pytest.raises(RuntimeError):
        device.fft(z)  # not a power of two
    e = torch.zeros(0, 16, dtype=torch.float64, device="cuda")
    device.iir(device.IIR_DC_BLOCK, 0.995, e)
    device.normalize(e, torch.zeros(0, dtype=torch.float64, device="cuda"))

@pytest.mark.parametrize("cfg", [
    dict(downsample=10, squelch_level=0, post_downsample=1, deemph_a=0, dc_block=False),   # == sdrx_rtlfm_dev
    dict(downsample=10, squelch_level=0, post_downsample=4, deemph_a=3, dc_block=True),    # wbfm-like post chain
    dict(downsample=7, squelch_level=0, post_downsample=3, deemph_a=0, dc_block=True, rotate90=True),
    dict(downsample=10, squelch_level=300, post_downsample=1, deemph_a=5, dc_block=False), # squelch around the level
    dict(downsample=4, squelch_level=150, post_downsample=2, deemph_a=2, dc_block=True),
    dict(downsample=1, squelch_level=40, post_downsample=5, deemph_a=7, dc_block=True),
    dict(downsample=3, squelch_level=0, post_downsample=2, deemph_a=4, dc_block=False),    # a power of two: exact quotients
    dict(downsample=5, squelch_level=0, post_downsample=1, deemph_a=64, dc_block=True),
])
def test_rtlfm_full_demod_vs_oracle(cfg, expect_both=True):
    """sdrx_rtlfm_full_dev against the literal restatement of rtl_fm's full_demod (parity unpinned: rtl_fm.c needs
    librtlsdr): 9 channels with amplitudes on both sides of the squelch level (some channels squelched in some calls
    only, so squelch_hits counts up and resets), four calls of ragged lengths with every state carried: pcm rows
    (the whole buffer, including what low_pass_simple leaves past its output), the 5 + 4 state words per channel."""
    rng = np.random.default_rng(hash(tuple(sorted(cfg.items()))) % (1 << 32))
    nchan, cap = 9, 6000
    fm = device.RtlFmFull(nchan, **cfg)
    sts = np.zeros((nchan, 5), dtype=np.int32)
    posts = np.zeros((nchan, 4), dtype=np.int32)
    pcm_ref = np.zeros((nchan, cap), dtype=np.int16)
    pcm = torch.zeros(nchan, cap, dtype=torch.int16, device="cuda")
    posts[1::2, 0] = 1000000  # a de-emphasis average far outside the int16 range: the kernel's plain-division path
    posts[2::4, 1] = -777     # a carried dc_avg
    fm.post_state.copy_(torch.tensor(posts))
    # rotate_90 works on groups of 4 samples (the reference only ever passes whole radio buffers)
    lengths = (5004, 4, 4000, 3000) if cfg.get("rotate90") else (5003, 1, 4000, 2999)
    for call, n in enumerate(lengths):
        raw = np.zeros((nchan, n, 2), dtype=np.uint8)
        for c in range(nchan):
            amp = (2.0, 8.0, 14.0, 20.0, 30.0, 45.0, 60.0, 90.0, 120.0)[(c + 3 * call) % 9]
            ph = np.cumsum(rng.normal(0.01 * (c + 1), 0.05, n))  # a slowly turning, frequency-modulated carrier
            raw[c, :, 0] = np.clip(np.round(amp * np.cos(ph) + rng.normal(0, 1, n) + 127), 0, 255)
            raw[c, :, 1] = np.clip(np.round(amp * np.sin(ph) + rng.normal(0, 1, n) + 127), 0, 255)
        got_len = fm.demod(torch.tensor(raw, device="cuda"), pcm)
        for c in range(nchan):
            want_len = O.rtlfm_full(raw[c], cfg["downsample"], sts[c], posts[c], pcm_ref[c],
                                    **{k: v for k, v in cfg.items() if k != "downsample"})
            assert want_len == got_len
        assert np.array_equal(pcm.cpu().numpy(), pcm_ref), call
        assert np.array_equal(fm.state[fm.cur].cpu().numpy(), sts), call
        assert np.array_equal(fm.post_state.cpu().numpy(), posts), call
    if cfg["squelch_level"] and expect_both:
        assert posts[:, 2].max() > 0 and posts[:, 2].min() == 0  # both squelched and open channels were seen


# =============================================================================== bit-domain back end

@pytest.mark.parametrize("crc_enable", [True, False])
def test_deframe_vs_oracle(crc_enable):
    """sdrx_deframe_dev against the oracle's literal restatement of PacketFramer::deserialize + the receiver's
    loop: 24 channels = 6 random streams (good packets, bad crc, bad length, bare unique words, dense and
    sparse) cut at 4 lengths each, so that every end-of-stream case (a header exactly one packet before the
    end, the 500-bit early return, partial passes) is hit: same events, payloads, zero returns and consumed bits."""
    n = 70000
    streams, cuts = [], []
    for seed in range(6):
        full = framed_bit_stream(n, 500 + seed, packet_every=2500 if seed % 2 else 300, crc_enable=crc_enable)
        for m in (n, n - 1 - 97 * seed, 2033 + 700 * seed, 40000 + seed):
            b = np.full(n, 2, dtype=np.uint8)  # beyond a channel's own length: never read
            b[:m] = full[:m]
            streams.append(b)
            cuts.append(m)
    # channels have different lengths: one launch per distinct length group (n is per launch)
    for m in sorted(set(cuts)):
        idx = [i for i, c in enumerate(cuts) if c == m]
        bits = torch.tensor(np.stack([streams[i][:m] for i in idx]), device="cuda")
        got = device.deframe(bits, crc_enable, max_events=256)
        for g, i in zip(got, idx):
            want = O.deframe(streams[i][:m], crc_enable, max_events=256)
            for k in ("events", "consumed", "zero_returns"):
                assert g[k] == want[k], (m, i, k, g[k], want[k])
            for k in ("pos", "kind", "len", "crc", "payload"):
                assert np.array_equal(g[k], want[k]), (m, i, k)


def test_deframe_dense_length_errors():
    """Streams made almost only of unique words followed by a length above 250 (the reference's return -1 that
    removes one bit), every 40 bits, on 96 channels: the event path without a crc check back to back with the
    next window refill (it once lacked a barrier between the header read and that refill)."""
    rng = np.random.default_rng(4242)
    n, nchan = 30000, 96
    uw = np.array([1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 0], dtype=np.uint8)
    streams = rng.integers(0, 2, (nchan, n)).astype(np.uint8)
    for c in range(nchan):
        for q in range(int(rng.integers(0, 40)), n - 24, 40):
            streams[c, q:q + 16] = uw
            streams[c, q + 16:q + 24] = [1, 1, 1, 1, 1, 1, int(rng.integers(0, 2)), 1] if rng.random() < 0.8 else rng.integers(0, 2, 8)
    got = device.deframe(torch.tensor(streams, device="cuda"), True, max_events=1024)
    for c in range(0, nchan, 5):
        want = O.deframe(streams[c], True, max_events=1024)
        for k in ("events", "consumed", "zero_returns"):
            assert got[c][k] == want[k], (c, k, got[c][k], want[k])
        for k in ("pos", "kind", "len", "crc", "payload"):
            assert np.array_equal(got[c][k], want[k]), (c, k)


def test_deframe_edge_lengths_and_golden():
    """Every stream length around one and two packets (the receiver's loop needs more than 2032 bits; deserialize
    itself at least 2032), the reference's TEST_PacketFramer stream, streams without any header, and an event
    buffer that is too small (all events are still counted)."""
    first64 = [1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
               1, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 1]
    gold = np.zeros(5000, dtype=np.uint8)
    gold[:64] = first64
    gold[2032:2032 + 64] = first64
    noise = np.random.default_rng(1).integers(0, 2, 5000).astype(np.uint8)
    zeros = np.zeros(5000, dtype=np.uint8)
    for stream in (gold, noise, zeros, framed_bit_stream(5000, 9, packet_every=10)):
        for m in list(range(2025, 2045)) + list(range(4057, 4072)) + [0, 1, 16, 2531, 2532, 2533, 3100, 5000]:
            for crc_enable in (False, True):
                bits = torch.tensor(stream[:m].copy().reshape(1, -1), device="cuda")
                g = device.deframe(bits, crc_enable, max_events=8)[0]
                w = O.deframe(stream[:m], crc_enable, max_events=8)
                assert (g["events"], g["consumed"], g["zero_returns"]) == (w["events"], w["consumed"], w["zero_returns"]), (m, crc_enable)
                assert np.array_equal(g["pos"], w["pos"]) and np.array_equal(g["kind"], w["kind"]), (m, crc_enable)
                assert np.array_equal(g["payload"], w["payload"]), (m, crc_enable)
    dense = framed_bit_stream(60000, 3, packet_every=5, p_bad_crc=0.0, p_bad_len=0.0, p_false_uw=0.0)
    g = device.deframe(torch.tensor(dense.reshape(1, -1), device="cuda"), True, max_events=4)[0]
    w = O.deframe(dense, True, max_events=4)
    assert g["events"] == w["events"] > 4 and g["consumed"] == w["consumed"] and np.array_equal(g["payload"], w["payload"])


# =============================================================================== fp32 twins

@pytest.mark.parametrize("D,T", [(1, 64), (4, 129), (10, 64), (8, 64), (5, 33), (12, 100), (3, 64), (100, 64)])
def test_f32_twins_within_1e5(D, T):
    """dtype = SDRX_F32 (float samples, taps, state, outputs; fused multiply-add).  There is no fp32
    reference: the checker is the fp64 CPU path, tolerance north_star's 1e-5 relative --
    filtered samples: |err| <= 1e-5 * sum|h||x|; discriminator angle: <= 1e-5 * pi (mod 2 pi) on a
    constant-envelope FM signal; AM bits: equal wherever the fp64 envelope is not within 1e-4 of the
    threshold.  Two calls, history carried; split and interleaved layouts."""
    err, taps = O.fir_design(O.LPF, T, 2.4e6, 1.0e5)
    assert err == 0
    C_, n = 2, 40000 + 3
    res, ims = zip(*[fm_iq(n, seed=60 + c) for c in range(C_)])
    re64, im64 = np.stack(res), np.stack(ims)
    re = torch.tensor(re64, device="cuda", dtype=torch.float32)
    im = torch.tensor(im64, device="cuda", dtype=torch.float32)
    iq = torch.stack([re, im], dim=-1).contiguous()
    n_out = (n + D - 1) // D
    scale_i = [np.convolve(np.abs(np.concatenate([r, r])), np.abs(taps))[:2 * n] + 1e-30 for r in res]
    scale_q = [np.convolve(np.abs(np.concatenate([q, q])), np.abs(taps))[:2 * n] + 1e-30 for q in ims]
    for x in ((re, im), iq):
        ch = device.FirChain(taps, nchan=C_, decim=D, dtype=torch.float32)
        cd = device.FirChain(taps, nchan=C_, decim=D, dtype=torch.float32)
        ca = device.FirChain(taps, nchan=C_, decim=D, dtype=torch.float32)
        got = []
        for _ in range(2):
            o0 = torch.zeros(C_, n_out, dtype=torch.float32, device="cuda")
            o1 = torch.zeros(C_, n_out, dtype=torch.float32, device="cuda")
            ang = torch.zeros(C_, n_out, dtype=torch.float32, device="cuda")
            bits = torch.zeros(C_, n_out, dtype=torch.uint8, device="cuda")
            ch.fir(x, o0, o1)
            cd.fir_fmdisc(x, ang)
            ca.fir_am(x, bits)
            got.append((o0.cpu().numpy(), o1.cpu().numpy(), ang.cpu().numpy(), bits.cpu().numpy()))
        for c in range(C_):
            fi, fq = O.Fir(O.LPF, T, 2.4e6, 1.0e5), O.Fir(O.LPF, T, 2.4e6, 1.0e5)
            pre = np.zeros(2)
            for call, (o0, o1, ang, bits) in enumerate(got):
                yi = fi.execute(res[c], fast=True)[::D]
                yq = fq.execute(ims[c], fast=True)[::D]
                si = scale_i[c][call * n:(call + 1) * n][::D]
                sq = scale_q[c][call * n:(call + 1) * n][::D]
                assert np.max(np.abs(o0[c] - yi) / si) <= 1e-5, (D, T, c, call)
                assert np.max(np.abs(o1[c] - yq) / sq) <= 1e-5, (D, T, c, call)
                want = np.zeros(n_out)
                L.orc_fm_discriminator(yi.copy(), yq.copy(), n_out, pre, want, O.MATH_LIBM)
                mag = np.hypot(yi, yq)
                ok = mag > 10.0  # where the filtered vector is tiny its angle is ill-conditioned in any precision
                ok[1:] &= mag[:-1] > 10.0
                d = np.angle(np.exp(1j * (ang[c] - want)))
                assert np.max(np.abs(d[ok])) <= 1e-5 * np.pi, (D, T, c, call, np.max(np.abs(d[ok])))
                env = np.sqrt(yi * yi + yq * yq)
                sure = np.abs(env - 80.0) > 1e-4 * 80.0
                assert np.array_equal(bits[c][sure], (env >= 80.0).astype(np.uint8)[sure]), (D, T, c, call)


def test_host_chain_long_call_is_chunked_and_u8_entry_points():
    """The host layer cuts a long call into 2^22-sample chunks on two staging slots (state carried from
    chunk to chunk): 2 * 2^22 + 12345 samples in one call equal the oracle's single pass; the raw-byte
    entry points equal the oracle's conversion + chain; a second call continues the stream."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    n, D = 2 * (1 << 22) + 12345, 10
    raw = np.random.default_rng(77).integers(0, 256, (n, 2), dtype=np.uint8)
    ph = 25 * np.sin(2 * np.pi * np.arange(n) / 3000.0)
    raw[:, 0] = np.clip(np.round(127.0 + 90 * np.cos(ph)) + (raw[:, 0] & 7), 0, 255).astype(np.uint8)
    raw[:, 1] = np.clip(np.round(127.0 + 90 * np.sin(ph)) + (raw[:, 1] & 7), 0, 255).astype(np.uint8)
    re, im = O.u8_to_iq(raw)
    hI, hQ, pre = np.zeros(63), np.zeros(63), np.zeros(2)
    aI, aQ = np.zeros(63), np.zeros(63)
    want_ang = _oracle_chain_disc(taps, re, im, D, hI, hQ, pre)
    want_bits = _oracle_chain_am(taps, re, im, D, aI, aQ)
    c1, c2, c3, c4 = host.Chain(taps, D), host.Chain(taps, D), host.Chain(taps, D), host.Chain(taps, D)
    assert np.array_equal(c1.fir_fmdisc(re, im), want_ang)
    assert np.array_equal(c2.fir_am(re, im), want_bits)
    assert np.array_equal(c3.fir_fmdisc_u8(raw), want_ang)
    assert np.array_equal(c4.fir_am_u8(raw), want_bits)
    # continue all four streams with a short second call
    m = 50000
    want_ang2 = _oracle_chain_disc(taps, re[:m], im[:m], D, hI, hQ, pre)
    want_bits2 = _oracle_chain_am(taps, re[:m], im[:m], D, aI, aQ)
    assert np.array_equal(c1.fir_fmdisc(re[:m].copy(), im[:m].copy()), want_ang2)
    assert np.array_equal(c3.fir_fmdisc_u8(raw[:m].copy()), want_ang2)
    assert np.array_equal(c2.fir_am(re[:m].copy(), im[:m].copy()), want_bits2)
    assert np.array_equal(c4.fir_am_u8(raw[:m].copy()), want_bits2)
    # load_buffer_and_rotate90 over a call longer than two chunks, D = 3 (chunks must stay whole groups of 4 samples)
    err3, taps3 = O.fir_design(O.LPF, 33, 2.4e6, 1.0e5)
    n4 = n - n % 4
    r3, i3 = O.u8_to_iq(raw[:n4], 127.4, signed_char=True, rotate90=True)
    w3 = _oracle_chain_disc(taps3, r3, i3, 3, np.zeros(32), np.zeros(32), np.zeros(2))
    assert np.array_equal(host.Chain(taps3, 3).fir_fmdisc_u8(raw[:n4].copy(), 127.4, signed_char=True, rotate90=True), w3)
    # Baseband_Stream::load_buffer's conversion through the host entry point
    r2, i2 = O.u8_to_iq(raw[:m], 127.4, signed_char=True)
    w = _oracle_chain_disc(taps, r2, i2, D, np.zeros(63), np.zeros(63), np.zeros(2))
    assert np.array_equal(host.Chain(taps, D).fir_fmdisc_u8(raw[:m].copy(), 127.4, signed_char=True), w)


def test_fir_shapes_fuzz():
    """80 seeded random shapes through the device layer: taps 1..200, decimation 1..40 (every kernel family
    and every hand-over between them: systolic D in {5,8,10,12} with few and with too many stages,
    register blockings, gather tiles), 1..5 channels, ragged lengths, 3 calls with carried state, a random
    layout and epilogue each -- all bit-identical to the oracle."""
    rng = np.random.default_rng(20260101)
    for case in range(80):
        T = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 65, 100, 129, 200]))
        D = int(rng.choice([1, 2, 3, 4, 5, 5, 6, 8, 8, 10, 10, 12, 12, 16, 20, 25, 40]))
        C_ = int(rng.integers(1, 6))
        epi = ("store", "am", "disc")[int(rng.integers(0, 3))]
        layout = ("split", "inter")[int(rng.integers(0, 2))]
        err, taps = O.fir_design(O.LPF if T > 1 else O.HPF, T, 2.4e6, 1.0e5)
        assert err == 0
        lens = [int(rng.integers(1, 6000)) for _ in range(3)]
        data = [[fm_iq(n, seed=1000 * case + 10 * k + c) for c in range(C_)] for k, n in enumerate(lens)]
        ch = device.FirChain(taps, nchan=C_, decim=D)
        H = max(T - 1, 1)
        hist = [(np.zeros(H), np.zeros(H), np.zeros(2)) for _ in range(C_)]
        fir_i = [O.Fir(O.LPF if T > 1 else O.HPF, T, 2.4e6, 1.0e5) for _ in range(C_)]
        fir_q = [O.Fir(O.LPF if T > 1 else O.HPF, T, 2.4e6, 1.0e5) for _ in range(C_)]
        for k, n in enumerate(lens):
            re = torch.tensor(np.stack([d[0] for d in data[k]]), device="cuda")
            im = torch.tensor(np.stack([d[1] for d in data[k]]), device="cuda")
            x = (re, im) if layout == "split" else torch.stack([re, im], dim=-1).contiguous()
            n_out = (n + D - 1) // D
            tag = (case, T, D, C_, epi, layout, k, n)
            if epi == "store":
                o0 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
                o1 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
                ch.fir(x, o0, o1)
                for c in range(C_):
                    assert np.array_equal(o0[c].cpu().numpy(), fir_i[c].execute(data[k][c][0], fast=True)[::D]), tag
                    assert np.array_equal(o1[c].cpu().numpy(), fir_q[c].execute(data[k][c][1], fast=True)[::D]), tag
            elif epi == "am":
                bits = torch.zeros(C_, n_out, dtype=torch.uint8, device="cuda")
                ch.fir_am(x, bits)
                for c in range(C_):
                    want = _oracle_chain_am(taps, data[k][c][0], data[k][c][1], D, hist[c][0], hist[c][1])
                    assert np.array_equal(bits[c].cpu().numpy(), want), tag
            else:
                ang = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
                ch.fir_fmdisc(x, ang)
                for c in range(C_):
                    want = _oracle_chain_disc(taps, data[k][c][0], data[k][c][1], D, hist[c][0], hist[c][1], hist[c][2])
                    assert np.array_equal(ang[c].cpu().numpy(), want), tag


def test_more_channels_than_a_grid_dimension():
    """70000 channels (> 65535, the limit of grid.y) through the workgroup-tile kernel, the systolic kernel
    and the rtl_fm chain: a few channels spot-checked against the oracle, all equal to their duplicates."""
    C_, n = 70000, 96
    rng = np.random.default_rng(8)
    base = rng.integers(-127, 128, (7, n, 2)).astype(np.float64)
    idx = np.arange(C_) % 7
    x = torch.tensor(base, device="cuda")[torch.tensor(idx, device="cuda")].contiguous()
    for D, T in ((1, 8), (10, 16)):
        err, taps = O.fir_design(O.LPF, T, 2.4e6, 1.0e5)
        ch = device.FirChain(taps, nchan=C_, decim=D)
        n_out = (n + D - 1) // D
        o0 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
        o1 = torch.zeros(C_, n_out, dtype=torch.float64, device="cuda")
        ch.fir(x, o0, o1)
        got = o0.cpu().numpy()
        for c in (0, 1, 6, 65534, 65535, 65536, 69999):
            want = O.Fir(O.LPF, T, 2.4e6, 1.0e5).execute(base[idx[c], :, 0], fast=True)[::D]
            assert np.array_equal(got[c], want), (D, T, c)
        assert np.array_equal(got[7:], got[:-7])  # channel c + 7 carries the same data as channel c
    raw = torch.tensor((base + 127).astype(np.uint8), device="cuda")[torch.tensor(idx, device="cuda")].contiguous()
    fm = device.RtlFm(C_, 4)
    pcm = torch.zeros(C_, n // 4 + 2, dtype=torch.int16, device="cuda")
    m = fm.demod(raw, pcm)
    got = pcm.cpu().numpy()[:, :m]
    st = np.zeros(5, dtype=np.int32)
    assert np.array_equal(got[69999], O.rtlfm_chain((base[idx[69999]] + 127).astype(np.uint8), 4, st))
    assert np.array_equal(got[7:], got[:-7])


def test_empty_and_tiny_inputs():
    """Edge cases the reference's loops accept silently: no channels, no samples, fewer samples than the
    decimation factor or than the taps -- through every FIR kernel family; state is carried unchanged by
    an empty call and correctly by tiny ones."""
    for D, T in ((10, 64), (1, 64), (4, 129), (100, 64)):
        err, taps = O.fir_design(O.LPF, T, 2.4e6, 1.0e5)
        re, im = fm_iq(300, seed=5)
        ch = host.Chain(taps, D)
        hI, hQ, pre = np.zeros(T - 1), np.zeros(T - 1), np.zeros(2)
        for lo, hi in ((0, 0), (0, 1), (1, 1), (1, 3), (3, 3 + D - 1 if D > 1 else 4), (120, 300), (300, 300)):
            r, i = re[lo:hi].copy(), im[lo:hi].copy()
            got = ch.fir_fmdisc(r, i)
            want = _oracle_chain_disc(taps, r, i, D, hI, hQ, pre) if hi > lo else np.zeros(0)
            assert got.size == want.size and np.array_equal(got, want), (D, T, lo, hi)
    dev_ch = device.FirChain(taps, nchan=4, decim=10)
    x = torch.zeros(4, 0, 2, dtype=torch.float64, device="cuda")
    out = torch.zeros(4, 1, dtype=torch.float64, device="cuda")
    dev_ch.fir_fmdisc(x, out)  # n = 0: a no-op that keeps the state
    torch.cuda.synchronize()
    f = host.Filter(O.LPF, 64, 2.0e6, 1.0e5)
    f.execute(np.zeros(0))
    out = np.zeros(0, dtype=np.uint8)
    assert host.AM().demodulate(np.zeros(0), np.zeros(0), out, 0, 8) == 0


def test_fast_fma_within_tolerance():
    """fast_fma = 1 fuses multiply-add: not bit-exact; bound |err| <= 1e-12 * sum|h||x| (north_star's fp64 tolerance)."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    n = 100000
    re, im = radio_iq(n, 9)
    x = torch.tensor(np.stack([re, im], axis=-1)[None], device="cuda")
    ch = device.FirChain(taps, nchan=1, decim=1)
    o0 = torch.zeros(1, n, dtype=torch.float64, device="cuda")
    o1 = torch.zeros(1, n, dtype=torch.float64, device="cuda")
    ch.fir(x, o0, o1, fast_fma=True)
    torch.cuda.synchronize()
    want = O.Fir(O.LPF, 64, 2.0e6, 1.0e5).execute(re, fast=True)
    scale = np.convolve(np.abs(re), np.abs(taps))[:n] + 1e-300
    assert np.max(np.abs(o0[0].cpu().numpy() - want) / scale) <= 1e-12


def test_full_size_streaming_property_and_spot_checks():
    """BASELINE config 2 at its full size (one 2^28-sample launch, what bench.py times): (i) one launch == the same
    stream in 2^18-sample chunks with state carried, bit for bit; (ii) 4000 outputs spot-checked against the oracle,
    which recomputes each from its own 64-sample input window."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    n, D = 1 << 28, 10
    g = torch.Generator(device="cuda").manual_seed(1234)
    iq = torch.randint(-127, 128, (1, n, 2), generator=g, device="cuda").to(torch.float64)
    n_out = (n + D - 1) // D
    one = torch.zeros(1, n_out, dtype=torch.float64, device="cuda")
    device.FirChain(taps, 1, D).fir_fmdisc(iq, one)
    ch = device.FirChain(taps, 1, D)
    parts = []
    chunk = 1 << 18  # not a multiple of 10: use a chunk that is
    chunk = chunk - chunk % D
    for k in range(0, n, chunk):
        x = iq[:, k:k + chunk].contiguous()
        o = torch.zeros(1, (x.shape[1] + D - 1) // D, dtype=torch.float64, device="cuda")
        ch.fir_fmdisc(x, o)
        parts.append(o)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts, dim=1), one)
    rng = np.random.default_rng(0)
    ms = np.unique(np.concatenate([[0, 1, 2, n_out - 1], rng.integers(7, n_out, 4000)]))
    got = one[0].cpu().numpy()
    for m in ms:
        lo = max(0, (m - 1) * D - 63)
        seg = iq[0, lo:m * D + 1].cpu().numpy()
        re, im = np.ascontiguousarray(seg[:, 0]), np.ascontiguousarray(seg[:, 1])
        # filter the window from zero history; exact whenever the window covers all 64 taps or starts at 0
        fi = O.Fir(O.LPF, 64, 2.4e6, 1.0e5).execute(re, fast=True)
        fq = O.Fir(O.LPF, 64, 2.4e6, 1.0e5).execute(im, fast=True)
        cur = (fi[-1], fq[-1])
        prev = (fi[-1 - D], fq[-1 - D]) if m > 0 else (0.0, 0.0)
        want = O.fm_discriminator(np.array([prev[0], cur[0]]), np.array([prev[1], cur[1]]), math_mode=O.MATH_SHARED)[1]
        assert got[m] == want, m


# =============================================================================== PSK

def test_bpsk_golden_stages(golden):
    """TEST(BPSK, ...) ladder of TEST_BPSK.cpp + Full_ModDemod, through the drop-in stage methods."""
    b = host.BPSK()
    ps = G(golden, "TEST_BPSK__BPSK__decode_pulseShape__pulseShaped")[:800]
    iq = np.zeros(1600)
    iq[0::2] = ps
    re, im = np.zeros(200), np.zeros(200)
    assert b.decode_pulseShape(iq, re, im) == 200
    assert np.allclose(re[:198], G(golden, "TEST_BPSK__BPSK__decode_pulseShape__RxCorrelated")[:198], rtol=0, atol=1e-5)
    re = G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxCorrelated_real")[:1000].copy()
    im = G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxCorrelated_imag")[:1000].copy()
    host.BPSK().symbol_timing_synchronization(re, im, 1000)
    assert np.allclose(re, G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxSymbolSynchronized_real")[:1000], rtol=0, atol=1e-8)
    assert np.allclose(im, G(golden, "TEST_BPSK__BPSK__symbol_timing_synchronization__RxSymbolSynchronized_imag")[:1000], rtol=0, atol=1e-8)
    re = G(golden, "TEST_BPSK__BPSK__rx_phase_correction__RxCorrelated_real")[:1000].copy()
    im = G(golden, "TEST_BPSK__BPSK__rx_phase_correction__RxCorrelated_imag")[:1000].copy()
    assert host.BPSK().phase_shift_correction(re, im, 1000) == 500
    assert np.array_equal(re[1:500], G(golden, "TEST_BPSK__BPSK__rx_phase_correction__RxSymbolSynchronized_real")[1:500])
    x = G(golden, "TEST_BPSK__BPSK__phase_ambiguity_correction__RX_PHASE_CORRECTED")[:1000].copy()
    host.BPSK().phase_ambiguity_correction(x, None, 1000)
    assert np.array_equal(x[537:737], G(golden, "TEST_BPSK__BPSK__phase_ambiguity_correction__RX_PHASE_AMBIGUITY_DETECTED")[:200])
    # Full_ModDemod: TEST_Modulator_Demodulator.cpp:12-61
    bits = G(golden, "TEST_Modulator_Demodulator__BPSK__Full_ModDemod__input_stream")
    bp = host.BPSK(0x14EB, 16)
    mod = np.zeros(8000)
    bp.modulate(bits, 1000, mod, 8000)
    assert np.allclose(mod[:800], G(golden, "TEST_Modulator_Demodulator__BPSK__Full_ModDemod__modulated_stream")[:800], rtol=0, atol=1e-6)
    out = bp.demodulate(np.repeat(mod, 2))
    assert out.size == 1000
    assert np.array_equal(out[:990], G(golden, "TEST_Modulator_Demodulator__BPSK__Full_ModDemod__output_stream")[:990])


def test_qpsk_golden_stages(golden):
    """TEST(QPSK, Deconstructed_ModDemod) TEST_QPSK.cpp:409-507 and Full_ModDemod."""
    src = "TEST_QPSK__QPSK__Deconstructed_ModDemod"
    sym = G(golden, f"{src}__input_stream")
    q = host.QPSK()
    mr, mi = np.zeros(8000), np.zeros(8000)
    q.modulate(sym, 1000, mr, mi, 8000)
    assert np.allclose(mr, G(golden, f"{src}__real_modulated_stream"), rtol=0, atol=1e-8)
    assert np.allclose(mi, G(golden, f"{src}__imag_modulated_stream"), rtol=0, atol=1e-8)
    iq = np.empty(16000)
    iq[0::2], iq[1::2] = mr, mi
    re, im = np.zeros(2000), np.zeros(2000)
    assert q.decode_pulseShape(iq, re, im) == 2000
    assert np.allclose(re, G(golden, f"{src}__real_correlated"), rtol=0, atol=1e-7)
    assert np.allclose(im, G(golden, f"{src}__imag_correlated"), rtol=0, atol=1e-7)
    q.symbol_timing_synchronization(re, im, 2000)
    assert np.allclose(re, G(golden, f"{src}__real_timesync"), rtol=0, atol=1e-7)
    assert np.allclose(im, G(golden, f"{src}__imag_timesync"), rtol=0, atol=1e-7)
    assert q.phase_shift_correction(re, im, 2000) == 1000
    assert np.array_equal(re[:1000], G(golden, f"{src}__real_phasecorrect"))
    q.phase_ambiguity_correction(re, im, 1000)
    amb = G(golden, f"{src}__real_phaseambcorrect")
    assert np.array_equal(re[81:1000], amb[:919])
    out = host.QPSK().demodulate(iq)
    assert out.size == 1000 and np.array_equal(out[81:], amb[:919])


@pytest.mark.parametrize("kind", [O.BPSK, O.QPSK])
@pytest.mark.parametrize("phase_deg", [0.0, 90.0, 180.0, 270.0, 33.0])
def test_psk_blocks_vs_oracle(kind, phase_deg):
    """Three consecutive noisy blocks: decisions, all 24 loop-state doubles, counters and UW hits are
    IDENTICAL to the oracle (shared-math mode); the oracle's libm mode (== compiled reference,
    tests/test_oracle_vs_ref.py) makes the same decisions."""
    uw = 0xDEADBEEF600DF00D
    if kind == O.BPSK:
        obj, st, st_libm = host.BPSK(uw, 64), O.psk_new(O.BPSK, uw, 64, O.MATH_SHARED), O.psk_new(O.BPSK, uw, 64, O.MATH_LIBM)
    else:
        obj, st, st_libm = host.QPSK(), O.psk_new(O.QPSK, None, 16, O.MATH_SHARED), O.psk_new(O.QPSK, None, 16, O.MATH_LIBM)
    for blk in range(3):
        iq = psk_test_signal(kind, 2048, seed=100 * blk + int(phase_deg), phase_deg=phase_deg, snr_db=20.0,
                             timing_offset=3, uw_bpsk=uw, freq_offset=1e-5)
        got = obj.demodulate(iq)
        want = O.psk_demodulate(st, iq)
        assert np.array_equal(got, want)
        assert np.array_equal(got, O.psk_demodulate(st_libm, iq))
        assert np.array_equal(obj.loop_state(), st.loop_state())
        ncc = 1 if kind == O.BPSK else 4
        assert list(obj.state().cc)[:ncc] == list(st.cc)[:ncc]


def test_psk_bank_multichannel_device():
    """Device layer: 70 channels (two waves, ragged) of QPSK and BPSK, two blocks each, vs the oracle."""
    for kind in (O.BPSK, O.QPSK):
        C_, nsym = 70, 512
        n = nsym * 8
        bank = device.PskBank(kind, C_, n)
        sts = [O.psk_new(kind, math_mode=O.MATH_SHARED) for _ in range(C_)]
        for blk in range(2):
            sig = np.stack([psk_test_signal(kind, nsym, seed=1000 * blk + c, phase_deg=(37 * c) % 360, snr_db=18.0,
                                            timing_offset=c % 8).reshape(n, 2) for c in range(C_)])
            iq = torch.tensor(sig, device="cuda")
            out = torch.zeros(C_, bank.n_dec, dtype=torch.uint8, device="cuda")
            bank.demod(iq, out)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            hs = bank.host_states()
            for c in range(C_):
                want = O.psk_demodulate(sts[c], np.ascontiguousarray(sig[c].reshape(-1)))
                assert np.array_equal(got[c].astype(np.float64), want), (kind, blk, c)
                mine = np.frombuffer(bytes(hs[c])[: 8 * 24], dtype=np.float64)
                assert np.array_equal(mine, sts[c].loop_state())


def test_config3_full_shape_bank():
    """BASELINE config 3 at the shape bench.py times (2^14 QPSK channels x 2^12 samples, two blocks): the bank is
    filled with 64 distinct signals repeated 256 times, so (i) every copy must give the same decisions and state as
    the first (channels are independent wherever they sit in the grid), and (ii) the 64 distinct channels are checked
    against the oracle's full chain, decisions and all loop-state doubles, bit for bit."""
    nchan, n, distinct = 1 << 14, 1 << 12, 64
    bank = device.PskBank(_lib.QPSK, nchan, n)
    sts = [O.psk_new(O.QPSK, math_mode=O.MATH_SHARED) for _ in range(distinct)]
    sts_libm = [O.psk_new(O.QPSK, math_mode=O.MATH_LIBM) for _ in range(distinct)]  # the CPU path's own cos / sin
    for blk in range(2):
        sig = np.stack([psk_test_signal(O.QPSK, n // 8, seed=7000 + 100 * blk + c, phase_deg=(29 * c) % 360, snr_db=16.0,
                                        timing_offset=c % 8).reshape(n, 2) for c in range(distinct)])
        iq = torch.tensor(sig, device="cuda").repeat(nchan // distinct, 1, 1).contiguous()
        out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device="cuda")
        bank.demod(iq, out)
        torch.cuda.synchronize()
        o = out.view(nchan // distinct, distinct, -1)
        assert torch.equal(o, o[0:1].expand_as(o))
        stt = bank.states.view(nchan // distinct, distinct, -1)
        assert torch.equal(stt, stt[0:1].expand_as(stt))
        got, hs = out[:distinct].cpu().numpy(), bank.states[:distinct].cpu().numpy()
        for c in range(distinct):
            want = O.psk_demodulate(sts[c], np.ascontiguousarray(sig[c].reshape(-1)))
            assert np.array_equal(got[c][: want.size].astype(np.float64), want), (blk, c)
            assert np.array_equal(hs[c][: 8 * 24].view(np.float64), sts[c].loop_state()), (blk, c)
            # north_star's bit-exact bitstream is against the CPU path, i.e. libm's cos / sin in the carrier loop
            want_libm = O.psk_demodulate(sts_libm[c], np.ascontiguousarray(sig[c].reshape(-1)))
            assert np.array_equal(got[c][: want_libm.size].astype(np.float64), want_libm), (blk, c, "libm")


def test_config4_full_shape_bank():
    """BASELINE config 4 at the shape bench.py times (1024 channels x 2^20 samples of the config-2 chain in one launch): 16 distinct
    streams repeated 64 times: every copy equals the first; the 16 distinct rows equal the same streams run one channel
    at a time (a different grid, the same bits) and the oracle on their first 3000 outputs; carried state included."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    nchan, n, D, distinct = 1024, 1 << 20, 10, 16  # 16 GiB of input, 0.8 GiB of output
    g = torch.Generator(device="cuda").manual_seed(99)
    base = torch.randint(-127, 128, (distinct, n, 2), generator=g, device="cuda").to(torch.float64)
    iq = base.repeat(nchan // distinct, 1, 1).contiguous()
    n_out = (n + D - 1) // D
    out = torch.zeros(nchan, n_out, dtype=torch.float64, device="cuda")
    bankc = device.FirChain(taps, nchan, D)
    bankc.fir_fmdisc(iq, out)
    o = out.view(nchan // distinct, distinct, -1)
    assert torch.equal(o, o[0:1].expand_as(o))
    for c in range(distinct):
        one = torch.zeros(1, n_out, dtype=torch.float64, device="cuda")
        device.FirChain(taps, 1, D).fir_fmdisc(base[c:c + 1].contiguous(), one)
        assert torch.equal(one[0], out[c]), c
        m = 3000
        seg = base[c, : m * D].cpu().numpy()
        hi, hq, pre = np.zeros(63), np.zeros(63), np.zeros(2)
        want = np.zeros(m)
        L.orc_chain_fir_decim_disc(taps, 64, hi, hq, pre, np.ascontiguousarray(seg[:, 0]), np.ascontiguousarray(seg[:, 1]),
                                   m * D, D, want, O.MATH_SHARED)
        assert np.array_equal(out[c, :m].cpu().numpy(), want), c


def test_psk_loops_one_wave_and_two_wave_forms_agree(tune):
    """psk_loops_kernel (both loops interleaved in one wave), psk_loops_split_kernel (timing and carrier loop on two
    waves through an LDS ring) and psk_loops_tri_kernel (the interpolator's mu-independent prefix on a third wave, a
    second ring) are picked by channel count; forced each way they give the same decisions and state, and all match
    the oracle (odd lengths incl. ones that end inside a ring block of either kernel, 3 blocks, 70 channels)."""
    nchan = 70
    for n in (4096, 4096 - 8 * 5, 4 * (16 * 7 + 2), 4 * 14, 4 * 34):
        iqs = [psk_test_signal(O.BPSK if c % 2 else O.QPSK, n // 8, seed=300 + c, phase_deg=45.0 * c, snr_db=15.0,
                               timing_offset=c % 4) for c in range(nchan)]
        for kind, lk in ((O.QPSK, _lib.QPSK), (O.BPSK, _lib.BPSK)):
            x = torch.tensor(np.stack([np.asarray(z).reshape(-1, 2) for z in iqs]), device="cuda")
            res = {}
            for force in ("0", "1", "2"):
                tune("SDRX_PSK_SPLIT", force)
                b = device.PskBank(lk, nchan, n)
                outs = []
                for _ in range(3):
                    o = torch.zeros(nchan, b.n_dec, dtype=torch.uint8, device="cuda")
                    b.demod(x, o)
                    outs.append(o.cpu().numpy())
                assert ("psk_loops_kernel", "psk_loops_split_kernel", "psk_loops_tri_kernel")[int(force)] in _lib.lib().sdrx_last_kernel().decode()
                res[force] = (outs, b.states.cpu().numpy())
            for other in ("1", "2"):
                for a, bb in zip(res["0"][0], res[other][0]):
                    assert np.array_equal(a, bb), (kind, n, other)
                assert np.array_equal(res["0"][1], res[other][1]), (kind, n, other)
            st = O.psk_new(kind, math_mode=O.MATH_SHARED)
            for blk in range(3):
                want = O.psk_demodulate(st, iqs[0])
                assert np.array_equal(res["1"][0][blk][0][: want.size], want), (kind, n, blk)


def test_psk_stage_entry_points_equal_the_fused_call():
    """sdrx_psk_matched_filter_dev -> sdrx_psk_loops_dev (timing, then carrier, as separate launches and as
    one) -> sdrx_psk_ambiguity_dev give the decisions and loop state of sdrx_psk_demod_dev, bit for bit."""
    Lx = _lib.lib()
    nchan, n = 70, 4096
    iqs = [psk_test_signal(O.QPSK, n // 8, seed=200 + c, phase_deg=90.0 * (c % 4), snr_db=20.0, timing_offset=c % 3)
           for c in range(nchan)]
    iq = np.stack([np.asarray(z).reshape(-1, 2) for z in iqs])
    x = torch.tensor(np.ascontiguousarray(iq), device="cuda")
    ref = device.PskBank(_lib.QPSK, nchan, n)
    want = torch.zeros(nchan, ref.n_dec, dtype=torch.uint8, device="cuda")
    ref.demod(x, want)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for split in (False, True):
        b = device.PskBank(_lib.QPSK, nchan, n)
        mf = torch.zeros(nchan, n // 4, 2, dtype=torch.float64, device="cuda")
        got = torch.zeros(nchan, ref.n_dec, dtype=torch.uint8, device="cuda")
        _lib.check(Lx.sdrx_psk_matched_filter_dev(vp(x), n, nchan, n, vp(mf), n // 4, None))
        if split:
            _lib.check(Lx.sdrx_psk_loops_dev(_lib.QPSK, vp(mf), n // 4, nchan, n // 4, vp(b.states), vp(got), got.shape[1], 1, 0, None))
            _lib.check(Lx.sdrx_psk_loops_dev(_lib.QPSK, vp(mf), n // 4, nchan, n // 4, vp(b.states), vp(got), got.shape[1], 0, 1, None))
        else:
            _lib.check(Lx.sdrx_psk_loops_dev(_lib.QPSK, vp(mf), n // 4, nchan, n // 4, vp(b.states), vp(got), got.shape[1], 1, 1, None))
        _lib.check(Lx.sdrx_psk_ambiguity_dev(_lib.QPSK, vp(got), (n // 4 + 1) // 2, nchan, got.shape[1], vp(b.states), None))
        torch.cuda.synchronize()
        assert torch.equal(got, want), split
        assert torch.equal(b.states, ref.states), split


def test_psk_modulators_vs_oracle():
    """BPSK/QPSK::modulate incl. the byte-count memset (prior contents beyond len_mod/8 accumulate)."""
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, 500).astype(np.uint8)
    sym = rng.integers(0, 4, 500).astype(np.uint8)
    prior = rng.uniform(-1, 1, 4000)
    a, b = prior.copy(), prior.copy()
    host.BPSK().modulate(bits, 500, a, 4000)
    L.orc_bpsk_modulate(bits, 500, b, 4000)
    assert np.array_equal(a, b)
    ar, ai, br, bi = prior.copy(), prior[::-1].copy(), prior.copy(), prior[::-1].copy()
    host.QPSK().modulate(sym, 500, ar, ai, 4000)
    L.orc_qpsk_modulate(sym, 500, br, bi, 4000)
    assert np.array_equal(ar, br) and np.array_equal(ai, bi)


# =============================================================================== round 2: modulators, convolve, full shapes

def test_am_fm_modulator_kernels_vs_oracle():
    """sdrx_am_modulate_dev / sdrx_fm_modulate_dev over banks of bit streams (AM.h:41-54, FM.h:44-74): every row equals
    the oracle's modulate bit for bit; outputs the reference leaves unwritten keep their contents; repeat count 0,
    lengths that do not divide, bit times longer than the LDS tables."""
    rng = np.random.default_rng(5)
    for nchan, nb, len_mod in ((3, 100, 800), (5, 37, 300), (2, 10, 7), (1, 1, 1), (4, 1000, 2999), (2, 4097, 4097 * 3 + 1)):
        bits = rng.integers(0, 2, (nchan, nb)).astype(np.uint8)
        out = torch.full((nchan, len_mod + 5), -5.5, dtype=torch.float64, device="cuda")
        device.am_modulate(_rows_dev(bits), out, len_mod)
        got = out.cpu().numpy()
        for c in range(nchan):
            want = O.am_modulate(bits[c], len_mod, np.full(len_mod + 5, -5.5))
            assert np.array_equal(got[c], want), (nchan, nb, len_mod, c)
    for spb, tx, mark, space, nb in ((100, 100.0, 20.0, 10.0, 70), (1, 10e6, 19000.0, 20000.0, 70), (7, 2e6, 19000.0, 20000.0, 70),
                                     (2000, 10e6, 19000.0, 20000.0, 70), (9001, 10e6, 19000.0, 20000.0, 70),
                                     (16, 2e6, 19000.0, 20000.0, 3001), (1, 2e6, 19000.0, 20000.0, 2500),
                                     (101, 2e6, 19000.0, 20000.0, 500)):  # several workgroups per row, odd run starts
        nchan = 3
        bits = rng.integers(0, 2, (nchan, nb)).astype(np.uint8)
        bits[0, :3] = [0, 1, 255]  # any non-zero byte is a mark (FM.h:62 tests == 0)
        w1, w2 = device.fm_tables(tx, mark, space, spb, True)
        out = torch.full((nchan, nb * spb + 3), 9.75, dtype=torch.float64, device="cuda")
        device.fm_modulate(_rows_dev(bits), torch.tensor(w1, device="cuda"), torch.tensor(w2, device="cuda"), out)
        got = out.cpu().numpy()
        for c in range(nchan):
            assert np.array_equal(got[c, : nb * spb], O.fm_modulate(tx, mark, space, bits[c], spb)), (spb, c)
            assert np.all(got[c, nb * spb:] == 9.75)
        d1, d2 = device.fm_tables(tx, mark, space, spb, False)  # the demodulator's tables: another operation order
        o1, o2 = np.zeros(spb), np.zeros(spb)
        L.orc_fm_waves(tx, mark, space, spb, o1, o2)
        assert np.array_equal(d1, o1) and np.array_equal(d2, o2)


def test_fm_convolve_vs_oracle_and_demodulate_identity():
    """FM<T>::convolve (FM.h:83-108 with the repair's `ds`): sums start from the outputs' prior values; windows longer
    and shorter than ds; fewer samples than one bit; and demodulate == compare(convolve(mark), convolve(space))."""
    rng = np.random.default_rng(6)
    fm = host.FM(100, 100, 20, 10)
    for len1, len2, ds in ((1000, 100, 100), (1000, 50, 100), (1003, 150, 100), (10, 3, 1), (99, 100, 100), (4096, 16, 8), (777, 7, 50)):
        x = rng.uniform(-100, 100, len1)
        w = rng.uniform(-1, 1, len2)
        n_out = len(range(ds - 1, len1, ds))
        prior = rng.uniform(-5, 5, n_out + 2)
        want = prior.copy()
        L.orc_fm_convolve(x, len1, w, len2, want, ds)
        got = prior.copy()
        fm.convolve(x, len1, w, len2, got, n_out, ds)
        assert np.array_equal(got, want), (len1, len2, ds)
        dgot = torch.tensor(prior, device="cuda")
        device.fm_convolve(torch.tensor(x, device="cuda"), torch.tensor(w, device="cuda"), dgot, ds)
        assert np.array_equal(dgot.cpu().numpy(), want), (len1, len2, ds)
    bits = rng.integers(0, 2, 40).astype(np.uint8)
    mod = O.fm_modulate(100.0, 20.0, 10.0, bits, 100) + rng.normal(0, 0.05, 4000)
    w1, w2 = device.fm_tables(100.0, 20.0, 10.0, 100, False)
    op1, op2 = np.zeros(40), np.zeros(40)
    fm.convolve(mod, 4000, w1, 100, op1, 40, 100)
    fm.convolve(mod, 4000, w2, 100, op2, 40, 100)
    out = np.zeros(40, dtype=np.uint8)
    assert fm.demodulate(mod, mod, out, 4000, 100) == 40
    assert np.array_equal(out, (op1 > op2).astype(np.uint8)) and np.array_equal(out, bits)


def test_u8_input_with_fast_mode_2_runs_as_mode_1():
    """sdrx.h: shapes the MFMA kernel lacks run as mode 1.  Raw-byte input is one of them (the MFMA kernel reads fp64
    rails): FirChain.fir(uint8, fast_fma=2) must equal the fused VALU chain on (byte - bias), not fault."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    rng = np.random.default_rng(8)
    raw = rng.integers(0, 256, (2, 5003, 2)).astype(np.uint8)
    ch = device.FirChain(taps, nchan=2, decim=1)
    o0 = torch.zeros(2, 5003, dtype=torch.float64, device="cuda")
    o1 = torch.zeros_like(o0)
    ch.fir(torch.tensor(raw, device="cuda"), o0, o1, fast_fma=2)
    assert b"fir_tile_kernel" in _lib.lib().sdrx_last_kernel() and b"fma" in _lib.lib().sdrx_last_kernel()
    for c in range(2):
        re, im = O.u8_to_iq(raw[c])
        assert np.array_equal(o0[c].cpu().numpy(), O.Fir(O.LPF, 64, 2.0e6, 1.0e5).execute_fma(re))
        assert np.array_equal(o1[c].cpu().numpy(), O.Fir(O.LPF, 64, 2.0e6, 1.0e5).execute_fma(im))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_config5_full_shape_bank(mode):
    """BASELINE config 5 at the shape bench.py times (1024 channels x 2^18 real samples, T = 1024): 8 distinct rows
    repeated 128 times: every copy equals the first; the first 3000 outputs and the last 500 of the 8 distinct rows equal
    the oracle (mode 0: the reference's two-rounding chain; modes 1 and 2: the fused chain, bit for bit)."""
    T, nchan, n, distinct = 1024, 1024, 1 << 18, 8
    err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5, max_taps=4096)
    assert err == 0
    g = torch.Generator(device="cuda").manual_seed(55)
    base = torch.randint(-127, 128, (distinct, n), generator=g, device="cuda").to(torch.float64)
    x = base.repeat(nchan // distinct, 1).contiguous()
    out = torch.zeros(nchan, n, dtype=torch.float64, device="cuda")
    device.FirChain(taps, nchan, 1).fir(x, out, fast_fma=mode)
    o = out.view(nchan // distinct, distinct, -1)
    assert torch.equal(o, o[0:1].expand_as(o))
    hb = base.cpu().numpy()
    for c in range(distinct):
        f = O.Fir(O.LPF, T, 2.0e6, 1.0e5, max_taps=4096)
        head = f.execute_fma(hb[c, :3000]) if mode else f.execute(hb[c, :3000], fast=True)
        assert np.array_equal(out[c, :3000].cpu().numpy(), head), (mode, c)
        f2 = O.Fir(O.LPF, T, 2.0e6, 1.0e5, max_taps=4096)
        seg = hb[c, n - 500 - (T - 1):]
        tail = f2.execute_fma(seg) if mode else f2.execute(seg, fast=True)
        assert np.array_equal(out[c, n - 500:].cpu().numpy(), tail[T - 1:]), (mode, c)


def test_config3_full_shape_bank_bpsk():
    """test_config3_full_shape_bank for BPSK (2^14 channels x 2^12 samples, two blocks, 64-bit unique word as the
    receiver constructs it, rtlsdr_receiver.cpp:230)."""
    nchan, n, distinct, uw = 1 << 14, 1 << 12, 64, 0xDEADBEEF600DF00D
    bank = device.PskBank(_lib.BPSK, nchan, n, unique_word=uw, uw_len=64)
    sts = [O.psk_new(O.BPSK, uw, 64, O.MATH_SHARED) for _ in range(distinct)]
    sts_libm = [O.psk_new(O.BPSK, uw, 64, O.MATH_LIBM) for _ in range(distinct)]
    for blk in range(2):
        sig = np.stack([psk_test_signal(O.BPSK, n // 8, seed=8000 + 100 * blk + c, phase_deg=(31 * c) % 360, snr_db=16.0,
                                        timing_offset=c % 8, uw_bpsk=uw).reshape(n, 2) for c in range(distinct)])
        iq = torch.tensor(sig, device="cuda").repeat(nchan // distinct, 1, 1).contiguous()
        out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device="cuda")
        bank.demod(iq, out)
        torch.cuda.synchronize()
        o = out.view(nchan // distinct, distinct, -1)
        assert torch.equal(o, o[0:1].expand_as(o))
        stt = bank.states.view(nchan // distinct, distinct, -1)
        assert torch.equal(stt, stt[0:1].expand_as(stt))
        got, hs = out[:distinct].cpu().numpy(), bank.states[:distinct].cpu().numpy()
        for c in range(distinct):
            want = O.psk_demodulate(sts[c], np.ascontiguousarray(sig[c].reshape(-1)))
            assert np.array_equal(got[c][: want.size].astype(np.float64), want), (blk, c)
            assert np.array_equal(hs[c][: 8 * 24].view(np.float64), sts[c].loop_state()), (blk, c)
            assert int(hs[c][8 * 24: 8 * 24 + 4].view(np.int32)[0]) == sts[c].cc[0]
            want_libm = O.psk_demodulate(sts_libm[c], np.ascontiguousarray(sig[c].reshape(-1)))
            assert np.array_equal(got[c][: want_libm.size].astype(np.float64), want_libm), (blk, c, "libm")
            assert sts_libm[c].cc[0] == sts[c].cc[0]


@pytest.mark.parametrize("kind,lk", [(O.QPSK, _lib.QPSK), (O.BPSK, _lib.BPSK)])
def test_config3_in_surveys_split_1024_channels_x_2_16(kind, lk):
    """BASELINE config 3 in SURVEY 8d's own split -- 1024 channels x 2^16 samples, the shape bench.py's cfg3_1024ch line
    times and the one that takes the BLOCK PIPELINE by default (m = 2^14 >= 2048, nchan <= 2^13): 16 distinct framed
    signals repeated 64 times.  Every copy equals the first (decisions and states); the 16 distinct channels equal the
    oracle's full chain: decisions with shared math AND with libm, all 24 loop-state doubles, the confidence counters."""
    nchan, n, distinct = 1024, 1 << 16, 16
    bank = device.PskBank(lk, nchan, n)
    sts = [O.psk_new(kind, math_mode=O.MATH_SHARED) for _ in range(distinct)]
    sts_libm = [O.psk_new(kind, math_mode=O.MATH_LIBM) for _ in range(distinct)]
    sig = np.stack([psk_test_signal(kind, n // 8, seed=9100 + c, phase_deg=90.0 * (c % 4) + (0.0 if c < 8 else 7.0), snr_db=20.0,
                                    timing_offset=c % 8).reshape(n, 2) for c in range(distinct)])
    iq = torch.tensor(sig, device="cuda").repeat(nchan // distinct, 1, 1).contiguous()
    out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device="cuda")
    bank.demod(iq, out)
    chain = _lib.lib().sdrx_last_kernel().decode()
    assert chain.count("fir_tile_kernel") >= 2, chain  # cut in time into pieces: the pipelined path
    torch.cuda.synchronize()
    o = out.view(nchan // distinct, distinct, -1)
    assert torch.equal(o, o[0:1].expand_as(o))
    stt = bank.states.view(nchan // distinct, distinct, -1)
    assert torch.equal(stt, stt[0:1].expand_as(stt))
    got, hs = out[:distinct].cpu().numpy(), bank.states[:distinct].cpu().numpy()
    hits = 0
    for c in range(distinct):
        want = O.psk_demodulate(sts[c], np.ascontiguousarray(sig[c].reshape(-1)))
        assert np.array_equal(got[c][: want.size].astype(np.float64), want), c
        assert np.array_equal(hs[c][: 8 * 24].view(np.float64), sts[c].loop_state()), c
        ncc = 1 if kind == O.BPSK else 4
        assert list(hs[c][8 * 24: 8 * 24 + 4 * ncc].view(np.int32)) == list(sts[c].cc[:ncc]), c
        want_libm = O.psk_demodulate(sts_libm[c], np.ascontiguousarray(sig[c].reshape(-1)))
        assert np.array_equal(got[c][: want_libm.size].astype(np.float64), want_libm), (c, "libm")
        hits += int(hs[c][8 * 24 + 20: 8 * 24 + 24].view(np.int32)[0])
    assert hits > 0  # the unique word was found: the ambiguity fold ran on real hits


def test_config1_at_its_stated_size():
    """BASELINE config 1 as stated: ONE 2^20-sample I/Q buffer, Filter(LPF,64,2e6,1e5) on I and Q, AM threshold with
    ds = 1 and ds = 8, through sdrx_fir_am_dev; every bit equals the oracle's chain (OOK carrier of amplitude 100 on
    random bits + noise, so both bit values and samples near the threshold occur)."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    n = 1 << 20
    re, im, _ = ook_iq(n, seed=1, ds=8)
    x = torch.tensor(np.stack([re, im], axis=-1)[None], device="cuda")
    for ds in (1, 8):
        ch = device.FirChain(taps, 1, ds)
        bits = torch.zeros(1, ch.n_out(n), dtype=torch.uint8, device="cuda")
        ch.fir_am(x, bits)
        want = np.zeros(n + 1, dtype=np.uint8)
        m = L.orc_chain_fir_am(taps, 64, np.zeros(63), np.zeros(63), re, im, n, ds, want)
        assert m == bits.shape[1] and np.array_equal(bits[0].cpu().numpy(), want[:m]), ds
        assert 0.2 < want[:m].mean() < 0.8


@pytest.mark.parametrize("rate_out,rate_out2", [(170000, 32000), (48000, 48000), (240000, 24000), (100, 33), (7, 3), (1000000, 1)])
def test_rtlfm_low_pass_real_vs_oracle(rate_out, rate_out2):
    """low_pass_real (rtl_fm.c:341-361) behind sdrx_rtlfm_full_dev's rows: 70 channels of different lengths (0, 1,
    shorter than one window, thousands), three calls with now_lpr / prev_lpr_index carried; rows bit for bit (the whole
    buffer: what the in-place walk leaves behind its output included), result_len and the two state words.  Literal
    restatement as the checker (parity unpinned: rtl_fm.c is not buildable here)."""
    rng = np.random.default_rng(rate_out + rate_out2)
    nchan, cap = 70, 5000
    lpr = np.zeros((nchan, 2), dtype=np.int32)
    d_lpr = torch.zeros(nchan, 2, dtype=torch.int32, device="cuda")
    for call in range(3):
        lens = rng.integers(0, cap + 1, nchan)
        lens[:4] = [0, 1, 2, cap]
        rows = rng.integers(-32768, 32768, (nchan, cap)).astype(np.int16)
        post = np.zeros((nchan, 4), dtype=np.int32)
        post[:, 3] = lens
        d_rows, d_post = torch.tensor(rows, device="cuda"), torch.tensor(post, device="cuda")
        device.rtlfm_low_pass_real(d_rows, rate_out, rate_out2, d_post, d_lpr)
        for c in range(nchan):
            post[c, 3] = L.orc_rtlfm_low_pass_real(rows[c], int(lens[c]), rate_out, rate_out2, lpr[c])
        assert np.array_equal(d_rows.cpu().numpy(), rows), call
        assert np.array_equal(d_post.cpu().numpy(), post) and np.array_equal(d_lpr.cpu().numpy(), lpr), call
    rc = _lib.lib().sdrx_rtlfm_low_pass_real_dev(C.c_void_p(d_rows.data_ptr()), nchan, cap, 100, 101, C.c_void_p(d_post.data_ptr()),
                                                 C.c_void_p(d_lpr.data_ptr()), None)
    assert rc == -1  # rate_out2 > rate_out divides by zero in the reference: rejected


def test_fir_execute_batch_host_layer():
    """sdrx_fir_execute_batch: many Filter objects in one call (one staging block, one launch per design).  Seven
    objects -- four of one design, two of another, one with its error flag set -- three calls of ragged lengths; every
    buffer equals the same object run alone through the oracle, shift registers carried; afterwards the objects still
    work one at a time (their own state rows were updated)."""
    rng = np.random.default_rng(12)
    specs = [(O.LPF, 64, 2.4e6, 1.0e5, None)] * 3 + [(O.HPF, 51, 2.0e6, 3.0e5, None), (O.LPF, 64, 2.4e6, 1.0e5, None),
                                                     (O.HPF, 51, 2.0e6, 3.0e5, None), (O.LPF, 1001, 2.0e6, 1.0e5, None)]
    fs = [host.Filter(*sp) for sp in specs]
    os_ = [O.Fir(*sp) for sp in specs]
    assert fs[-1].get_error_flag() == -3
    for n in (1, 777, 5000):
        xs = [rng.integers(-127, 128, n).astype(np.float64) + rng.uniform(-0.5, 0.5, n) for _ in specs]
        bufs = [x.copy() for x in xs]
        host.fir_execute_batch(fs, bufs)
        for c, (o, x) in enumerate(zip(os_, xs)):
            assert np.array_equal(bufs[c], o.execute(x, fast=True)), (n, c)
    x = rng.uniform(-100, 100, 300)
    a = x.copy()
    fs[1].execute(a)
    assert np.array_equal(a, os_[1].execute(x, fast=True))
    with pytest.raises(RuntimeError):
        host.fir_execute_batch([fs[0], fs[0]], [np.zeros(4), np.zeros(4)])  # one object, two rows of state: refused


def test_legacy_entry_points_from_concurrent_radio_threads():
    """The receiver runs one demodulating host thread per RADIO_ID (rtlsdr_receiver.cpp:191-194, 2016-2019): four threads
    call am_gpu_demodulation / fm_gpu_demodulation with their own RADIO_ID at once, the FM threads with two different
    samples-per-bit so that the shared mark / space tables are rebuilt under them.  Every result equals the oracle's."""
    import threading
    Lx = _lib.lib()
    n, radios, rounds = 200000, 4, 6
    Lx.am_gpu_initialize(n, radios)
    Lx.fm_gpu_initialize(radios, n, 2000000, 100, 19000.0, 20000.0)
    rng = np.random.default_rng(13)
    res = [rng.integers(-127, 128, n).astype(np.float64) for _ in range(radios)]
    ims = [rng.integers(-127, 128, n).astype(np.float64) for _ in range(radios)]
    want_am = [O.am_demodulate(res[r], ims[r], 1 + r) for r in range(radios)]
    ds_of = [100, 50, 100, 50]
    want_fm = [O.fm_demodulate(2000000.0, 19000.0, 20000.0, res[r], ds_of[r]) for r in range(radios)]
    errors = []

    def radio_thread(r):
        try:
            for _ in range(rounds):
                out = np.zeros(n, dtype=np.uint8)
                Lx.am_gpu_demodulation(r, res[r].ctypes.data, ims[r].ctypes.data, out.ctypes.data, n, 1 + r)
                if not np.array_equal(out[: n // (1 + r)], want_am[r][: n // (1 + r)]):
                    errors.append(("am", r))
                out = np.zeros(n, dtype=np.uint8)
                Lx.fm_gpu_demodulation(r, res[r].ctypes.data, ims[r].ctypes.data, out.ctypes.data, n, ds_of[r])
                if not np.array_equal(out[: n // ds_of[r]], want_fm[r]):
                    errors.append(("fm", r))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=radio_thread, args=(r,)) for r in range(radios)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    Lx.am_gpu_free()
    Lx.fm_gpu_free()
    assert not errors, errors


@pytest.mark.parametrize("chunks,use_graph", [(8, 2), (5, 2), (5, 0), (1, 2), (8, 1), (5, 1)])
@pytest.mark.parametrize("epilogue,D", [("fmdisc", 10), ("am", 1), ("fir", 4)])
def test_fir_ring_replays_equal_one_long_call(epilogue, D, chunks, use_graph):
    """sdrx_fir_ring_*: `chunks` chunks of one stream per call: replayed from a HIP graph (use_graph 2; odd ring lengths
    alternate two graphs, one per state-buffer parity), issued one by one (0), or -- the ring being contiguous -- as one
    launch over the whole ring (1).  Three replays over fresh data == the same 3 * chunks * n samples in ONE call
    of the plain entry point on a second chain, bit for bit (Filter::execute is a stream: filt.cpp:256-259), final
    state included; the systolic, the workgroup-tile (first-use table creation before the capture) kernels."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    n = 4000
    n_out = (n + D - 1) // D
    g = torch.Generator(device="cuda").manual_seed(chunks * 100 + D)
    ring_x = torch.zeros(chunks, n, 2, dtype=torch.float64, device="cuda")
    odt = torch.uint8 if epilogue == "am" else torch.float64
    width = 2 * n_out if epilogue == "fir" else n_out  # "fir" stores interleaved complex
    ring_o = torch.zeros(chunks, width, dtype=odt, device="cuda")
    ch = device.FirChain(taps, 1, D)
    ring = device.FirRing(ch, ring_x, ring_o, epilogue, use_graph)
    ref = device.FirChain(taps, 1, D)
    for rep in range(3):
        data = torch.randint(-127, 128, (chunks, n, 2), generator=g, device="cuda").to(torch.float64)
        ring_x.copy_(data)
        ring.launch()
        whole = data.view(1, chunks * n, 2)
        if epilogue == "fmdisc":
            want = torch.zeros(1, chunks * n_out, dtype=torch.float64, device="cuda")
            ref.fir_fmdisc(whole, want)
        elif epilogue == "am":
            want = torch.zeros(1, chunks * n_out, dtype=torch.uint8, device="cuda")
            ref.fir_am(whole, want)
        else:
            want = torch.zeros(1, chunks * n_out, 2, dtype=torch.float64, device="cuda")
            ref.fir(whole, want, None)
        torch.cuda.synchronize()
        assert torch.equal(ring_o.view(-1), want.view(-1)), (rep, epilogue, chunks)
        assert torch.equal(ch.state[ch.cur], ref.state[ref.cur]), (rep, "carried state")


@pytest.mark.parametrize("kind,lk", [(O.QPSK, _lib.QPSK), (O.BPSK, _lib.BPSK)])
def test_psk_block_pipeline_equals_the_one_stream_form(kind, lk, tune):
    """sdrx_psk_demod_dev's block pipeline (a call cut in time into pieces: the matched filter of piece k+1 on the
    caller's stream beside the loops of piece k on the library's side stream; the matched filter's history taken from the
    samples in front of the piece, sdrx_fir_job.n_before; the carrier loop's call-locals carried from piece to piece):
    forced on and forced off give the same decisions and states, both equal the oracle; ragged piece lengths (a block
    that is no multiple of the piece size), two calls in a row, a call on a non-default stream."""
    nchan = 9
    for n in (4 * 2048, 4 * (2048 + 24) + 3, 4 * 513):
        iqs = [psk_test_signal(kind, n // 8 + 1, seed=400 + c, phase_deg=50.0 * c, snr_db=17.0, timing_offset=c % 5)[: 2 * n]
               for c in range(nchan)]
        x = torch.tensor(np.stack([np.asarray(z).reshape(-1, 2) for z in iqs]), device="cuda")
        res = {}
        for force in (0, 1):
            tune("SDRX_PSK_OVERLAP", force)
            b = device.PskBank(lk, nchan, n)
            outs = []
            for call in range(2):
                o = torch.zeros(nchan, b.n_dec, dtype=torch.uint8, device="cuda")
                if call == 1:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())  # o's zero fill and call 0's state writes come first
                    with torch.cuda.stream(side):
                        b.demod(x, o)
                    torch.cuda.synchronize()
                else:
                    b.demod(x, o)
                outs.append(o.cpu().numpy())
            chain = _lib.lib().sdrx_last_kernel().decode()
            assert chain.count("fir_tile_kernel") == (4 if force else 1), chain
            res[force] = (outs, b.states.cpu().numpy())
        for a, bb in zip(res[0][0], res[1][0]):
            assert np.array_equal(a, bb), (kind, n)
        assert np.array_equal(res[0][1], res[1][1]), (kind, n)
        for c in (0, nchan - 1):
            st = O.psk_new(kind, math_mode=O.MATH_SHARED)
            for call in range(2):
                want = O.psk_demodulate(st, np.ascontiguousarray(x[c].cpu().numpy().reshape(-1)))
                assert np.array_equal(res[1][0][call][c][: want.size].astype(np.float64), want), (kind, n, c, call)


@pytest.mark.timeout(120)
def test_psk_block_pipeline_with_many_live_streams(tune):
    """The block pipeline forks onto a side stream of the library and joins back through events.  Every caller stream has
    a side stream and events of its own (api.hip psk_pipe_for).  Here: 40 live non-blocking streams (the box gives a
    process 4 hardware queues, so streams share them), pipelined calls on 8 of them -- back to back on one stream, and
    interleaved across streams so that pieces of different calls sit in the queues at once -- plus the null stream, then
    the same on a second round; every result bit-identical to the one-stream form (SDRX_PSK_OVERLAP = 0) and the states too.
    Run once, under a timeout: a fork / join that could not be satisfied would show as a hang in the final synchronize."""
    nchan, n = 16, 1 << 15  # the shape that hung in round 2 (16 x 2^15: m = 8192 -> pipelined by default)
    iqs = [psk_test_signal(O.QPSK, n // 8 + 1, seed=700 + c, phase_deg=41.0 * c, snr_db=18.0, timing_offset=c % 5)[: 2 * n]
           for c in range(nchan)]
    x = torch.tensor(np.stack([np.asarray(z).reshape(-1, 2) for z in iqs]), device="cuda")
    streams = [torch.cuda.Stream() for _ in range(40)]  # all stay alive to the end of the test
    for s in streams:  # make every one of them a live queue user
        with torch.cuda.stream(s):
            torch.zeros(8, device="cuda").add_(1.0)
    users = [None] + streams[3::5][:8]  # the null stream + 8 of the 40
    ROUNDS, CALLS = 2, 3

    def run(overlap):
        tune("SDRX_PSK_OVERLAP", overlap)
        banks = [device.PskBank(_lib.QPSK, nchan, n) for _ in users]
        outs = [[torch.zeros(nchan, banks[0].n_dec, dtype=torch.uint8, device="cuda") for _ in range(ROUNDS * CALLS)] for _ in users]
        torch.cuda.synchronize()
        for r in range(ROUNDS):
            for call in range(CALLS):  # interleaved: call `call` of every user before anybody's next call
                for u, s in enumerate(users):
                    ctx = torch.cuda.stream(s) if s is not None else torch.cuda.stream(torch.cuda.default_stream())
                    with ctx:
                        banks[u].demod(x, outs[u][r * CALLS + call])
                        if overlap:
                            assert _lib.lib().sdrx_last_kernel().decode().count("fir_tile_kernel") >= 2  # it was pipelined
            with torch.cuda.stream(users[1]):  # and back to back on one stream
                for call in range(CALLS):
                    banks[1].demod(x, outs[1][r * CALLS + call])
        torch.cuda.synchronize()
        return [[o.cpu().numpy() for o in row] for row in outs], [b.states.cpu().numpy() for b in banks]

    o1, s1 = run(-1)  # the default choice: pipelined at this shape
    o0, s0 = run(0)
    for u in range(len(users)):
        assert np.array_equal(s0[u], s1[u]), u
        for k in range(ROUNDS * CALLS):
            assert np.array_equal(o0[u][k], o1[u][k]), (u, k)
    # all users but users[1] ran the same 6 calls from the same initial state
    for u in range(2, len(users)):
        assert np.array_equal(s1[u], s1[0]) and all(np.array_equal(a, b) for a, b in zip(o1[u], o1[0]))


def test_fir_job_n_before_tiles_a_stream_in_time():
    """sdrx_fir_job.n_before: a block that is a slice of a longer resident buffer takes its history from the samples in
    front of it.  A stream cut into 5 ragged slices, each its own launch with n_before = its offset, reproduces the
    one-launch outputs bit for bit: FIR store (tile kernel), decimating discriminator (systolic kernel, needs the
    previous output as well), raw-byte input with rotate_90 (slices at multiples of 4)."""
    err, taps = O.fir_design(O.LPF, 64, 2.4e6, 1.0e5)
    n = 40000
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randint(-127, 128, (2, n, 2), generator=g, device="cuda").to(torch.float64)
    raw = torch.randint(0, 256, (2, n, 2), generator=g, device="cuda").to(torch.uint8)
    cuts = [0, 8000, 8040, 20000, 39000, n]  # multiples of 40: whole decimation groups and whole rotate_90 groups
    Lx = _lib.lib()
    for name, D, inp in (("fir", 1, x), ("fmdisc", 10, x), ("fir_u8_rot", 4, raw)):
        n_out = (n + D - 1) // D
        def outs():
            return (torch.zeros(2, n_out, dtype=torch.float64, device="cuda"), torch.zeros(2, n_out, dtype=torch.float64, device="cuda"))
        ch = device.FirChain(taps, 2, D)
        if name == "fir_u8_rot":
            ch.set_u8_conversion(127.4, signed_char=True, rotate90=True)
        w0, w1 = outs()
        if name == "fmdisc":
            ch.fir_fmdisc(inp, w0)
        else:
            ch.fir(inp, w0, w1)
        g0, g1 = outs()
        ch2 = device.FirChain(taps, 2, D)
        if name == "fir_u8_rot":
            ch2.set_u8_conversion(127.4, signed_char=True, rotate90=True)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            layout = _lib.IQ_U8 if name == "fir_u8_rot" else _lib.IQ_INTERLEAVED
            j = ch2._job(layout, inp, None, hi - lo, n, g0, None if name == "fmdisc" else g1, n_out, False, 0)
            esz = inp.element_size() * 2
            j.in0 = C.c_void_p(inp.data_ptr() + lo * esz)
            j.out0 = C.c_void_p(g0.data_ptr() + (lo // D) * 8)
            if name != "fmdisc":
                j.out1 = C.c_void_p(g1.data_ptr() + (lo // D) * 8)
            j.n_before = lo
            fn = Lx.sdrx_fir_fmdisc_dev if name == "fmdisc" else Lx.sdrx_fir_dev
            _lib.check(fn(C.byref(j)), name)
            ch2.cur ^= 1
        torch.cuda.synchronize()
        assert torch.equal(g0, w0), name
        if name != "fmdisc":
            assert torch.equal(g1, w1), name
        assert torch.equal(ch2.state[ch2.cur], ch.state[ch.cur]), (name, "state after the last slice")
    j = ch2._job(_lib.IQ_INTERLEAVED, x, None, 100, n, g0, g1, n_out, False, 0)
    j.n_before = 10  # shorter than the filter's history: refused
    assert Lx.sdrx_fir_dev(C.byref(j)) == -1


def test_psk_carrier_loop_outside_the_common_range_of_sincos(tune):
    """The split and three-wave kernels run each block of the carrier loop on sincos's straight-line common case
    (|phase| <= 1.5e6) and repeat the block through the full function if any lane left it.  Channels that start
    near the limit (and cross it inside a block), far beyond it, at an absurd phase, and one fed an infinite sample
    give the oracle's decisions and loop state bit for bit (NaNs in the same places), in both kernel forms, next to
    ordinary channels in the same wave."""
    n = 4096
    starts = [0.0, 1.4999999e6, -1.49999995e6, 3.0e7, -2.5e9, 1.0e301, 0.0, 0.0]
    nchan = len(starts)
    for kind, lk in ((O.QPSK, _lib.QPSK), (O.BPSK, _lib.BPSK)):
        iqs = [np.asarray(psk_test_signal(kind, n // 8, seed=900 + c, phase_deg=30.0 * c, snr_db=12.0, timing_offset=c % 4),
                          dtype=np.float64).copy() for c in range(nchan)]
        iqs[6][2 * 2000] = np.inf  # one infinite I sample: the loops run on NaN from there on
        x = torch.tensor(np.stack([z.reshape(-1, 2) for z in iqs]), device="cuda")
        sts = []
        wants = []
        for c in range(nchan):
            st = O.psk_new(kind, math_mode=O.MATH_SHARED)
            st.dds_delay = starts[c]
            wants.append([O.psk_demodulate(st, iqs[c]) for _ in range(2)])
            sts.append(st)
        for force in ("2", "1", "0"):
            tune("SDRX_PSK_SPLIT", force)
            b = device.PskBank(lk, nchan, n)
            rows = b.states.cpu().numpy().copy()
            rows[:, : 8 * 24].view(np.float64)[:, 21] = starts
            b.states.copy_(torch.tensor(rows))
            for blk in range(2):
                o = torch.zeros(nchan, b.n_dec, dtype=torch.uint8, device="cuda")
                b.demod(x, o)
                got = o.cpu().numpy()
                for c in range(nchan):
                    assert np.array_equal(got[c][: wants[c][blk].size], wants[c][blk]), (kind, force, blk, c)
            hs = b.states.cpu().numpy()
            for c in range(nchan):
                assert np.array_equal(hs[c][: 8 * 24].view(np.float64), sts[c].loop_state(), equal_nan=True), (kind, force, c)
        assert np.isnan(sts[6].loop_state()).any() and not np.isnan(sts[7].loop_state()).any()


def _same_bits_or_both_nan(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def test_psk_loops_signs_of_zeros_and_nans_in_every_kernel_form(tune):
    """The loops' speculative blocks take SIGN_FUNC as 1.0 with the operand's sign and are repeated with the real
    three-valued function when a block met a zero (carrier and timing loop) or a NaN (timing loop, whose detector
    multiplies OLDER samples by the new sign).  Channels of silence, of signal with silent stretches (the matched
    filter's output becomes exactly zero 129 samples in), of negative zeros, with a NaN and with an infinite sample,
    next to ordinary ones in the same wave: decisions equal the oracle's and the 24 loop-state doubles are the same
    BITS (the sign of a zero included; NaNs in the same places), over two calls, in all three kernel forms."""
    n = 4096
    for kind, lk in ((O.QPSK, _lib.QPSK), (O.BPSK, _lib.BPSK)):
        iqs = [np.asarray(psk_test_signal(kind, n // 8, seed=700 + c, phase_deg=50.0 * c, snr_db=14.0, timing_offset=c % 4),
                          dtype=np.float64).copy() for c in range(10)]
        iqs[1][:] = 0.0                      # silence
        iqs[2][2 * 1000: 2 * 1700] = 0.0     # a silent stretch: zeros reach the loops, then signal again
        iqs[3][: 2 * 700] = 0.0              # zero padding in front
        iqs[4][2 * 1500 + 1] = np.nan        # one NaN on the Q rail
        iqs[5][:] = -0.0                     # negative zeros
        iqs[6][2 * 900] = -np.inf
        iqs[7][1::2] = 0.0                   # one rail silent throughout
        iqs[8][2 * 2048:] = 0.0              # silent from the middle of the call on
        nchan = len(iqs)
        x = torch.tensor(np.stack([z.reshape(-1, 2) for z in iqs]), device="cuda")
        sts, wants = [], []
        for c in range(nchan):
            st = O.psk_new(kind, math_mode=O.MATH_SHARED)
            wants.append([O.psk_demodulate(st, iqs[c]) for _ in range(2)])
            sts.append(st)
        for force in ("2", "1", "0"):
            tune("SDRX_PSK_SPLIT", force)
            b = device.PskBank(lk, nchan, n)
            for blk in range(2):
                o = torch.zeros(nchan, b.n_dec, dtype=torch.uint8, device="cuda")
                b.demod(x, o)
                got = o.cpu().numpy()
                for c in range(nchan):
                    assert np.array_equal(got[c][: wants[c][blk].size], wants[c][blk]), (kind, force, blk, c)
            hs = b.states.cpu().numpy()
            for c in range(nchan):
                assert _same_bits_or_both_nan(hs[c][: 8 * 24].view(np.float64), sts[c].loop_state()), (kind, force, c)
        assert np.isnan(sts[4].loop_state()).any() and not np.isnan(sts[9].loop_state()).any()


@pytest.mark.parametrize("kind,lk", [(O.BPSK, _lib.BPSK), (O.QPSK, _lib.QPSK)])
def test_psk_ambiguity_stage_unique_word_lengths_and_rows(kind, lk):
    """sdrx_psk_ambiguity_dev alone, on decision rows with planted unique words in every rotation (some overlapping,
    some cut by the end of the row): unique-word lengths that are and are not multiples of four (the one-wave form
    compares four symbols per operation), 0 and 64, row lengths on both sides of the one-wave / 256-thread switch, a
    QPSK unique word with out-of-range entries (its four rows coincide there: the symbol-by-symbol loop), two calls in
    a row (confidence counters carried).  Decisions, counters and hit count equal the oracle's."""
    Lx = _lib.lib()
    rng = np.random.default_rng(77 + kind)
    nsym = 2 if kind == O.BPSK else 4
    cases = [(8, None), (13, None), (16, None), (31, None), (64, None), (0, None), (1, None)]
    if kind == O.QPSK:
        bad = np.array(rng.integers(0, 4, 64), dtype=np.float64)
        bad[60] = 7.0  # not a symbol: QPSK.cpp:103 leaves this index at 0 in all four rows
        cases.append((16, bad))
    # (rows, channels): few channels take the 256-thread form past 2048 decisions, banks of >= 256 channels the one-wave
    # form for any length (rows staged through LDS in segments of 1920 positions)
    shapes = [(m, 5) for m in (255, 256, 300, 301, 2048, 2049, 5000)] + [(m, 257) for m in (1921, 3905, 5001)]
    for U, uw in cases:
        for m, nchan in shapes:
            if nchan > 5 and U not in (13, 64, 16):
                continue
            if kind == O.BPSK:
                word = int(rng.integers(0, 2 ** 63)) if uw is None else uw
                sts = [O.psk_new(kind, uw=word, uw_len=U, math_mode=O.MATH_SHARED) for _ in range(nchan)]
                b = device.PskBank(lk, nchan, 8 * m, unique_word=word, uw_len=U)
            else:
                word = np.array(rng.integers(0, 4, 64), dtype=np.float64) if uw is None else uw
                sts = [O.psk_new(kind, uw=word, uw_len=U, math_mode=O.MATH_SHARED) for _ in range(nchan)]
                b = device.PskBank(lk, nchan, 8 * m, unique_word=word, uw_len=U)
            rows = [np.array([sts[0].uw[k][(U - 1 - j) if kind == O.BPSK else (63 - j)] for j in range(U)])
                    for k in range(4 if kind == O.QPSK else 2)]
            for call in range(2):
                dec = rng.integers(0, nsym, (nchan, m)).astype(np.uint8)
                for c in range(nchan):
                    for _ in range(0 if U == 0 else 1 + 3 * c):
                        k = int(rng.integers(0, len(rows)))
                        pos = int(rng.integers(0, m))
                        seg = rows[k][: max(0, min(U, m - pos))]
                        dec[c, pos: pos + seg.size] = seg.astype(np.uint8)
                d = torch.tensor(dec, device="cuda")
                _lib.check(Lx.sdrx_psk_ambiguity_dev(lk, C.c_void_p(d.data_ptr()), m, nchan, m, C.c_void_p(b.states.data_ptr()), None),
                           "ambiguity")
                got = d.cpu().numpy()
                hs = b.host_states()
                for c in range(nchan):
                    want = dec[c].astype(np.float64)
                    nh = O.lib().orc_psk_ambiguity(C.byref(sts[c]), want, m, None, 0)
                    assert np.array_equal(got[c], want.astype(np.uint8)), (U, m, call, c)
                    assert list(hs[c].cc) == list(sts[c].cc), (U, m, call, c)
                    assert hs[c].n_hits == nh, (U, m, call, c)


def test_fir_job_without_state_rows_is_a_one_shot_call():
    """state_in = state_out = NULL: zero history, nothing carried -- the same outputs as a call on zeroed state rows,
    for the workgroup-tile and the systolic kernel, exact and fma modes; refused for the AM / discriminator chains
    and for fast mode 2."""
    Lx = _lib.lib()
    rng = np.random.default_rng(5)
    for T, D, nchan, n in ((129, 4, 3, 5000), (64, 10, 2, 70000), (64, 1, 2, 3000), (7, 3, 1, 100)):
        taps = rng.normal(0, 1, T)
        x = torch.tensor(rng.normal(0, 50, (nchan, n, 2)), device="cuda")
        n_out = (n + D - 1) // D
        for fma in (0, 1):
            ch = device.FirChain(taps, nchan, D)
            w0, w1 = (torch.zeros(nchan, n_out, dtype=torch.float64, device="cuda") for _ in range(2))
            ch.fir(x, w0, w1, fast_fma=fma)
            g0, g1 = (torch.full((nchan, n_out), 7.0, dtype=torch.float64, device="cuda") for _ in range(2))
            j = ch._job(_lib.IQ_INTERLEAVED, x, None, n, n, g0, g1, n_out, fma, 0)
            j.state_in = None
            j.state_out = None
            _lib.check(Lx.sdrx_fir_dev(C.byref(j)), "stateless")
            torch.cuda.synchronize()
            assert torch.equal(g0, w0) and torch.equal(g1, w1), (T, D, fma)
    j.fast_fma = 2
    assert Lx.sdrx_fir_dev(C.byref(j)) == -1
    j.fast_fma = 0
    assert Lx.sdrx_fir_fmdisc_dev(C.byref(j)) == -1
    j.state_in = C.c_void_p(ch.state[0].data_ptr())
    assert Lx.sdrx_fir_dev(C.byref(j)) == -1  # one of the two rows only


@pytest.mark.parametrize("ds", [64, 65, 100, 333, 2000])
def test_fm_demod_bits_per_wave_forms_match_oracle(ds, tune):
    """The FSK correlator's wave kernel with 64 and with 16 bits per wave (the second is picked for calls with few
    bits): the oracle's bits through the host class either way, ragged bit counts, and the same bits as each other at
    the device layer on integer noise (exact ties) with random taps."""
    rng = np.random.default_rng(2000 + ds)
    nbits = 333
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = O.fm_modulate(2.0e6, 19000.0, 20000.0, bits, ds) * 40 + rng.normal(0, 1.0, nbits * ds)
    x = np.concatenate([x, rng.normal(0, 1, ds // 2)])
    want = O.fm_demodulate(2.0e6, 19000.0, 20000.0, x, ds)
    for rows in ("64", "16", "0"):
        tune("SDRX_FSK_ROWS", rows)
        out = np.zeros(nbits + 4, dtype=np.uint8)
        m = host.FM(10e6, 2.0e6).demodulate(x, x, out, x.size, ds)
        assert m == nbits and np.array_equal(out[:m], want), rows
    w1 = torch.tensor(rng.normal(0, 1, ds), device="cuda")
    w2 = torch.tensor(rng.normal(0, 1, ds), device="cuda")
    for nb in (1, 15, 16, 17, 63, 65, 1000):
        noise = torch.tensor(rng.integers(-3, 4, nb * ds).astype(np.float64), device="cuda")
        got = {}
        for rows in ("64", "16"):
            tune("SDRX_FSK_ROWS", rows)
            o = torch.full((nb + 3,), 9, dtype=torch.uint8, device="cuda")
            device.fm_demod(noise, w1, w2, o, ds)
            got[rows] = o.cpu().numpy()
            assert (got[rows][nb:] == 9).all()
        assert np.array_equal(got["64"], got["16"]), nb


def test_compile_time_window_walk_instances_equal_the_run_time_walk(tune):
    """The two compile-time instances of the workgroup-tile kernel (T = 64, D = 1 with the AM epilogue: BASELINE config
    1; T = 129, D = 4 storing: the PSK matched filter) against the run-time walk (SDRX_FIR_CT=0) over three ragged
    calls with carried state, 3 channels: same bits / samples and same final state, and the kernel that ran is the one
    meant."""
    Lx = _lib.lib()
    rng = np.random.default_rng(64129)
    for T, D, epi in ((64, 1, "am"), (129, 4, "store")):
        taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5)[1] if T == 64 else rng.normal(0, 1, T)
        res = {}
        for ct in ("1", "0"):
            tune("SDRX_FIR_CT", ct)
            ch = device.FirChain(taps, 3, D)
            outs = []
            rs = np.random.default_rng(7)
            for n in (5000, 3001, 7777):
                x = torch.tensor(rs.integers(-127, 128, (3, n, 2)).astype(np.float64) * (3.0 if epi == "am" else 0.01), device="cuda")
                n_out = (n + D - 1) // D
                if epi == "am":
                    o = torch.zeros(3, n_out, dtype=torch.uint8, device="cuda")
                    ch.fir_am(x, o)
                    outs.append(o.cpu().numpy())
                else:
                    o0, o1 = (torch.zeros(3, n_out, dtype=torch.float64, device="cuda") for _ in range(2))
                    ch.fir(x, o0, o1)
                    outs.append(np.stack([o0.cpu().numpy(), o1.cpu().numpy()]))
                name = Lx.sdrx_last_kernel().decode()
                assert ("compile-time" in name) == (ct == "1"), (name, ct)
            res[ct] = (outs, ch.state[ch.cur].cpu().numpy())
        for a, b in zip(res["1"][0], res["0"][0]):
            assert np.array_equal(a, b), (T, D)
        assert np.array_equal(res["1"][1], res["0"][1]), (T, D, "state")
        if epi == "am":
            assert 0.05 < np.concatenate([o.ravel() for o in res["1"][0]]).mean() < 0.95
