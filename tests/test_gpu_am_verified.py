"""The verified fp32 forms of BASELINE config 1 (fir_amv_kernel: packed fma; fir_amv_mfma_kernel: matrix cores) against
the oracle's chain -- Filter::execute on I and Q (filt.cpp:248-273) + AM<T>::demodulate (AM.h:63-77) -- on inputs
ENGINEERED to sit on the decision threshold, on non-finite and extreme samples, with a wrong fp32 tap table, and
streamed in ragged chunks.  Bar: every bit equal to the oracle's, in every form (SDRX_FIR_AMV = 0 exact fp64 kernel,
2 packed fma, 3 fp32 matrix cores with 2 and 4 blocks per wave, 4 int8 matrix cores on the tiles that are 8-bit integers,
1 the default choice)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from signals import ook_iq

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from sdr_amd import _lib, device  # noqa: E402

L = O.lib()
FORMS = [(0, 4), (1, None), (2, 4), (3, 2), (3, 4), (4, None)]  # (SDRX_FIR_AMV, SDRX_AMV_NB); 4 = the int8 matrix form


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    device.init(0)
    yield
    torch.cuda.synchronize()


def _oracle_bits(taps, re, im, hI=None, hQ=None):
    T = taps.size
    hI = np.zeros(max(T - 1, 1)) if hI is None else hI
    hQ = np.zeros(max(T - 1, 1)) if hQ is None else hQ
    bits = np.zeros(re.size + 1, dtype=np.uint8)
    m = L.orc_chain_fir_am(taps, T, hI, hQ, np.ascontiguousarray(re), np.ascontiguousarray(im), re.size, 1, bits)
    return bits[:m]


def _stats(reset=True):
    a = (C.c_uint64 * 4)()
    _lib.check(_lib.lib().sdrx_fir_am_verify_stats(a, 1 if reset else 0), "stats")
    return list(a)


def _run_forms(tune, taps, re_rows, im_rows, want_rows, min_recomputed=0, split=False):
    """re_rows, im_rows: (nchan, n) float64; every form's bits against want_rows."""
    nchan, n = re_rows.shape
    re = torch.tensor(re_rows, device="cuda")
    im = torch.tensor(im_rows, device="cuda")
    x = (re, im) if split else torch.stack([re, im], dim=-1).contiguous()
    tune("SDRX_AMV_STATS", 1)
    for form, nb in FORMS:
        tune("SDRX_FIR_AMV", form)
        tune("SDRX_AMV_NB", nb)
        _stats()
        ch = device.FirChain(taps, nchan=nchan, decim=1)
        bits = torch.zeros(nchan, n, dtype=torch.uint8, device="cuda")
        ch.fir_am(x, bits)
        torch.cuda.synchronize()
        got = bits.cpu().numpy()
        kern = _lib.lib().sdrx_last_kernel().decode()
        if form == 0:
            assert "fir_amv" not in kern, kern
        elif form == 2:
            assert "fir_amv_kernel" in kern, kern
        elif form == 3:
            assert "fir_amv_mfma_kernel" in kern and "NB=%d" % nb in kern, kern
        elif form == 4:
            assert "fir_amv_i8_kernel" in kern, kern
        bad = np.argwhere(got != want_rows)
        assert bad.size == 0, (form, nb, kern, bad[:5])
        st = _stats()
        if form in (2, 3, 4):
            assert st[0] >= min_recomputed, (form, nb, st)
    return st


def _steady_bit(taps, c_re, c_im):
    n = taps.size + 6
    return int(_oracle_bits(taps, np.full(n, c_re), np.full(n, c_im))[-1])


@pytest.mark.parametrize("T,angle", [(64, 0.0), (64, 0.7), (64, np.pi / 4), (33, 2.1), (129, 0.3), (1, 0.0), (17, 1.0)])
def test_constant_carriers_bisected_onto_the_threshold(T, angle, tune):
    """A constant carrier a * (cos, sin) settles at |y| = a * |sum h| in the reference's summation order.  Bisect `a` over
    the doubles until two NEIGHBOURING doubles give different steady-state bits in the oracle, then run the 121 doubles
    around that point as 121 channels: every steady-state output sits within a few ulp of the threshold, far inside the
    fp32 bound, so every one of them must take the recomputation and come out with the oracle's bit."""
    if T == 1:
        taps = np.array([1.0])
    else:
        err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5)
        assert err == 0
    cr, ci = np.cos(angle), np.sin(angle)
    lo, hi = 1.0, 1000.0
    assert _steady_bit(taps, lo * cr, lo * ci) == 0 and _steady_bit(taps, hi * cr, hi * ci) == 1
    while np.nextafter(lo, np.inf) < hi:
        mid = 0.5 * (lo + hi)
        if _steady_bit(taps, mid * cr, mid * ci):
            hi = mid
        else:
            lo = mid
    amps = [hi]
    for _ in range(60):
        amps.append(np.nextafter(amps[-1], np.inf))
        amps.insert(0, np.nextafter(amps[0], 0.0))
    amps = np.array(amps)
    n = 300
    re = np.repeat((amps * cr)[:, None], n, axis=1)
    im = np.repeat((amps * ci)[:, None], n, axis=1)
    want = np.stack([_oracle_bits(taps, re[c], im[c]) for c in range(amps.size)])
    steady = want[:, T + 5:]
    assert steady.min() == 0 and steady.max() == 1  # the 121 channels straddle the threshold
    _run_forms(tune, taps, re, im, want, min_recomputed=amps.size * (n - T - 5))


def test_non_finite_and_extreme_samples(tune):
    """NaN, +-Inf, huge, tiny and denormal samples dropped into an OOK stream: the outputs they reach fail both certainty
    tests (or make the tile's bound unusable) and are recomputed in the reference's arithmetic."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    n = 40000
    rows_re, rows_im = [], []
    specials = [np.nan, np.inf, -np.inf, 1e300, -1e300, 1e-300, 5e-324, 1e30, 3.5e38, 7e38, 1e39, -1e39, 2.0 ** 127, 65504.0 * 4]
    for c, v in enumerate(specials):
        re, im, _ = ook_iq(n, seed=200 + c, ds=8)
        rng = np.random.default_rng(300 + c)
        for p in rng.integers(0, n, 7):
            (re if p & 1 else im)[p] = v
        re[n - 3] = v  # inside the carried history too
        rows_re.append(re)
        rows_im.append(im)
    re, im = np.stack(rows_re), np.stack(rows_im)
    with np.errstate(all="ignore"):
        want = np.stack([_oracle_bits(taps, re[c], im[c]) for c in range(re.shape[0])])
    _run_forms(tune, taps, re, im, want)
    _run_forms(tune, taps, re, im, want, split=True)


def test_a_wrong_fp32_tap_table_only_costs_time(tune):
    """sdrx_fir_job.taps_f32 is the caller's: a table that is not the fp64 taps rounded to float makes every workgroup
    refuse its bound (stats[2]) and recompute; the bits do not change."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    n = 30000
    re, im, _ = ook_iq(n, seed=9, ds=8)
    want = _oracle_bits(taps, re, im)
    x = torch.tensor(np.stack([re, im], axis=-1)[None], device="cuda")
    tune("SDRX_AMV_STATS", 1)
    for form, nb in FORMS[2:]:
        tune("SDRX_FIR_AMV", form)
        tune("SDRX_AMV_NB", nb)
        for which in ("scaled", "one_tap", "nan"):
            ch = device.FirChain(taps, nchan=1, decim=1)
            t32 = ch.taps_f32.clone()
            if which == "scaled":
                t32 *= 1.001
            elif which == "one_tap":
                t32[40] = torch.nextafter(t32[40], t32[40] * 4)
                t32[40] = torch.nextafter(t32[40], t32[40] * 4)
            else:
                t32[3] = float("nan")
            ch.taps_f32 = t32
            _stats()
            bits = torch.zeros(1, n, dtype=torch.uint8, device="cuda")
            ch.fir_am(x, bits)
            torch.cuda.synchronize()
            st = _stats()
            assert np.array_equal(bits[0].cpu().numpy(), want), (form, nb, which)
            assert st[2] > 0 and st[0] == n, (form, nb, which, st)


@pytest.mark.parametrize("T", [1, 2, 7, 16, 63, 64, 65, 129, 1000])
def test_verified_forms_streamed_in_ragged_chunks(T, tune):
    """State carried over ragged calls (1 sample, fewer than T, not a multiple of anything, more than one tile), and taps
    whose sum|h| puts the OOK envelope on both sides of the threshold."""
    if T <= 2:
        taps = np.array([0.9, -0.4][:T])
    else:
        err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5)
        taps = taps * (1.3 / (np.abs(taps).sum() if T <= 129 else abs(taps.sum())))  # (a long sinc: scaled by its DC gain)
    n = 150000
    re, im, _ = ook_iq(n, seed=50 + T, ds=8)
    # (... and a call that ends exactly on a tile boundary of every form -- 8192 = 4 x 2048 outputs: its last tile is an
    # interior one and still has to carry the state; found by tests/soak/soak_amv.py)
    cuts = [0, 1, 3, 40, 40 + 2 * T + 1, 9000, 9000 + 8192, 9000 + 8192 + 4097, 77777, 77777 + 8192, 77777 + 8192 + 16384, n]
    xd = torch.tensor(np.stack([re, im], axis=-1)[None], device="cuda")
    hI, hQ = np.zeros(max(T - 1, 1)), np.zeros(max(T - 1, 1))
    want = np.concatenate([_oracle_bits(taps, re[a:b], im[a:b], hI, hQ) for a, b in zip(cuts[:-1], cuts[1:])])
    assert 0.02 < want.mean() < 0.98
    for form, nb in FORMS[1:]:
        tune("SDRX_FIR_AMV", form)
        tune("SDRX_AMV_NB", nb)
        ch = device.FirChain(taps, nchan=1, decim=1)
        got = torch.zeros(1, n, dtype=torch.uint8, device="cuda")
        for a, b in zip(cuts[:-1], cuts[1:]):
            ch.fir_am(xd[:, a:b].contiguous(), got[:, a:b])  # (a view of the output row: the pitch is the view's stride)
        torch.cuda.synchronize()
        g = got[0].cpu().numpy()
        assert np.array_equal(g, want), (form, nb, np.flatnonzero(g != want)[:5])


def test_int8_form_on_integer_streams_and_mixed_tiles(tune):
    """The int8 matrix form (fixed-point taps, exact integer products) on what it is for -- 8-bit integer samples -- and on
    streams where only SOME tiles are: integer OOK with fractions / +-128 / 200 / NaN dropped into chosen tiles, a stream
    that starts integer and turns fractional half way (the call's probe says 'int8', the tiles decide), several channels,
    interleaved and split rails, ragged length.  Every bit equals the oracle's; the integer tiles did take the int8 path
    (stats[3] counts them)."""
    err, taps = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    n = 6 * 2048 + 1234
    # the plan kernel probes sample floor(u n / 1024) of channel u % nchan, u = 0 .. 1023: the anomalies go where it does not look
    probed = {c: {(u * n) // 1024 for u in range(1024) if u % 5 == c} for c in range(5)}

    def free(c, lo, k=1):  # the first run of k unprobed positions of channel c at or after lo
        p = lo
        while any(q in probed[c] for q in range(p, p + k)):
            p += 1
        return p
    rows_re, rows_im = [], []
    for c in range(5):
        re, im, _ = ook_iq(n, seed=400 + c, ds=8)
        re, im = np.round(re).clip(-127, 127), np.round(im).clip(-127, 127)
        if c == 1:
            re[free(c, 3000)] += 0.5   # one fraction: tile 1 takes the fp32 path
            im[free(c, 9000)] = 128.0  # +-128 is outside the symmetric range: tile 4
            re[free(c, 9001)] = -128.0
        if c == 2:
            p = free(c, 5000, 3)       # tile 2
            im[p:p + 3] = [200.0, np.nan, 1e9]
        if c == 3:
            p = free(c, 2048 * 3 + 5, 20)
            re[p:p + 20] += 0.25       # a run of fractions between two probed positions: tile 3
        rows_re.append(re)
        rows_im.append(im)
    re, im = np.stack(rows_re), np.stack(rows_im)
    with np.errstate(all="ignore"):
        want = np.stack([_oracle_bits(taps, re[c], im[c]) for c in range(re.shape[0])])
    tune("SDRX_AMV_STATS", 1)
    tune("SDRX_FIR_AMV", 4)
    for split in (False, True):
        d_re, d_im = torch.tensor(re, device="cuda"), torch.tensor(im, device="cuda")
        x = (d_re, d_im) if split else torch.stack([d_re, d_im], dim=-1).contiguous()
        _stats()
        ch = device.FirChain(taps, nchan=5, decim=1)
        bits = torch.zeros(5, n, dtype=torch.uint8, device="cuda")
        ch.fir_am(x, bits)
        ch.fir_am(x, bits)  # a second call continues the streams: compare that one too
        torch.cuda.synchronize()
        assert "fir_amv_i8_kernel" in _lib.lib().sdrx_last_kernel().decode()
        hI = [np.zeros(63) for _ in range(5)]
        hQ = [np.zeros(63) for _ in range(5)]
        with np.errstate(all="ignore"):
            for c in range(5):
                _oracle_bits(taps, re[c], im[c], hI[c], hQ[c])
                w2 = _oracle_bits(taps, re[c], im[c], hI[c], hQ[c])
                assert np.array_equal(bits[c].cpu().numpy(), w2), (split, c)
        st = _stats()
        # tiles 1 .. 5 of a channel are interior (tile 0 takes its history from the state rows, tile 6 is ragged); of those,
        # channel 1 loses two to the fp32 path, channels 2 and 3 one each (an anomaly within 112 samples of a tile's start
        # would take the tile before it along: none is); two calls
        assert st[3] == 2 * (5 + 3 + 4 + 4 + 5), st


@pytest.mark.parametrize("T", [16, 17, 49, 64, 65, 113, 129, 241, 242])
def test_int8_form_filter_lengths(T, tune):
    """Filter lengths around the int8 form's operand boundaries (T <= 64 NIT - 15: 49 | 50, 113 | 114, 177 | 178, 241 | 242;
    beyond 241 taps the call falls back to the fp32 matrix form), taps of very different magnitudes (fixed point relative to
    the largest), integer samples, two channels."""
    rng = np.random.default_rng(T)
    taps = rng.normal(0, 1, T) * np.exp(rng.uniform(-12, 0, T))  # magnitudes over five decades
    n = 5 * 2048 + 77
    re = rng.integers(-127, 128, (2, n)).astype(np.float64)
    im = rng.integers(-127, 128, (2, n)).astype(np.float64)
    env = np.abs(np.convolve(re[0] + 1j * im[0], taps)[T:n])
    taps *= 80.0 / np.median(env)  # the envelope straddles the threshold: about half the bits are ones
    want = np.stack([_oracle_bits(taps, re[c], im[c]) for c in range(2)])
    assert 0.2 < want.mean() < 0.8
    tune("SDRX_FIR_AMV", 4)
    tune("SDRX_AMV_STATS", 1)
    _stats()
    ch = device.FirChain(taps, nchan=2, decim=1)
    bits = torch.zeros(2, n, dtype=torch.uint8, device="cuda")
    ch.fir_am(torch.tensor(np.stack([re, im], axis=-1), device="cuda"), bits)
    torch.cuda.synchronize()
    kern = _lib.lib().sdrx_last_kernel().decode()
    assert ("fir_amv_i8_kernel" in kern) == (T <= 241), kern
    assert np.array_equal(bits.cpu().numpy(), want)
    st = _stats()
    assert (st[3] > 0) == (T <= 241), st


@pytest.mark.parametrize("T", [64, 129, 17])
def test_verified_forms_from_the_radios_raw_bytes(T, tune):
    """SDRX_IQ_U8 input with the plain conversion (byte - 127: rtlsdr_receiver.cpp:1587-1588): every verified form reads
    the bytes directly (the int8 form filters them as the integers they are; a tile that holds a byte of 255 = +128 is
    outside its symmetric range and takes the fp32 form), streamed over ragged calls incl. one that ends on a tile
    boundary; every bit equals the oracle's chain on the converted samples.  Other conversions (bias 127.4, signed char,
    rotate_90) keep the exact kernel."""
    if T == 64:
        err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5)
    else:
        err, taps = O.fir_design(O.LPF, T, 2.0e6, 1.0e5)
        taps = taps * 1.1
    n, nchan = 9 * 2048 + 333, 3
    rng = np.random.default_rng(T)
    raw = np.zeros((nchan, n, 2), dtype=np.uint8)
    for c in range(nchan):
        re, im, _ = ook_iq(n, seed=600 + c + T, ds=8)
        raw[c, :, 0] = np.round(re).clip(-127, 127) + 127
        raw[c, :, 1] = np.round(im).clip(-127, 127) + 127
    raw[1, 5000, 0] = 255  # +128: tile 2 of channel 1 is not int8
    raw[1, 5001, 1] = 0
    raw[2, rng.integers(0, n, 40), 1] = rng.integers(0, 256, 40).astype(np.uint8)
    x = raw.astype(np.float64) - 127.0
    cuts = [0, 4096, 4096 + 8192, 4096 + 8192 + 777, n]
    hI = [np.zeros(max(T - 1, 1)) for _ in range(nchan)]
    hQ = [np.zeros(max(T - 1, 1)) for _ in range(nchan)]
    want = np.stack([np.concatenate([_oracle_bits(taps, x[c, a:b, 0], x[c, a:b, 1], hI[c], hQ[c]) for a, b in zip(cuts[:-1], cuts[1:])])
                     for c in range(nchan)])
    assert 0.1 < want.mean() < 0.9
    d_raw = torch.tensor(raw, device="cuda")
    tune("SDRX_AMV_STATS", 1)
    for form, nb in FORMS:
        tune("SDRX_FIR_AMV", form)
        tune("SDRX_AMV_NB", nb)
        _stats()
        ch = device.FirChain(taps, nchan=nchan, decim=1)
        got = torch.zeros(nchan, n, dtype=torch.uint8, device="cuda")
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = torch.zeros(nchan, b - a, dtype=torch.uint8, device="cuda")
            ch.fir_am(d_raw[:, a:b].contiguous(), o)
            got[:, a:b] = o
        torch.cuda.synchronize()
        kern = _lib.lib().sdrx_last_kernel().decode()
        assert ("fir_amv" in kern) == (form != 0) and "IQ_U8" in kern, (form, kern)
        g = got.cpu().numpy()
        assert np.array_equal(g, want), (form, nb, np.argwhere(g != want)[:5])
        if form == 4:
            assert _stats()[3] > 0
    # another conversion: the exact kernel, still the oracle's bits
    tune("SDRX_FIR_AMV", None)
    ch = device.FirChain(taps, nchan=nchan, decim=1)
    ch.set_u8_conversion(bias=127.4, signed_char=True)
    o = torch.zeros(nchan, n, dtype=torch.uint8, device="cuda")
    ch.fir_am(d_raw, o)
    torch.cuda.synchronize()
    assert "fir_amv" not in _lib.lib().sdrx_last_kernel().decode()
    x2 = raw.view(np.int8).astype(np.float64) - 127.4
    w2 = np.stack([_oracle_bits(taps, x2[c, :, 0], x2[c, :, 1]) for c in range(nchan)])
    assert np.array_equal(o.cpu().numpy(), w2)


def test_raw_byte_kernel_edges_refused_plans_and_history_in_front(tune):
    """fir_amv_i8_bytes_kernel takes the tiles inside a raw-byte call whose rows sit on 16-byte boundaries; the first tile
    and a ragged last one go to fir_amv_i8_kernel on a grid of their own.  Bits equal the oracle's chain on byte - 127
      * for lengths around the tile (2048) and the edge rules: one tile, two, whole tiles only, a ragged tail, 3 channels
        with a row pitch wider than the call;
      * when the plan REFUSES the taps (all zero; a NaN tap; taps so small that the fixed-point scale overflows): every
        output of the inner tiles is recomputed in the reference's arithmetic;
      * when the call's history lies in FRONT of it in the same buffer (sdrx_fir_job.n_before): no edge tile at the front."""
    rng = np.random.default_rng(77)
    err, taps0 = O.fir_design(O.LPF, 64, 2.0e6, 1.0e5)
    assert err == 0
    Lx = _lib.lib()
    tune("SDRX_FIR_AMV", 4)
    tune("SDRX_AMV_STATS", 1)

    def raw_rows(nchan, n):
        raw = np.zeros((nchan, n, 2), dtype=np.uint8)
        for c in range(nchan):
            re, im, _ = ook_iq(n, seed=4000 + c + n, ds=8)
            raw[c, :, 0] = np.round(re).clip(-127, 128) + 127
            raw[c, :, 1] = np.round(im).clip(-127, 128) + 127
        raw[:, rng.integers(0, n, 30), rng.integers(0, 2, 30)] = rng.integers(0, 256, 30).astype(np.uint8)[None, :]
        raw[0, n // 2, 0] = 255
        return raw

    # ---- lengths and pitches
    for n, nchan, pitch in ((2048, 1, 2048), (4096, 2, 4096), (8 * 2048, 3, 8 * 2048 + 64), (5 * 2048 + 8 * 31, 2, 5 * 2048 + 8 * 31),
                            (40 * 2048, 1, 40 * 2048)):
        raw = raw_rows(nchan, pitch)[:, :, :]
        x = raw.astype(np.float64) - 127.0
        want = np.stack([_oracle_bits(taps0, x[c, :n, 0], x[c, :n, 1]) for c in range(nchan)])
        d_raw = torch.tensor(raw, device="cuda")
        ch = device.FirChain(taps0, nchan=nchan, decim=1)
        out = torch.zeros(nchan, n, dtype=torch.uint8, device="cuda")
        j = ch._job(_lib.IQ_U8, d_raw, None, n, pitch, out, None, n, False, 0)
        _stats()
        assert Lx.sdrx_fir_am_dev(C.byref(j)) == 0
        torch.cuda.synchronize()
        kern = Lx.sdrx_last_kernel().decode()
        assert ("fir_amv_i8_bytes_kernel" in kern) == (n > 2048), (n, kern)
        assert np.array_equal(out.cpu().numpy(), want), (n, nchan, pitch)
        assert _stats()[3] == nchan * (n // 2048 - 1), (n, "tiles on the int8 path")

    # ---- plans that refuse the taps
    n, nchan = 6 * 2048, 2
    raw = raw_rows(nchan, n)
    x = raw.astype(np.float64) - 127.0
    d_raw = torch.tensor(raw, device="cuda")
    nan_taps = taps0.copy()
    nan_taps[17] = np.nan
    for name, taps in (("zero", np.zeros(64)), ("nan", nan_taps), ("tiny", taps0 * 1e-306)):
        with np.errstate(all="ignore"):
            want = np.stack([_oracle_bits(taps, x[c, :, 0], x[c, :, 1]) for c in range(nchan)])
        ch = device.FirChain(taps, nchan=nchan, decim=1)
        out = torch.full((nchan, n), 7, dtype=torch.uint8, device="cuda")
        _stats()
        ch.fir_am(d_raw, out)
        torch.cuda.synchronize()
        assert "fir_amv_i8_bytes_kernel" in Lx.sdrx_last_kernel().decode(), name
        assert np.array_equal(out.cpu().numpy(), want), name
        st = _stats()
        assert st[3] == 0 and st[0] >= nchan * 4 * 2048, (name, st)  # no tile on the integer path; the inner tiles recomputed

    # ---- history in front of the call
    N, off, nchan = 12 * 2048, 2048 + 8 * 40, 2
    raw = raw_rows(nchan, N)
    x = raw.astype(np.float64) - 127.0
    want = np.stack([_oracle_bits(taps0, x[c, :, 0], x[c, :, 1]) for c in range(nchan)])[:, off:]
    d_raw = torch.tensor(raw, device="cuda")
    ch = device.FirChain(taps0, nchan=nchan, decim=1)
    out = torch.zeros(nchan, N - off, dtype=torch.uint8, device="cuda")
    j = ch._job(_lib.IQ_U8, d_raw[:, off:], None, N - off, N, out, None, N - off, False, 0)
    j.n_before = off
    _stats()
    assert Lx.sdrx_fir_am_dev(C.byref(j)) == 0
    torch.cuda.synchronize()
    assert "fir_amv_i8_bytes_kernel" in Lx.sdrx_last_kernel().decode()
    assert np.array_equal(out.cpu().numpy(), want)
    assert _stats()[3] == nchan * ((N - off) // 2048)  # the first tile is an inner tile here; the ragged last one is not
