"""The oracle (and, on a GPU, the HIP path) against vectors produced by the REFERENCE ITSELF:
tests/golden/ref_fixtures.npz was written in the build container by tests/golden/gen_ref_fixtures.py from the
reference's own classes compiled out of /root/reference (oracle/_ref).  The fixture is data (inputs and outputs);
it pins the restatements wherever these tests run, also where neither /root/reference nor oracle/_ref exists.
Bar: bit-exact (the oracle in libm mode is the reference's arithmetic); the fft in shared-math mode within 1e-12."""
import os

import numpy as np
import pytest

import _oracle as O

F = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures.npz"))
L = O.lib()


@pytest.mark.parametrize("name", ["lpf64", "hpf51", "bpf129"])
def test_filter_fixture(name):
    ftype, T, Fs, Fx, Fu = F[f"fir_{name}_params"]
    f = O.Fir(int(ftype), int(T), Fs, Fx, None if Fu < 0 else Fu)
    assert f.err == 0 and np.array_equal(f.taps, F[f"fir_{name}_taps"])
    x = F[f"fir_{name}_in"]
    got = np.concatenate([f.execute(x[:300]), f.execute(x[300:])])
    assert np.array_equal(got, F[f"fir_{name}_out"])


def test_am_fixture():
    for ds in (1, 8, 7):
        assert np.array_equal(O.am_demodulate(F["am_re"], F["am_im"], ds), F[f"am_bits_ds{ds}"])


@pytest.mark.parametrize("kind,tag", [(O.BPSK, "bpsk"), (O.QPSK, "qpsk")])
def test_psk_block_fixture(kind, tag):
    st = O.psk_new(kind, 0xDEADBEEF600DF00D, 64, O.MATH_LIBM) if kind == O.BPSK else O.psk_new(kind, None, 16, O.MATH_LIBM)
    st2 = O.psk_new(kind, 0xDEADBEEF600DF00D, 64, O.MATH_SHARED) if kind == O.BPSK else O.psk_new(kind, None, 16, O.MATH_SHARED)
    for blk in range(2):
        iq = np.ascontiguousarray(F[f"{tag}_iq{blk}"])
        assert np.array_equal(O.psk_demodulate(st, iq), F[f"{tag}_dec{blk}"])
        assert np.array_equal(st.loop_state(), F[f"{tag}_state{blk}"])
        ncc = 1 if kind == O.BPSK else 4
        assert list(st.cc)[:ncc] == list(F[f"{tag}_cc{blk}"][:ncc])
        assert np.array_equal(O.psk_demodulate(st2, iq), F[f"{tag}_dec{blk}"])  # the device's math makes the same decisions


def test_deframe_crc_confidence_fixtures():
    ev = O.deframe(np.ascontiguousarray(F["deframe_bits"]), True, max_events=64)
    for k in ("pos", "kind", "len", "crc", "payload"):
        assert np.array_equal(np.asarray(ev[k]), F[f"deframe_{k}"]), k
    assert [ev["events"], ev["consumed"], ev["zero_returns"]] == list(F["deframe_summary"])
    msg = np.ascontiguousarray(F["crc_msg"])
    assert [O.crc8(msg), O.crc8(msg[:17])] == list(F["crc_val"])
    assert L.orc_ones_zeros_confidence(np.ascontiguousarray(F["conf_re"]), np.ascontiguousarray(F["conf_im"]), 3000, 37, 5) == F["conf_out"][0]


def test_dsp_primitive_fixtures():
    x = np.ascontiguousarray(F["dsp_in"])
    assert np.array_equal(O.dc_block(x, 0.995), F["dsp_dc_block"])
    assert np.array_equal(O.lowpass1(x, L.orc_lowpass1_coef(3000.0, 1.0 / 48000.0), np.zeros(1)), F["dsp_lowpass1"])
    assert np.array_equal(O.notch(x, O.notch_design(0.9, 0.05), np.zeros(4)), F["dsp_notch"])
    assert np.array_equal(O.integrate(x, 1.0 / 48000.0), F["dsp_integrate"])
    y, m = O.normalize(x, 0.0)
    assert m == F["dsp_normalize_max"][0] and np.array_equal(y, F["dsp_normalize"])
    y, r = O.fifth_order_decimate(x, 4, np.zeros(6))
    assert r == F["dsp_fifth_len"][0] and np.array_equal(y, F["dsp_fifth"])
    y, r = O.fifth_order_decimate_block(np.ascontiguousarray(F["dsp_fifth_block_in"]), 257, 4, np.zeros(12))
    assert r == F["dsp_fifth_block_len"][0] and np.array_equal(y, F["dsp_fifth_block"])
    z = np.ascontiguousarray(F["dsp_fft_in"])
    assert np.array_equal(O.fft(z, O.MATH_LIBM), F["dsp_fft"])
    assert np.max(np.abs(O.fft(z, O.MATH_SHARED) - F["dsp_fft"])) <= 1e-12 * 8 * np.abs(F["dsp_fft"]).max()


def _long_fft_inputs():
    rng3 = np.random.default_rng(20260303)  # gen_ref_fixtures.py's round-3 generator, same draw order
    return {n: (rng3.normal(0, 1, n) + 1j * rng3.normal(0, 1, n)).astype(np.complex128) for n in (8192, 65536)}


def test_long_fft_fixtures():
    """fft above 4096 points, pinned to the compiled reference (FFT.cpp:9-29): the oracle in libm mode reproduces its 8192
    and 65536-point outputs bit for bit (the latter through every 8th value and the sha256 of all of them); the
    shared-math mode, which the device computes, within 1e-12 x log2 N of the largest bin."""
    import hashlib
    zs = _long_fft_inputs()
    a = O.fft(zs[8192], O.MATH_LIBM)
    assert np.array_equal(a, F["dsp_fft_8192"])
    b = O.fft(zs[65536], O.MATH_LIBM)
    assert np.array_equal(b[::8], F["dsp_fft_65536_every8"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(b.tobytes()).digest(), dtype=np.uint8), F["dsp_fft_65536_sha256"])
    for n, want in ((8192, a), (65536, b)):
        got = O.fft(zs[n], O.MATH_SHARED)
        assert np.max(np.abs(got - want)) <= 1e-12 * np.log2(n) * np.abs(want).max()


@pytest.mark.gpu
def test_device_long_fft_against_reference_fixtures():
    """sdrx_fft_dev at 8192 and 65536 points against the reference's outputs: within 1e-12 x log2 N of the largest bin
    (the twiddles are sdrx_math.h's, not libm's), and bit for bit against the oracle in shared-math mode."""
    import torch
    from sdr_amd import device
    device.init(0)
    zs = _long_fft_inputs()
    for n in (8192, 65536):
        zd = torch.tensor(zs[n][None, :], device="cuda")
        device.fft(zd)
        got = zd[0].cpu().numpy()
        assert np.array_equal(got, O.fft(zs[n], O.MATH_SHARED))
        if n == 8192:
            assert np.max(np.abs(got - F["dsp_fft_8192"])) <= 1e-12 * 13 * np.abs(F["dsp_fft_8192"]).max()
        else:
            assert np.max(np.abs(got[::8] - F["dsp_fft_65536_every8"])) <= 1e-12 * 16 * np.abs(F["dsp_fft_65536_every8"]).max()


def test_u8_ingest_fixture():
    """orc_u8_* against what Baseband_Stream::load_buffer / load_buffer_and_rotate90 produced (every byte value on both
    rails): the `char` reading, the 127.4 bias and rotate_90's byte order are the reference's, not a reading of them."""
    raw = np.ascontiguousarray(F["u8_raw"])
    for rot, tag in ((False, "u8"), (True, "u8_rot")):
        re, im = O.u8_to_iq(raw, 127.4, signed_char=True, rotate90=rot)
        assert np.array_equal(re, F[f"{tag}_re"]) and np.array_equal(im, F[f"{tag}_im"])


def test_resample_fixtures():
    x = np.ascontiguousarray(F["rs_in"])
    for name, fn in (("raw_upsample", L.orc_raw_upsample), ("fill_w_zeros", L.orc_upsample_fill_w_zeros), ("upsample", L.orc_upsample)):
        for k in range(4):
            n_in, n_out, ratio = F[f"rs_{name}_{k}_args"]
            o = np.full(int(n_out) + 4, 7.25)
            fn(x.copy(), int(n_in), o, int(n_out), float(ratio))
            assert np.array_equal(o, F[f"rs_{name}_{k}"]), (name, k)
    z = np.ascontiguousarray(F["rs_cplx_in"])
    for k in range(4):
        for name, fn, src in (("decimate_complex", L.orc_decimate_complex, z), ("decimate_block", L.orc_decimate_block, z),
                              ("decimate", L.orc_decimate, z[:13])):
            n, by, r = (int(v) for v in F[f"rs_{name}_{k}_args"])
            a = src.copy()
            assert fn(a, n, by) == r, (name, k)
            assert np.array_equal(a, F[f"rs_{name}_{k}"]), (name, k)


def test_am_modulate_and_bpsk_overload_2_fixtures():
    for k in range(3):
        bits = np.ascontiguousarray(F[f"am_mod_bits_{k}"])
        want = F[f"am_mod_out_{k}"]
        o = np.full(want.size, -5.5)
        L.orc_am_modulate(bits, bits.size, o, want.size - 3)
        assert np.array_equal(o, want), k
    import ctypes as C
    st = O.psk_new(O.BPSK, 0x14EB, 16)
    for call in range(2):
        n, ds, m = (int(v) for v in F[f"bpsk2_args{call}"])
        z = np.ascontiguousarray(F[f"bpsk2_iq{call}"]).copy()
        points = L.orc_decimate_complex(z, n, ds)
        assert np.array_equal(z, F[f"bpsk2_iq_after{call}"])
        out, tr, ti = np.zeros(n), np.zeros(n), np.zeros(n)
        assert L.orc_psk_demodulate(C.byref(st), z[:2 * points].copy(), points, out, tr.ctypes.data, ti.ctypes.data) == m
        assert np.array_equal((out[:m] > 0.5).astype(np.uint8), F[f"bpsk2_bits{call}"])
        assert np.array_equal(tr[:points // 4], F[f"bpsk2_plot_re{call}"]) and np.array_equal(ti[:points // 4], F[f"bpsk2_plot_im{call}"])
    assert np.array_equal(O.dc_block_inplace(F["dsp_in"], 0.995), F["dsp_dc_block_inplace"])


@pytest.mark.gpu
def test_device_psk_u8_and_new_members_against_reference_fixtures():
    """The HIP path against the reference's own outputs directly, second batch: whole BPSK / QPSK blocks (decisions,
    confidence counters, the timing loop's state bit for bit; the carrier loop's three state doubles to 1e-9 -- the
    device's sincos is include/sdrx_math.h, the reference's is libm), the raw-byte conversion of the FIR kernels, AM
    modulate, BPSK::demodulate's second overload, DC_Filter called in place."""
    torch = pytest.importorskip("torch")
    from sdr_amd import _lib, device, host
    device.init(0)
    for tag, obj in (("bpsk", host.BPSK(0xDEADBEEF600DF00D, 64)), ("qpsk", host.QPSK())):
        for blk in range(2):
            got = obj.demodulate(np.ascontiguousarray(F[f"{tag}_iq{blk}"]))
            assert np.array_equal(got, F[f"{tag}_dec{blk}"]), (tag, blk)
            ls, want = obj.loop_state(), F[f"{tag}_state{blk}"]
            assert np.array_equal(ls[:21], want[:21]), (tag, blk, "timing-loop state")
            assert np.allclose(ls[21:], want[21:], rtol=0, atol=1e-9), (tag, blk, "carrier-loop state")
            ncc = 1 if tag == "bpsk" else 4
            assert list(obj.state().cc)[:ncc] == list(F[f"{tag}_cc{blk}"][:ncc])
    # raw bytes through the FIR kernels' load path: a one-tap filter h = {1} returns the converted samples themselves
    raw = torch.tensor(np.ascontiguousarray(F["u8_raw"])[None], device="cuda")
    n = raw.shape[1]
    for rot, tag in ((False, "u8"), (True, "u8_rot")):
        for D in (1, 10):  # tile kernel, systolic kernel (its staging converts too)
            taps = np.array([1.0]) if D == 1 else np.concatenate([[1.0], np.zeros(63)])
            ch = device.FirChain(taps, nchan=1, decim=D)
            ch.set_u8_conversion(127.4, signed_char=True, rotate90=rot)
            o0 = torch.zeros(1, ch.n_out(n), dtype=torch.float64, device="cuda")
            o1 = torch.zeros_like(o0)
            ch.fir(raw, o0, o1)
            assert np.array_equal(o0[0].cpu().numpy(), F[f"{tag}_re"][::D]) and np.array_equal(o1[0].cpu().numpy(), F[f"{tag}_im"][::D])
    for k in range(3):
        bits = np.ascontiguousarray(F[f"am_mod_bits_{k}"])
        want = F[f"am_mod_out_{k}"]
        o = np.full(want.size, -5.5)
        host.AM().modulate(bits, bits.size, o, None, want.size - 3)
        assert np.array_equal(o, want), k
    b = host.BPSK(0x14EB, 16)
    for call in range(2):
        n, ds, m = (int(v) for v in F[f"bpsk2_args{call}"])
        z = np.ascontiguousarray(F[f"bpsk2_iq{call}"]).copy()
        bits, pr, pi = np.zeros(n, dtype=np.uint8), np.zeros(n), np.zeros(n)
        assert b.demodulate_array(z, bits, n, ds, pr, pi) == m
        assert np.array_equal(z, F[f"bpsk2_iq_after{call}"])
        assert np.array_equal(bits[:m], F[f"bpsk2_bits{call}"])
        k = F[f"bpsk2_plot_re{call}"].size
        assert np.array_equal(pr[:k], F[f"bpsk2_plot_re{call}"]) and np.array_equal(pi[:k], F[f"bpsk2_plot_im{call}"])
    x = torch.tensor(np.ascontiguousarray(F["dsp_in"])[None], device="cuda")
    device.iir(device.IIR_DC_BLOCK, [0.995], x)  # in place
    assert np.array_equal(x[0].cpu().numpy(), F["dsp_dc_block_inplace"])


@pytest.mark.gpu
def test_device_against_reference_fixtures():
    """The HIP path against the reference's own outputs directly (no oracle in between): Filter through the drop-in
    host class, AM bits, the deframer's events."""
    torch = pytest.importorskip("torch")
    from sdr_amd import device, host
    device.init(0)
    for name in ("lpf64", "hpf51", "bpf129"):
        ftype, T, Fs, Fx, Fu = F[f"fir_{name}_params"]
        f = host.Filter(int(ftype), int(T), Fs, Fx) if Fu < 0 else host.Filter(int(ftype), int(T), Fs, Fx, Fu)
        x = F[f"fir_{name}_in"]
        a, b = x[:300].copy(), x[300:].copy()
        f.execute(a)
        f.execute(b)
        assert np.array_equal(np.concatenate([a, b]), F[f"fir_{name}_out"])
    for ds in (1, 8, 7):
        out = np.zeros(1003, dtype=np.uint8)
        nb = host.AM().demodulate(np.ascontiguousarray(F["am_re"]), np.ascontiguousarray(F["am_im"]), out, 1003, ds)
        assert np.array_equal(out[:nb], F[f"am_bits_ds{ds}"])
    ev = device.deframe(torch.tensor(np.ascontiguousarray(F["deframe_bits"])[None, :], device="cuda"), True, max_events=64)[0]
    for k in ("pos", "kind", "len", "crc", "payload"):
        assert np.array_equal(np.asarray(ev[k]), F[f"deframe_{k}"]), k
    assert [ev["events"], ev["consumed"], ev["zero_returns"]] == list(F["deframe_summary"])
