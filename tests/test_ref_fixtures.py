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
