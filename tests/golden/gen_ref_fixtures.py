#!/usr/bin/env python3
"""Golden vectors from the REFERENCE ITSELF, run in the build container.

The reference's own unit tests hold no vector for `Filter`, `PacketFramer` streams, the DSP primitives or whole
BPSK / QPSK blocks with their loop state.  This script drives the reference's classes, compiled from
/root/reference into oracle/_ref/libsdr_ref.so (`make -C oracle ref`), over small seeded inputs and stores inputs
and outputs in tests/golden/ref_fixtures.npz.  tests/test_ref_fixtures.py checks the oracle (and, on a GPU, the
HIP path) against them wherever the tests run -- the compiled reference itself cannot travel.

    python tests/golden/gen_ref_fixtures.py         (needs oracle/_ref/libsdr_ref.so)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
from signals import framed_bit_stream, psk_test_signal  # noqa: E402

R = O.ref()
rng = np.random.default_rng(20260101)
out = {}

# ---- Filter: taps, two execute() blocks with the state carried (filt.cpp)
for name, (ftype, T, Fs, Fx, Fu) in {"lpf64": (O.LPF, 64, 2.4e6, 1.0e5, None), "hpf51": (O.HPF, 51, 2.0e6, 3.0e5, None),
                                     "bpf129": (O.BPF, 129, 2.0e6, 1.0e5, 4.0e5)}.items():
    h = R.ref_fir_new(ftype, 0 if Fu is None else 1, T, Fs, Fx, 0.0 if Fu is None else Fu)
    assert R.ref_fir_error(h) == 0
    taps = np.zeros(T)
    R.ref_fir_get_taps(h, taps)
    x = rng.integers(-127, 128, 700).astype(np.float64)
    y = x.copy()
    a, b = y[:300].copy(), y[300:].copy()
    R.ref_fir_execute(h, a, a.size)
    R.ref_fir_execute(h, b, b.size)
    R.ref_fir_free(h)
    out[f"fir_{name}_params"] = np.array([ftype, T, Fs, Fx, -1.0 if Fu is None else Fu])
    out[f"fir_{name}_taps"], out[f"fir_{name}_in"], out[f"fir_{name}_out"] = taps, x, np.concatenate([a, b])

# ---- AM demodulate (AM.h:63-77)
re, im = rng.integers(-127, 128, 1003).astype(np.float64), rng.integers(-127, 128, 1003).astype(np.float64)
re[:4], im[:4] = [80.0, 48.0, 79.99999999999999, 0.0], [0.0, 64.0, 0.0, 80.00000000000001]
out["am_re"], out["am_im"] = re, im
for ds in (1, 8, 7):
    bits = np.zeros(1003, dtype=np.uint8)
    nb = R.ref_am_demodulate(re, im, bits, 1003, ds)
    out[f"am_bits_ds{ds}"] = bits[:nb]

# ---- BPSK / QPSK: two consecutive blocks, decisions + the 24 loop-state doubles + confidence counters
uw = 0xDEADBEEF600DF00D
for kind, tag in ((O.BPSK, "bpsk"), (O.QPSK, "qpsk")):
    h = R.ref_bpsk_new(uw, 64) if kind == O.BPSK else R.ref_qpsk_new(None, 16)
    for blk in range(2):
        iq = psk_test_signal(kind, nsym=256, seed=40 + blk, phase_deg=33.0, snr_db=18.0, timing_offset=3, uw_bpsk=uw)
        n = iq.size // 2
        dec = np.zeros(n)
        nd = (R.ref_bpsk_demodulate if kind == O.BPSK else R.ref_qpsk_demodulate)(h, iq, n, dec, None)
        s, cc = np.zeros(24), np.zeros(4, dtype=np.int32)
        (R.ref_bpsk_state if kind == O.BPSK else R.ref_qpsk_state)(h, s, cc)
        out[f"{tag}_iq{blk}"], out[f"{tag}_dec{blk}"], out[f"{tag}_state{blk}"], out[f"{tag}_cc{blk}"] = iq, dec[:nd], s, cc
    (R.ref_bpsk_free if kind == O.BPSK else R.ref_qpsk_free)(h)

# ---- PacketFramer / RingBuffer / crc8 over a bit stream (rtlsdr_receiver.cpp:2197-2230)
bits = framed_bit_stream(9000, 77, packet_every=600)
ev = O.ref_deframe(bits, True, max_events=64)
out["deframe_bits"] = bits
for k in ("pos", "kind", "len", "crc", "payload"):
    out[f"deframe_{k}"] = np.asarray(ev[k])
out["deframe_summary"] = np.array([ev["events"], ev["consumed"], ev["zero_returns"]], dtype=np.int64)
msg = rng.integers(0, 256, 250).astype(np.uint8)
out["crc_msg"], out["crc_val"] = msg, np.array([R.ref_crc8(msg.copy(), 250), R.ref_crc8(msg[:17].copy(), 17)])

# ---- ones/zeros confidence walk over the reference's confidence_counter
mag = np.where(rng.random(3000) < 0.6, 100.0, 10.0)
cre, cim = np.ascontiguousarray(mag * np.cos(0.3)), np.ascontiguousarray(mag * np.sin(0.3))
out["conf_re"], out["conf_im"] = cre, cim
out["conf_out"] = np.array([R.ref_ones_zeros_confidence(cre, cim, 3000, 37, 0, 37, 5)])

# ---- DigitalSignalProcessing primitives
x = rng.normal(0, 40, 257) + 3.0
out["dsp_in"] = x
y = np.zeros(257)
R.ref_dc_block(x.copy(), y, 257, 0.995)
out["dsp_dc_block"] = y
h = R.ref_lowpass1_new(3000.0, 1.0 / 48000.0)
y = np.zeros(257)
R.ref_lowpass1(h, x, y, 257)
R.ref_lowpass1_free(h)
out["dsp_lowpass1"] = y
y = x.copy()
R.ref_notch(0.9, 0.05, y, 257)  # (first and only call in this process: the reference's delay line is static)
out["dsp_notch"] = y
y = x.copy()
R.ref_integrate(48000.0, y, 257)
out["dsp_integrate"] = y
y = x.copy()
out["dsp_normalize_max"] = np.array([R.ref_normalize(y, 257, 0.0)])
out["dsp_normalize"] = y
h = R.ref_fifth_new()
y = x.copy()
out["dsp_fifth_len"] = np.array([R.ref_fifth_decimate(h, y, 257, 4)])
out["dsp_fifth"] = y
R.ref_fifth_free(h)
h = R.ref_fifth_new()
z = (rng.normal(0, 10, 300) + 1j * rng.normal(0, 10, 300)).astype(np.complex128)
out["dsp_fifth_block_in"] = z.copy()
out["dsp_fifth_block_len"] = np.array([R.ref_fifth_decimate_block(h, z.view(np.float64), 300, 257, 4)])
out["dsp_fifth_block"] = z
R.ref_fifth_free(h)
z = (rng.normal(0, 1, 256) + 1j * rng.normal(0, 1, 256)).astype(np.complex128)
out["dsp_fft_in"] = z.copy()
R.ref_fft(z.view(np.float64), 256)
out["dsp_fft"] = z

# ==== round 2 additions (their own generator: the arrays above keep their values) ====
rng2 = np.random.default_rng(20260202)

# ---- u8 I/Q ingest: Baseband_Stream::load_buffer / load_buffer_and_rotate90 (Baseband_Stream.cpp:82-148)
allb = np.arange(256, dtype=np.uint8)
raw = np.concatenate([np.stack([allb, allb[::-1]], axis=1), rng2.integers(0, 256, (768, 2)).astype(np.uint8)])
out["u8_raw"] = raw
for rot, tag in ((0, "u8"), (1, "u8_rot")):
    re, im = np.zeros(raw.shape[0]), np.zeros(raw.shape[0])
    assert R.ref_baseband_load(np.ascontiguousarray(raw).reshape(-1), raw.size, rot, re, im) == raw.shape[0]
    out[f"{tag}_re"], out[f"{tag}_im"] = re, im

# ---- Resample.{h,cpp}: all six helpers (inputs, outputs incl. untouched tails, return values)
x = rng2.uniform(-3, 3, 80)
out["rs_in"] = x
for name, fn in (("raw_upsample", R.ref_raw_upsample), ("fill_w_zeros", R.ref_upsample_fill_w_zeros), ("upsample", R.ref_upsample)):
    for k, (n_in, n_out, ratio) in enumerate(((10, 80, 8.0), (10, 75, 7.5), (5, 23, 4.6), (3, 2, 0.9))):
        o = np.full(n_out + 4, 7.25)
        fn(x.copy(), n_in, o, n_out, ratio)
        out[f"rs_{name}_{k}"] = o
        out[f"rs_{name}_{k}_args"] = np.array([n_in, n_out, ratio])
z = rng2.uniform(-1, 1, 2 * 13)
out["rs_cplx_in"] = z
for k, (n, by) in enumerate(((13, 3), (12, 4), (13, 1), (5, 9))):
    a = z.copy()
    r = R.ref_decimate_complex(a, n, by)
    out[f"rs_decimate_complex_{k}"], out[f"rs_decimate_complex_{k}_args"] = a, np.array([n, by, r])
    a = z.copy()
    r = R.ref_decimate_block(a, n, by)
    out[f"rs_decimate_block_{k}"], out[f"rs_decimate_block_{k}_args"] = a, np.array([n, by, r])
    a = z[:13].copy()
    r = R.ref_decimate(a, n, by)
    out[f"rs_decimate_{k}"], out[f"rs_decimate_{k}_args"] = a, np.array([n, by, r])

# ---- AM modulate (AM.h:41-54) incl. a length that is not a multiple
for k, (nb, len_mod) in enumerate(((100, 800), (37, 300), (10, 7))):
    bits = rng2.integers(0, 2, nb).astype(np.uint8)
    o = np.full(len_mod + 3, -5.5)
    R.ref_am_modulate(bits.copy(), nb, o, len_mod)
    out[f"am_mod_bits_{k}"], out[f"am_mod_out_{k}"] = bits, o

# ---- BPSK::demodulate overload 2 (BPSK.cpp:178-211), two calls on one object
h = R.ref_bpsk_new(0x14EB, 16)
for call, ds in enumerate((2, 3)):
    base = np.asarray(psk_test_signal(O.BPSK, 300, seed=70 + call, phase_deg=20.0, snr_db=25.0, uw_bpsk=0x14EB, uw_len=16))
    zz = np.repeat(base.reshape(-1, 2), ds, axis=0).reshape(-1).copy()
    n = zz.size // 2
    out[f"bpsk2_iq{call}"] = zz.copy()
    bits, pr, pi = np.zeros(n, dtype=np.uint8), np.zeros(n), np.zeros(n)
    m = R.ref_bpsk_demodulate2(h, zz, n, ds, bits, pr, pi)
    points = -(-n // ds)
    out[f"bpsk2_args{call}"] = np.array([n, ds, m])
    out[f"bpsk2_iq_after{call}"], out[f"bpsk2_bits{call}"] = zz, bits[:m]
    out[f"bpsk2_plot_re{call}"], out[f"bpsk2_plot_im{call}"] = pr[:points // 4], pi[:points // 4]
R.ref_bpsk_free(h)

# ---- DC_Filter::update called in place (input == output): input[i-1] is read after it was overwritten
y = out["dsp_in"].copy()
R.ref_dc_block(y, y, 257, 0.995)
out["dsp_dc_block_inplace"] = y

# ==== round 3 addition: fft above 4096 points (FFT.cpp:9-29 recurses for any power of two).  Inputs are regenerated from the
# seed by the tests; 8192 points are kept whole, of 65536 every 8th output and the sha256 of all of them.
import hashlib  # noqa: E402

rng3 = np.random.default_rng(20260303)
for n in (8192, 65536):
    z = (rng3.normal(0, 1, n) + 1j * rng3.normal(0, 1, n)).astype(np.complex128)
    R.ref_fft(z.view(np.float64), n)
    if n == 8192:
        out["dsp_fft_8192"] = z
    else:
        out["dsp_fft_65536_every8"] = z[::8].copy()
        out["dsp_fft_65536_sha256"] = np.frombuffer(hashlib.sha256(z.tobytes()).digest(), dtype=np.uint8).copy()

np.savez_compressed(os.path.join(HERE, "ref_fixtures.npz"), **out)
print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "ref_fixtures.npz")), "bytes")
