"""Deterministic synthetic inputs for the parity tests, numpy only.

PSK inputs are produced with the oracle's modulators (tests only), the rest with plain
numpy so that the same arrays can be regenerated on the GPU box from the seed.
"""
import numpy as np

import _oracle as O

QPSK_DEFAULT_UW_TAIL = [3, 0, 2, 1, 0, 3, 2, 1, 2, 0, 1, 2, 3, 2, 2, 3]  # QPSK.cpp:27-28 (last 16 of 64)


def radio_iq(n, seed, lo=-127, hi=127):
    """u8-like radio samples: integer values in [lo, hi] as fp64 (rtlsdr_receiver.cpp:1587-1588)."""
    rng = np.random.default_rng(seed)
    return (rng.integers(lo, hi + 1, n).astype(np.float64), rng.integers(lo, hi + 1, n).astype(np.float64))


def ook_iq(n, seed, ds=8, amp=100.0, noise=10.0):
    """cfg1: on-off keyed carrier, amplitude `amp` on bits from `seed`, `ds` samples per bit, +-noise."""
    rng = np.random.default_rng(seed)
    nb = (n + ds - 1) // ds
    bits = rng.integers(0, 2, nb).astype(np.uint8)
    env = np.repeat(bits.astype(np.float64), ds)[:n] * amp
    ph = rng.uniform(0, 2 * np.pi)
    re = env * np.cos(ph) + rng.uniform(-noise, noise, n)
    im = env * np.sin(ph) + rng.uniform(-noise, noise, n)
    return re, im, bits


def fm_iq(n, seed, fs=2.4e6, dev=75e3, tone=1000.0, amp=100.0, noise=1.0):
    """cfg2: FM-modulated tone (deviation +-dev) at sample rate fs, small additive noise."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    phase = 2 * np.pi * dev / (2 * np.pi * tone) * np.sin(2 * np.pi * tone * t) * 1.0
    re = amp * np.cos(phase) + rng.normal(0, noise, n)
    im = amp * np.sin(phase) + rng.normal(0, noise, n)
    return re, im


def psk_symbols(kind, nsym, seed, uw_bpsk=0x4E, uw_len=None, uw_at=96):
    """Random payload with the unique word placed so that the reference's reversed compare
    (BPSK.cpp:456, QPSK.cpp:428) finds it at symbol index `uw_at`."""
    rng = np.random.default_rng(seed)
    if kind == O.BPSK:
        uw_len = 8 if uw_len is None else uw_len
        sym = rng.integers(0, 2, nsym).astype(np.uint8)
        uw = [(uw_bpsk >> (uw_len - 1 - j)) & 1 for j in range(uw_len)]
    else:
        uw_len = 16 if uw_len is None else uw_len
        sym = rng.integers(0, 4, nsym).astype(np.uint8)
        uw = [QPSK_DEFAULT_UW_TAIL[15 - j] for j in range(uw_len)]
    if uw_at + uw_len <= nsym:
        sym[uw_at:uw_at + uw_len] = uw
    return sym


def psk_test_signal(kind, nsym, seed, phase_deg=0.0, snr_db=None, timing_offset=0, uw_bpsk=0x4E, uw_len=None,
                    freq_offset=0.0):
    """Interleaved complex fp64 [re0, im0, re1, ...] of nsym*8 samples."""
    L = O.lib()
    if kind == O.BPSK and uw_len is None:
        uw_len = 64 if uw_bpsk > 0xFFFF else (16 if uw_bpsk > 0xFF else 8)
    sym = psk_symbols(kind, nsym, seed, uw_bpsk, uw_len)
    n = nsym * 8
    if kind == O.BPSK:
        re = np.zeros(n)
        L.orc_bpsk_modulate(sym, nsym, re, n)
        z = re + 1j * re  # both rails carry the same data (TEST_Modulator_Demodulator.cpp:41-46)
    else:
        re, im = np.zeros(n), np.zeros(n)
        L.orc_qpsk_modulate(sym, nsym, re, im, n)
        z = re + 1j * im
    if timing_offset:
        z = np.concatenate([np.zeros(timing_offset, dtype=complex), z])[:n]
    k = np.arange(n)
    z = z * np.exp(1j * (np.deg2rad(phase_deg) + 2 * np.pi * freq_offset * k))
    if snr_db is not None:
        rng = np.random.default_rng(seed + 7919)
        p = np.mean(np.abs(z) ** 2)
        sigma = np.sqrt(p / (10 ** (snr_db / 10)) / 2)
        z = z + rng.normal(0, sigma, n) + 1j * rng.normal(0, sigma, n)
    iq = np.empty(2 * n)
    iq[0::2], iq[1::2] = z.real, z.imag
    return iq


def framed_bit_stream(n_bits, seed, packet_every=3000, p_bad_crc=0.15, p_bad_len=0.1, p_false_uw=0.002, crc_enable=True):
    """A bit stream (uint8 0/1) with framed packets (PacketFramer layout: 16-bit unique word, length, crc8,
    250 data bytes) scattered in noise; some with a corrupted CRC or a length > 250, plus bare unique words."""
    import _oracle as O
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, n_bits).astype(np.uint8)
    uw = np.array([1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 0], dtype=np.uint8)
    p = int(rng.integers(0, packet_every))
    while p + 2032 <= n_bits:
        ln = int(rng.integers(0, 251))
        data = rng.integers(0, 256, ln).astype(np.uint8)
        pkt = O.frame(data, crc_enable).copy()
        pkt[32 + 8 * ln:] = rng.integers(0, 2, 2032 - 32 - 8 * ln)  # the reference reads all 250 bytes: fill the rest
        u = rng.random()
        if u < p_bad_crc:
            pkt[24 + int(rng.integers(0, 8))] ^= 1
        elif u < p_bad_crc + p_bad_len:
            pkt[16:24] = [1, 1, 1, 1, 1, 1, int(rng.integers(0, 2)), 1]  # 253 or 255
        bits[p:p + 2032] = pkt
        p += 2032 + int(rng.integers(0, packet_every))
    for q in np.flatnonzero(rng.random(n_bits // 64) < p_false_uw * 64):
        q = int(q) * 64 + int(rng.integers(0, 48))
        if q + 16 <= n_bits:
            bits[q:q + 16] = uw
    return bits

