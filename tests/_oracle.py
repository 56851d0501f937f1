"""ctypes bindings for the CPU checker (oracle/liboracle.so) and, when it was built in
this container, the compiled reference (oracle/_ref/libsdr_ref.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libsdr_ref.so")

MATH_LIBM, MATH_SHARED = 0, 1
LPF, HPF, BPF = 0, 1, 2
BPSK, QPSK = 0, 1

_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i16 = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u64 = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_sz = C.c_size_t


class Interp(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("d1", "d2", "d3", "d4", "d5")]


class PskState(C.Structure):
    """Mirror of orc_psk (oracle/sdr_oracle.h)."""

    _fields_ = (
        [(n, C.c_double) for n in ("nco_delay", "reg_mu", "reg_Q", "reg_I", "mu_delay")]
        + [("interp_q", Interp), ("interp_i", Interp)]
        + [(n, C.c_double) for n in ("g1", "g2", "g3", "g4", "lf_x", "lf_y", "dds_delay", "plf_x", "plf_y")]
        + [("cc", C.c_int32 * 4), ("kind", C.c_int32), ("uw_size", C.c_int32), ("math_mode", C.c_int32),
           ("pad_", C.c_int32), ("uw", (C.c_double * 64) * 4)]
    )

    N_LOOP_DOUBLES = 24

    def loop_state(self):
        """The 24 loop doubles, in declaration order."""
        return np.frombuffer(bytes(self)[: 8 * self.N_LOOP_DOUBLES], dtype=np.float64).copy()


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "sdr_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB)
    L.orc_fir_design.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _f64]
    L.orc_fir_design.restype = C.c_int
    L.orc_fir_execute.argtypes = [_f64, C.c_int, _f64, _f64, _sz]
    L.orc_fir_execute.restype = None
    L.orc_fir_execute_fast.argtypes = [_f64, C.c_int, _f64, _f64, _sz]
    L.orc_fir_execute_fast.restype = None
    L.orc_u8_rotate_90.argtypes = [_u8, _sz]
    L.orc_u8_rotate_90.restype = None
    L.orc_u8_to_iq.argtypes = [_u8, _sz, C.c_double, C.c_int, _f64, _f64]
    L.orc_u8_to_iq.restype = None
    _i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    _u64 = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    L.orc_crc8.argtypes = [_u8, _sz]
    L.orc_crc8.restype = C.c_uint8
    L.orc_deframe.argtypes = [_u8, _sz, C.c_int, _sz, _i64, _i32, _u8, _u8, _u8, _u64]
    L.orc_deframe.restype = None
    L.orc_frame.argtypes = [_u8, _sz, C.c_int, _u8]
    L.orc_frame.restype = C.c_int
    L.orc_fir_execute_fma.argtypes = [_f64, C.c_int, _f64, _f64, _sz]
    L.orc_fir_execute_fma.restype = None
    L.orc_dc_block.argtypes = [_f64, _f64, _sz, C.c_double]
    L.orc_dc_block.restype = None
    L.orc_lowpass1_coef.argtypes = [C.c_double, C.c_double]
    L.orc_lowpass1_coef.restype = C.c_double
    L.orc_lowpass1.argtypes = [_f64, _f64, _sz, C.c_double, _f64]
    L.orc_lowpass1.restype = None
    L.orc_notch_design.argtypes = [C.c_double, C.c_double, _f64]
    L.orc_notch_design.restype = None
    L.orc_notch.argtypes = [_f64, _sz, _f64, _f64]
    L.orc_notch.restype = None
    L.orc_integrate.argtypes = [_f64, _sz, C.c_double]
    L.orc_integrate.restype = None
    L.orc_normalize.argtypes = [_f64, _sz, C.c_double]
    L.orc_normalize.restype = C.c_double
    L.orc_fifth_order_decimate.argtypes = [_f64, _sz, _sz, _f64]
    L.orc_fifth_order_decimate.restype = _sz
    L.orc_fifth_order_decimate_block.argtypes = [_f64, _sz, _sz, _f64]
    L.orc_fifth_order_decimate_block.restype = _sz
    L.orc_fft.argtypes = [_f64, _sz, C.c_int]
    L.orc_fft.restype = None
    L.orc_am_modulate.argtypes = [_u8, _sz, _f64, _sz]
    L.orc_am_modulate.restype = None
    L.orc_am_demodulate.argtypes = [_f64, _f64, _u8, _sz, _sz]
    L.orc_am_demodulate.restype = _sz
    L.orc_ones_zeros.argtypes = [_f64, _f64, _sz]
    L.orc_ones_zeros.restype = C.c_int32
    L.orc_ones_zeros_confidence.argtypes = [_f64, _f64, _sz, C.c_int32, C.c_int32]
    L.orc_ones_zeros_confidence.restype = C.c_int32
    L.orc_fm_modulate.argtypes = [C.c_double, C.c_double, C.c_double, _u8, _sz, _f64, _sz]
    L.orc_fm_modulate.restype = None
    L.orc_fm_demodulate.argtypes = [C.c_double, C.c_double, C.c_double, _f64, _u8, _sz, _sz]
    L.orc_fm_demodulate.restype = _sz
    L.orc_fm_waves.argtypes = [C.c_double, C.c_double, C.c_double, _sz, _f64, _f64]
    L.orc_fm_waves.restype = None
    L.orc_fm_discriminator.argtypes = [_f64, _f64, _sz, _f64, _f64, C.c_int]
    L.orc_fm_discriminator.restype = None
    L.orc_fm_discriminator_i16.argtypes = [_i16, _sz, _i16, _i16, C.c_int]
    L.orc_fm_discriminator_i16.restype = None
    L.orc_rtlfm_chain.argtypes = [_u8, _sz, C.c_int, C.c_int, _i32, _i16, C.c_int]
    L.orc_rtlfm_chain.restype = _sz
    L.orc_rtlfm_full.argtypes = [_u8, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32, _i32, _i16, _sz, C.c_int]
    L.orc_rtlfm_full.restype = _sz
    L.orc_rtlfm_low_pass_real.argtypes = [_i16, C.c_int, C.c_int, C.c_int, _i32]
    L.orc_rtlfm_low_pass_real.restype = C.c_int
    L.orc_sincos.argtypes = [_f64, _sz, C.c_int, _f64, _f64]
    L.orc_sincos.restype = None
    L.orc_decimate.argtypes = [_f64, _sz, _sz]
    L.orc_decimate.restype = _sz
    L.orc_upsample_fill_w_zeros.argtypes = [_f64, _sz, _f64, _sz, C.c_double]
    L.orc_upsample_fill_w_zeros.restype = _sz
    L.orc_raw_upsample.argtypes = [_f64, _sz, _f64, _sz, C.c_double]
    L.orc_raw_upsample.restype = _sz
    L.orc_upsample.argtypes = [_f64, _sz, _f64, _sz, C.c_double]
    L.orc_upsample.restype = _sz
    L.orc_decimate_block.argtypes = [_f64, _sz, _sz]
    L.orc_decimate_block.restype = _sz
    L.orc_decimate_complex.argtypes = [_f64, _sz, _sz]
    L.orc_decimate_complex.restype = _sz
    L.orc_fm_convolve.argtypes = [_f64, _sz, _f64, _sz, _f64, _sz]
    L.orc_fm_convolve.restype = None
    L.orc_chain_fir_am.argtypes = [_f64, C.c_int, _f64, _f64, _f64, _f64, _sz, _sz, _u8]
    L.orc_chain_fir_am.restype = _sz
    L.orc_chain_fir_decim_disc.argtypes = [_f64, C.c_int, _f64, _f64, _f64, _f64, _f64, _sz, _sz, _f64, C.c_int]
    L.orc_chain_fir_decim_disc.restype = _sz
    P = C.POINTER(PskState)
    L.orc_bpsk_init.argtypes = [P, C.c_uint64, _sz, C.c_int]
    L.orc_bpsk_init.restype = None
    L.orc_qpsk_init.argtypes = [P, C.c_void_p, _sz, C.c_int]
    L.orc_qpsk_init.restype = None
    L.orc_psk_encode_lut.argtypes = [C.c_int, _u8, _sz, _f64, C.c_void_p]
    L.orc_psk_encode_lut.restype = None
    L.orc_psk_pulse_shape.argtypes = [_f64, _sz, _f64, _sz]
    L.orc_psk_pulse_shape.restype = None
    L.orc_psk_decode_pulse_shape.argtypes = [_f64, _sz, _f64, _f64]
    L.orc_psk_decode_pulse_shape.restype = _sz
    L.orc_psk_timing.argtypes = [P, _f64, _f64, _sz]
    L.orc_psk_timing.restype = None
    L.orc_psk_phase_correction.argtypes = [P, _f64, _f64, _sz]
    L.orc_psk_phase_correction.restype = _sz
    L.orc_psk_ambiguity.argtypes = [P, _f64, _sz, C.c_void_p, _sz]
    L.orc_psk_ambiguity.restype = _sz
    L.orc_psk_demodulate.argtypes = [P, _f64, _sz, _f64, C.c_void_p, C.c_void_p]
    L.orc_psk_demodulate.restype = _sz
    L.orc_bpsk_modulate.argtypes = [_u8, _sz, _f64, _sz]
    L.orc_bpsk_modulate.restype = None
    L.orc_qpsk_modulate.argtypes = [_u8, _sz, _f64, _f64, _sz]
    L.orc_qpsk_modulate.restype = None
    _lib = L
    return L


# ------------------------------------------------------------------ convenience wrappers

def fir_design(ftype, ntaps, Fs, Fx, Fu=None, max_taps=1000):
    taps = np.zeros(max(int(ntaps), 1), dtype=np.float64)
    err = lib().orc_fir_design(ftype, 0 if Fu is None else 1, int(ntaps), float(Fs), float(Fx),
                               0.0 if Fu is None else float(Fu), int(max_taps), taps)
    return err, taps


class Fir:
    """Stateful stand-in for the reference Filter object (literal shift register)."""

    def __init__(self, ftype, ntaps, Fs, Fx, Fu=None, max_taps=1000):
        self.err, self.taps = fir_design(ftype, ntaps, Fs, Fx, Fu, max_taps)
        self.T = int(ntaps)
        self.sr = np.zeros(max(self.T, 1), dtype=np.float64)

    def execute_fma(self, buf):
        """fused-chain variant (definition of the fast_fma modes); history carried like execute(fast=True)"""
        buf = np.ascontiguousarray(buf, dtype=np.float64).copy()
        hist = self.sr[: self.T - 1].copy() if self.T > 1 else np.zeros(1)
        lib().orc_fir_execute_fma(self.taps, self.T, hist, buf, buf.size)
        self.sr[: self.T - 1] = hist[: self.T - 1]
        return buf

    def execute(self, buf, fast=False):
        buf = np.ascontiguousarray(buf, dtype=np.float64).copy()
        if self.err != 0:
            buf[:] = 0.0  # do_sample returns 0 when the flag is set (filt.cpp:254)
            return buf
        if fast:
            hist = self.sr[: self.T - 1].copy() if self.T > 1 else np.zeros(1)
            lib().orc_fir_execute_fast(self.taps, self.T, hist, buf, buf.size)
            self.sr[: self.T - 1] = hist[: self.T - 1]
        else:
            lib().orc_fir_execute(self.taps, self.T, self.sr, buf, buf.size)
        return buf


def am_demodulate(re, im, ds):
    re = np.ascontiguousarray(re, dtype=np.float64)
    im = np.ascontiguousarray(im, dtype=np.float64)
    out = np.zeros((re.size + ds - 1) // ds + 1, dtype=np.uint8)
    n = lib().orc_am_demodulate(re, im, out, re.size, ds)
    return out[:n]


def am_modulate(bits, len_mod, out=None):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    if out is None:
        out = np.zeros(len_mod, dtype=np.float64)
    lib().orc_am_modulate(bits, bits.size, out, len_mod)
    return out


def fm_modulate(tx_rate, mark, space, bits, spb):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(bits.size * spb, dtype=np.float64)
    lib().orc_fm_modulate(tx_rate, mark, space, bits, bits.size, out, spb)
    return out


def fm_demodulate(rx_rate, mark, space, re, ds):
    re = np.ascontiguousarray(re, dtype=np.float64)
    out = np.zeros(re.size // ds + 1, dtype=np.uint8)
    n = lib().orc_fm_demodulate(rx_rate, mark, space, re, out, re.size, ds)
    return out[:n]


def u8_to_iq(buf, bias=127.0, signed_char=False, rotate90=False):
    """raw radio bytes (n, 2) -> (re, im) doubles, optionally after rotate_90 (n a multiple of 4)"""
    b = np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1).copy()
    if rotate90:
        lib().orc_u8_rotate_90(b, b.size)
    n = b.size // 2
    re, im = np.zeros(n), np.zeros(n)
    lib().orc_u8_to_iq(b, n, float(bias), int(bool(signed_char)), re, im)
    return re, im


def rtlfm_chain(raw, downsample, st, rotate90=False, math_mode=MATH_SHARED):
    """rtl_fm's int16 chain on raw bytes (n, 2); st = int32[5] carried state, updated in place"""
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1)
    n = raw.size // 2
    pcm = np.zeros(n // max(downsample, 1) + 2, dtype=np.int16)
    m = lib().orc_rtlfm_chain(raw, n, int(downsample), int(bool(rotate90)), st, pcm, math_mode)
    return pcm[:m]


def rtlfm_full(raw, downsample, st, post, pcm, rotate90=False, squelch_level=0, post_downsample=1, deemph_a=0,
               dc_block=False, math_mode=MATH_SHARED):
    """rtl_fm's full_demod on raw bytes (n, 2); st = int32[5], post = int32[4] carried, pcm = the int16 result buffer
    (kept between calls like rtl_fm's result[]), all updated in place; returns result_len"""
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1)
    return lib().orc_rtlfm_full(raw, raw.size // 2, int(downsample), int(bool(rotate90)), int(squelch_level),
                                int(post_downsample), int(deemph_a), int(bool(dc_block)), st, post, pcm, pcm.size, math_mode)


def fm_discriminator(re, im, pre=None, math_mode=MATH_SHARED):
    re = np.ascontiguousarray(re, dtype=np.float64)
    im = np.ascontiguousarray(im, dtype=np.float64)
    pre = np.zeros(2) if pre is None else pre
    out = np.zeros(max(re.size, 1), dtype=np.float64)
    lib().orc_fm_discriminator(re, im, re.size, pre, out, math_mode)
    return out[: re.size]


def psk_new(kind, uw=None, uw_len=None, math_mode=MATH_LIBM):
    st = PskState()
    if kind == BPSK:
        lib().orc_bpsk_init(C.byref(st), 0x4E if uw is None else uw, 8 if uw_len is None else uw_len, math_mode)
    else:
        ptr = None
        if uw is not None:
            uw = np.ascontiguousarray(uw, dtype=np.float64)
            ptr = uw.ctypes.data
        lib().orc_qpsk_init(C.byref(st), ptr, 16 if uw_len is None else uw_len, math_mode)
    return st


def psk_demodulate(st, iq):
    iq = np.ascontiguousarray(iq, dtype=np.float64)
    n = iq.size // 2
    out = np.zeros(max(n, 1), dtype=np.float64)
    m = lib().orc_psk_demodulate(C.byref(st), iq, n, out, None, None)
    return out[:m]


# ------------------------------------------------------------------ compiled reference

_ref = None


def ref_available():
    return os.path.exists(REF_LIB)


def _deframe_with(fn, bits, crc_enable=True, max_events=4096):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    pos = np.zeros(max_events, dtype=np.int64)
    kind = np.zeros(max_events, dtype=np.int32)
    ln = np.zeros(max_events, dtype=np.uint8)
    crc = np.zeros(max_events, dtype=np.uint8)
    payload = np.zeros(max_events * 250, dtype=np.uint8)
    summary = np.zeros(4, dtype=np.uint64)
    fn(bits, bits.size, int(bool(crc_enable)), max_events, pos, kind, ln, crc, payload, summary)
    k = int(min(summary[0], max_events))
    return {"pos": pos[:k], "kind": kind[:k], "len": ln[:k], "crc": crc[:k], "payload": payload.reshape(-1, 250)[:k],
            "events": int(summary[0]), "consumed": int(summary[1]), "zero_returns": int(summary[2])}


def deframe(bits, crc_enable=True, max_events=4096):
    """receiver loop + PacketFramer::deserialize over a bit stream (oracle restatement)"""
    return _deframe_with(lib().orc_deframe, bits, crc_enable, max_events)


def ref_deframe(bits, crc_enable=True, max_events=4096):
    """the same through the reference's own PacketFramer / RingBuffer / crc8 (oracle/_ref)"""
    return _deframe_with(ref().ref_deframe, bits, crc_enable, max_events)


def frame(data, crc_enable=True):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(2032, dtype=np.uint8)
    r = lib().orc_frame(data, data.size, int(bool(crc_enable)), out)
    return out if r > 0 else None


def crc8(data):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().orc_crc8(data, data.size))


# ---- DSP primitives (DigitalSignalProcessing/*.cpp), single channel, numpy in / numpy out

def dc_block(x, R=0.995):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().orc_dc_block(x, y, x.size, R)
    return y


def dc_block_inplace(x, R=0.995):
    """DC_Filter::update(buf, buf, n): the aliased call (input[i-1] is read after output[i-1] overwrote it)"""
    y = np.array(x, dtype=np.float64)
    lib().orc_dc_block(y, y, y.size, R)
    return y


def lowpass1(x, epow, state):
    """state: float64[1], updated"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().orc_lowpass1(x, y, x.size, epow, state)
    return y


def notch_design(bw, fc):
    c = np.zeros(5)
    lib().orc_notch_design(bw, fc, c)
    return c


def notch(x, coef, state):
    """state: float64[4] = x_1,x_2,y_1,y_2, updated"""
    y = np.array(x, dtype=np.float64)
    lib().orc_notch(y, y.size, np.ascontiguousarray(coef, dtype=np.float64), state)
    return y


def integrate(x, tstep):
    y = np.array(x, dtype=np.float64)
    lib().orc_integrate(y, y.size, tstep)
    return y


def normalize(x, prev_max=0.0):
    y = np.array(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        m = lib().orc_normalize(y, y.size, prev_max)
    return y, m


def fifth_order_decimate(x, by, state):
    """state: float64[6] = r_a..r_f, updated; returns (array after the in-place passes, returned length)"""
    y = np.array(x, dtype=np.float64)
    r = lib().orc_fifth_order_decimate(y, y.size, by, state)
    return y, r


def fifth_order_decimate_block(z, n_points, by, state):
    """z: complex128[capacity]; state: float64[12], updated; returns (array afterwards, n_points afterwards)"""
    y = np.array(z, dtype=np.complex128)
    r = lib().orc_fifth_order_decimate_block(y.view(np.float64), n_points, by, state)
    return y, r


def fft(iq, math_mode=MATH_SHARED):
    """iq: complex128[n], n a power of two"""
    y = np.array(iq, dtype=np.complex128)
    lib().orc_fft(y.view(np.float64), y.size, math_mode)
    return y


def ref():
    global _ref
    if _ref is not None:
        return _ref
    R = C.CDLL(REF_LIB)
    _i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    _u64 = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    R.ref_crc8.argtypes = [_u8, _sz]
    R.ref_crc8.restype = C.c_uint8
    R.ref_frame.argtypes = [_u8, _sz, C.c_int, _u8]
    R.ref_frame.restype = C.c_int
    R.ref_deframe.argtypes = [_u8, _sz, C.c_int, _sz, _i64, _i32, _u8, _u8, _u8, _u64]
    R.ref_deframe.restype = None
    R.ref_ones_zeros_confidence.argtypes = [_f64, _f64, _sz, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    R.ref_ones_zeros_confidence.restype = C.c_int32
    R.ref_dc_block.argtypes = [_f64, _f64, _sz, C.c_double]
    R.ref_lowpass1_new.argtypes = [C.c_double, C.c_double]
    R.ref_lowpass1_new.restype = C.c_void_p
    R.ref_lowpass1_free.argtypes = [C.c_void_p]
    R.ref_lowpass1.argtypes = [C.c_void_p, _f64, _f64, _sz]
    R.ref_notch.argtypes = [C.c_double, C.c_double, _f64, _sz]
    R.ref_integrate.argtypes = [C.c_double, _f64, _sz]
    R.ref_normalize.argtypes = [_f64, _sz, C.c_double]
    R.ref_normalize.restype = C.c_double
    R.ref_fifth_new.restype = C.c_void_p
    R.ref_fifth_free.argtypes = [C.c_void_p]
    R.ref_fifth_decimate.argtypes = [C.c_void_p, _f64, _sz, _sz]
    R.ref_fifth_decimate.restype = _sz
    R.ref_fifth_decimate_block.argtypes = [C.c_void_p, _f64, _sz, _sz, _sz]
    R.ref_fifth_decimate_block.restype = _sz
    R.ref_fft.argtypes = [_f64, _sz]
    R.ref_fir_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    R.ref_fir_new.restype = C.c_void_p
    R.ref_fir_error.argtypes = [C.c_void_p]
    R.ref_fir_error.restype = C.c_int
    R.ref_fir_get_taps.argtypes = [C.c_void_p, _f64]
    R.ref_fir_execute.argtypes = [C.c_void_p, _f64, _sz]
    R.ref_fir_free.argtypes = [C.c_void_p]
    R.ref_am_modulate.argtypes = [_u8, _sz, _f64, _sz]
    R.ref_am_demodulate.argtypes = [_f64, _f64, _u8, _sz, _sz]
    R.ref_am_demodulate.restype = _sz
    R.ref_decimate.argtypes = [_f64, _sz, _sz]
    R.ref_decimate.restype = _sz
    R.ref_upsample_fill_w_zeros.argtypes = [_f64, _sz, _f64, _sz, C.c_double]
    R.ref_upsample_fill_w_zeros.restype = _sz
    R.ref_raw_upsample.argtypes = [_f64, _sz, _f64, _sz, C.c_double]
    R.ref_raw_upsample.restype = _sz
    R.ref_upsample.argtypes = [_f64, _sz, _f64, _sz, C.c_double]
    R.ref_upsample.restype = _sz
    R.ref_decimate_block.argtypes = [_f64, _sz, _sz]
    R.ref_decimate_block.restype = _sz
    R.ref_decimate_complex.argtypes = [_f64, _sz, _sz]
    R.ref_decimate_complex.restype = _sz
    R.ref_baseband_load.argtypes = [_u8, _sz, C.c_int, _f64, _f64]
    R.ref_baseband_load.restype = _sz
    R.ref_bpsk_demodulate2.argtypes = [C.c_void_p, _f64, _sz, _sz, _u8, _f64, _f64]
    R.ref_bpsk_demodulate2.restype = _sz
    R.ref_bpsk_new.argtypes = [C.c_uint64, _sz]
    R.ref_bpsk_new.restype = C.c_void_p
    R.ref_bpsk_free.argtypes = [C.c_void_p]
    R.ref_bpsk_modulate.argtypes = [C.c_void_p, _u8, _sz, _f64, _sz]
    R.ref_bpsk_demodulate.argtypes = [C.c_void_p, _f64, _sz, _f64, C.c_void_p]
    R.ref_bpsk_demodulate.restype = _sz
    R.ref_bpsk_decode_pulse_shape.argtypes = [C.c_void_p, _f64, _sz, _f64, _f64]
    R.ref_bpsk_decode_pulse_shape.restype = _sz
    R.ref_bpsk_timing.argtypes = [C.c_void_p, _f64, _f64, _sz]
    R.ref_bpsk_phase.argtypes = [C.c_void_p, _f64, _f64, _sz]
    R.ref_bpsk_phase.restype = _sz
    R.ref_bpsk_ambiguity.argtypes = [C.c_void_p, _f64, _sz]
    R.ref_bpsk_state.argtypes = [C.c_void_p, _f64, np.ctypeslib.ndpointer(dtype=np.int32)]
    R.ref_qpsk_new.argtypes = [C.c_void_p, _sz]
    R.ref_qpsk_new.restype = C.c_void_p
    R.ref_qpsk_free.argtypes = [C.c_void_p]
    R.ref_qpsk_modulate.argtypes = [C.c_void_p, _u8, _sz, _f64, _f64, _sz]
    R.ref_qpsk_demodulate.argtypes = [C.c_void_p, _f64, _sz, _f64, C.c_void_p]
    R.ref_qpsk_demodulate.restype = _sz
    R.ref_qpsk_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _f64, _f64, _sz]
    R.ref_qpsk_stage.restype = _sz
    R.ref_qpsk_state.argtypes = [C.c_void_p, _f64, np.ctypeslib.ndpointer(dtype=np.int32)]
    _ref = R
    return R
