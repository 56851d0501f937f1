"""Python mirror of the reference's class interface for the hot path, over libsdrx's host layer.

Same names, argument meaning and error behaviour as RadioNode's C++ classes
(FIR-filter-class/filt.h:133-162, streams/AM.h, FM.h, BPSK.h, QPSK.h) so that the parity tests
read like the reference's own unit tests.  Every method runs on the GPU through the C ABI
(host pointers in, pinned staging + async copies inside); nothing here computes on the CPU.
numpy arrays stand in for the reference's raw pointers and are modified in place where the
reference modifies in place.
"""
import ctypes as C

import numpy as np

from . import _lib

LPF, HPF, BPF = _lib.LPF, _lib.HPF, _lib.BPF


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "expects contiguous float64 (RADIO_DATA_TYPE)"
    return a


def _u8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"], "expects contiguous uint8"
    return a


class Filter:
    """filt.h:133-162.  Filter(LPF|HPF, taps, Fs, Fx) or Filter(BPF, taps, Fs, Fl, Fu)."""

    def __init__(self, filt_t, num_taps, Fs, Fx, Fu=None, max_taps=_lib.REF_MAX_TAPS):
        self._L = _lib.lib()
        self._num_taps = int(num_taps)
        self._h = self._L.sdrx_fir_create(int(filt_t), 0 if Fu is None else 1, int(num_taps), float(Fs), float(Fx),
                                          0.0 if Fu is None else float(Fu), int(max_taps))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.sdrx_fir_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def get_error_flag(self):
        return self._L.sdrx_fir_error_flag(self._h)

    def get_taps(self, taps=None):
        if taps is None:
            taps = np.zeros(max(self._num_taps, 1))
        self._L.sdrx_fir_get_taps(self._h, _p(_f64(taps)))
        return taps

    def init(self):
        self._L.sdrx_fir_reset(self._h)

    def do_sample(self, data_sample):
        return self._L.sdrx_fir_do_sample(self._h, float(data_sample))

    def execute(self, buf, length=None):
        """In place, like Filter::execute(RADIO_DATA_TYPE*, size_t)."""
        _f64(buf)
        n = buf.size if length is None else int(length)
        _lib.check(self._L.sdrx_fir_execute(self._h, _p(buf), n), "sdrx_fir_execute")


def fir_execute_batch(filters, bufs, length=None):
    """Filter::execute on many Filter objects at once (sdrx_fir_execute_batch): bufs[c] is filtered in place by
    filters[c]; one staging block, one launch per filter design."""
    nchan = len(filters)
    assert len(bufs) == nchan
    n = (bufs[0].size if nchan else 0) if length is None else int(length)
    hs = (C.c_void_p * max(nchan, 1))(*[f._h for f in filters])
    ps = (C.c_void_p * max(nchan, 1))(*[_f64(b).ctypes.data for b in bufs])
    _lib.check(_lib.lib().sdrx_fir_execute_batch(hs, ps, n, nchan), "sdrx_fir_execute_batch")


class AM:
    """AM<RADIO_DATA_TYPE>, streams/AM.h:19-81."""

    THRESHOLD = 80.0

    def __init__(self):
        self._L = _lib.lib()

    def modulate(self, data_in, length, modulated_real, modulated_imag, length_modulated):
        self._L.sdrx_am_modulate(_p(_u8(data_in)), int(length), _p(_f64(modulated_real)), int(length_modulated))

    def demodulate(self, real, imag, output, number_of_points, downsample):
        return self._L.sdrx_am_demodulate(_p(_f64(real)), _p(_f64(imag)), _p(_u8(output)), int(number_of_points),
                                          int(downsample))


class FM:
    """FM<RADIO_DATA_TYPE>, streams/FM.h:16-157."""

    def __init__(self, tx_sample_rate, rx_sample_rate, MarkFreq=19000.0, SpaceFreq=20000.0):
        self._L = _lib.lib()
        self.tx, self.rx, self.mark, self.space = float(tx_sample_rate), float(rx_sample_rate), float(MarkFreq), float(SpaceFreq)

    def modulate(self, data_in, length, modulated_real, modulated_imag, length_modulated, samples_per_bit):
        self._L.sdrx_fm_modulate(self.tx, self.mark, self.space, _p(_u8(data_in)), int(length),
                                 _p(_f64(modulated_real)), int(samples_per_bit))

    def demodulate(self, real, imag, output, number_of_points, ds):
        return self._L.sdrx_fm_demodulate(self.rx, self.mark, self.space, _p(_f64(real)), _p(_u8(output)),
                                          int(number_of_points), int(ds))


    def convolve(self, input1, length1, input2, length2, output, lengthOutput, ds):
        """FM<T>::convolve with the `ds` parameter of the repair (FM.h:83-108): output[k] += ..., in place."""
        self._L.sdrx_fm_convolve(_p(_f64(input1)), int(length1), _p(_f64(input2)), int(length2), _p(_f64(output)),
                                 int(lengthOutput), int(ds))


def decimateComplexArray(block, length, decimate_by=2):
    """Resample.cpp:50-70 on an interleaved complex array (host index shuffle, as in include/sdrx/Resample.h): keeps
    samples 0, by, 2*by, ...; clears only element [kept] after them.  Returns the kept count."""
    z = block.reshape(-1, 2)
    kept = -(-int(length) // int(decimate_by)) if length else 0
    z[:kept] = z[0:length:decimate_by].copy()
    if kept < length:
        z[kept] = 0.0
    return kept


class _PSK:
    KIND = None

    def __init__(self, bpsk_uw, qpsk_uw, uw_len):
        self._L = _lib.lib()
        ptr = None
        if qpsk_uw is not None:
            self._uw = np.ascontiguousarray(qpsk_uw, dtype=np.float64)
            ptr = _p(self._uw)
        self._h = self._L.sdrx_psk_create(self.KIND, int(bpsk_uw), ptr, int(uw_len))
        if not self._h:
            raise RuntimeError("sdrx_psk_create failed: " + self._L.sdrx_last_error().decode())
        self.last_hits = []

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.sdrx_psk_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- state (the reference exposes every loop struct as a public member)
    def state(self):
        st = _lib.PskState()
        self._L.sdrx_psk_get_state(self._h, C.byref(st))
        return st

    def set_state(self, st):
        self._L.sdrx_psk_set_state(self._h, C.byref(st))

    def loop_state(self):
        return np.frombuffer(bytes(self.state())[: 8 * _lib.PskState.N_LOOP_DOUBLES], dtype=np.float64).copy()

    # ---- block level
    def demodulate(self, iq):
        """demodulate(BBP_Block* in, BBP_Block* out): iq = interleaved complex; returns the real parts
        of the output block (decisions as doubles), n/8 of them."""
        _f64(iq)
        n = iq.size // 2
        out = np.zeros(max(n // 8 + 2, 1))
        hits = np.zeros(4096, dtype=np.uint64)
        m = self._L.sdrx_psk_demodulate(self._h, _p(iq), n, _p(out), _p(hits), hits.size)
        nh = self.state().n_hits
        self.last_hits = [(int(h) >> 3, int(h) & 7) for h in hits[: min(nh, hits.size)]]
        return out[:m]

    # ---- stages (BPSK.h:86-117 / QPSK.h:79-91)
    def decode_pulseShape(self, iq, real, imag):
        return self._L.sdrx_psk_decode_pulse_shape(self._h, _p(_f64(iq)), iq.size // 2, _p(_f64(real)), _p(_f64(imag)))

    def symbol_timing_synchronization(self, real, imag, num_of_points):
        self._L.sdrx_psk_timing(self._h, _p(_f64(real)), _p(_f64(imag)), int(num_of_points))

    def phase_shift_correction(self, real, imag, num_of_points):
        return self._L.sdrx_psk_phase_correction(self._h, _p(_f64(real)), _p(_f64(imag)), int(num_of_points))

    def phase_ambiguity_correction(self, real, imag, num_of_points):
        hits = np.zeros(4096, dtype=np.uint64)
        nh = self._L.sdrx_psk_ambiguity(self._h, _p(_f64(real)), int(num_of_points), _p(hits), hits.size)
        self.last_hits = [(int(h) >> 3, int(h) & 7) for h in hits[: min(nh, hits.size)]]


class BPSK(_PSK):
    """streams/BPSK.h:20-309."""

    KIND = _lib.BPSK

    def __init__(self, UniqueWord=0x4E, UniqueWordLength=8):
        super().__init__(UniqueWord, None, UniqueWordLength)

    def demodulate_array(self, input_iq, output, number_of_points, downsample, real_plot_data=None, imag_plot_data=None):
        """BPSK::demodulate(Complex_Array&, uint8_t*, n, downsample, real_plot, imag_plot), BPSK.cpp:178-211: the input is
        decimated in place like the caller's valarray; returns the number of bits written to `output`."""
        _f64(input_iq)
        points = decimateComplexArray(input_iq, number_of_points, downsample)
        hits = np.zeros(4096, dtype=np.uint64)
        m = self._L.sdrx_psk_demodulate_plot(self._h, _p(input_iq), points, _p(_u8(output)),
                                             None if real_plot_data is None else _p(_f64(real_plot_data)),
                                             None if imag_plot_data is None else _p(_f64(imag_plot_data)), _p(hits), hits.size)
        nh = self.state().n_hits
        self.last_hits = [(int(h) >> 3, int(h) & 7) for h in hits[: min(nh, hits.size)]]
        return m

    def modulate(self, data_in, length, modulated_real, length_modulated):
        _lib.check(self._L.sdrx_psk_modulate(_lib.BPSK, _p(_u8(data_in)), int(length), _p(_f64(modulated_real)), None,
                                             int(length_modulated)), "sdrx_psk_modulate")


class QPSK(_PSK):
    """streams/QPSK.h:20-289."""

    KIND = _lib.QPSK

    def __init__(self, UniqueWord=None, UniqueWordLength=16):
        super().__init__(0, UniqueWord, UniqueWordLength)

    def modulate(self, data_in, length, modulated_real, modulated_imag, length_modulated):
        _lib.check(self._L.sdrx_psk_modulate(_lib.QPSK, _p(_u8(data_in)), int(length), _p(_f64(modulated_real)),
                                             _p(_f64(modulated_imag)), int(length_modulated)), "sdrx_psk_modulate")


class Chain:
    """BASELINE configs 1 and 2: Filter on I and Q, then AM threshold or /D + FM discriminator,
    with the Filter state carried across calls (streaming)."""

    def __init__(self, taps, decim=1):
        self._L = _lib.lib()
        self._taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.decim = int(decim)
        self._h = self._L.sdrx_chain_create(_p(self._taps), self._taps.size, self.decim)
        if not self._h:
            raise RuntimeError("sdrx_chain_create failed: " + self._L.sdrx_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.sdrx_chain_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        self._L.sdrx_chain_reset(self._h)

    def fir_am(self, re, im):
        n = re.size
        out = np.zeros((n + self.decim - 1) // self.decim + 8, dtype=np.uint8)
        m = self._L.sdrx_chain_fir_am(self._h, _p(_f64(re)), _p(_f64(im)), n, _p(out))
        return out[:m]

    def fir_fmdisc(self, re, im):
        n = re.size
        out = np.zeros((n + self.decim - 1) // self.decim + 8)
        m = self._L.sdrx_chain_fir_fmdisc(self._h, _p(_f64(re)), _p(_f64(im)), n, _p(out))
        return out[:m]

    @staticmethod
    def _u8_flags(signed_char, rotate90):
        return (_lib.U8_SIGNED_CHAR if signed_char else 0) | (_lib.U8_ROTATE90 if rotate90 else 0)

    def fir_am_u8(self, raw, bias=127.0, signed_char=False, rotate90=False):
        """raw: (n, 2) uint8 radio bytes, converted on the device (x = byte - bias)"""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        n = raw.size // 2
        out = np.zeros((n + self.decim - 1) // self.decim + 8, dtype=np.uint8)
        m = self._L.sdrx_chain_fir_am_u8(self._h, _p(raw), n, float(bias), self._u8_flags(signed_char, rotate90), _p(out))
        return out[:m]

    def fir_fmdisc_u8(self, raw, bias=127.0, signed_char=False, rotate90=False):
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        n = raw.size // 2
        out = np.zeros((n + self.decim - 1) // self.decim + 8)
        m = self._L.sdrx_chain_fir_fmdisc_u8(self._h, _p(raw), n, float(bias), self._u8_flags(signed_char, rotate90),
                                             _p(out))
        return out[:m]
