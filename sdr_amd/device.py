"""Device-resident entry points of libsdrx (include/sdrx.h, layer 1) driven from torch tensors.

torch is plumbing only: device memory, streams, torch.distributed.  Every compute call goes to
the hand-written HIP kernels through the C ABI with raw device pointers.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(cond, what):
    """The kernels trust the shapes they are given (a wrong one is an out-of-bounds access on the GPU): checked here."""
    if not cond:
        raise ValueError("sdr_amd.device: " + what)


def init(device_index=0):
    _lib.check(_lib.lib().sdrx_init(int(device_index)), "sdrx_init")


class FirChain:
    """FIR (+ fused demod) over `nchan` independent channels with carried state.

    Mirrors Filter objects on I and Q (filt.cpp:236-273) per channel: hist = the shift register,
    pre = the discriminator's previous sample (rtl_fm.c:523,543-544)."""

    def __init__(self, taps, nchan=1, decim=1, device="cuda", dtype=torch.float64):
        """dtype = torch.float64: the reference's arithmetic; torch.float32: the fp32 twins (float samples,
        taps, state and outputs)."""
        self.L = _lib.lib()
        assert dtype in (torch.float64, torch.float32)
        self.dtype = dtype
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.T = int(taps.size)
        self.decim = int(decim)
        self.nchan = int(nchan)
        self.device = torch.device(device)
        self.taps = torch.from_numpy(taps).to(self.device).to(dtype)
        # the taps rounded to float: with them sdrx_fir_am_dev filters in fp32 and verifies every decision (same bits)
        self.taps_f32 = torch.from_numpy(taps.astype(np.float32)).to(self.device) if dtype == torch.float64 else None
        # ... and a work buffer for the plan of its int8 form (tiles of 8-bit integer samples: the matrix cores)
        self.am_work = torch.zeros(_lib.FIR_AM_WORK_BYTES, dtype=torch.uint8, device=self.device) if dtype == torch.float64 else None
        self.pitch = int(self.L.sdrx_fir_state_doubles(self.T))
        self.state = [torch.zeros(self.nchan, self.pitch, dtype=dtype, device=self.device) for _ in range(2)]
        self.cur = 0
        self.u8_bias, self.u8_flags = 127.0, 0  # for uint8 {I,Q} inputs: x = byte - 127 (rtlsdr_receiver.cpp:1587-1588)

    def set_u8_conversion(self, bias=127.0, signed_char=False, rotate90=False):
        self.u8_bias = float(bias)
        self.u8_flags = (_lib.U8_SIGNED_CHAR if signed_char else 0) | (_lib.U8_ROTATE90 if rotate90 else 0)

    def reset(self):
        for s in self.state:
            s.zero_()
        self.cur = 0

    def n_out(self, n):
        return (n + self.decim - 1) // self.decim

    def _job(self, layout, in0, in1, n, in_pitch, out0, out1, out_pitch, fast_fma, n_out):
        j = _lib.FirJob()
        j.layout = layout
        j.fast_fma = int(fast_fma)  # False/0 exact, True/1 fma chain, 2 fp64 MFMA
        j.in0, j.in1 = _ptr(in0), _ptr(in1)
        j.n, j.nchan, j.in_pitch = n, self.nchan, in_pitch
        j.taps, j.num_taps, j.decim = _ptr(self.taps), self.T, self.decim
        j.state_in, j.state_out = _ptr(self.state[self.cur]), _ptr(self.state[self.cur ^ 1])
        j.state_pitch = self.pitch
        j.out0, j.out1, j.out_pitch, j.n_out = _ptr(out0), _ptr(out1), out_pitch, n_out
        j.stream = _stream()
        j.dtype = _lib.F32 if self.dtype == torch.float32 else _lib.F64
        j.u8_bias, j.u8_flags = self.u8_bias, self.u8_flags
        j.taps_f32 = _ptr(self.taps_f32)
        j.am_work = _ptr(self.am_work)
        return j

    def _run(self, fn, name, layout, in0, in1, n, in_pitch, out0, out1, out_pitch, fast_fma=False, n_out=0):
        j = self._job(layout, in0, in1, n, in_pitch, out0, out1, out_pitch, fast_fma, n_out)
        _lib.check(fn(C.byref(j)), name)
        self.cur ^= 1

    # x: (nchan, n, 2) interleaved complex, or (re, im) tuple of (nchan, n), or a single real (nchan, n),
    # or (nchan, n, 2) uint8 = raw RTL-SDR bytes, converted on load
    def _inputs(self, x):
        if isinstance(x, (tuple, list)):
            re, im = x
            assert re.dtype == self.dtype == im.dtype and re.is_contiguous() and im.is_contiguous()
            _need(re.shape == im.shape and re.numel() == self.nchan * re.shape[-1], "re / im must be (nchan, n)")
            return _lib.IQ_SPLIT, re, im, re.shape[-1], re.shape[-1]
        if x.dtype == torch.uint8:  # (nchan, n, 2) raw radio bytes
            assert x.is_contiguous() and x.dim() == 3 and x.shape[-1] == 2
            _need(x.shape[0] == self.nchan, "input must have nchan rows")
            return _lib.IQ_U8, x, None, x.shape[1], x.shape[1]
        assert x.dtype == self.dtype and x.is_contiguous()
        if x.dim() == 3 and x.shape[-1] == 2:
            _need(x.shape[0] == self.nchan, "input must have nchan rows")
            return _lib.IQ_INTERLEAVED, x, None, x.shape[1], x.shape[1]
        _need(x.numel() == self.nchan * x.shape[-1], "input must be (nchan, n)")
        return _lib.REAL, x, None, x.shape[-1], x.shape[-1]

    def _outputs(self, n, n_out, *outs):
        want = int(n_out) if n_out else self.n_out(n)
        for o in outs:
            if o is not None:
                _need(o.is_contiguous() and o.dim() >= 2 and o.shape[0] == self.nchan and o.shape[1] >= want,
                      "output must be (nchan, >= %d) and contiguous" % want)

    def fir(self, x, out0, out1=None, fast_fma=False, n_out=0):
        layout, in0, in1, n, pitch = self._inputs(x)
        self._outputs(n, n_out, out0, out1)
        op = out0.shape[1]
        self._run(self.L.sdrx_fir_dev, "sdrx_fir_dev", layout, in0, in1, n, pitch, out0, out1, op, fast_fma, n_out)

    def fir_am(self, x, bits, fast_fma=False):
        layout, in0, in1, n, pitch = self._inputs(x)
        self._outputs(n, 0, bits)
        self._run(self.L.sdrx_fir_am_dev, "sdrx_fir_am_dev", layout, in0, in1, n, pitch, bits, None, bits.shape[1],
                  fast_fma)

    def fir_fmdisc(self, x, angle, fast_fma=False):
        layout, in0, in1, n, pitch = self._inputs(x)
        self._outputs(n, 0, angle)
        self._run(self.L.sdrx_fir_fmdisc_dev, "sdrx_fir_fmdisc_dev", layout, in0, in1, n, pitch, angle, None,
                  angle.shape[1], fast_fma)


class FirRing:
    """`chunks` consecutive chunk launches of one FirChain's stream, replayed from a HIP graph (sdrx_fir_ring_*).
    x: (chunks, n, 2) interleaved complex (or (chunks, n) real) ring of input chunks; out: (chunks, n_out_per_chunk)
    ring of outputs; epilogue: "fir" | "am" | "fmdisc".  The chain must have nchan == 1; its state is the stream's.
    use_graph: 1 = one launch over a contiguous ring, else a HIP graph; 2 = always the graph; 0 = launches one by one."""

    def __init__(self, chain, x, out, epilogue="fmdisc", use_graph=1):
        assert chain.nchan == 1 and x.is_contiguous() and out.is_contiguous() and x.shape[0] == out.shape[0]
        self.chain, self.x, self.out = chain, x, out
        layout, in0, in1, n, pitch = chain._inputs(x[0:1])
        assert in1 is None
        _need(out.dim() == 2 and out.shape[1] >= chain.n_out(n), "FirRing: out must be (chunks, >= %d)" % chain.n_out(n))
        j = chain._job(layout, x[0:1], None, n, pitch, out[0:1], None, out.shape[1], False, 0)
        j.stream = None
        self.L = _lib.lib()
        self.h = self.L.sdrx_fir_ring_create(C.byref(j), {"fir": 0, "am": 1, "fmdisc": 2}[epilogue], x.shape[0],
                                             x.stride(0) * x.element_size(), out.stride(0) * out.element_size(),
                                             int(use_graph))
        if not self.h:
            raise RuntimeError("sdrx_fir_ring_create failed: " + self.L.sdrx_last_error().decode())
        self._state0 = chain.state[chain.cur].data_ptr()

    def launch(self):
        _lib.check(self.L.sdrx_fir_ring_launch(self.h, _stream()), "sdrx_fir_ring_launch")
        self.chain.cur = 0 if self.L.sdrx_fir_ring_state(self.h) == self.chain.state[0].data_ptr() else 1

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.sdrx_fir_ring_destroy(self.h)
                self.h = None
        except Exception:
            pass


def am_demod(re, im, out, ds):
    _need(im.numel() == re.numel() and int(ds) >= 1 and out.numel() >= (re.numel() + int(ds) - 1) // int(ds), "am_demod: sizes")
    _lib.check(_lib.lib().sdrx_am_demod_dev(_ptr(re), _ptr(im), _ptr(out), re.numel(), int(ds), _stream()),
               "sdrx_am_demod_dev")


def ones_zeros(re, im, count):
    _need(im.numel() == re.numel() and count.numel() >= 1, "ones_zeros: sizes")
    _lib.check(_lib.lib().sdrx_ones_zeros_dev(_ptr(re), _ptr(im), re.numel(), _ptr(count), _stream()),
               "sdrx_ones_zeros_dev")


def ones_zeros_confidence(re, im, cmax, counters):
    """re, im: (nchan, n) float64; counters: (nchan,) int32, walked in place like the receiver's CPU branch walks its
    confidence_counter(cmax, ...) (rtlsdr_receiver.cpp:2063-2074)."""
    assert re.dtype == torch.float64 and re.dim() == 2 and re.is_contiguous() and im.shape == re.shape and im.is_contiguous()
    nchan, n = re.shape
    assert counters.dtype == torch.int32 and counters.numel() == nchan and counters.is_contiguous()
    _lib.check(_lib.lib().sdrx_ones_zeros_confidence_dev(_ptr(re), _ptr(im), n, nchan, n, int(cmax), _ptr(counters), _stream()),
               "sdrx_ones_zeros_confidence_dev")
    return counters


def fm_demod(re, w1, w2, out, ds):
    _need(int(ds) >= 1 and w1.numel() >= int(ds) and w2.numel() >= int(ds) and out.numel() >= re.numel() // int(ds), "fm_demod: sizes")
    _lib.check(_lib.lib().sdrx_fm_demod_dev(_ptr(re), _ptr(w1), _ptr(w2), _ptr(out), re.numel(), int(ds), _stream()),
               "sdrx_fm_demod_dev")


def am_modulate(bits, out, len_mod=None):
    """AM<T>::modulate over every row of bits (nchan, len) uint8 -> out (nchan, >= len_mod) float64."""
    assert bits.dtype == torch.uint8 and bits.dim() == 2 and bits.is_contiguous()
    assert out.dtype == torch.float64 and out.dim() == 2 and out.is_contiguous() and out.shape[0] == bits.shape[0]
    len_mod = out.shape[1] if len_mod is None else int(len_mod)
    _need(0 <= len_mod <= out.shape[1], "am_modulate: len_mod beyond the output rows")
    _lib.check(_lib.lib().sdrx_am_modulate_dev(_ptr(bits), bits.shape[1], bits.shape[0], bits.shape[1], _ptr(out), len_mod,
                                               out.shape[1], _stream()), "sdrx_am_modulate_dev")
    return out


def fm_tables(rate, mark, space, n, for_modulator):
    """The reference's host-built cosine tables (FM.h:47-57 / FM.h:127-131) as numpy arrays."""
    w1, w2 = np.zeros(max(n, 1)), np.zeros(max(n, 1))
    _lib.lib().sdrx_fm_tables(float(rate), float(mark), float(space), int(n), int(bool(for_modulator)), w1.ctypes.data,
                              w2.ctypes.data)
    return w1[:n], w2[:n]


def fm_modulate(bits, w_mark, w_space, out):
    """FM<T>::modulate over every row of bits (nchan, len) uint8; w_mark / w_space: (spb,) float64 device tables;
    out (nchan, >= len * spb) float64."""
    assert bits.dtype == torch.uint8 and bits.dim() == 2 and bits.is_contiguous()
    assert out.dtype == torch.float64 and out.dim() == 2 and out.is_contiguous() and out.shape[0] == bits.shape[0]
    spb = w_mark.numel()
    assert w_space.numel() == spb and out.shape[1] >= bits.shape[1] * spb
    _lib.check(_lib.lib().sdrx_fm_modulate_dev(_ptr(bits), bits.shape[1], bits.shape[0], bits.shape[1], _ptr(w_mark),
                                               _ptr(w_space), spb, _ptr(out), out.shape[1], _stream()), "sdrx_fm_modulate_dev")
    return out


def fm_convolve(x, w, out, ds):
    """FM<T>::convolve (repaired): out[k] += sum_j x[(k+1)ds-1-j] * w[len2-1-j]; 1-D float64 device tensors."""
    ds = int(ds)
    _need(ds >= 1, "fm_convolve: ds must be >= 1")
    for t in (x, w, out):
        _need(t.dtype == torch.float64 and t.is_contiguous(), "fm_convolve: contiguous float64 tensors")
    n_out = (x.numel() - ds + 1 + ds - 1) // ds if x.numel() >= ds else 0  # outputs at ds-1, 2ds-1, ... < len1
    _need(out.numel() >= n_out, "fm_convolve: out must hold %d doubles" % n_out)
    _lib.check(_lib.lib().sdrx_fm_convolve_dev(_ptr(x), x.numel(), _ptr(w), w.numel(), _ptr(out), int(ds), _stream()),
               "sdrx_fm_convolve_dev")
    return out


class RtlFm:
    """rtl_fm's int16 chain (u8 bytes -> -127 -> boxcar decimate -> polar discriminant -> int16 pcm) over
    `nchan` independent byte streams, state carried like struct demod_state (rtl_fm.c:113-147)."""

    def __init__(self, nchan=1, downsample=10, rotate90=False, device="cuda"):
        self.L = _lib.lib()
        self.nchan, self.ds, self.rot = int(nchan), int(downsample), bool(rotate90)
        self.state = [torch.zeros(self.nchan, 5, dtype=torch.int32, device=device) for _ in range(2)]
        self.cur = 0
        self.prev_index = 0

    def demod(self, raw, pcm):
        """raw: (nchan, n, 2) uint8; pcm: (nchan, >= (prev_index + n) // downsample) int16.  Returns the count."""
        assert raw.dtype == torch.uint8 and raw.is_contiguous() and pcm.dtype == torch.int16 and pcm.is_contiguous()
        n = raw.shape[1]
        _need(raw.dim() == 3 and raw.shape[0] == self.nchan and raw.shape[2] == 2, "raw must be (nchan, n, 2)")
        _need(pcm.dim() == 2 and pcm.shape[0] == self.nchan and pcm.shape[1] >= (self.prev_index + n) // self.ds,
              "pcm must be (nchan, >= (prev_index + n) // downsample)")
        _lib.check(self.L.sdrx_rtlfm_dev(_ptr(raw), n, self.nchan, n, self.ds, int(self.rot), _ptr(self.state[self.cur]),
                                         _ptr(self.state[self.cur ^ 1]), _ptr(pcm), pcm.shape[1], _stream()),
                   "sdrx_rtlfm_dev")
        self.cur ^= 1
        m = (self.prev_index + n) // self.ds
        self.prev_index = (self.prev_index + n) % self.ds
        return m


class RtlFmFull(RtlFm):
    """RtlFm plus the rest of rtl_fm's full_demod: power squelch, post_downsample (low_pass_simple), de-emphasis and
    DC block (sdrx_rtlfm_full_dev).  post_state: (nchan, 4) int32 = deemph avg, dc_avg, squelch_hits, result_len."""

    def __init__(self, nchan=1, downsample=10, rotate90=False, squelch_level=0, post_downsample=1, deemph_a=0,
                 dc_block=False, device="cuda"):
        super().__init__(nchan, downsample, rotate90, device)
        self.params = _lib.RtlFmParams(int(downsample), int(bool(rotate90)), int(squelch_level), int(post_downsample),
                                       int(deemph_a), int(bool(dc_block)))
        self.post_state = torch.zeros(self.nchan, 4, dtype=torch.int32, device=device)
        self.work = torch.zeros(self.nchan, 2, dtype=torch.int64, device=device)

    def demod(self, raw, pcm):
        """As RtlFm.demod; returns result_len (the same for every channel)."""
        assert raw.dtype == torch.uint8 and raw.is_contiguous() and pcm.dtype == torch.int16 and pcm.is_contiguous()
        n = raw.shape[1]
        _need(raw.dim() == 3 and raw.shape[0] == self.nchan and raw.shape[2] == 2, "raw must be (nchan, n, 2)")
        _need(pcm.dim() == 2 and pcm.shape[0] == self.nchan and pcm.shape[1] >= (self.prev_index + n) // self.ds,
              "pcm must be (nchan, >= (prev_index + n) // downsample)")
        _lib.check(self.L.sdrx_rtlfm_full_dev(C.byref(self.params), _ptr(raw), n, self.nchan, n, _ptr(self.state[self.cur]),
                                              _ptr(self.state[self.cur ^ 1]), _ptr(self.post_state), _ptr(self.work),
                                              _ptr(pcm), pcm.shape[1], _stream()), "sdrx_rtlfm_full_dev")
        self.cur ^= 1
        m = (self.prev_index + n) // self.ds
        self.prev_index = (self.prev_index + n) % self.ds
        return m // self.params.post_downsample if self.params.post_downsample > 1 else m


def rtlfm_low_pass_real(pcm, rate_out, rate_out2, post_state, lpr_state):
    """low_pass_real (rtl_fm.c:341-361) in place on the rows of pcm (nchan, pitch) int16; post_state (nchan, 4) int32 as
    RtlFmFull keeps it (row length at [:, 3], updated); lpr_state (nchan, 2) int32 = now_lpr, prev_lpr_index."""
    assert pcm.dtype == torch.int16 and pcm.dim() == 2 and pcm.is_contiguous()
    assert post_state.dtype == torch.int32 and lpr_state.dtype == torch.int32 and lpr_state.numel() == 2 * pcm.shape[0]
    _need(post_state.is_contiguous() and post_state.numel() >= 4 * pcm.shape[0] and lpr_state.is_contiguous(),
          "rtlfm_low_pass_real: post_state must be (nchan, 4), both state tensors contiguous")
    _lib.check(_lib.lib().sdrx_rtlfm_low_pass_real_dev(_ptr(pcm), pcm.shape[0], pcm.shape[1], int(rate_out), int(rate_out2),
                                                       _ptr(post_state), _ptr(lpr_state), _stream()),
               "sdrx_rtlfm_low_pass_real_dev")


def deframe(bits, crc_enable=True, max_events=1024):
    """bits: (nchan, n) uint8 0/1 on the device.  Returns per channel a dict like the reference's loop would leave:
    events (pos, kind, len, crc as numpy arrays), payload (k, 250), counts; see sdrx_deframe_dev."""
    assert bits.dtype == torch.uint8 and bits.is_contiguous() and bits.dim() == 2
    nchan, n = bits.shape
    ev = torch.zeros(nchan, max_events, 16, dtype=torch.uint8, device=bits.device)
    pay = torch.zeros(nchan, max_events, 250, dtype=torch.uint8, device=bits.device)
    summ = torch.zeros(nchan, 4, dtype=torch.int64, device=bits.device)
    _lib.check(_lib.lib().sdrx_deframe_dev(_ptr(bits), n, nchan, n, int(bool(crc_enable)), _ptr(ev), _ptr(pay), max_events,
                                           _ptr(summ), _stream()), "sdrx_deframe_dev")
    summ_h, ev_h, pay_h = summ.cpu().numpy(), ev.cpu().numpy(), pay.cpu().numpy()
    out = []
    for c in range(nchan):
        k = int(min(summ_h[c, 0], max_events))
        raw = ev_h[c, :k]
        out.append({"pos": raw[:, :8].copy().view(np.int64).reshape(-1), "kind": raw[:, 8:12].copy().view(np.int32).reshape(-1),
                    "len": raw[:, 12].copy(), "crc": raw[:, 13].copy(), "payload": pay_h[c, :k],
                    "events": int(summ_h[c, 0]), "consumed": int(summ_h[c, 1]), "zero_returns": int(summ_h[c, 2])})
    return out


# ---- DigitalSignalProcessing primitives on (nchan, n) float64 device tensors (sdrx.h: sdrx_iir_dev ...) ----
IIR_DC_BLOCK, IIR_LOWPASS, IIR_NOTCH, IIR_INTEGRATE = 0, 1, 2, 3
FFT_MAX = 4096


def _rows(x):
    assert x.dtype == torch.float64 and x.dim() == 2 and x.is_contiguous()
    return x.shape


def iir(kind, coef, x, y=None, state=None):
    """DC_Filter / LowPassFilter / NotchFilter / FirstOrderDigitalIntegration over every row of x (in place when y is
    None).  coef: host floats; state: (nchan, 4) float64 device tensor carried between calls (lowpass, notch)."""
    nchan, n = _rows(x)
    y = x if y is None else y
    assert y.shape == x.shape and y.dtype == torch.float64 and y.is_contiguous()
    if state is not None:
        assert state.dtype == torch.float64 and state.is_contiguous() and state.numel() == nchan * 4
    coef = np.ascontiguousarray(np.atleast_1d(coef), dtype=np.float64)
    _lib.check(_lib.lib().sdrx_iir_dev(int(kind), coef.ctypes.data, _ptr(x), _ptr(y), n, nchan, n,
                                       _ptr(state) if state is not None else None, _stream()), "sdrx_iir_dev")
    return y


def lowpass1_coef(cutoff, dt):
    return float(_lib.lib().sdrx_lowpass1_coef(cutoff, dt))


def notch_design(bw, fc):
    c = np.zeros(5)
    _lib.lib().sdrx_notch_design(bw, fc, c.ctypes.data)
    return c


def normalize(x, max_inout):
    """normalize() over every row of x in place; max_inout: (nchan,) float64 device tensor, prev_max in, max out."""
    nchan, n = _rows(x)
    assert max_inout.dtype == torch.float64 and max_inout.is_contiguous() and max_inout.numel() == nchan
    _lib.check(_lib.lib().sdrx_normalize_dev(_ptr(x), n, nchan, n, _ptr(max_inout), _stream()), "sdrx_normalize_dev")
    return x


def fifth_order_decimate(x, by, state):
    """fifth_order_filter::decimate over every row in place; state: (nchan, 6) float64 device tensor.  Returns n // by."""
    nchan, n = _rows(x)
    assert state.dtype == torch.float64 and state.is_contiguous() and state.numel() == nchan * 6
    _lib.check(_lib.lib().sdrx_fifth_order_decimate_dev(_ptr(x), n, nchan, n, int(by), _ptr(state), _stream()),
               "sdrx_fifth_order_decimate_dev")
    return n // int(by)


def fifth_order_decimate_block(z, n_points, by, state):
    """fifth_order_filter::decimate(BBP_Block*) over every row of z (complex128, (nchan, capacity)) of which n_points
    are valid; state: (nchan, 12) float64 device tensor.  Returns the block's n_points afterwards."""
    assert z.dtype == torch.complex128 and z.dim() == 2 and z.is_contiguous()
    nchan, cap = z.shape
    assert state.dtype == torch.float64 and state.is_contiguous() and state.numel() == nchan * 12
    out = C.c_size_t(0)
    _lib.check(_lib.lib().sdrx_fifth_order_decimate_block_dev(_ptr(z), int(n_points), nchan, cap, int(by), _ptr(state),
                                                              C.byref(out), _stream()), "sdrx_fifth_order_decimate_block_dev")
    return int(out.value)


def fft(z):
    """fft() over every row of z (complex128, (nchan, n), n a power of two <= FFT_MAX), in place."""
    assert z.dtype == torch.complex128 and z.dim() == 2 and z.is_contiguous()
    nchan, n = z.shape
    _lib.check(_lib.lib().sdrx_fft_dev(_ptr(z), n, nchan, n, _stream()), "sdrx_fft_dev")
    return z


class PskBank:
    """nchan independent BPSK or QPSK demodulators (BPSK.cpp:148-176 / QPSK.cpp:190-219), device resident."""

    def __init__(self, kind, nchan, n, unique_word=None, uw_len=None, device="cuda"):
        self.L = _lib.lib()
        self.kind, self.nchan, self.n = int(kind), int(nchan), int(n)
        self.device = torch.device(device)
        st = _lib.PskState()
        if kind == _lib.BPSK:
            self.L.sdrx_psk_state_init_bpsk(C.byref(st), 0x4E if unique_word is None else int(unique_word),
                                            8 if uw_len is None else int(uw_len))
        else:
            uw = None
            if unique_word is not None:
                self._uw = np.ascontiguousarray(unique_word, dtype=np.float64)
                uw = self._uw.ctypes.data_as(C.c_void_p)
            self.L.sdrx_psk_state_init_qpsk(C.byref(st), uw, 16 if uw_len is None else int(uw_len))
        self.state_bytes = C.sizeof(_lib.PskState)
        one = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8)
        self.states = one.repeat(self.nchan, 1).contiguous().to(self.device)
        self.work = torch.empty(int(self.L.sdrx_psk_work_bytes(self.n, self.nchan)), dtype=torch.uint8,
                                device=self.device)
        self.n_dec = (self.n // 4 + 1) // 2

    def demod(self, iq, out):
        """iq: (nchan, n, 2) f64; out: (nchan, >= n/8) uint8."""
        assert iq.dtype == torch.float64 and iq.is_contiguous() and iq.shape[1] == self.n
        _need(iq.shape[0] == self.nchan and iq.numel() == self.nchan * self.n * 2, "iq must be (nchan, n, 2)")
        _need(out.dtype == torch.uint8 and out.is_contiguous() and out.shape[0] == self.nchan and out.shape[1] >= self.n_dec,
              "out must be (nchan, >= n_dec) uint8")
        _lib.check(self.L.sdrx_psk_demod_dev(self.kind, _ptr(iq), self.n, self.nchan, self.n, _ptr(self.states),
                                             _ptr(out), out.shape[1], _ptr(self.work), _stream()),
                   "sdrx_psk_demod_dev")

    def host_states(self):
        raw = self.states.cpu().numpy().tobytes()
        return [_lib.PskState.from_buffer_copy(raw[i * self.state_bytes:(i + 1) * self.state_bytes])
                for i in range(self.nchan)]
