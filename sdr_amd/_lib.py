"""ctypes view of libsdrx.so (include/sdrx.h).  The library is the product: if it is missing
or does not load, importing this module's `lib()` raises -- there is no CPU fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDRX_LIB_PATH") or os.path.join(HERE, "libsdrx.so")  # override: experiments only

REAL, IQ_SPLIT, IQ_INTERLEAVED, IQ_U8 = 0, 1, 2, 3
U8_SIGNED_CHAR, U8_ROTATE90 = 1, 2
LPF, HPF, BPF = 0, 1, 2
BPSK, QPSK = 0, 1
F64, F32 = 0, 1
REF_MAX_TAPS = 1000

_vp = C.c_void_p
_sz = C.c_size_t


FIR_AM_WORK_BYTES = 16384  # SDRX_FIR_AM_WORK_BYTES


class FirJob(C.Structure):
    """sdrx_fir_job"""

    _fields_ = [
        ("layout", C.c_int), ("fast_fma", C.c_int),
        ("in0", _vp), ("in1", _vp),
        ("n", _sz), ("nchan", _sz), ("in_pitch", _sz),
        ("taps", _vp), ("num_taps", C.c_int), ("decim", C.c_uint),
        ("state_in", _vp), ("state_out", _vp), ("state_pitch", _sz),
        ("out0", _vp), ("out1", _vp), ("out_pitch", _sz), ("n_out", _sz),
        ("stream", _vp), ("dtype", C.c_int), ("u8_flags", C.c_int), ("u8_bias", C.c_double),
        ("n_before", _sz), ("taps_f32", _vp), ("am_work", _vp),
    ]


class RtlFmParams(C.Structure):
    _fields_ = [("downsample", C.c_int32), ("rotate90", C.c_int32), ("squelch_level", C.c_int32),
                ("post_downsample", C.c_int32), ("deemph_a", C.c_int32), ("dc_block", C.c_int32)]


class FrameEvent(C.Structure):
    """sdrx_frame_event"""

    _fields_ = [("pos", C.c_int64), ("kind", C.c_int32), ("len", C.c_uint8), ("crc", C.c_uint8), ("pad", C.c_uint8 * 2)]


class PskState(C.Structure):
    """sdrx_psk_state"""

    _fields_ = (
        [(n, C.c_double) for n in ("nco_delay", "reg_mu", "reg_Q", "reg_I", "mu_delay")]
        + [("iq_d", C.c_double * 5), ("ii_d", C.c_double * 5)]
        + [(n, C.c_double) for n in ("g1", "g2", "g3", "g4", "lf_x", "lf_y", "dds_delay", "plf_x", "plf_y")]
        + [("cc", C.c_int32 * 4), ("uw_size", C.c_int32), ("n_hits", C.c_int32), ("uw", (C.c_double * 64) * 4)]
    )
    N_LOOP_DOUBLES = 24


# name -> (restype, argtypes): every symbol include/sdrx.h declares
SIGNATURES = {
    "sdrx_init": (C.c_int, [C.c_int]),
    "sdrx_device_count": (C.c_int, []),
    "sdrx_last_error": (C.c_char_p, []),
    "sdrx_version": (C.c_char_p, []),
    "sdrx_tune": (C.c_int, [C.c_char_p, C.c_char_p]),
    "sdrx_last_kernel": (C.c_char_p, []),
    "sdrx_fir_dev": (C.c_int, [C.POINTER(FirJob)]),
    "sdrx_fir_am_dev": (C.c_int, [C.POINTER(FirJob)]),
    "sdrx_fir_fmdisc_dev": (C.c_int, [C.POINTER(FirJob)]),
    "sdrx_fir_am_verify_stats": (C.c_int, [_vp, C.c_int]),
    "sdrx_fir_state_doubles": (_sz, [C.c_int]),
    "sdrx_fir_ring_create": (_vp, [C.POINTER(FirJob), C.c_int, _sz, _sz, _sz, C.c_int]),
    "sdrx_fir_ring_launch": (C.c_int, [_vp, _vp]),
    "sdrx_fir_ring_state": (_vp, [_vp]),
    "sdrx_fir_ring_destroy": (None, [_vp]),
    "sdrx_am_demod_dev": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "sdrx_ones_zeros_dev": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "sdrx_ones_zeros_confidence_dev": (C.c_int, [_vp, _vp, _sz, _sz, _sz, C.c_int32, _vp, _vp]),
    "sdrx_fm_demod_dev": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "sdrx_rtlfm_dev": (C.c_int, [_vp, _sz, _sz, _sz, C.c_int, C.c_int, _vp, _vp, _vp, _sz, _vp]),
    "sdrx_rtlfm_full_dev": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sdrx_rtlfm_low_pass_real_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, C.c_int, _vp, _vp, _vp]),
    "sdrx_deframe_dev": (C.c_int, [_vp, _sz, _sz, _sz, C.c_int, _vp, _vp, _sz, _vp, _vp]),
    "sdrx_iir_dev": (C.c_int, [C.c_int, _vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "sdrx_lowpass1_coef": (C.c_double, [C.c_double, C.c_double]),
    "sdrx_notch_design": (None, [C.c_double, C.c_double, _vp]),
    "sdrx_normalize_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _vp]),
    "sdrx_fifth_order_decimate_dev": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    "sdrx_fifth_order_decimate_block_dev": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _vp, C.POINTER(C.c_size_t), _vp]),
    "sdrx_fft_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp]),
    "sdrx_am_modulate_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    "sdrx_fm_modulate_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _vp, _sz, _vp, _sz, _vp]),
    "sdrx_fm_convolve_dev": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "sdrx_fm_tables": (None, [C.c_double, C.c_double, C.c_double, _sz, C.c_int, _vp, _vp]),
    "sdrx_rrc_coeff": (C.POINTER(C.c_double), []),
    "sdrx_qpsk_default_unique_word": (C.POINTER(C.c_double), []),
    "sdrx_psk_state_init_bpsk": (None, [C.POINTER(PskState), C.c_uint64, _sz]),
    "sdrx_psk_state_init_qpsk": (None, [C.POINTER(PskState), _vp, _sz]),
    "sdrx_psk_work_bytes": (_sz, [_sz, _sz]),
    "sdrx_psk_demod_dev": (C.c_int, [C.c_int, _vp, _sz, _sz, _sz, _vp, _vp, _sz, _vp, _vp]),
    "sdrx_psk_demod_noalloc_max_bytes": (_sz, []),
    "sdrx_psk_demod_noalloc_dev": (C.c_int, [C.c_int, _vp, _sz, _sz, _sz, _vp, _vp, _sz, _vp]),
    "sdrx_psk_matched_filter_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _sz, _vp]),
    "sdrx_psk_loops_dev": (C.c_int, [C.c_int, _vp, _sz, _sz, _sz, _vp, _vp, _sz, C.c_int, C.c_int, _vp]),
    "sdrx_psk_ambiguity_dev": (C.c_int, [C.c_int, _vp, _sz, _sz, _sz, _vp, _vp]),
    "sdrx_psk_modulate_dev": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp, _sz, _vp]),
    "sdrx_fir_design": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _vp]),
    "sdrx_fir_create": (_vp, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]),
    "sdrx_fir_destroy": (None, [_vp]),
    "sdrx_fir_error_flag": (C.c_int, [_vp]),
    "sdrx_fir_get_taps": (None, [_vp, _vp]),
    "sdrx_fir_reset": (None, [_vp]),
    "sdrx_fir_execute": (C.c_int, [_vp, _vp, _sz]),
    "sdrx_fir_do_sample": (C.c_double, [_vp, C.c_double]),
    "sdrx_fir_execute_batch": (C.c_int, [_vp, _vp, _sz, _sz]),
    "sdrx_am_demodulate": (_sz, [_vp, _vp, _vp, _sz, _sz]),
    "sdrx_am_modulate": (None, [_vp, _sz, _vp, _sz]),
    "sdrx_fm_demodulate": (_sz, [C.c_double, C.c_double, C.c_double, _vp, _vp, _sz, _sz]),
    "sdrx_fm_modulate": (None, [C.c_double, C.c_double, C.c_double, _vp, _sz, _vp, _sz]),
    "sdrx_fm_convolve": (None, [_vp, _sz, _vp, _sz, _vp, _sz, _sz]),
    "sdrx_chain_create": (_vp, [_vp, C.c_int, C.c_uint]),
    "sdrx_chain_destroy": (None, [_vp]),
    "sdrx_chain_reset": (None, [_vp]),
    "sdrx_chain_fir_am": (_sz, [_vp, _vp, _vp, _sz, _vp]),
    "sdrx_chain_fir_fmdisc": (_sz, [_vp, _vp, _vp, _sz, _vp]),
    "sdrx_chain_fir_am_u8": (_sz, [_vp, _vp, _sz, C.c_double, C.c_int, _vp]),
    "sdrx_chain_fir_fmdisc_u8": (_sz, [_vp, _vp, _sz, C.c_double, C.c_int, _vp]),
    "sdrx_psk_create": (_vp, [C.c_int, C.c_uint64, _vp, _sz]),
    "sdrx_psk_destroy": (None, [_vp]),
    "sdrx_psk_get_state": (None, [_vp, C.POINTER(PskState)]),
    "sdrx_psk_set_state": (None, [_vp, C.POINTER(PskState)]),
    "sdrx_psk_demodulate": (_sz, [_vp, _vp, _sz, _vp, _vp, _sz]),
    "sdrx_psk_demodulate_plot": (_sz, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz]),
    "sdrx_psk_decode_pulse_shape": (_sz, [_vp, _vp, _sz, _vp, _vp]),
    "sdrx_psk_timing": (None, [_vp, _vp, _vp, _sz]),
    "sdrx_psk_phase_correction": (_sz, [_vp, _vp, _vp, _sz]),
    "sdrx_psk_ambiguity": (_sz, [_vp, _vp, _sz, _vp, _sz]),
    "sdrx_psk_modulate": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp, _sz]),
    "am_gpu_initialize": (None, [_sz, _sz]),
    "am_gpu_free": (None, []),
    "am_gpu_demodulation": (C.c_double, [C.c_int, _vp, _vp, _vp, _sz, _sz]),
    "fm_gpu_initialize": (None, [_sz, _sz, _sz, _sz, C.c_double, C.c_double]),
    "fm_gpu_free": (None, []),
    "fm_gpu_set_samples_per_bit": (None, [_sz, _sz]),
    "fm_gpu_demodulation": (C.c_double, [C.c_int, _vp, _vp, _vp, _sz, _sz]),
    "ones_zeros_demodulation": (C.c_int32, [_vp, _vp, _sz]),
}

_lib = None


def lib():
    """Load libsdrx.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m sdr_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # torch first: it ships its own HIP runtime, and one process must not initialise two of them.  With
    # torch's loaded, libsdrx.so binds to it (same soname); loaded the other way round, the second
    # runtime to start sees no device.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libsdrx {what} failed ({rc}): {lib().sdrx_last_error().decode()}")
