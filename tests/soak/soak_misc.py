"""One-off soak of the stand-alone kernels: the parametrized tests of tests/test_gpu_parity.py called with random
parameters (shapes, decimations, divisors, rates) and shifted seeds (other data), each against the oracle bit for bit.
    python tests/soak/soak_misc.py [seconds]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_parity as T
from sdr_amd import _lib

orig = np.random.default_rng
pick = orig(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)


def tune(name, value):
    _lib.check(_lib.lib().sdrx_tune(name.encode(), None if value is None else str(value).encode()), "sdrx_tune")


def r(lo, hi):
    return int(pick.integers(lo, hi + 1))


def log_uniform(lo, hi):
    return int(round(np.exp(pick.uniform(np.log(lo), np.log(hi)))))


def one_round():
    cases = []
    rot = bool(r(0, 1))
    cases.append(("rtlfm_full", lambda: T.test_rtlfm_full_demod_vs_oracle(dict(
        downsample=r(1, 16), squelch_level=[0, r(5, 400)][r(0, 1)], post_downsample=r(1, 6),
        deemph_a=[0, r(1, 70), 1 << r(0, 10), r(2040, 3000)][r(0, 3)], dc_block=bool(r(0, 1)), rotate90=rot), expect_both=False)))
    ro = log_uniform(2, 2000000)
    cases.append(("low_pass_real", lambda: T.test_rtlfm_low_pass_real_vs_oracle(ro, r(1, ro))))
    cases.append(("rtlfm_int16", lambda: T.test_rtlfm_int16_chain(log_uniform(1, 600), bool(r(0, 1)))))
    cases.append(("fm_demod", lambda: T.test_fm_demod_vs_oracle(log_uniform(1, 3000))))
    cases.append(("am_demod", lambda: T.test_am_demod_vs_oracle(log_uniform(1, 500))))
    cases.append(("fifth", lambda: T.test_dsp_fifth_order_decimate_bit_exact(log_uniform(1, 40000), 1 << r(0, 4))))
    cases.append(("fifth_block", lambda: T.test_dsp_fifth_order_block_overload_bit_exact(log_uniform(1, 40000), 1 << r(0, 4))))
    cases.append(("confidence", lambda: T.test_ones_zeros_confidence_walk(log_uniform(1, 400000))))
    cases.append(("confidence_segments", lambda: T.test_ones_zeros_confidence_walk_in_segments(r(1, 12), r(262144, 1500000))))
    cases.append(("iir", lambda: T.test_dsp_recurrences_bit_exact(log_uniform(1, 9000), [4, 16, 64, -4, -16][r(0, 4)], tune)))
    cases.append(("normalize", lambda: T.test_dsp_normalize_bit_exact(log_uniform(1, 20000), float(np.exp(pick.uniform(-3, 6))),
                                                                      float(pick.uniform(0, 300)))))
    cases.append(("fft", lambda: T.test_dsp_fft(1 << r(0, 12))))
    nf = 1 << r(1, 12)
    cases.append(("fft_many_rows", lambda: T.test_dsp_fft_many_rows(nf, log_uniform(1, max(1, (1 << 22) // nf)), tune)))
    cases.append(("modulators", T.test_am_fm_modulator_kernels_vs_oracle))
    cases.append(("psk_modulators", T.test_psk_modulators_vs_oracle))
    cases.append(("fm_convolve", T.test_fm_convolve_vs_oracle_and_demodulate_identity))
    cases.append(("deframe", lambda: T.test_deframe_vs_oracle(bool(r(0, 1)))))
    cases.append(("fsk_rows", lambda: T.test_fm_demod_bits_per_wave_forms_match_oracle(log_uniform(64, 3000), tune)))
    for name, fn in cases:
        try:
            fn()
        except BaseException:
            print("FAILED in", name, flush=True)
            raise
        finally:
            for sw in ("SDRX_IIR_CW", "SDRX_IIR_PAIR", "SDRX_FSK_ROWS", "SDRX_FFT_NT", "SDRX_FFT_CT"):
                _lib.lib().sdrx_tune(sw.encode(), None)


t_end = time.time() + float(sys.argv[1] if len(sys.argv) > 1 else 60)
k = 0
while time.time() < t_end:
    k += 1
    np.random.default_rng = lambda s=None, _k=k: orig(None if s is None else (int(s) + 7919 * _k) % (1 << 63))
    one_round()
    print("round", k, "ok", flush=True)
print("soak_misc: %d rounds, every case bit-identical" % k)
