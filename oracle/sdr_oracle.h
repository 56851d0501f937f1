/*
 * sdr_oracle.h -- CPU restatement of RadioNode's FIR + AM/FM/BPSK/QPSK hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (sdr_amd/, include/) may
 * link, load or call this: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, as the checker.  It is a from-scratch C restatement of
 * the reference's sample-at-a-time fp64 algorithms with the reference's exact
 * operation order; each function cites the reference file:line it follows.
 * Build with -O2 -ffp-contract=off (oracle/Makefile).
 *
 * Parity pinning: checked against every golden vector of the reference's own
 * unit tests (tests/golden/reference_vectors.npz, extracted from
 * RadioNode/src/test/TEST_*.cpp) and, in the build container, against the
 * reference classes compiled from /root/reference (oracle/_ref, see
 * oracle/Makefile and tests/test_oracle_vs_ref.py).
 */
#ifndef SDR_ORACLE_H
#define SDR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
/* DSP primitives: DigitalSignalProcessing/{DC_Filter,LowPassFilter,NotchFilter,
 * FirstOrderDigitalIntegration,Normalize,fifth_order_filter,FFT}.cpp */
void orc_dc_block(const double *in, double *out, size_t n, double R);            /* DC_Filter.cpp:12-19 */
double orc_lowpass1_coef(double cutoff, double dt);                              /* LowPassFilter.cpp:7-11 */
void orc_lowpass1(const double *in, double *out, size_t n, double ePow, double *state1); /* LowPassFilter.cpp:13-16, per sample */
void orc_notch_design(double notch_bw, double center_freq, double *coef5);       /* NotchFilter.cpp:7-16: a0,a1,a2,b1,b2 */
void orc_notch(double *block, size_t n, const double *coef5, double *state4);    /* NotchFilter.cpp:18-37; state = x_1,x_2,y_1,y_2 */
void orc_integrate(double *array, size_t n, double tStep);                       /* FirstOrderDigitalIntegration.cpp:12-18 */
double orc_normalize(double *block, size_t n, double prev_max);                  /* Normalize.cpp:7-22 */
size_t orc_fifth_order_decimate(double *data, size_t n, size_t decimate_by, double *state6); /* fifth_order_filter.cpp:65-88; state = r_a..r_f */
size_t orc_fifth_order_decimate_block(double *iq, size_t n_points, size_t decimate_by, double *state12); /* :28-63 */
void orc_fft(double *iq, size_t n, int math_mode);                               /* FFT.cpp:7-28, n a power of two, interleaved complex */

#endif

/* math_mode for the transcendental calls inside loops */
#define ORC_MATH_LIBM 0   /* libm cos/sin/atan2 -- what the reference calls */
#define ORC_MATH_SHARED 1 /* include/sdrx_math.h -- what the GPU path computes */

/* ---- Filter (FIR-filter-class/filt.cpp) ---- */
enum { ORC_LPF = 0, ORC_HPF = 1, ORC_BPF = 2 };

/* filt.cpp:45-102 validation + :110-154 design.  Returns the reference's
 * error flag (0, -1..-5, -10..-16).  two_freq != 0 selects the 5-argument
 * (BPF) constructor.  max_taps: 1000 = reference (filt.h:121); larger lifts the cap. */
int orc_fir_design(int type, int two_freq, int num_taps, double Fs, double Fx, double Fu,
                   int max_taps, double *taps);
/* filt.cpp:248-273: in-place, literal shift register sr[num_taps] carried across calls. */
void orc_fir_execute(const double *taps, int num_taps, double *sr, double *buf, size_t n);
/* Same arithmetic (same sums, same order) without the O(T) register shuffle; hist[T-1]
 * holds x[-1], x[-2], ... (newest first) and is updated.  Used for large-N checks. */
/* u8 I/Q ingest: Baseband_Stream.cpp:82-148, rtlsdr_receiver.cpp:1587-1588, rtl_fm.c:282-301,812-815 */
void orc_u8_rotate_90(uint8_t *buf, size_t len);
void orc_u8_to_iq(const uint8_t *buf, size_t n, double bias, int signed_char, double *re, double *im);
void orc_fir_execute_fast(const double *taps, int num_taps, double *hist, double *buf, size_t n);
/* fused variant (one rounding per term): the definition of the library's fast_fma modes, not of filt.cpp */
void orc_fir_execute_fma(const double *taps, int num_taps, double *hist, double *buf, size_t n);

/* ---- AM (streams/AM.h) ---- */
void orc_am_modulate(const uint8_t *bits, size_t len, double *out_real, size_t len_mod); /* AM.h:41-54 */
size_t orc_am_demodulate(const double *re, const double *im, uint8_t *out, size_t n, size_t ds); /* AM.h:63-77 */
/* rtlsdr_receiver.cpp:2063-2074: number of samples with envelope >= THRESHOLD */
int32_t orc_ones_zeros(const double *re, const double *im, size_t n);
int32_t orc_ones_zeros_confidence(const double *re, const double *im, size_t n, int32_t max, int32_t counter);

/* ---- FM / FSK (streams/FM.h, repaired semantics: SURVEY section 8c) ---- */
void orc_fm_modulate(double tx_rate, double mark, double space, const uint8_t *bits, size_t len,
                     double *out_real, size_t spb); /* FM.h:44-74 */
size_t orc_fm_demodulate(double rx_rate, double mark, double space, const double *re, uint8_t *out,
                         size_t n, size_t ds); /* FM.h:83-148 */
void orc_fm_waves(double rx_rate, double mark, double space, size_t ds, double *w1, double *w2); /* FM.h:125-131 */

/* ---- phase-difference discriminator (rtl_fm.c:422-435, 518-546), float restatement ----
 * pre[2] = previous sample (re, im), carried; out[i] = atan2(Im(z[i] conj z[i-1]), Re(..)) */
void orc_fm_discriminator(const double *re, const double *im, size_t n, double *pre, double *out,
                          int math_mode);
/* rtl_fm.c:428-435 on int16 I/Q pairs: (int)(angle / 3.14159 * (1<<14)) */
void orc_fm_discriminator_i16(const int16_t *iq, size_t n, int16_t *pre, int16_t *out, int math_mode);
/* rtl_fm.c:808-822 (u8 -> int16), :303-323 (boxcar low_pass), :518-546 (fm_demod); st = now_r, now_j, prev_index, pre_r, pre_j */
size_t orc_rtlfm_chain(const uint8_t *iq_u8, size_t n, int downsample, int rotate90, int32_t *st, int16_t *pcm,
                       int math_mode);

/* ---- Resample.{h,cpp} ---- */
size_t orc_rtlfm_full(const uint8_t *iq_u8, size_t n, int downsample, int rotate90, int squelch_level, int post_downsample,
                      int deemph_a, int dc_block, int32_t *st, int32_t *post, int16_t *pcm, size_t cap, int math_mode);
int orc_rtlfm_low_pass_real(int16_t *result, int result_len, int rate_out, int rate_out2, int32_t *lpr); /* rtl_fm.c:341-361 */
size_t orc_decimate(double *block, size_t length, size_t by);                                  /* Resample.h:33-52 */
size_t orc_upsample_fill_w_zeros(const double *in, size_t in_len, double *out, size_t out_len, double ratio); /* Resample.cpp:24-36 */
size_t orc_raw_upsample(const double *in, size_t in_len, double *out, size_t out_len, double ratio);          /* Resample.h:15-27 */
size_t orc_upsample(const double *in, size_t in_len, double *out, size_t out_len, double ratio);              /* Resample.cpp:10-22 */
size_t orc_decimate_block(double *iq, size_t n_points, size_t by);  /* decimate(BBP_Block*), Resample.cpp:38-48; returns n_points */
size_t orc_decimate_complex(double *iq, size_t length, size_t by);  /* decimateComplexArray, Resample.cpp:50-70 */
/* FM<T>::convolve, FM.h:83-108 (repaired): output[k] += ..., k = 0 .. while ds-1 + k*ds < length1 */
void orc_fm_convolve(const double *input1, size_t length1, const double *input2, size_t length2, double *output, size_t ds);

/* ---- composed chains (BASELINE configs 1 and 2/4) ---- */
/* Filter::execute on I and on Q (state srI/srQ carried) then AM::demodulate(ds). */
size_t orc_chain_fir_am(const double *taps, int T, double *histI, double *histQ, const double *re,
                        const double *im, size_t n, size_t ds, uint8_t *bits);
/* Filter::execute on I and Q, decimate<T>(.., D) (keeps 0,D,2D..), discriminator. Returns ceil(n/D). */
size_t orc_chain_fir_decim_disc(const double *taps, int T, double *histI, double *histQ, double *pre,
                                const double *re, const double *im, size_t n, size_t D, double *out,
                                int math_mode);

/* ---- PSK (streams/BPSK.cpp, QPSK.cpp) ---- */
typedef struct {
    double d1, d2, d3, d4, d5;
} orc_interp; /* BPSK.h:148-165 */

typedef struct {
    /* timing loop, BPSK.h:130-245 */
    double nco_delay, reg_mu, reg_Q, reg_I, mu_delay;
    orc_interp interp_q, interp_i;
    double g1, g2, g3, g4; /* GardnerDetectorVariables.delay1..4 */
    double lf_x, lf_y;
    /* carrier loop, BPSK.h:249-276 */
    double dds_delay, plf_x, plf_y;
    /* ambiguity, confidence_counter(10,5,5): BPSK uses cc[0]; QPSK 000,090,180,270 */
    int32_t cc[4];
    int32_t kind; /* 0 BPSK, 1 QPSK */
    int32_t uw_size;
    int32_t math_mode;
    int32_t pad_;
    double uw[4][64];
} orc_psk;

#define ORC_BPSK 0
#define ORC_QPSK 1

extern const double ORC_RRC[129]; /* BPSK.cpp:18-50 == QPSK.cpp:32-64 */

void orc_bpsk_init(orc_psk *s, uint64_t unique_word, size_t uw_len, int math_mode); /* BPSK.cpp:52-90 */
void orc_qpsk_init(orc_psk *s, const double *unique_word64, size_t uw_len, int math_mode); /* QPSK.cpp:66-120; NULL = DEFAULT_UNIQUE_WORD */

/* BPSK.cpp:112-146 / QPSK.cpp:144-188.  sym: LUT-encoded +-1 rails (imag NULL for BPSK).
 * out_* hold prior contents; the first zero_prefix elements are treated as zeroed
 * (the reference's memset(out, 0, output_length) clears output_length BYTES = output_length/8 doubles). */
void orc_psk_encode_lut(int kind, const uint8_t *data, size_t len, double *re, double *im);
void orc_psk_pulse_shape(const double *sym, size_t len, double *out, size_t zero_prefix);
/* matched filter + keep 0,4,8..: returns n/4.  BPSK.cpp:257-300 / QPSK.cpp:222-262.  iq interleaved complex. */
size_t orc_psk_decode_pulse_shape(const double *iq, size_t n, double *re, double *im);
void orc_psk_timing(orc_psk *s, double *re, double *im, size_t n);           /* BPSK.cpp:302-389 */
size_t orc_psk_phase_correction(orc_psk *s, double *re, double *im, size_t n); /* BPSK.cpp:393-444 / QPSK.cpp:355-414 */
/* BPSK.cpp:446-487 / QPSK.cpp:416-509.  events (may be NULL): per UW hit, (position<<3 | code). Returns hit count. */
size_t orc_psk_ambiguity(orc_psk *s, double *re, size_t n, uint64_t *events, size_t max_events);
/* BPSK.cpp:148-176 / QPSK.cpp:190-219: iq[n] interleaved -> decisions (0/1 or 0..3) as doubles; returns count.
 * timing_re/timing_im (may be NULL) receive the post-timing samples (n/4 each), as the plot buffers of BPSK.cpp:194-195. */
size_t orc_psk_demodulate(orc_psk *s, const double *iq, size_t n, double *out, double *timing_re,
                          double *timing_im);
/* full modulators: bits/symbols -> shaped rails, 8 samples per symbol */
void orc_bpsk_modulate(const uint8_t *bits, size_t len, double *out_real, size_t len_mod);               /* BPSK.cpp:96-108 */
void orc_qpsk_modulate(const uint8_t *sym, size_t len, double *out_real, double *out_imag, size_t len_mod); /* QPSK.cpp:127-142 */

#ifdef __cplusplus
}
/* DSP primitives: DigitalSignalProcessing/{DC_Filter,LowPassFilter,NotchFilter,
 * FirstOrderDigitalIntegration,Normalize,fifth_order_filter,FFT}.cpp */
void orc_dc_block(const double *in, double *out, size_t n, double R);            /* DC_Filter.cpp:12-19 */
double orc_lowpass1_coef(double cutoff, double dt);                              /* LowPassFilter.cpp:7-11 */
void orc_lowpass1(const double *in, double *out, size_t n, double ePow, double *state1); /* LowPassFilter.cpp:13-16, per sample */
void orc_notch_design(double notch_bw, double center_freq, double *coef5);       /* NotchFilter.cpp:7-16: a0,a1,a2,b1,b2 */
void orc_notch(double *block, size_t n, const double *coef5, double *state4);    /* NotchFilter.cpp:18-37; state = x_1,x_2,y_1,y_2 */
void orc_integrate(double *array, size_t n, double tStep);                       /* FirstOrderDigitalIntegration.cpp:12-18 */
double orc_normalize(double *block, size_t n, double prev_max);                  /* Normalize.cpp:7-22 */
size_t orc_fifth_order_decimate(double *data, size_t n, size_t decimate_by, double *state6); /* fifth_order_filter.cpp:65-88; state = r_a..r_f */
void orc_fft(double *iq, size_t n, int math_mode);                               /* FFT.cpp:7-28, n a power of two, interleaved complex */

#endif
/* bit-domain back end: crc/crc8.cpp:6-44, network/PacketFramer.h:363-513, common/RingBuffer.h,
 * RtlSdr_Receiver/rtlsdr_receiver.cpp:2197-2230 */
uint8_t orc_crc8(const uint8_t *message, size_t length);
void orc_deframe(const uint8_t *bits, size_t n, int crc_enable, size_t max_events, int64_t *ev_pos, int32_t *ev_kind,
                 uint8_t *ev_len, uint8_t *ev_crc, uint8_t *payload, uint64_t *summary);
int orc_frame(const uint8_t *data, size_t length, int crc_enable, uint8_t *bits_out);

/* DSP primitives: DigitalSignalProcessing/{DC_Filter,LowPassFilter,NotchFilter,
 * FirstOrderDigitalIntegration,Normalize,fifth_order_filter,FFT}.cpp */
void orc_dc_block(const double *in, double *out, size_t n, double R);            /* DC_Filter.cpp:12-19 */
double orc_lowpass1_coef(double cutoff, double dt);                              /* LowPassFilter.cpp:7-11 */
void orc_lowpass1(const double *in, double *out, size_t n, double ePow, double *state1); /* LowPassFilter.cpp:13-16, per sample */
void orc_notch_design(double notch_bw, double center_freq, double *coef5);       /* NotchFilter.cpp:7-16: a0,a1,a2,b1,b2 */
void orc_notch(double *block, size_t n, const double *coef5, double *state4);    /* NotchFilter.cpp:18-37; state = x_1,x_2,y_1,y_2 */
void orc_integrate(double *array, size_t n, double tStep);                       /* FirstOrderDigitalIntegration.cpp:12-18 */
double orc_normalize(double *block, size_t n, double prev_max);                  /* Normalize.cpp:7-22 */
size_t orc_fifth_order_decimate(double *data, size_t n, size_t decimate_by, double *state6); /* fifth_order_filter.cpp:65-88; state = r_a..r_f */
void orc_fft(double *iq, size_t n, int math_mode);                               /* FFT.cpp:7-28, n a power of two, interleaved complex */

/* DSP primitives: DigitalSignalProcessing/{DC_Filter,LowPassFilter,NotchFilter,
 * FirstOrderDigitalIntegration,Normalize,fifth_order_filter,FFT}.cpp */
void orc_dc_block(const double *in, double *out, size_t n, double R);            /* DC_Filter.cpp:12-19 */
double orc_lowpass1_coef(double cutoff, double dt);                              /* LowPassFilter.cpp:7-11 */
void orc_lowpass1(const double *in, double *out, size_t n, double ePow, double *state1); /* LowPassFilter.cpp:13-16, per sample */
void orc_notch_design(double notch_bw, double center_freq, double *coef5);       /* NotchFilter.cpp:7-16: a0,a1,a2,b1,b2 */
void orc_notch(double *block, size_t n, const double *coef5, double *state4);    /* NotchFilter.cpp:18-37; state = x_1,x_2,y_1,y_2 */
void orc_integrate(double *array, size_t n, double tStep);                       /* FirstOrderDigitalIntegration.cpp:12-18 */
double orc_normalize(double *block, size_t n, double prev_max);                  /* Normalize.cpp:7-22 */
size_t orc_fifth_order_decimate(double *data, size_t n, size_t decimate_by, double *state6); /* fifth_order_filter.cpp:65-88; state = r_a..r_f */
void orc_fft(double *iq, size_t n, int math_mode);                               /* FFT.cpp:7-28, n a power of two, interleaved complex */

/* include/sdrx_math.h's sincos forms over an array: form 0 = sdrx_sincos, 1 = sdrx_sincos_tab (the PSK carrier loop's) */
void orc_sincos(const double *x, size_t n, int form, double *sn, double *cs);

#endif
