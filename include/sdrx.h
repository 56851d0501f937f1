/*
 * sdrx.h -- C ABI of libsdrx.so: the MI355X (gfx950) implementation of RadioNode's
 * baseband DSP hot path (FIR Filter + AM / FM / BPSK / QPSK modulators, demodulators), and of
 * the steps on either side of it (raw-byte ingest in front; packet deframing and the ones/zeros
 * confidence walk behind; rtl_fm's full_demod; the DigitalSignalProcessing primitives).
 *
 * Plain pointers and sizes only; no C++ or torch types.  Three layers:
 *   1. device layer  (sdrx_*_dev)  : device pointers + a hipStream_t (void*), asynchronous,
 *                                    nothing allocated per call.  This is what bench.py times.
 *   2. host layer    (sdrx_fir_*, sdrx_am_*, sdrx_fm_*, sdrx_psk_*) : host pointers, the
 *                                    reference's class methods one-to-one; pinned staging +
 *                                    hipMemcpyAsync on a per-object stream; synchronous return.
 *                                    include/sdrx/ (C++ headers) wraps these in the reference's class names.
 *   3. legacy layer  (am_gpu_*, fm_gpu_*, ones_zeros_demodulation) : the reference's own
 *                                    extern "C" GPU entry points, byte-for-byte signatures.
 *
 * Every entry point cites the reference interface it replaces.  By default all arithmetic is
 * IEEE fp64 with the reference's operation order and no FMA contraction, so filtered samples,
 * bits and symbol decisions are bit-identical to RadioNode's CPU path (see DESIGN.md for the two
 * transcendental exceptions: device sincos/atan2 are include/sdrx_math.h).  Opt-in variants
 * (sdrx_fir_job.fast_fma, .dtype = SDRX_F32) trade that for speed within stated tolerances.
 *
 * Return convention for new symbols: 0 = SDRX_OK, negative = error (sdrx_last_error()).
 */
#ifndef SDRX_H
#define SDRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRX_OK 0
#define SDRX_EINVAL (-1)      /* bad argument (null pointer, zero taps, ...) */
#define SDRX_EHIP (-2)        /* a HIP runtime call failed */
#define SDRX_EUNSUPPORTED (-3) /* shape outside the kernels' limits */
#define SDRX_ENOMEM (-4)

/* sample type of a device-layer FIR job */
#define SDRX_F64 0
#define SDRX_F32 1

/* sample layouts at the device boundary */
#define SDRX_REAL 0           /* one rail: in0 = T[n] */
#define SDRX_IQ_SPLIT 1       /* in0 = re[n], in1 = im[n]  (AM.h:63, FM.h:112, rtlsdr_receiver.cpp:1994-1995) */
#define SDRX_IQ_INTERLEAVED 2 /* in0 = {re,im}[n] = std::complex<double>[n] (Common_Deffinitions.h:12-13, BBP_Block.h:14) */
#define SDRX_IQ_U8 3          /* in0 = uint8_t {I,Q}[n]: the RTL-SDR wire format, converted on load (device layer only) */
#define SDRX_U8_SIGNED_CHAR 1 /* read bytes as signed char, like Baseband_Stream::load_buffer(char*, ...) does on x86 */
#define SDRX_U8_ROTATE90 2    /* apply rotate_90 (Baseband_Stream.cpp:82-101, rtl_fm.c:282-301) before converting */

/* Filter::filterType, filt.h:131 */
#define SDRX_LPF 0
#define SDRX_HPF 1
#define SDRX_BPF 2
/* filt.h:121 */
#define SDRX_REF_MAX_TAPS 1000

#define SDRX_BPSK 0
#define SDRX_QPSK 1

/* ------------------------------------------------------------------ library */
int sdrx_init(int device);           /* hipSetDevice + constants; idempotent per device */
int sdrx_device_count(void);
const char *sdrx_last_error(void);   /* thread-local message of the last failure */
const char *sdrx_version(void);
/* Experiment switches (DESIGN.md "Experiment switches"): the SDRX_* environment variables are read once, when the
 * library is first used; sdrx_tune(name, value) sets one afterwards (value NULL or "" = its default).  None of them
 * changes results.  Not to be called while launches are in flight on other threads. */
int sdrx_tune(const char *name, const char *value);
/* The kernel (with its template arguments) the last launch on this thread picked: bench.py's roofline.kernel. */
const char *sdrx_last_kernel(void);

/* ------------------------------------------------------------------ 1. device layer */

/*
 * One FIR (+ fused epilogue) job over `nchan` independent channels.
 *   y[c][m] = sum_{k=0}^{T-1} x[c][m*D - k] * h[k]              (k ascending: filt.cpp:261-262)
 * x[c][i<0] comes from the channel's carried history (Filter's shift register, filt.cpp:256-259).
 * State per channel (doubles): hist0[T-1] (x[-1], x[-2], ...), hist1[T-1] (second rail), pre[2].
 * state_in and state_out must be different buffers (a launch reads one and writes the other);
 * state_pitch >= 2*(T-1)+2.  Outputs never alias inputs.  sdrx_fir_dev only, modes 0 and 1: state_in = state_out = NULL
 * is a one-shot call: zero history, no state kept (what decode_pulseShape does on every call, BPSK.cpp:267-268).
 */
typedef struct sdrx_fir_job {
    int layout;               /* SDRX_REAL / SDRX_IQ_SPLIT / SDRX_IQ_INTERLEAVED / SDRX_IQ_U8 */
    int fast_fma;             /* 0: multiply then add, two roundings, like the CPU reference (bit-exact with it);
                                 1: acc = fma(x, h, acc), k ascending (<= 1e-12 of sum|h||x| from mode 0);
                                 2: the same fma chain on the fp64 matrix cores (sdrx_fir_dev, decim 1,
                                    num_taps >= 32; other shapes run as mode 1).  Bit-identical to mode 1
                                    for finite inputs */
    const void *in0, *in1;    /* device */
    size_t n;                 /* input samples per channel */
    size_t nchan;
    size_t in_pitch;          /* samples between consecutive channels */
    const void *taps;         /* device, num_taps samples of the job's dtype */
    int num_taps;
    unsigned decim;           /* D >= 1 */
    const void *state_in;     /* device, nchan * state_pitch samples of the job's dtype */
    void *state_out;
    size_t state_pitch;
    void *out0, *out1;        /* device; meaning depends on the entry point */
    size_t out_pitch;         /* output elements between channels */
    size_t n_out;             /* 0 = ceil(n / D); otherwise exactly this many outputs per channel */
    void *stream;             /* hipStream_t, NULL = default stream */
    int dtype;                /* SDRX_F64 (0, default): the reference's fp64; SDRX_F32: the fp32 twins -- float samples,
                                 taps, state and outputs, fused multiply-add, checked to 1e-5 relative */
    int u8_flags;             /* SDRX_IQ_U8 only: SDRX_U8_SIGNED_CHAR | SDRX_U8_ROTATE90 */
    double u8_bias;           /* SDRX_IQ_U8 only: x = (T)byte - u8_bias (127: rtlsdr_receiver.cpp:1587-1588, rtl_fm.c:812-815;
                                 127.4 with SDRX_U8_SIGNED_CHAR: Baseband_Stream.cpp:128-148) */
    size_t n_before;          /* 0 (default): x[c][i < 0] is the carried state.  Otherwise the block is a slice of a longer
                                 resident buffer: n_before >= num_taps - 1 (+ decim for sdrx_fir_fmdisc_dev) samples of the
                                 same channel lie in front of in0 and ARE the history (state_in is not read): a stream
                                 tiled in time, every tile its own launch, gives the one-launch results */
    const void *taps_f32;     /* optional (NULL = absent), sdrx_fir_am_dev with fp64 samples, decim 1, fast_fma 0: device, the
                                 num_taps taps rounded to float.  With it the chain filters in fp32 and VERIFIES every
                                 decision against a rigorous error bound; outputs inside the bound are recomputed in the
                                 reference's fp64 order, so the bits are those of the exact path for every input (a table
                                 that is not the rounded taps is detected on the device and only costs time) */
    void *am_work;            /* optional (NULL = absent), with taps_f32: device, SDRX_FIR_AM_WORK_BYTES the library may write
                                 during the call (one buffer per concurrently running job).  With it, tiles whose samples
                                 are 8-bit integers (a radio's) are filtered on the int8 matrix cores with fixed-point taps
                                 the call derives from `taps`, verified and recomputed like the fp32 form: the same bits */
} sdrx_fir_job;
#define SDRX_FIR_AM_WORK_BYTES 16384

/* Filter::execute (filt.cpp:267-273) on 1 rail (out0) or on I and Q (out0, out1), optionally
 * keeping every D-th output (decimate<T>, Resample.h:33-52).  Output type double. */
int sdrx_fir_dev(const sdrx_fir_job *job);
/* Filter::execute on I and Q, then AM<T>::demodulate with downsample = decim (AM.h:63-77):
 * out0 = uint8 0/1 per kept sample. */
int sdrx_fir_am_dev(const sdrx_fir_job *job);
/* Debug counters of the verified form of sdrx_fir_am_dev (job.taps_f32), kept per GPU while
 * sdrx_tune("SDRX_AMV_STATS", "1"): out[0] outputs recomputed in fp64, out[1] waves that recomputed any, out[2] tiles
 * whose bound was unusable (all outputs recomputed), out[3] unused.  Synchronises the device.  No counterpart in the
 * reference: there to report the fallback rate next to a measurement. */
int sdrx_fir_am_verify_stats(uint64_t out[4], int reset);
/* Filter::execute on I and Q, decimate by D, phase-difference discriminator
 * (rtl_fm.c:422-435, 518-546 in fp64): out0 = double angle (rad) per kept sample. */
int sdrx_fir_fmdisc_dev(const sdrx_fir_job *job);
size_t sdrx_fir_state_doubles(int num_taps); /* 2*(T-1)+2 */

/*
 * One radio delivers its stream in blocks (rtlsdr_receiver.cpp:1974-2116: one block per call), and a block of 2^18
 * samples is a microsecond of GPU work behind several microseconds of launch: launch-bound.  A ring processes `chunks`
 * consecutive chunks of ONE stream per launch call -- chunk k reads in0 + k * in_stride bytes, writes out0 + k * out_stride
 * bytes, the Filter / discriminator state carried from chunk to chunk through the job's two state buffers and from
 * call to call.  use_graph = 1 (recommended): chunks that abut in memory (strides = chunk sizes, n a multiple of decim)
 * are filtered by ONE launch over the whole ring -- a ring of chunks of one stream is one longer block of it --, other
 * rings replay a HIP graph captured once at creation; 2: always the graph; 0: the chunk launches issued one by one.
 * Measured (cfg2 chain, 2^18-sample chunks): 5.8 us per chunk one by one, 5.6 us from the graph (the cost is the GPU's
 * own latency between dependent launches), 0.78 us in one launch.  epilogue: 0 sdrx_fir_dev, 1 sdrx_fir_am_dev,
 * 2 sdrx_fir_fmdisc_dev.  Results are those of the plain calls, bit for bit.  sdrx_fir_ring_state(): which of the two
 * state buffers holds the stream's state now.
 */
typedef struct sdrx_fir_ring sdrx_fir_ring_t;
sdrx_fir_ring_t *sdrx_fir_ring_create(const sdrx_fir_job *chunk0, int epilogue, size_t chunks, size_t in_stride,
                                      size_t out_stride, int use_graph);
int sdrx_fir_ring_launch(sdrx_fir_ring_t *r, void *stream);
const void *sdrx_fir_ring_state(const sdrx_fir_ring_t *r);
void sdrx_fir_ring_destroy(sdrx_fir_ring_t *r);

/* AM<T>::demodulate (AM.h:63-77) / kernel_am_demod (am.cu:72-81): out[k] = |x[k*ds]| >= 80 */
int sdrx_am_demod_dev(const double *re, const double *im, uint8_t *out, size_t n, size_t ds, void *stream);
/* ones/zeros count (rtlsdr_receiver.cpp:2063-2074, ones_zeros.cu:28-52): *count (device int32, zeroed by the call) */
int sdrx_ones_zeros_dev(const double *re, const double *im, size_t n, int32_t *count, void *stream);
/* The same branch as the reference's CPU path runs it (rtlsdr_receiver.cpp:2063-2074): a confidence_counter walked
 * by increment(1) / decrement(1) per sample, clamped to [0, max] after every step (confidence_counter.h:50-72); nchan
 * independent streams (rows at `pitch`), counters[c] (device int32) read and written.  The walk is an ordered
 * reduction of clamp maps: exact, whatever the rails do.  With fewer than 256 streams of >= 262144 samples the streams
 * are cut into segments that share one per-GPU array of maps; the library orders such calls on the GPU with an event
 * (a later call waits for the earlier one's last kernel, whichever streams they are on), and runs the one-workgroup
 * form while `stream` is being captured. */
int sdrx_ones_zeros_confidence_dev(const double *re, const double *im, size_t n, size_t nchan, size_t pitch, int32_t max,
                                   int32_t *counters, void *stream);
/* FM<T>::demodulate (FM.h:83-148, repaired): w1/w2 device tables of ds doubles; out[n/ds] bits */
int sdrx_fm_demod_dev(const double *re, const double *w1, const double *w2, uint8_t *out, size_t n, size_t ds,
                      void *stream);

/* rtl_fm's own int16 chain on raw radio bytes, fused (third-party/RTLSDR/rtl_fm.c): rtlsdr_callback's
 * conversion (:808-822: optional rotate_90, (int16)byte - 127), low_pass (:303-323: boxcar sum of
 * `downsample` samples, narrowed to int16) and fm_demod with polar_discriminant (:422-435, :518-546):
 * pcm = (int)(atan2(Im, Re of z[m] conj z[m-1]) / 3.14159 * 2^14).  State per channel, 5 x int32:
 * now_r, now_j, prev_index (the partial boxcar sum), pre_r, pre_j; state_in != state_out.  Writes
 * (prev_index + n) / downsample pcm samples per channel (the caller tracks prev_index = (prev_index + n) % downsample). */
int sdrx_rtlfm_dev(const uint8_t *iq_u8, size_t n, size_t nchan, size_t in_pitch, int downsample, int rotate90,
                   const int32_t *state_in, int32_t *state_out, int16_t *pcm, size_t out_pitch, void *stream);
/* The rest of rtl_fm's full_demod (rtl_fm.c:730-780) around that chain, for downsample_passes == 0 and fm_demod:
 * power squelch between low_pass and fm_demod (rms, :650-667: below squelch_level the call's lowpassed[] is zeroed and
 * squelch_hits counts up, else it is reset), then on the pcm row low_pass_simple (:325-339, post_downsample > 1),
 * deemph_filter (:598-617, deemph_a > 0; rtl_fm.c:1221 computes deemph_a) and dc_block_filter (:619-632).
 * post_state, 4 x int32 per channel, read and written: deemph avg (a function static in rtl_fm), dc_avg, squelch_hits,
 * result_len (out: the number of pcm samples this call left).  work: nchan x 2 int64 of scratch (only for squelch_level
 * != 0).  With post_downsample = s > 1, out_pitch must hold ceil(m / s) * s and ceil(m / s) + 2 samples (the reference's
 * loop reads whole windows and writes one sample past its output), m = (prev_index + n) / downsample. */
typedef struct sdrx_rtlfm_params {
    int32_t downsample, rotate90, squelch_level, post_downsample, deemph_a, dc_block;
} sdrx_rtlfm_params;
int sdrx_rtlfm_full_dev(const sdrx_rtlfm_params *params, const uint8_t *iq_u8, size_t n, size_t nchan, size_t in_pitch,
                        const int32_t *state_in, int32_t *state_out, int32_t *post_state, int64_t *work, int16_t *pcm,
                        size_t out_pitch, void *stream);

/* low_pass_real (rtl_fm.c:341-361), full_demod's last stage when rate_out2 > 0 (:788-789): in place on the pcm rows
 * sdrx_rtlfm_full_dev left (row length = post_state[4c+3], which receives the new result_len), a boxcar over the
 * rate change rate_out -> rate_out2: result[k] = (int16)(window sum / (rate_out / rate_out2)).  lpr_state, 2 x int32
 * per channel, read and written: now_lpr, prev_lpr_index (zeros for a new demod_state).  0 < rate_out2 <= rate_out. */
int sdrx_rtlfm_low_pass_real_dev(int16_t *pcm, size_t nchan, size_t out_pitch, int rate_out, int rate_out2,
                                 int32_t *post_state, int32_t *lpr_state, void *stream);

/* The bit-domain back end of the receive path: what RtlSdr_Receiver::deserializeData (rtlsdr_receiver.cpp:2197-2230)
 * does with PacketFramer<T>::deserialize (network/PacketFramer.h:424-513), RingBuffer<T> and crc8 (crc/crc8.cpp), on
 * `nchan` independent bit streams (one byte per bit, as the demodulators write them).  The n bits of a call are the ring
 * buffer's content.  Per channel: events in the order the reference would emit them (at most max_events recorded, all
 * counted), the 250 data bytes of every valid packet in payloads[(chan * max_events + k) * 250], and
 * summary[4 * chan + {0,1,2}] = {events, bits consumed = where the ring buffer's front would stand, calls that
 * returned 0}: the caller feeds bits[consumed .. n) again in front of the next bits. */
typedef struct sdrx_frame_event {
    int64_t pos;      /* index of the unique word's first bit */
    int32_t kind;     /* 1 valid packet, 2 length > 250, 3 crc mismatch (the reference's return -1 cases) */
    uint8_t len, crc; /* the header's length and crc fields */
    uint8_t pad[2];
} sdrx_frame_event;
int sdrx_deframe_dev(const uint8_t *bits, size_t n, size_t nchan, size_t in_pitch, int crc_enable, sdrx_frame_event *events,
                     uint8_t *payloads, size_t max_events, uint64_t *summary, void *stream);

/* The reference's DigitalSignalProcessing primitives on `nchan` rows of n doubles (row c at c * pitch), in place or
 * x -> y; every row is computed in the reference's operation order (bit-identical to the CPU classes).
 *   SDRX_IIR_DC_BLOCK   DC_Filter::update (DC_Filter.cpp:12-19), coef = {R}; restarts at every call like the reference;
 *                       called in place (y == x) it is the reference's aliased call: input[i-1] is read after
 *                       output[i-1] overwrote it, y(n) = x(n) - y(n-1) + R*y(n-1)
 *   SDRX_IIR_LOWPASS    LowPassFilter::update per sample (LowPassFilter.cpp:13-16), coef = {ePow = sdrx_lowpass1_coef()},
 *                       state[4c] = `output`
 *   SDRX_IIR_NOTCH      NotchFilter::filter (NotchFilter.cpp:18-37), coef = sdrx_notch_design(); state[4c..4c+3] =
 *                       x_1, x_2, y_1, y_2 (function statics in the reference, per channel here); the first two samples
 *                       of every call pass through and do not enter the delay line
 *   SDRX_IIR_INTEGRATE  FirstOrderDigitalIntegration::integrate (:12-18), coef = {tStep = 1.0f / sampleRate}; restarts
 * coef is host memory; state is device memory, nchan x 4 doubles, read and written (NULL = zeros, not kept). */
enum { SDRX_IIR_DC_BLOCK = 0, SDRX_IIR_LOWPASS = 1, SDRX_IIR_NOTCH = 2, SDRX_IIR_INTEGRATE = 3 };
int sdrx_iir_dev(int kind, const double *coef, const double *x, double *y, size_t n, size_t nchan, size_t pitch,
                 double *state, void *stream);
double sdrx_lowpass1_coef(double cutoff_frequency, double delta_time);          /* LowPassFilter.cpp:7-11 */
void sdrx_notch_design(double notch_bw, double center_freq, double *coef5);      /* NotchFilter.cpp:7-16: a0 a1 a2 b1 b2 */
/* normalize (Normalize.cpp:7-22; the BBP_Block overload :24-43 is the same over the 2n doubles of the points):
 * max_inout[c] holds prev_max on entry and the returned maximum (an integer value: the reference keeps it in a size_t)
 * on exit; x = (x / max) * 2 - 1. */
int sdrx_normalize_dev(double *x, size_t n, size_t nchan, size_t pitch, double *max_inout, void *stream);
/* fifth_order_filter::decimate(double*, size_t, size_t) (fifth_order_filter.cpp:65-88), in place, log2(decimate_by)
 * passes over the same n like the reference; state6[6c..] = r_a..r_f carried (zeros for a new filter).  The reference
 * returns n / decimate_by; samples beyond the last written one keep their values. */
int sdrx_fifth_order_decimate_dev(double *data, size_t n, size_t nchan, size_t pitch, size_t decimate_by, double *state6,
                                  void *stream);
/* fifth_order_filter::decimate(BBP_Block*, size_t) (fifth_order_filter.cpp:28-63): rows of interleaved complex points
 * (pitch in points, at least max(n_points, 3)); both rails, state12[12c..] = r_a..r_f, i_a..i_f; n_points shrinks to
 * dst + 1 after every pass like block->n_points, the final value goes to *n_points_out (host). */
int sdrx_fifth_order_decimate_block_dev(double *iq, size_t n_points, size_t nchan, size_t pitch, size_t decimate_by,
                                        double *state12, size_t *n_points_out, void *stream);
/* fft (FFT.cpp:7-28): in-place radix-2 transforms of nchan rows of n complex doubles (pitch in complex elements),
 * n any power of two up to 2^SDRX_FFT_MAX_LOG2_ANY, like the reference's recursion; twiddles from sdrx_math.h's sincos
 * (<= 1e-12 of the libm build).  Up to SDRX_FFT_MAX points a transform runs in LDS (one launch for all rows); longer
 * ones as permutation + 4096-point blocks in LDS + the upper stages from global memory (launches per row).
 * The first call for a length <= SDRX_FFT_MAX on a device allocates that length's twiddle table: sdrx_init() makes all
 * of them, so a call under stream capture needs sdrx_init() (or one uncaptured call) before it. */
#define SDRX_FFT_MAX 4096
#define SDRX_FFT_MAX_LOG2_ANY 28
int sdrx_fft_dev(double *iq, size_t n, size_t nchan, size_t pitch, void *stream);

/* AM<T>::modulate (AM.h:41-54) = raw_upsample (Resample.h:15-27) over nchan bit streams (one byte per bit): every bit
 * repeated floor(len_mod / len) times as a double; outputs past len * floor(len_mod / len) are left untouched. */
int sdrx_am_modulate_dev(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, double *out_real, size_t len_mod,
                         size_t out_pitch, void *stream);
/* FM<T>::modulate (FM.h:44-74) over nchan bit streams: bit != 0 copies w_mark[0 .. spb), else w_space (device tables of
 * spb doubles, e.g. from sdrx_fm_tables(..., for_modulator = 1)) to out_real[i * spb ...]. */
int sdrx_fm_modulate_dev(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, const double *w_mark,
                         const double *w_space, size_t spb, double *out_real, size_t out_pitch, void *stream);
/* FM<T>::convolve (FM.h:83-108 with the `ds` parameter of the repair): for k = 0, 1, ... while
 * i = ds-1 + k*ds < len1: out[k] += sum_{j=0}^{len2-1} x[i-j] * w[len2-1-j] (x = 0 before its start), j ascending. */
int sdrx_fm_convolve_dev(const double *x, size_t len1, const double *w, size_t len2, double *out, size_t ds, void *stream);
/* The mark / space cosine tables as the reference builds them on the host (libm): for_modulator = 1: FM.h:47-57
 * (sqrt(2)/sqrt(n) * cos(2 PI f dt i)); 0: FM.h:127-131 / fm.cu:62-68 (sqrt(2/n) * cos(2 PI f i dt)). */
void sdrx_fm_tables(double sample_rate, double mark, double space, size_t n, int for_modulator, double *w_mark,
                    double *w_space);
/* BPSK::RRC_COEFF / QPSK::RRC_COEFF (BPSK.cpp:18-50, QPSK.cpp:32-64): the 129 root-raised-cosine taps (host table). */
const double *sdrx_rrc_coeff(void);
/* QPSK::DEFAULT_UNIQUE_WORD (QPSK.h:30, QPSK.cpp:20-28): 64 symbols, the constructor's default argument. */
const double *sdrx_qpsk_default_unique_word(void);

/* PSK loop state of one channel: the reference's public loop structs (BPSK.h:130-276) flattened. */
typedef struct sdrx_psk_state {
    double nco_delay, reg_mu, reg_Q, reg_I, mu_delay; /* NcoControlVariables.delay, Register_*, muDelay */
    double iq_d[5], ii_d[5];                           /* Interp_Q_Channel, Interp_I_Channel delay1..5 */
    double g1, g2, g3, g4;                             /* GardnerDetectorVariables.delay1..4 (start at 1) */
    double lf_x, lf_y;                                 /* LoopFilterVariables */
    double dds_delay, plf_x, plf_y;                    /* PhaseCorrectionVariables.dds_delay, PhaseLoopFilterVariables */
    int32_t cc[4];                                     /* confidence counters: BPSK cc[0]; QPSK 000/090/180/270 */
    int32_t uw_size;
    int32_t n_hits;                                    /* UW hits found by the last ambiguity pass */
    double uw[4][64];                                  /* uw_0/uw_180 or uw_000/090/180/270 (BPSK.h:307-308) */
} sdrx_psk_state;

void sdrx_psk_state_init_bpsk(sdrx_psk_state *s, uint64_t unique_word, size_t uw_len); /* BPSK.cpp:52-90 (host) */
void sdrx_psk_state_init_qpsk(sdrx_psk_state *s, const double *uw64, size_t uw_len);   /* QPSK.cpp:66-120 (host) */

/*
 * BPSK::demodulate / QPSK::demodulate (BPSK.cpp:148-176, QPSK.cpp:190-219) over nchan channels.
 *   iq        device, interleaved complex fp64, channel c at iq + 2*c*in_pitch doubles, n samples each
 *   states    device, nchan sdrx_psk_state (read and updated in place)
 *   out       device uint8: decisions 0/1 (BPSK) or 0..3 (QPSK), n/8 per channel, pitch out_pitch
 *   work      device workspace of sdrx_psk_work_bytes(n, nchan) bytes
 * Banks of few channels with long blocks (nchan <= 8192 and n >= 8192) are cut in time into pieces whose matched
 * filter runs beside the previous piece's loops on a side stream of the library (created on the first such call on a
 * device; ordered with `stream` by events, so the call still behaves as one operation of `stream`; results are those
 * of the one-stream form bit for bit).  sdrx_tune("SDRX_PSK_OVERLAP", "0") keeps everything on `stream`; a caller that
 * captures `stream` into a HIP graph should set that, or make one call before the capture begins.
 */
size_t sdrx_psk_work_bytes(size_t n, size_t nchan);
int sdrx_psk_demod_dev(int kind, const double *iq, size_t n, size_t nchan, size_t in_pitch,
                       sdrx_psk_state *states, uint8_t *out, size_t out_pitch, void *work, void *stream);
/*
 * sdrx_psk_demod_dev for callers that cannot allocate and bring no workspace: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Same arguments and results as
 * sdrx_psk_demod_dev minus `work`: the matched-filter output and the filter's zero-history rows live in a static
 * device array of SDRX_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds until the process ends.  The
 * call makes no hipMalloc / hipFree (neither does any routine it calls; the array's per-device address and the matched
 * filter's tap table are looked up by sdrx_init, or by the first call on a device).  A call under HIP stream capture must
 * therefore be preceded, on that device, by sdrx_init() or by one uncaptured call (symbol lookups do not belong inside a
 * capture).  Shapes whose workspace -- sdrx_psk_work_bytes(n, nchan) -- exceeds
 * sdrx_psk_demod_noalloc_max_bytes() are rejected with SDRX_EUNSUPPORTED.  Every call on a GPU shares that GPU's
 * array: calls that overlap on one GPU (two streams, two threads) race; serialise them or use sdrx_psk_demod_dev.
 */
#define SDRX_NOALLOC_SCRATCH_BYTES (16u << 20)
size_t sdrx_psk_demod_noalloc_max_bytes(void);
int sdrx_psk_demod_noalloc_dev(int kind, const double *iq, size_t n, size_t nchan, size_t in_pitch,
                               sdrx_psk_state *states, uint8_t *out, size_t out_pitch, void *stream);
/* stage entry points (the reference keeps every stage public and unit-tested) */
int sdrx_psk_matched_filter_dev(const double *iq, size_t n, size_t nchan, size_t in_pitch, double *mf_iq,
                                size_t mf_pitch, void *stream); /* decode_pulseShape: n -> n/4 interleaved */
int sdrx_psk_loops_dev(int kind, double *mf_iq, size_t m, size_t nchan, size_t mf_pitch, sdrx_psk_state *states,
                       uint8_t *decisions, size_t out_pitch, int run_timing, int run_phase,
                       void *stream); /* timing (in place on mf_iq) + carrier loop + /2 */
int sdrx_psk_ambiguity_dev(int kind, uint8_t *decisions, size_t m, size_t nchan, size_t out_pitch,
                           sdrx_psk_state *states, void *stream);
/* encode_LUT + encode_pulseShape (BPSK.cpp:112-146, QPSK.cpp:144-188): sym uint8 -> 8*len samples per rail.
 * out_im NULL for BPSK.  zero_prefix: leading elements treated as zero, the rest accumulate onto prior contents. */
int sdrx_psk_modulate_dev(int kind, const uint8_t *sym, size_t len, double *out_re, double *out_im,
                          size_t zero_prefix, void *stream);

/* ------------------------------------------------------------------ 2. host layer (drop-in) */

typedef struct sdrx_fir sdrx_fir_t;

/* Filter::Filter validation + designLPF/HPF/BPF (filt.cpp:45-154), host arithmetic (libm sin). */
int sdrx_fir_design(int type, int two_freq, int num_taps, double Fs, double Fx, double Fu, int max_taps,
                    double *taps);
sdrx_fir_t *sdrx_fir_create(int type, int two_freq, int num_taps, double Fs, double Fx, double Fu, int max_taps);
void sdrx_fir_destroy(sdrx_fir_t *f);                     /* Filter::~Filter */
int sdrx_fir_error_flag(const sdrx_fir_t *f);             /* Filter::get_error_flag, filt.h:156 */
void sdrx_fir_get_taps(const sdrx_fir_t *f, double *taps); /* filt.cpp:156-166 */
void sdrx_fir_reset(sdrx_fir_t *f);                       /* Filter::init, filt.cpp:236-246 */
int sdrx_fir_execute(sdrx_fir_t *f, double *buf, size_t n); /* Filter::execute, in place, state carried */
double sdrx_fir_do_sample(sdrx_fir_t *f, double x);       /* Filter::do_sample (one-sample launch) */
/* Filter::execute on nchan Filter objects at once (e.g. the I and Q rails of every radio): bufs[c] (n doubles each) is
 * filtered in place by h[c], every object's shift register carried as in sdrx_fir_execute.  All buffers go through one
 * pinned block and objects of the same design (identical taps) share ONE launch.  An object may appear only once. */
int sdrx_fir_execute_batch(sdrx_fir_t *const *h, double *const *bufs, size_t n, size_t nchan);

size_t sdrx_am_demodulate(const double *re, const double *im, uint8_t *out, size_t n, size_t ds); /* AM.h:63-77 */
void sdrx_am_modulate(const uint8_t *bits, size_t len, double *out_real, size_t len_mod);         /* AM.h:41-54 */
size_t sdrx_fm_demodulate(double rx_rate, double mark, double space, const double *re, uint8_t *out, size_t n,
                          size_t ds);                                                             /* FM.h:112-148 */
void sdrx_fm_modulate(double tx_rate, double mark, double space, const uint8_t *bits, size_t len,
                      double *out_real, size_t spb);                                              /* FM.h:44-74 */
void sdrx_fm_convolve(const double *input1, size_t length1, const double *input2, size_t length2, double *output,
                      size_t length_output, size_t ds);                                           /* FM.h:83-108 */
/* Chains of BASELINE configs 1 and 2 on host buffers (state in the two Filter-like objects). */
typedef struct sdrx_chain sdrx_chain_t;
sdrx_chain_t *sdrx_chain_create(const double *taps, int num_taps, unsigned decim);
void sdrx_chain_destroy(sdrx_chain_t *c);
void sdrx_chain_reset(sdrx_chain_t *c);
size_t sdrx_chain_fir_am(sdrx_chain_t *c, const double *re, const double *im, size_t n, uint8_t *bits);
size_t sdrx_chain_fir_fmdisc(sdrx_chain_t *c, const double *re, const double *im, size_t n, double *angle);
/* The same two chains fed with the radio's raw bytes (uint8 {I,Q} pairs as rtlsdr_read_async delivers them):
 * x = byte - bias on the device (bias 127: rtlsdr_receiver.cpp:1587-1588; 127.4 with SDRX_U8_SIGNED_CHAR:
 * Baseband_Stream.cpp:128-148), an eighth of the PCIe traffic of the fp64 entry points. */
size_t sdrx_chain_fir_am_u8(sdrx_chain_t *c, const uint8_t *iq_u8, size_t n, double bias, int flags, uint8_t *bits);
size_t sdrx_chain_fir_fmdisc_u8(sdrx_chain_t *c, const uint8_t *iq_u8, size_t n, double bias, int flags, double *angle);

typedef struct sdrx_psk sdrx_psk_t;
sdrx_psk_t *sdrx_psk_create(int kind, uint64_t bpsk_uw, const double *qpsk_uw64, size_t uw_len);
void sdrx_psk_destroy(sdrx_psk_t *p);
void sdrx_psk_get_state(sdrx_psk_t *p, sdrx_psk_state *out);
void sdrx_psk_set_state(sdrx_psk_t *p, const sdrx_psk_state *in);
/* BPSK::demodulate(BBP_Block*,BBP_Block*) / QPSK::demodulate: iq = n interleaved complex;
 * out[n/8] decisions as doubles (real part of the output block; imag is 0).  Returns the count.
 * hits (may be NULL, capacity max_hits): UW hits as (position << 3 | code) for the "Phase ..." prints. */
size_t sdrx_psk_demodulate(sdrx_psk_t *p, const double *iq, size_t n, double *out, uint64_t *hits, size_t max_hits);
/* BPSK::demodulate(Complex_Array&, uint8_t*, n, downsample, real_plot, imag_plot) (BPSK.cpp:178-211) after its
 * decimateComplexArray (done by the caller: include/sdrx/Resample.h): bits out, the timing loop's n/4 output samples
 * to plot_re / plot_im (either may be NULL).  Returns the number of bits. */
size_t sdrx_psk_demodulate_plot(sdrx_psk_t *p, const double *iq, size_t n, uint8_t *out_bits, double *plot_re,
                                double *plot_im, uint64_t *hits, size_t max_hits);
/* stage methods on host buffers (BPSK.cpp:213-487 / QPSK.cpp:222-509) */
size_t sdrx_psk_decode_pulse_shape(sdrx_psk_t *p, const double *iq, size_t n, double *re, double *im);
void sdrx_psk_timing(sdrx_psk_t *p, double *re, double *im, size_t n);
size_t sdrx_psk_phase_correction(sdrx_psk_t *p, double *re, double *im, size_t n);
size_t sdrx_psk_ambiguity(sdrx_psk_t *p, double *re, size_t n, uint64_t *hits, size_t max_hits);
/* BPSK::modulate / QPSK::modulate (out_imag NULL for BPSK); prior contents beyond len_mod/8 elements accumulate */
int sdrx_psk_modulate(int kind, const uint8_t *sym, size_t len, double *out_real, double *out_imag, size_t len_mod);

/* ------------------------------------------------------------------ 3. legacy layer */
/* am.cu:94-152 */
void am_gpu_initialize(size_t MAX_NUM_POINTS, size_t num_of_radios);
void am_gpu_free(void);
double am_gpu_demodulation(int RADIO_ID, double *real, double *imag, uint8_t *output, size_t number_of_points,
                           size_t ds);
/* fm.cu:48-167 (demodulation follows the CPU FM<T>::demodulate semantics: see DESIGN.md) */
void fm_gpu_initialize(size_t num_of_radios, size_t MaxBufferSize, size_t rx_sample_rate, size_t samples_per_bit,
                       double MARK_FREQ, double SPACE_FREQ);
void fm_gpu_free(void);
void fm_gpu_set_samples_per_bit(size_t samples_per_bit, size_t rx_sample_rate);
double fm_gpu_demodulation(int RADIO_ID, double *real, double *imag, uint8_t *output, size_t number_of_points,
                           size_t ds);
/* ones_zeros.cu:28-52 */
int32_t ones_zeros_demodulation(double *real, double *imag, size_t number_of_points);

#ifdef __cplusplus
}
#endif
#endif /* SDRX_H */
