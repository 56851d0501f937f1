/*
 * sdr_oracle.c -- CPU restatement of the reference hot path (see sdr_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product.
 *
 * All arithmetic is IEEE fp64, one rounding per operation, in the reference's
 * source order (the reference's unit-test build is -O0, RadioNode/CMakeLists.txt:8-10),
 * so compile with -ffp-contract=off.
 *
 * Pinned by (a) every golden vector of the reference's own unit tests (tests/golden/reference_vectors.npz,
 * tests/test_oracle_golden.py), (b) the reference's classes compiled from /root/reference (oracle/_ref,
 * tests/test_oracle_vs_ref.py, build container only) and (c) vectors those compiled classes produced
 * (tests/golden/ref_fixtures.npz, tests/test_ref_fixtures.py, anywhere).  The u8 conversions (orc_u8_*) are pinned to
 * the reference's Baseband_Stream::load_buffer / load_buffer_and_rotate90 the same two ways.  Parity unpinned (no
 * reference vector and not buildable here: rtl_fm.c needs librtlsdr): rtl_fm's chain (orc_fm_discriminator*,
 * orc_rtlfm_*), checked by its defining properties only.
 */
#include "sdr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sdrx_math.h"

static const double ORC_PI = 3.141592653589793238460; /* Common_Deffinitions.h:14 */
#define SGN(v) (((v) == 0.0) ? 0.0 : (((v) < 0.0) ? -1.0 : 1.0)) /* BPSK.h:16 */

/* ------------------------------------------------------------------ Filter */

int orc_fir_design(int type, int two_freq, int num_taps, double Fs, double Fx, double Fu,
                   int max_taps, double *taps)
{
    double lambda = M_PI * Fx / (Fs / 2); /* filt.cpp:52 / :82 */
    if (!two_freq) {                      /* filt.cpp:45-70 */
        if (Fs <= 0) return -1;
        if (Fx <= 0 || Fx >= Fs / 2) return -2;
        if (num_taps <= 0 || num_taps > max_taps) return -3;
        if (type != ORC_LPF && type != ORC_HPF) return -5;
        for (int n = 0; n < num_taps; n++) {
            double mm = n - (num_taps - 1.0) / 2.0;
            if (type == ORC_LPF) { /* filt.cpp:116-120 */
                if (mm == 0.0) taps[n] = lambda / M_PI;
                else taps[n] = sin(mm * lambda) / (mm * M_PI);
            } else { /* filt.cpp:131-135 */
                if (mm == 0.0) taps[n] = 1.0 - lambda / M_PI;
                else taps[n] = -sin(mm * lambda) / (mm * M_PI);
            }
        }
        return 0;
    }
    /* filt.cpp:73-102 */
    double phi = M_PI * Fu / (Fs / 2);
    if (Fs <= 0) return -10;
    if (Fx >= Fu) return -11;
    if (Fx <= 0 || Fx >= Fs / 2) return -12;
    if (Fu <= 0 || Fu >= Fs / 2) return -13;
    if (num_taps <= 0 || num_taps > max_taps) return -14;
    if (type != ORC_BPF) return -16;
    for (int n = 0; n < num_taps; n++) { /* filt.cpp:146-151 */
        double mm = n - (num_taps - 1.0) / 2.0;
        if (mm == 0.0) taps[n] = (phi - lambda) / M_PI;
        else taps[n] = (sin(mm * phi) - sin(mm * lambda)) / (mm * M_PI);
    }
    return 0;
}

void orc_fir_execute(const double *taps, int T, double *sr, double *buf, size_t n)
{
    for (size_t s = 0; s < n; s++) { /* filt.cpp:267-273 */
        for (int i = T - 1; i >= 1; i--) sr[i] = sr[i - 1]; /* filt.cpp:256-258 */
        sr[0] = buf[s];
        double result = 0;
        for (int i = 0; i < T; i++) result += sr[i] * taps[i]; /* filt.cpp:262 */
        buf[s] = result;
    }
}

void orc_fir_execute_fast(const double *taps, int T, double *hist, double *buf, size_t n)
{
    /* x laid out oldest..newest: [x[-(T-1)] .. x[-1] | x[0] .. x[n-1]] */
    size_t H = (size_t)(T - 1);
    double *x = (double *)malloc((H + n) * sizeof(double));
    for (size_t k = 0; k < H; k++) x[H - 1 - k] = hist[k];
    memcpy(x + H, buf, n * sizeof(double));
    for (size_t s = 0; s < n; s++) {
        const double *p = x + H + s;
        double result = 0;
        for (int i = 0; i < T; i++) result += p[-i] * taps[i];
        buf[s] = result;
    }
    for (size_t k = 0; k < H; k++) hist[k] = x[H + n - 1 - k];
    free(x);
}

/* The same sum with one rounding per term: result = fma(x[n-k], h[k], result), k ascending.  NOT what
 * filt.cpp computes (it rounds the product and the sum separately); this is the definition of the
 * library's opt-in fast modes (sdrx_fir_job.fast_fma = 1: VALU fma chain, 2: fp64 MFMA, which on gfx950
 * is the same chain), so that those modes can be checked bit for bit too. */
void orc_fir_execute_fma(const double *taps, int T, double *hist, double *buf, size_t n)
{
    size_t H = (size_t)(T - 1);
    double *x = (double *)malloc((H + n) * sizeof(double));
    for (size_t k = 0; k < H; k++) x[H - 1 - k] = hist[k];
    memcpy(x + H, buf, n * sizeof(double));
    for (size_t s = 0; s < n; s++) {
        const double *p = x + H + s;
        double result = 0;
        for (int i = 0; i < T; i++) result = fma(p[-i], taps[i], result);
        buf[s] = result;
    }
    for (size_t k = 0; k < H; k++) hist[k] = x[H + n - 1 - k];
    free(x);
}

/* --------------------------------------------------------- u8 I/Q ingest */

/* Baseband_Stream::rotate_90 (Baseband_Stream.cpp:82-101) == rtl_fm.c:282-301: in place, groups of 8 bytes
 * (4 samples): x1, xj, x(-1), x(-j) with 255 - byte as negation.  len = bytes, a multiple of 8. */
void orc_u8_rotate_90(uint8_t *buf, size_t len)
{
    for (size_t i = 0; i + 8 <= len; i += 8) {
        uint8_t tmp = (uint8_t)(255 - buf[i + 3]);
        buf[i + 3] = buf[i + 2];
        buf[i + 2] = tmp;
        buf[i + 4] = (uint8_t)(255 - buf[i + 4]);
        buf[i + 5] = (uint8_t)(255 - buf[i + 5]);
        tmp = (uint8_t)(255 - buf[i + 6]);
        buf[i + 6] = buf[i + 7];
        buf[i + 7] = tmp;
    }
}

/* bytes -> I/Q doubles.  signed_char = 0, bias 127: rtlsdr_receiver.cpp:1587-1588 (`buffer[i] - 127` on
 * unsigned bytes) and rtl_fm.c:812-815.  signed_char = 1, bias 127.4: Baseband_Stream::load_buffer
 * (Baseband_Stream.cpp:128-148), whose parameter is `char*`: on x86 bytes >= 128 read as negative. */
void orc_u8_to_iq(const uint8_t *buf, size_t n, double bias, int signed_char, double *re, double *im)
{
    for (size_t p = 0; p < n; p++) {
        double a = signed_char ? (double)(signed char)buf[2 * p] : (double)buf[2 * p];
        double b = signed_char ? (double)(signed char)buf[2 * p + 1] : (double)buf[2 * p + 1];
        re[p] = a - bias;
        im[p] = b - bias;
    }
}

/* ---------------------------------------------------------------------- AM */

size_t orc_raw_upsample(const double *in, size_t in_len, double *out, size_t out_len, double ratio)
{
    for (size_t oi = 0, ii = 0; oi < out_len && ii < in_len; ii++) /* Resample.h:18-24 */
        for (int j = 0; (j < floor(ratio)) && (oi < out_len); j++) out[oi++] = in[ii];
    return out_len;
}

void orc_am_modulate(const uint8_t *bits, size_t len, double *out_real, size_t len_mod)
{
    double *tmp = (double *)malloc((len ? len : 1) * sizeof(double)); /* AM.h:43-48 */
    for (size_t i = 0; i < len; i++) tmp[i] = bits[i];
    orc_raw_upsample(tmp, len, out_real, len_mod, (double)len_mod / (double)len); /* AM.h:51 */
    free(tmp);
}

size_t orc_am_demodulate(const double *re, const double *im, uint8_t *out, size_t n, size_t ds)
{
    size_t oi = 0;
    for (size_t i = 0; i < n; i += ds, oi++) /* AM.h:68-74 */
        out[oi] = ((sqrt(re[i] * re[i] + im[i] * im[i]) >= 80.0) ? 1.0 : 0.0);
    return oi;
}

int32_t orc_ones_zeros(const double *re, const double *im, size_t n)
{
    int32_t ones = 0;
    for (size_t i = 0; i < n; i++) /* rtlsdr_receiver.cpp:2063-2074 */
        if (sqrt(re[i] * re[i] + im[i] * im[i]) >= 80.0) ones++;
    return ones;
}

/* the same branch as the CPU path runs it: confidence_counter::increment(1) / decrement(1) per sample
 * (rtlsdr_receiver.cpp:2063-2074, confidence_counter.h:50-72); returns the counter */
int32_t orc_ones_zeros_confidence(const double *re, const double *im, size_t n, int32_t max, int32_t counter)
{
    for (size_t i = 0; i < n; i++) {
        if (sqrt(re[i] * re[i] + im[i] * im[i]) >= 80.0) {
            counter += 1;
            if (counter > max) counter = max;
        } else {
            counter -= 1;
            if (counter < 0) counter = 0;
        }
    }
    return counter;
}

/* ---------------------------------------------------------------------- FM */

void orc_fm_modulate(double tx_rate, double mark, double space, const uint8_t *bits, size_t len,
                     double *out_real, size_t spb)
{
    const double delta_t = 1.0 / tx_rate; /* FM.h:46-47 */
    const double root_raised = sqrt(2) / sqrt(spb);
    double *w1 = (double *)malloc((spb ? spb : 1) * sizeof(double));
    double *w2 = (double *)malloc((spb ? spb : 1) * sizeof(double));
    for (size_t i = 0; i < spb; i++) { /* FM.h:53-57 */
        w1[i] = root_raised * cos(2.0 * ORC_PI * mark * delta_t * i);
        w2[i] = root_raised * cos(2.0 * ORC_PI * space * delta_t * i);
    }
    for (size_t i = 0; i < len; i++) /* FM.h:59-70 */
        memcpy(&out_real[i * spb], bits[i] == 0 ? w2 : w1, spb * sizeof(double));
    free(w1);
    free(w2);
}

void orc_fm_waves(double rx_rate, double mark, double space, size_t ds, double *w1, double *w2)
{
    const double delta_t = 1.0 / rx_rate; /* FM.h:125 */
    for (size_t i = 0; i < ds; i++) {     /* FM.h:127-131 */
        w1[i] = sqrt(2.0 / ds) * cos(2.0 * ORC_PI * mark * i * delta_t);
        w2[i] = sqrt(2.0 / ds) * cos(2.0 * ORC_PI * space * i * delta_t);
    }
}

size_t orc_fm_demodulate(double rx_rate, double mark, double space, const double *re, uint8_t *out,
                         size_t n, size_t ds)
{
    size_t length = n / ds; /* FM.h:114 */
    double *w1 = (double *)malloc((ds ? ds : 1) * sizeof(double));
    double *w2 = (double *)malloc((ds ? ds : 1) * sizeof(double));
    orc_fm_waves(rx_rate, mark, space, ds, w1, w2);
    /* FM.h:94-100 with the 7th parameter ds restored: window [b*ds, b*ds+ds-1],
     * accumulated last sample first, accumulator starting at 0 */
    size_t b = 0;
    for (size_t i = ds - 1; i < n; i += ds, b++) {
        double op1 = 0, op2 = 0;
        for (size_t j = 0; j < ds; j++) op1 += re[i - j] * w1[ds - j - 1];
        for (size_t j = 0; j < ds; j++) op2 += re[i - j] * w2[ds - j - 1];
        out[b] = (op1 > op2) ? 1 : 0; /* FM.h:139 repaired */
    }
    free(w1);
    free(w2);
    return length;
}

void orc_fm_convolve(const double *input1, size_t length1, const double *input2, size_t length2, double *output,
                     size_t ds)
{
    /* FM.h:83-108 with the 7th parameter `ds` of the repair */
    double *real_data = (double *)calloc(length1 + 2 * length2 + 1, sizeof(double)); /* FM.h:85-86 */
    for (size_t i = 0; i < length1; i++) real_data[i + length2] = input1[i];          /* FM.h:89-92 */
    for (size_t i = ds - 1, oi = 0; i < length1; i += ds, oi++)                         /* FM.h:94 */
        for (size_t j = 0; j < length2; j++) output[oi] += real_data[i + length2 - j] * input2[length2 - j - 1]; /* FM.h:98 */
    free(real_data);
}

/* ----------------------------------------------------------- discriminator */

static double orc_atan2(double y, double x, int math_mode)
{
    return math_mode == ORC_MATH_SHARED ? sdrx_atan2(y, x) : atan2(y, x);
}

void orc_fm_discriminator(const double *re, const double *im, size_t n, double *pre, double *out,
                          int math_mode)
{
    double pr = pre[0], pj = pre[1];
    for (size_t i = 0; i < n; i++) {
        double ar = re[i], aj = im[i], br = pr, bj = -pj; /* rtl_fm.c:432 multiply(a, conj b) */
        double cr = ar * br - aj * bj;                    /* rtl_fm.c:424 */
        double cj = aj * br + ar * bj;                    /* rtl_fm.c:425 */
        out[i] = orc_atan2(cj, cr, math_mode);            /* rtl_fm.c:433 */
        pr = ar;
        pj = aj;
    }
    pre[0] = pr; /* rtl_fm.c:543-544 */
    pre[1] = pj;
}

void orc_fm_discriminator_i16(const int16_t *iq, size_t n, int16_t *pre, int16_t *out, int math_mode)
{
    int pr = pre[0], pj = pre[1];
    for (size_t i = 0; i < n; i++) {
        int ar = iq[2 * i], aj = iq[2 * i + 1], br = pr, bj = -pj;
        int cr = ar * br - aj * bj;
        int cj = aj * br + ar * bj;
        double angle = orc_atan2((double)cj, (double)cr, math_mode);
        out[i] = (int16_t)(int)(angle / 3.14159 * (1 << 14)); /* rtl_fm.c:434 */
        pr = ar;
        pj = aj;
    }
    pre[0] = (int16_t)pr;
    pre[1] = (int16_t)pj;
}

/* rtl_fm's own chain around the discriminator, restated on its int16 data path:
 * rtlsdr_callback (rtl_fm.c:808-822): optional rotate_90, buf16 = (int16)byte - 127;
 * low_pass (rtl_fm.c:303-323): boxcar sum of `downsample` samples, partial sum carried in now_r/now_j/prev_index,
 * stored back as int16; fm_demod (rtl_fm.c:518-546) with custom_atan == 0.  st = {now_r, now_j, prev_index, pre_r, pre_j}.
 * Returns the number of pcm samples written.  (rtl_fm.c needs librtlsdr and is not built here: parity unpinned.) */
/* rtlsdr_callback's conversion + low_pass (rtl_fm.c:808-822, 303-323): returns lowpassed[] (malloc'd, 2*m int16) */
static int16_t *rtlfm_lowpass_stage(const uint8_t *iq_u8, size_t n, int downsample, int rotate90, int32_t *st, size_t *m_out)
{
    size_t len = 2 * n;
    uint8_t *buf = (uint8_t *)malloc(len ? len : 1);
    int16_t *lp = (int16_t *)malloc((len ? len : 1) * sizeof(int16_t));
    memcpy(buf, iq_u8, len);
    if (rotate90) orc_u8_rotate_90(buf, len);
    for (size_t i = 0; i < len; i++) lp[i] = (int16_t)((int16_t)buf[i] - 127);
    int now_r = st[0], now_j = st[1], prev_index = st[2];
    size_t i = 0, i2 = 0;
    while (i < len) { /* low_pass */
        now_r += lp[i];
        now_j += lp[i + 1];
        i += 2;
        prev_index++;
        if (prev_index < downsample) continue;
        lp[i2] = (int16_t)now_r;
        lp[i2 + 1] = (int16_t)now_j;
        prev_index = 0;
        now_r = 0;
        now_j = 0;
        i2 += 2;
    }
    st[0] = now_r;
    st[1] = now_j;
    st[2] = prev_index;
    free(buf);
    *m_out = i2 / 2;
    return lp;
}

size_t orc_rtlfm_chain(const uint8_t *iq_u8, size_t n, int downsample, int rotate90, int32_t *st, int16_t *pcm,
                       int math_mode)
{
    size_t m;
    int16_t *lp = rtlfm_lowpass_stage(iq_u8, n, downsample, rotate90, st, &m);
    if (m) { /* fm_demod: with lp_len == 0 the reference would read lp[-2]; it is never called that way */
        int16_t pre[2] = {(int16_t)st[3], (int16_t)st[4]};
        orc_fm_discriminator_i16(lp, m, pre, pcm, math_mode);
        st[3] = pre[0];
        st[4] = pre[1];
    }
    free(lp);
    return m;
}

/* rms (rtl_fm.c:650-667) */
static int rtlfm_rms(const int16_t *samples, int len, int step)
{
    long p = 0, t = 0, s;
    for (int i = 0; i < len; i += step) {
        s = (long)samples[i];
        t += s;
        p += s * s;
    }
    double dc = (double)(t * step) / (double)len;
    double err = t * 2 * dc - dc * dc * len;
    double v = (p - err) / len;
    double r = sqrt(v);
    return r == r && r > -2147483648.0 ? (int)r : (-2147483647 - 1); /* (int)NaN on x86: INT_MIN */
}

/* full_demod (rtl_fm.c:730-780) on one callback's bytes, downsample_passes == 0, mode_demod == fm_demod:
 * low_pass -> power squelch (rms, :650-667) -> fm_demod -> low_pass_simple (:325-339) -> deemph_filter (:598-617)
 * -> dc_block_filter (:619-632).  post[4] = {deemph avg (a function static in rtl_fm), dc_avg, squelch_hits,
 * result_len}; pcm is result[] (capacity cap, contents beyond the written part are read by low_pass_simple when
 * result_len is not a multiple of the step, like the reference reads its buffer).  Returns result_len.
 * PARITY UNPINNED: rtl_fm.c needs librtlsdr and its vendored copy does not compile (low_pass_real). */
size_t orc_rtlfm_full(const uint8_t *iq_u8, size_t n, int downsample, int rotate90, int squelch_level, int post_downsample,
                      int deemph_a, int dc_block, int32_t *st, int32_t *post, int16_t *pcm, size_t cap, int math_mode)
{
    size_t m;
    int16_t *lp = rtlfm_lowpass_stage(iq_u8, n, downsample, rotate90, st, &m);
    (void)cap;
    if (squelch_level) {
        int sr = rtlfm_rms(lp, (int)(2 * m), 1);
        if (sr < squelch_level) {
            post[2]++;
            for (size_t i = 0; i < 2 * m; i++) lp[i] = 0;
        } else {
            post[2] = 0;
        }
    }
    if (m) {
        int16_t pre[2] = {(int16_t)st[3], (int16_t)st[4]};
        orc_fm_discriminator_i16(lp, m, pre, pcm, math_mode);
        st[3] = pre[0];
        st[4] = pre[1];
    }
    free(lp);
    int result_len = (int)m;
    if (post_downsample > 1) { /* low_pass_simple */
        int i, i2, sum, step = post_downsample;
        for (i = 0; i < result_len; i += step) {
            sum = 0;
            for (i2 = 0; i2 < step; i2++) sum += (int)pcm[i + i2];
            pcm[i / step] = (int16_t)(sum);
        }
        pcm[i / step + 1] = pcm[i / step];
        result_len = result_len / step;
    }
    if (deemph_a > 0) { /* deemph_filter */
        int avg = post[0];
        for (int i = 0; i < result_len; i++) {
            int d = pcm[i] - avg;
            if (d > 0) avg += (d + deemph_a / 2) / deemph_a;
            else avg += (d - deemph_a / 2) / deemph_a;
            pcm[i] = (int16_t)avg;
        }
        post[0] = avg;
    }
    if (dc_block && result_len > 0) { /* dc_block_filter (divides by result_len) */
        int64_t sum = 0;
        for (int i = 0; i < result_len; i++) sum += pcm[i];
        int avg = (int)(sum / result_len);
        avg = (avg + post[1] * 9) / 10;
        for (int i = 0; i < result_len; i++) pcm[i] = (int16_t)(pcm[i] - avg);
        post[1] = avg;
    }
    post[3] = result_len;
    return (size_t)result_len;
}

/* low_pass_real, rtl_fm.c:341-361: result[] in place, lpr = {now_lpr, prev_lpr_index}; returns the new result_len.
 * PARITY UNPINNED like the rest of the rtl_fm chain. */
int orc_rtlfm_low_pass_real(int16_t *result, int result_len, int rate_out, int rate_out2, int32_t *lpr)
{
    int i = 0, i2 = 0;
    int fast = rate_out;
    int slow = rate_out2;
    int now_lpr = lpr[0], prev_lpr_index = lpr[1];
    while (i < result_len) {
        now_lpr += result[i];
        i++;
        prev_lpr_index += slow;
        if (prev_lpr_index < fast) continue;
        result[i2] = (int16_t)(now_lpr / (fast / slow));
        prev_lpr_index -= fast;
        now_lpr = 0;
        i2 += 1;
    }
    lpr[0] = now_lpr;
    lpr[1] = prev_lpr_index;
    return i2;
}

/* ---------------------------------------------------------------- Resample */

size_t orc_decimate(double *block, size_t length, size_t by)
{
    size_t dst = 0, src = 0;
    while (src < length) { /* Resample.h:39-44 */
        block[dst] = block[src];
        src += by;
        dst++;
    }
    for (size_t j = dst; j < length; j++) block[dst] = 0; /* Resample.h:46-49 (sic: index dst) */
    return dst;
}

size_t orc_upsample_fill_w_zeros(const double *in, size_t in_len, double *out, size_t out_len, double ratio)
{
    for (size_t oi = 0, ii = 0; oi < out_len && ii < in_len; ii++) { /* Resample.cpp:26-33 */
        out[oi++] = in[ii];
        for (int j = 0; j < (ceil(ratio) - 1) && oi < out_len && ii < in_len; j++) out[oi++] = 0;
    }
    return out_len;
}

size_t orc_upsample(const double *in, size_t in_len, double *out, size_t out_len, double ratio)
{
    (void)in_len; /* Resample.cpp:10: the parameter is commented out, the loop runs on out_length alone */
    for (size_t oi = 0, ii = 0; oi < out_len; ii++) { /* Resample.cpp:12-19 */
        out[oi++] = in[ii];
        for (int j = 1; j < ceil(ratio) && oi < out_len; j++) out[oi++] = in[ii] + (ratio * j);
    }
    return out_len;
}

size_t orc_decimate_block(double *iq, size_t n_points, size_t by)
{
    size_t dst = 0, src = 0; /* Resample.cpp:38-48: points are {re, im} pairs; the first point is dropped */
    while (src + by < n_points) {
        src += by;
        iq[2 * dst] = iq[2 * src];
        iq[2 * dst + 1] = iq[2 * src + 1];
        dst++;
    }
    return dst + 1; /* block->n_points */
}

size_t orc_decimate_complex(double *iq, size_t length, size_t by)
{
    size_t dst = 0, src = 0; /* Resample.cpp:50-70 */
    while (src < length) {
        iq[2 * dst] = iq[2 * src];
        iq[2 * dst + 1] = iq[2 * src + 1];
        src += by;
        dst++;
    }
    for (size_t j = dst; j < length; j++) { /* (sic: index dst) */
        iq[2 * dst] = 0;
        iq[2 * dst + 1] = 0;
    }
    return dst;
}

/* ------------------------------------------------------------------ chains */

size_t orc_chain_fir_am(const double *taps, int T, double *histI, double *histQ, const double *re,
                        const double *im, size_t n, size_t ds, uint8_t *bits)
{
    double *fi = (double *)malloc((n ? n : 1) * sizeof(double));
    double *fq = (double *)malloc((n ? n : 1) * sizeof(double));
    memcpy(fi, re, n * sizeof(double));
    memcpy(fq, im, n * sizeof(double));
    orc_fir_execute_fast(taps, T, histI, fi, n);
    orc_fir_execute_fast(taps, T, histQ, fq, n);
    size_t r = orc_am_demodulate(fi, fq, bits, n, ds);
    free(fi);
    free(fq);
    return r;
}

size_t orc_chain_fir_decim_disc(const double *taps, int T, double *histI, double *histQ, double *pre,
                                const double *re, const double *im, size_t n, size_t D, double *out,
                                int math_mode)
{
    double *fi = (double *)malloc((n ? n : 1) * sizeof(double));
    double *fq = (double *)malloc((n ? n : 1) * sizeof(double));
    memcpy(fi, re, n * sizeof(double));
    memcpy(fq, im, n * sizeof(double));
    orc_fir_execute_fast(taps, T, histI, fi, n);
    orc_fir_execute_fast(taps, T, histQ, fq, n);
    size_t m = orc_decimate(fi, n, D);
    (void)orc_decimate(fq, n, D);
    orc_fm_discriminator(fi, fq, m, pre, out, math_mode);
    free(fi);
    free(fq);
    return m;
}

/* --------------------------------------------------------------------- PSK */

const double ORC_RRC[129] = { /* MATLAB rcosine(1/8,1,'sqrt',0.5,8): BPSK.cpp:18-50 */
    -0.000882663055055987, -0.000746201590523265, -0.000336789968798916, 0.000225038893553655,
    0.000757880681389978,  0.001077943504359050,  0.001062303725777150,  0.000695521249813576,
    8.24465490986367e-05,  -0.000579554041955506, -0.001062303725777150, -0.001183702054816000,
    -0.000874477709296126, -0.000211083163638599, 0.000599193283619645,  0.001279536876191750,
    0.001573979573701230,  0.001338143762729140,  0.000599193283619642,  -0.000438236415412023,
    -0.001446863119017230, -0.002078893324143190, -0.002084671221261930, -0.001408617284422990,
    -0.000227352605090172, 0.001088699270085450,  0.002084671221261940,  0.002371289907799650,
    0.001768388256576610,  0.000391988582678128,  -0.001351122110122730, -0.002869993360503870,
    -0.003572683794274230, -0.003072685906611650, -0.001351122110122720, 0.001186107048403510,
    0.003789403406949890,  0.005575980755162950,  0.005816670050933640,  0.004213278581633850,
    0.001071805138282270,  -0.002702696087965530, -0.005816670050933650, -0.006968745846090550,
    -0.005305164769729840, -0.000815536975401244, 0.005468827588591990,  0.011573823048484000,
    0.015005271935951800,  0.013430574641186900,  0.005468827588591980,  -0.008615615450369140,
    -0.026525823848649200, -0.044008760276746200, -0.055452254485567400, -0.054931750238658300,
    -0.037513179839879400, -0.000532386899804849, 0.055452254485567400,  0.126288206185736000,
    0.204577471545948000,  0.280728910881642000,  0.344536138081295000,  0.386975375596443000,
    0.401855774335913000,  0.386975375596443000,  0.344536138081295000,  0.280728910881642000,
    0.204577471545948000,  0.126288206185736000,  0.055452254485567400,  -0.000532386899804849,
    -0.037513179839879400, -0.054931750238658300, -0.055452254485567400, -0.044008760276746200,
    -0.026525823848649200, -0.008615615450369140, 0.005468827588591980,  0.013430574641186900,
    0.015005271935951800,  0.011573823048484000,  0.005468827588591990,  -0.000815536975401244,
    -0.005305164769729840, -0.006968745846090550, -0.005816670050933650, -0.002702696087965530,
    0.001071805138282270,  0.004213278581633850,  0.005816670050933640,  0.005575980755162950,
    0.003789403406949890,  0.001186107048403510,  -0.001351122110122720, -0.003072685906611650,
    -0.003572683794274230, -0.002869993360503870, -0.001351122110122730, 0.000391988582678128,
    0.001768388256576610,  0.002371289907799650,  0.002084671221261940,  0.001088699270085450,
    -0.000227352605090172, -0.001408617284422990, -0.002084671221261930, -0.002078893324143190,
    -0.001446863119017230, -0.000438236415412023, 0.000599193283619642,  0.001338143762729140,
    0.001573979573701230,  0.001279536876191750,  0.000599193283619645,  -0.000211083163638599,
    -0.000874477709296126, -0.001183702054816000, -0.001062303725777150, -0.000579554041955506,
    8.24465490986367e-05,  0.000695521249813576,  0.001062303725777150,  0.001077943504359050,
    0.000757880681389978,  0.000225038893553655,  -0.000336789968798916, -0.000746201590523265,
    -0.000882663055055987};

static const double QPSK_DEFAULT_UW[64] = { /* QPSK.cpp:20-28 */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 2, 1, 0, 3, 2, 1, 2, 0, 1, 2, 3, 2, 2, 3};
static const double LUT_B_REAL[2] = {1, -1};        /* BPSK.cpp:14 */
static const double LUT_B_180[2] = {1, 0};          /* BPSK.cpp:15 */
static const double LUT_Q_REAL[4] = {1, -1, 1, -1}; /* QPSK.cpp:14 */
static const double LUT_Q_IMAG[4] = {1, 1, -1, -1}; /* QPSK.cpp:15 */
static const double LUT_Q_000[4] = {0, 1, 2, 3};    /* QPSK.cpp:16 */
static const double LUT_Q_090[4] = {2, 0, 3, 1};    /* QPSK.cpp:17 */
static const double LUT_Q_180[4] = {3, 2, 1, 0};    /* QPSK.cpp:18 */
static const double LUT_Q_270[4] = {1, 3, 0, 2};    /* QPSK.cpp:19 */

static void psk_common_init(orc_psk *s, int kind, size_t uw_len, int math_mode)
{
    memset(s, 0, sizeof(*s));
    s->g1 = s->g2 = s->g3 = s->g4 = 1; /* BPSK.cpp:68-71 */
    for (int i = 0; i < 4; i++) s->cc[i] = 10 / 2; /* confidence_counter.cpp:7 */
    s->kind = kind;
    s->uw_size = (int32_t)uw_len;
    s->math_mode = math_mode;
}

void orc_bpsk_init(orc_psk *s, uint64_t unique_word, size_t uw_len, int math_mode)
{
    psk_common_init(s, ORC_BPSK, uw_len, math_mode);
    if (uw_len > 64) return; /* BPSK.cpp:79 */
    for (int i = (int)uw_len - 1; i >= 0; i--) { /* BPSK.cpp:83-89 */
        uint64_t mask = 1ULL << i;
        s->uw[0][i] = ((unique_word & mask) > 0) ? 1 : 0;
        s->uw[1][i] = ((unique_word & mask) > 0) ? 0 : 1;
    }
}

void orc_qpsk_init(orc_psk *s, const double *uw, size_t uw_len, int math_mode)
{
    psk_common_init(s, ORC_QPSK, uw_len, math_mode);
    if (!uw) uw = QPSK_DEFAULT_UW;
    if (uw_len > 64) return;                               /* QPSK.cpp:96 */
    for (int i = 64 - 1; i >= 64 - (int)uw_len && i >= 0; i--) { /* QPSK.cpp:102-118 */
        if (uw[i] >= 0 && uw[i] <= 3) {
            s->uw[0][i] = LUT_Q_000[(int)uw[i]];
            s->uw[1][i] = LUT_Q_270[(int)uw[i]]; /* uw_090 <- LUT_REAL_270 (QPSK.cpp:107) */
            s->uw[2][i] = LUT_Q_180[(int)uw[i]];
            s->uw[3][i] = LUT_Q_090[(int)uw[i]]; /* uw_270 <- LUT_REAL__90 (QPSK.cpp:109) */
        }
    }
}

void orc_psk_encode_lut(int kind, const uint8_t *data, size_t len, double *re, double *im)
{
    for (size_t i = 0; i < len; i++) {
        if (kind == ORC_BPSK) re[i] = LUT_B_REAL[data[i]]; /* BPSK.cpp:116-119 */
        else {
            re[i] = LUT_Q_REAL[data[i]]; /* QPSK.cpp:148-152 */
            im[i] = LUT_Q_IMAG[data[i]];
        }
    }
}

void orc_psk_pulse_shape(const double *sym, size_t len, double *out, size_t zero_prefix)
{
    size_t n = len * 8;
    double *data = (double *)calloc(n + 129 * 2, sizeof(double)); /* BPSK.cpp:132-133 */
    orc_upsample_fill_w_zeros(sym, len, &data[129], n, 8);       /* BPSK.cpp:135 */
    for (size_t i = 0; i < n && i < zero_prefix; i++) out[i] = 0; /* BPSK.cpp:131 (bytes, not doubles) */
    for (size_t i = 0; i < n; i++)                                /* BPSK.cpp:137-143 */
        for (int j = 0; j < 129; j++) out[i] += data[i + 129 - j] * ORC_RRC[129 - j - 1];
    free(data);
}

size_t orc_psk_decode_pulse_shape(const double *iq, size_t n, double *re, double *im)
{
    double *rd = (double *)calloc(n + 129 * 2, sizeof(double)); /* BPSK.cpp:262-270 */
    double *id = (double *)calloc(n + 129 * 2, sizeof(double));
    double *ri = (double *)calloc(n ? n : 1, sizeof(double));
    double *ii = (double *)calloc(n ? n : 1, sizeof(double));
    for (size_t i = 0; i < n; i++) { /* BPSK.cpp:273-277 */
        rd[i + 129] = iq[2 * i];
        id[i + 129] = iq[2 * i + 1];
    }
    for (size_t i = 0; i < n; i++) /* BPSK.cpp:279-286 */
        for (int j = 0; j < 129; j++) {
            ri[i] += rd[i + 129 - j] * ORC_RRC[129 - j - 1];
            ii[i] += id[i + 129 - j] * ORC_RRC[129 - j - 1];
        }
    (void)orc_decimate(ri, n, 4); /* BPSK.cpp:289-290 */
    (void)orc_decimate(ii, n, 4);
    memcpy(re, ri, sizeof(double) * (n / 4)); /* BPSK.cpp:292-293 */
    memcpy(im, ii, sizeof(double) * (n / 4));
    free(rd);
    free(id);
    free(ri);
    free(ii);
    return n / 4;
}

static double interp_filter(orc_interp *d, double in, double mu)
{
    const double K = -0.5; /* BPSK.cpp:319-344 */
    double V1 = d->d1 - K * in;
    double V2 = V1 + d->d2;
    double V3 = V2 - d->d3;
    double V4 = K * in - d->d1 + d->d4;
    double V5 = V4 + d->d2;
    double V6 = d->d3 + V5;
    double V7 = V3 * mu + V6;
    double V8 = d->d5 + mu * V7;
    d->d3 = d->d2;
    d->d2 = d->d1;
    d->d1 = K * in;
    d->d5 = d->d4;
    d->d4 = in;
    return V8;
}

static double strobe_reg(double input, double strobe, double *reg)
{
    if (strobe >= 1) *reg = input; /* BPSK.cpp:382-389 */
    return *reg;
}

void orc_psk_timing(orc_psk *s, double *re, double *im, size_t n)
{
    for (size_t i = 0; i < n; i++) { /* BPSK.cpp:307-316 */
        double strobe = (s->nco_delay < 0) ? 1 : 0;
        double mu = strobe_reg(s->mu_delay * 2, strobe, &s->reg_mu);
        double Q = interp_filter(&s->interp_q, re[i], mu); /* "Q" gets real (sic) */
        double I = interp_filter(&s->interp_i, im[i], mu);
        /* GardnerDetector, BPSK.cpp:346-363 */
        double e = 0;
        if (strobe >= 0.5)
            e = (s->g3 * (-SGN(Q) + SGN(s->g4)) + (s->g1 * (-SGN(I) + SGN(s->g2))));
        s->g2 = s->g1;
        s->g1 = I;
        s->g4 = s->g3;
        s->g3 = Q;
        /* LoopFilter, BPSK.cpp:365-373 */
        const double K1 = -0.00587880145347487, K2 = -2.35152058138995e-05;
        double v = s->lf_y + K2 * e + K1 * (e - s->lf_x);
        s->lf_x = e;
        s->lf_y = v;
        /* NcoControl, BPSK.cpp:375-380 */
        double eta = s->nco_delay - floor(s->nco_delay);
        s->nco_delay = eta - 0.5 - v;
        s->mu_delay = eta;
        re[i] = strobe_reg(Q, strobe, &s->reg_Q);
        im[i] = strobe_reg(I, strobe, &s->reg_I);
    }
}

static double phase_loop_filter(orc_psk *s, double val)
{
    const double K1 = 0.0620001240002480, K2 = 0.000992001984003968; /* BPSK.cpp:424-432 */
    double r = s->plf_y - K1 * s->plf_x + (K1 + K2) * val;
    s->plf_x = val;
    s->plf_y = r;
    return r;
}

/* libm's cos and sin as TWO calls, the way the reference's -O0 build makes them (RadioNode/CMakeLists.txt:8).  At -O2 gcc
 * merges a cos / sin pair of one argument into a single sincos() call, and glibc's sincos() differs from its own cos() /
 * sin() in the last bit for a few arguments per thousand (seen: 2 of 4096 twiddles of a 4096-point fft): calling through
 * volatile pointers keeps the calls apart. */
static double (*volatile libm_cos)(double) = cos;
static double (*volatile libm_sin)(double) = sin;

/* include/sdrx_math.h's two sincos forms over an array (tests: distance from libm).  form 0 = sdrx_sincos, 1 = sdrx_sincos_tab */
void orc_sincos(const double *x, size_t n, int form, double *sn, double *cs)
{
    for (size_t i = 0; i < n; i++) {
        if (form == 1) sdrx_sincos_tab(x[i], SDRX_SINCOS_TAB, &sn[i], &cs[i]);
        else sdrx_sincos(x[i], &sn[i], &cs[i]);
    }
}

static void dds(orc_psk *s, double arg, double *c, double *sn)
{
    if (s->math_mode == ORC_MATH_SHARED) sdrx_sincos_tab(arg, SDRX_SINCOS_TAB, sn, c); /* the carrier loop's form (psk_kernels.hip) */
    else {
        *c = libm_cos(arg);
        *sn = libm_sin(arg);
    }
}

size_t orc_psk_phase_correction(orc_psk *s, double *re, double *im, size_t n)
{
    double tr = 0, ti = 0, temp = 0;
    const double constant = 8.0;
    for (size_t i = 0; i < n; i++) {
        double c, sn;
        if (s->kind == ORC_BPSK) { /* BPSK.cpp:399-411 */
            dds(s, s->dds_delay + temp, &c, &sn);
            s->dds_delay = s->dds_delay + temp;
            tr = re[i] * c - im[i] * sn; /* BPSK.cpp:417-421 */
            ti = im[i] * c + re[i] * sn;
            temp = phase_loop_filter(s, constant * (SGN(ti) * tr - SGN(tr) * ti));
            re[i] = (SGN(tr) < 0) ? 1 : 0; /* BPSK.cpp:434-444 */
        } else {                           /* QPSK.cpp:361-373 */
            temp = phase_loop_filter(s, constant * (SGN(ti) * tr - SGN(tr) * ti));
            dds(s, s->dds_delay + temp, &c, &sn);
            s->dds_delay = s->dds_delay + temp;
            tr = re[i] * c - im[i] * sn;
            ti = im[i] * c + re[i] * sn;
            double sr = SGN(tr), si = SGN(ti); /* QPSK.cpp:396-414 */
            if (sr < 0 && si < 0) re[i] = 3;
            else if (sr > 0 && si < 0) re[i] = 2;
            else if (sr < 0 && si > 0) re[i] = 1;
            else re[i] = 0;
        }
        im[i] = 0;
    }
    return orc_decimate(re, n, 2); /* BPSK.cpp:413 */
}

static void cc_inc(int32_t *c) /* confidence_counter.cpp:41-50, max 10 */
{
    (*c)++;
    if (*c >= 10) *c = 10;
}
static void cc_dec(int32_t *c) /* confidence_counter.cpp:52-61 */
{
    (*c)--;
    if (*c <= 0) *c = 0;
}
#define CC_HIGH(c) ((c) >= 5) /* confidence_counter.cpp:23-30, high_threshold 5 */

size_t orc_psk_ambiguity(orc_psk *s, double *re, size_t n, uint64_t *events, size_t max_events)
{
    size_t U = (size_t)s->uw_size, hits = 0;
    if (s->kind == ORC_BPSK) {
        for (size_t i = 0; n >= U && i < n - U; i++) { /* BPSK.cpp:449-476 */
            size_t m0 = 0, m180 = 0;
            for (size_t j = 0; j < U; j++) {
                if (re[i + j] == s->uw[0][U - 1 - j]) m0++;
                if (re[i + j] == s->uw[1][U - 1 - j]) m180++;
            }
            int code = -1;
            if (m0 == U) { cc_dec(&s->cc[0]); code = 0; }
            else if (m180 == U) { cc_inc(&s->cc[0]); code = 2; }
            if (code >= 0) {
                if (events && hits < max_events) events[hits] = ((uint64_t)i << 3) | (uint64_t)code;
                hits++;
            }
        }
        if (CC_HIGH(s->cc[0])) /* BPSK.cpp:478-485 */
            for (size_t i = 0; i < n; i++) re[i] = LUT_B_180[(int)re[i]];
        return hits;
    }
    for (size_t i = 0; n >= U && i < n - U; i++) { /* QPSK.cpp:419-478 */
        size_t m[4] = {0, 0, 0, 0};
        for (size_t j = 0; j < U; j++) {
            double v = re[i + j];
            if (v == s->uw[0][63 - j]) m[0]++;
            else if (v == s->uw[1][63 - j]) m[1]++;
            else if (v == s->uw[2][63 - j]) m[2]++;
            else if (v == s->uw[3][63 - j]) m[3]++;
        }
        int code = -1;
        for (int k = 0; k < 4; k++)
            if (m[k] == U) { code = k; break; }
        if (code >= 0) {
            for (int k = 0; k < 4; k++) {
                if (k == code) cc_inc(&s->cc[k]);
                else cc_dec(&s->cc[k]);
            }
            if (events && hits < max_events) events[hits] = ((uint64_t)i << 3) | (uint64_t)code;
            hits++;
        }
    }
    const double *lut = NULL; /* QPSK.cpp:481-507 */
    if (CC_HIGH(s->cc[1])) lut = LUT_Q_090;
    else if (CC_HIGH(s->cc[2])) lut = LUT_Q_180;
    else if (CC_HIGH(s->cc[3])) lut = LUT_Q_270;
    if (lut)
        for (size_t i = 0; i < n; i++) re[i] = lut[(int)re[i]];
    return hits;
}

size_t orc_psk_demodulate(orc_psk *s, const double *iq, size_t n, double *out, double *timing_re,
                          double *timing_im)
{
    size_t cap = n ? n : 1;
    double *re = (double *)calloc(cap, sizeof(double)); /* BPSK.cpp:150-154 */
    double *im = (double *)calloc(cap, sizeof(double));
    size_t m = orc_psk_decode_pulse_shape(iq, n, re, im); /* BPSK.cpp:157 */
    orc_psk_timing(s, re, im, m);                         /* BPSK.cpp:159 */
    if (timing_re) memcpy(timing_re, re, m * sizeof(double));
    if (timing_im) memcpy(timing_im, im, m * sizeof(double));
    m = orc_psk_phase_correction(s, re, im, m); /* BPSK.cpp:161 */
    orc_psk_ambiguity(s, re, m, NULL, 0);       /* BPSK.cpp:163 */
    memcpy(out, re, m * sizeof(double));        /* BPSK.cpp:167-172 */
    free(re);
    free(im);
    return m;
}

void orc_bpsk_modulate(const uint8_t *bits, size_t len, double *out_real, size_t len_mod)
{
    if (len_mod < 8 * len) return; /* BPSK.cpp:125-129 */
    double *sym = (double *)calloc(len * 8 + 1, sizeof(double));
    orc_psk_encode_lut(ORC_BPSK, bits, len, sym, NULL);
    orc_psk_pulse_shape(sym, len, out_real, len_mod / 8);
    free(sym);
}

void orc_qpsk_modulate(const uint8_t *symb, size_t len, double *out_real, double *out_imag, size_t len_mod)
{
    if (len_mod != 8 * len) return; /* QPSK.cpp:158-162 */
    double *sr = (double *)calloc(len * 8 + 1, sizeof(double));
    double *si = (double *)calloc(len * 8 + 1, sizeof(double));
    orc_psk_encode_lut(ORC_QPSK, symb, len, sr, si);
    orc_psk_pulse_shape(sr, len, out_real, len_mod / 8);
    orc_psk_pulse_shape(si, len, out_imag, len_mod / 8);
    free(sr);
    free(si);
}

/* --------------------------------------------------- bit-domain back end
 *
 * crc8 (crc/crc8.cpp:6-44, CRC7_POLY 0x91, crc8.h:35), PacketFramer<T>::deserialize
 * (network/PacketFramer.h:424-513) over a RingBuffer<T> (common/RingBuffer.h), and the receiver's
 * loop around it (RtlSdr_Receiver/rtlsdr_receiver.cpp:2197-2230), restated literally: the same
 * passes over the 16 unique-word bits, the same restart at index 1 after a removed bit, the same
 * early returns (500 bad bits; fewer than a packet's bits left at the end of a pass).  The ring
 * buffer is modelled as a window [front, front + count) over the caller's bit array (the stream
 * appended at once); value(i) is only ever read for i < count here, as in the reference. */

static uint8_t orc_crc_table[256];
static int orc_crc_ready = 0;

static void orc_crc_init(void)
{
    for (int i = 0; i < 256; i++) { /* crc8::crc8 + getCRCForByte */
        uint8_t val = (uint8_t)i;
        for (int j = 0; j < 8; j++) {
            if (val & 1) val ^= 0x91;
            val >>= 1;
        }
        orc_crc_table[i] = val;
    }
    orc_crc_ready = 1;
}

uint8_t orc_crc8(const uint8_t *message, size_t length)
{
    if (!orc_crc_ready) orc_crc_init();
    uint8_t crc = 0;
    for (size_t i = 0; i < length; i++) crc = orc_crc_table[crc ^ message[i]];
    return crc;
}

const uint8_t ORC_FRAMER_UW[16] = {1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 0}; /* PacketFramer.h:17-19 */
#define ORC_PKT_BITS 2032 /* MAX_PKT_SIZE_BITS = (250 + 4) * 8 */

typedef struct {
    const uint8_t *d;
    size_t front, count;
} orc_ring;

static uint8_t ring_value(const orc_ring *rb, size_t i) { return i < rb->count ? rb->d[rb->front + i] : 0; }
static void ring_remove(orc_ring *rb, size_t k)
{
    while (k-- && rb->count) { rb->front++; rb->count--; }
}
static uint8_t ring_to_byte(const orc_ring *rb, size_t index) /* PacketFramer.h:537-549 */
{
    uint8_t c = 0;
    for (size_t i = 0; i < 8; ++i)
        if (ring_value(rb, index + i) == 1) c |= (uint8_t)(1 << (7 - i));
    return c;
}

/* PacketFramer<T>::deserialize.  Returns 254 (valid packet), 0 or -1 like the reference. */
static int orc_deserialize(orc_ring *buf, uint8_t *pkt, int crc_enable, int *validPacket, int *validCRC,
                           int *validHeader, int *bad_bits, uint8_t *len_out, uint8_t *crc_out)
{
    *validPacket = 0;
    *validCRC = 0;
    *validHeader = 0;
    *bad_bits = 0;
    if (buf->count < ORC_PKT_BITS) return 0;
    do {
        *validHeader = 1;
        for (size_t i = 0; i < 16; i++) {
            if (ring_value(buf, i) != ORC_FRAMER_UW[i]) {
                (*bad_bits)++;
                ring_remove(buf, 1);
                *validHeader = 0;
                i = 0; /* (sic) the loop's i++ makes the next comparison start at index 1 */
                if (*bad_bits >= 500) return 0;
            }
        }
        if (buf->count < ORC_PKT_BITS) return 0;
    } while (!*validHeader);
    *validHeader = 1;
    uint8_t pkt_length = ring_to_byte(buf, 16);
    *len_out = pkt_length;
    if (pkt_length > 250) {
        *bad_bits = 1;
        return -1;
    }
    uint8_t pkt_crc = ring_to_byte(buf, 24);
    *crc_out = pkt_crc;
    memset(pkt, 0, 250);
    for (size_t i = 0; i < 250; i++) pkt[i] = ring_to_byte(buf, i * 8 + 32);
    if (crc_enable) {
        if (orc_crc8(pkt, pkt_length) == pkt_crc) {
            *validCRC = 1;
        } else {
            *bad_bits = 1;
            return -1;
        }
    } else {
        *validCRC = 1;
    }
    *validPacket = 1;
    return 254;
}

/* The receiver's deserializeData loop over a stream of n bits appended at once.
 * Events (in stream order, at most max_events recorded, all counted): kind 1 = valid packet
 * (its 250 data bytes go to payload[250 * k]), 2 = length > 250, 3 = CRC mismatch; pos = index of
 * the header's first bit.  summary = {events, bits consumed (front at exit), calls that returned 0}. */
void orc_deframe(const uint8_t *bits, size_t n, int crc_enable, size_t max_events, int64_t *ev_pos, int32_t *ev_kind,
                 uint8_t *ev_len, uint8_t *ev_crc, uint8_t *payload, uint64_t *summary)
{
    orc_ring rb = {bits, 0, n};
    uint8_t pkt[250];
    uint64_t nev = 0, zero_returns = 0;
    while (rb.count > ORC_PKT_BITS) {
        int vp, vc, vh, bad;
        uint8_t len = 0, crc = 0;
        int r = orc_deserialize(&rb, pkt, crc_enable, &vp, &vc, &vh, &bad, &len, &crc);
        if (r > 0 || r == -1) {
            if (nev < max_events) {
                ev_pos[nev] = (int64_t)rb.front;
                ev_kind[nev] = r > 0 ? 1 : (len > 250 ? 2 : 3);
                ev_len[nev] = len;
                ev_crc[nev] = (r > 0 || len <= 250) ? crc : 0;
                if (r > 0) memcpy(payload + 250 * nev, pkt, 250);
            }
            nev++;
            ring_remove(&rb, r > 0 ? (size_t)r * 8 : 1);
        } else {
            zero_returns++;
        }
    }
    summary[0] = nev;
    summary[1] = rb.front;
    summary[2] = zero_returns;
}

/* PacketFramer<T>::serialize (PacketFramer.h:363-413): 2032 bits; returns -1 if length > 250 */
int orc_frame(const uint8_t *data, size_t length, int crc_enable, uint8_t *bits_out)
{
    if (length > 250) return -1;
    memset(bits_out, 0, ORC_PKT_BITS);
    memcpy(bits_out, ORC_FRAMER_UW, 16);
    uint8_t crc = crc_enable ? orc_crc8(data, length) : 0;
    for (int i = 0; i < 8; i++) {
        bits_out[16 + i] = (uint8_t)((length >> (7 - i)) & 1); /* binary_array[v][i]: MSB first */
        bits_out[24 + i] = (uint8_t)((crc >> (7 - i)) & 1);
    }
    for (size_t i = 0; i < length; i++)
        for (int j = 0; j < 8; j++) bits_out[32 + i * 8 + j] = (uint8_t)((data[i] >> (7 - j)) & 1);
    return ORC_PKT_BITS;
}


/* ------------------------------------------------------------------------------------------
 * DSP primitives.  None of them has a caller or a test in the reference;
 * the restatements are pinned, in the build container, to the reference's own files compiled
 * into oracle/_ref (tests/test_oracle_vs_ref.py).
 * ------------------------------------------------------------------------------------------ */

/* DC_Filter::update (DC_Filter.cpp:12-19): y(n) = x(n) - x(n-1) + R*y(n-1), y(0) = x(0); no carried state */
void orc_dc_block(const double *in, double *out, size_t n, double R)
{
    if (n == 0) return;
    out[0] = in[0];
    for (size_t i = 1; i < n; i++) out[i] = in[i] - in[i - 1] + R * out[i - 1];
}

double orc_lowpass1_coef(double cutoff, double dt) { return 1.0 - exp(-dt * cutoff); }

/* LowPassFilter::update (LowPassFilter.cpp:13-16) called once per sample; *state1 = `output` */
void orc_lowpass1(const double *in, double *out, size_t n, double ePow, double *state1)
{
    double o = *state1;
    for (size_t i = 0; i < n; i++) {
        o += (in[i] - o) * ePow;
        out[i] = o;
    }
    *state1 = o;
}

/* NotchFilter::NotchFilter (NotchFilter.cpp:7-16) */
void orc_notch_design(double notch_bw, double center_freq, double *coef5)
{
    double R = 1.0 * notch_bw;
    double K = (1 - 2 * R * cos(2.0 * 3.14159 * center_freq) + R * R) / (2 - 2 * cos(2 * 3.14159 * center_freq));
    coef5[0] = K;
    coef5[1] = -2.0 * K * cos(2.0 * 3.14159 * center_freq);
    coef5[2] = K;
    coef5[3] = 2.0 * R * cos(2.0 * 3.14159 * center_freq);
    coef5[4] = -(R * R);
}

/* NotchFilter::filter (NotchFilter.cpp:18-37): in place from index 2 (the first two samples of
 * every block neither change nor enter the delay line); the delays are function statics in the
 * reference (shared by all instances), per-channel state here. */
void orc_notch(double *b, size_t n, const double *c, double *st)
{
    double x_1 = st[0], x_2 = st[1], y_1 = st[2], y_2 = st[3];
    for (size_t i = 2; i < n; i++) {
        double delay = b[i];
        b[i] = c[0] * b[i] + c[1] * x_1 + c[2] * x_2 + c[3] * y_1 + c[4] * y_2;
        x_2 = x_1;
        x_1 = delay;
        y_2 = y_1;
        y_1 = b[i];
    }
    st[0] = x_1, st[1] = x_2, st[2] = y_1, st[3] = y_2;
}

/* FirstOrderDigitalIntegration::integrate (FirstOrderDigitalIntegration.cpp:12-18); tStep = 1.0f/sampleRate */
void orc_integrate(double *a, size_t n, double tStep)
{
    for (size_t i = 1; i < n; i++) a[i] = a[i - 1] + tStep * a[i];
}

/* normalize (Normalize.cpp:7-22): the running maximum is a size_t (truncated on every update);
 * abs() resolves to the double overload with libstdc++'s <cmath>/<stdlib.h>.  The BBP_Block
 * overload (:24-43) is the same loop over the 2n doubles of the complex points. */
double orc_normalize(double *block, size_t n, double prev_max)
{
    size_t max = (size_t)prev_max;
    for (size_t i = 0; i < n; i++)
        if ((double)max < fabs(block[i])) max = (size_t)fabs(block[i]);
    for (size_t i = 0; i < n; i++) block[i] = (block[i] / (double)max) * 2.0 - 1.0;
    return (double)max;
}

/* fifth_order_filter::decimate(double*, size_t, size_t) (fifth_order_filter.cpp:65-88), a=1 b=5 c=10 d=11 */
size_t orc_fifth_order_decimate(double *data, size_t n, size_t decimate_by, double *s)
{
    const double a = 1, b = 5, c = 10, d = 11;
    double r_a = s[0], r_b = s[1], r_c = s[2], r_d = s[3], r_e = s[4], r_f = s[5];
    if (n == 0) return 0; /* the reference writes data[0] even then */
    for (size_t dec = 2; dec <= decimate_by; dec *= 2) {
        data[0] = ((a * (r_a + r_f)) + ((r_b + r_e) * b) + ((r_c + r_d) * c)) / d;
        for (size_t i = 2; i < n; i += 2) {
            r_a = r_c;
            r_b = r_d;
            r_c = r_e;
            r_d = r_f;
            r_e = data[i - 2];
            r_f = data[i];
            data[i / 2] = ((a * (r_a + r_f)) + ((r_b + r_e) * b) + ((r_c + r_d) * c)) / d;
        }
    }
    s[0] = r_a, s[1] = r_b, s[2] = r_c, s[3] = r_d, s[4] = r_e, s[5] = r_f;
    return n / decimate_by;
}

/* fft (FFT.cpp:7-28): recursive radix-2 decimation in time on std::complex<double>;
 * twiddle = std::polar(1.0, -2*PI*k/N) = (cos, sin); complex product = (ac-bd, ad+bc). */
static void fft_rec(double *x, size_t N, int math_mode)
{
    if (N <= 1) return;
    size_t h = N / 2;
    double *even = (double *)malloc(sizeof(double) * 2 * h), *odd = (double *)malloc(sizeof(double) * 2 * h);
    for (size_t k = 0; k < h; k++) {
        even[2 * k] = x[4 * k], even[2 * k + 1] = x[4 * k + 1];
        odd[2 * k] = x[4 * k + 2], odd[2 * k + 1] = x[4 * k + 3];
    }
    fft_rec(even, h, math_mode);
    fft_rec(odd, h, math_mode);
    for (size_t k = 0; k < h; k++) {
        double theta = -2 * 3.141592653589793238460 * (double)k / (double)N, wr, wi;
        if (math_mode == ORC_MATH_SHARED) sdrx_sincos(theta, &wi, &wr);
        else wr = libm_cos(theta), wi = libm_sin(theta);
        wr = 1.0 * wr, wi = 1.0 * wi;
        double tr = wr * odd[2 * k] - wi * odd[2 * k + 1], ti = wr * odd[2 * k + 1] + wi * odd[2 * k];
        x[2 * k] = even[2 * k] + tr, x[2 * k + 1] = even[2 * k + 1] + ti;
        x[2 * (k + h)] = even[2 * k] - tr, x[2 * (k + h) + 1] = even[2 * k + 1] - ti;
    }
    free(even);
    free(odd);
}

void orc_fft(double *iq, size_t n, int math_mode) { fft_rec(iq, n, math_mode); }

/* fifth_order_filter::decimate(BBP_Block*, size_t) (fifth_order_filter.cpp:28-63): interleaved complex points,
 * state12 = r_a..r_f, i_a..i_f; returns block->n_points */
size_t orc_fifth_order_decimate_block(double *iq, size_t n_points, size_t decimate_by, double *s)
{
    const double a = 1, b = 5, c = 10, d = 11;
    double *r = s, *q = s + 6;
    for (size_t dec = 2; dec <= decimate_by; dec *= 2) {
        iq[0] = ((a * (r[0] + r[5])) + ((r[1] + r[4]) * b) + ((r[2] + r[3]) * c)) / d;
        iq[1] = ((a * (q[0] + q[5])) + ((q[1] + q[4]) * b) + ((q[2] + q[3]) * c)) / d;
        size_t dst = 2;
        for (size_t i = 2; i < n_points; i += 2) {
            dst = i / 2;
            r[0] = r[2], r[1] = r[3], r[2] = r[4], r[3] = r[5];
            r[4] = iq[2 * (i - 2)];
            r[5] = iq[2 * i];
            iq[2 * dst] = ((a * (r[0] + r[5])) + ((r[1] + r[4]) * b) + ((r[2] + r[3]) * c)) / d;
            q[0] = q[2], q[1] = q[3], q[2] = q[4], q[3] = q[5];
            q[4] = iq[2 * (i - 2) + 1];
            q[5] = iq[2 * i + 1];
            iq[2 * dst + 1] = ((a * (q[0] + q[5])) + ((q[1] + q[4]) * b) + ((q[2] + q[3]) * c)) / d;
        }
        n_points = dst + 1;
    }
    return n_points;
}
