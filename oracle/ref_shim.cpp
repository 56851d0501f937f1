// ref_shim.cpp -- extern "C" handles onto the REFERENCE's own classes.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the
// reference headers where they lie under /root/reference and is linked against the
// reference's own .cpp files by oracle/Makefile into oracle/_ref/libsdr_ref.so
// (git-ignored; outputs only).  tests/test_oracle_vs_ref.py uses it to validate the
// C restatement (sdr_oracle.c) bit-for-bit, and gen_fixtures.py to dump fixtures.
// streams/FM.h does not compile as shipped and is NOT built here;
// the FM restatement is pinned by the reference's TEST_FM golden vectors instead.
#include <cstring>
#include <cstdint>
#include <cstdio>

#include <FIR-filter-class/filt.h>
#include <streams/AM.h>
#include <streams/BPSK.h>
#include <streams/QPSK.h>
#include <DigitalSignalProcessing/Resample.h>
#include <common/BBP_Block.h>
#include <common/Baseband_Stream.h>
#include <common/SafeBufferPoolQueue.h>
#include <condition_variable>
#include <mutex>
#include <common/RingBuffer.h>
#include <common/confidence_counter.h>
#include <cmath>
#include <crc/crc8.h>
#include <network/PacketFramer.h>
#include <DigitalSignalProcessing/DC_Filter.h>
#include <DigitalSignalProcessing/LowPassFilter.h>
#include <DigitalSignalProcessing/NotchFilter.h>
#include <DigitalSignalProcessing/FirstOrderDigitalIntegration.h>
#include <DigitalSignalProcessing/Normalize.h>
#include <DigitalSignalProcessing/fifth_order_filter.h>
#include <DigitalSignalProcessing/FFT.h>

extern "C" {

// ---- Filter ----
void *ref_fir_new(int type, int two_freq, int taps, double Fs, double Fx, double Fu)
{
    if (two_freq) return new Filter((filterType)type, taps, Fs, Fx, Fu);
    return new Filter((filterType)type, taps, Fs, Fx);
}
int ref_fir_error(void *h) { return ((Filter *)h)->get_error_flag(); }
void ref_fir_get_taps(void *h, double *taps) { ((Filter *)h)->get_taps(taps); }
void ref_fir_execute(void *h, double *buf, size_t n) { ((Filter *)h)->execute(buf, n); }
double ref_fir_do_sample(void *h, double x) { return ((Filter *)h)->do_sample(x); }
void ref_fir_init(void *h) { ((Filter *)h)->init(); }
// The error-path constructor leaves m_taps/m_sr uninitialised and the destructor frees
// them (filt.cpp:54-58,104-108): leak errored objects instead of crashing the checker.
void ref_fir_free(void *h) { if (((Filter *)h)->get_error_flag() == 0) delete (Filter *)h; }

// ---- AM ----
void ref_am_modulate(uint8_t *bits, size_t len, double *out, size_t len_mod)
{
    AM<RADIO_DATA_TYPE> am;
    am.modulate(bits, len, out, nullptr, len_mod);
}
size_t ref_am_demodulate(double *re, double *im, uint8_t *out, size_t n, size_t ds)
{
    AM<RADIO_DATA_TYPE> am;
    return am.demodulate(re, im, out, n, ds);
}

// ---- Resample ----
size_t ref_decimate(double *block, size_t len, size_t by) { return decimate<RADIO_DATA_TYPE>(block, len, by); }
size_t ref_upsample_fill_w_zeros(double *in, size_t in_len, double *out, size_t out_len, double ratio)
{
    return upsample_fill_w_zeros(in, in_len, out, out_len, ratio);
}

size_t ref_raw_upsample(double *in, size_t in_len, double *out, size_t out_len, double ratio)
{
    return raw_upsample<RADIO_DATA_TYPE>(in, in_len, out, out_len, ratio);
}
size_t ref_upsample(double *in, size_t in_len, double *out, size_t out_len, double ratio)
{
    return upsample(in, in_len, out, out_len, ratio);
}
// decimate(BBP_Block*): iq = interleaved complex of n_points points, in place; returns block->n_points afterwards
size_t ref_decimate_block(double *iq, size_t n_points, size_t by)
{
    BBP_Block b(n_points ? n_points : 1);
    for (size_t i = 0; i < n_points; i++) b.points[i] = Complex(iq[2 * i], iq[2 * i + 1]);
    b.n_points = n_points;
    decimate(&b, by);
    for (size_t i = 0; i < n_points; i++) { iq[2 * i] = b.points[i].real(); iq[2 * i + 1] = b.points[i].imag(); }
    return b.n_points;
}
size_t ref_decimate_complex(double *iq, size_t length, size_t by)
{
    Complex_Array a(Complex(0, 0), length ? length : 1);
    for (size_t i = 0; i < length; i++) a[i] = Complex(iq[2 * i], iq[2 * i + 1]);
    const size_t r = decimateComplexArray(a, length, by);
    for (size_t i = 0; i < length; i++) { iq[2 * i] = a[i].real(); iq[2 * i + 1] = a[i].imag(); }
    return r;
}

// ---- Baseband_Stream::load_buffer / load_buffer_and_rotate90 (Baseband_Stream.cpp:82-148): the byte -> sample
// conversion, observed through the stream's own post callback.  n_bytes / 2 points must fit a default BBP_Block.
size_t ref_baseband_load(const uint8_t *bytes, size_t n_bytes, int rotate90, double *re, double *im)
{
    SafeBufferPoolQueue pool(1);
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    size_t n_points = 0;
    {
        Baseband_Stream stream([&](BBP_Block *b) {
            std::lock_guard<std::mutex> lk(mu);
            n_points = b->n_points;
            for (size_t i = 0; i < n_points; i++) { re[i] = b->points[i].real(); im[i] = b->points[i].imag(); }
            done = true;
            cv.notify_all();
        }, &pool);
        std::vector<char> buf(bytes, bytes + n_bytes); // rotate_90 works in place on the caller's bytes
        buf.resize(n_bytes + 8);
        if (rotate90) stream.load_buffer_and_rotate90(buf.data(), n_bytes);
        else stream.load_buffer(buf.data(), n_bytes);
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    } // ~Baseband_Stream joins its thread
    return n_points;
}

// ---- BPSK ----
void *ref_bpsk_new(uint64_t uw, size_t uw_len) { return new BPSK(uw, uw_len); }
void ref_bpsk_free(void *h) { delete (BPSK *)h; }
void ref_bpsk_modulate(void *h, uint8_t *bits, size_t len, double *out, size_t len_mod)
{
    ((BPSK *)h)->modulate(bits, len, out, len_mod);
}
static void load_block(BBP_Block &b, const double *iq, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        b.points[i].real(iq[2 * i]);
        b.points[i].imag(iq[2 * i + 1]);
        b.n_points++;
    }
}
size_t ref_bpsk_demodulate(void *h, const double *iq, size_t n, double *out_re, double *out_im)
{
    BBP_Block in(n ? n : 1), out(n ? n : 1);
    load_block(in, iq, n);
    ((BPSK *)h)->demodulate(&in, &out);
    for (size_t i = 0; i < out.n_points; i++) {
        out_re[i] = out.points[i].real();
        if (out_im) out_im[i] = out.points[i].imag();
    }
    return out.n_points;
}
// overload 2 (BPSK.cpp:178-211): iq is modified in place by its decimateComplexArray, like the caller's valarray
size_t ref_bpsk_demodulate2(void *h, double *iq, size_t n, size_t downsample, uint8_t *out, double *plot_re, double *plot_im)
{
    Complex_Array a(Complex(0, 0), n ? n : 1);
    for (size_t i = 0; i < n; i++) a[i] = Complex(iq[2 * i], iq[2 * i + 1]);
    const size_t m = ((BPSK *)h)->demodulate(a, out, n, downsample, plot_re, plot_im);
    for (size_t i = 0; i < n; i++) { iq[2 * i] = a[i].real(); iq[2 * i + 1] = a[i].imag(); }
    return m;
}
size_t ref_bpsk_decode_pulse_shape(void *h, const double *iq, size_t n, double *re, double *im)
{
    BBP_Block in(n ? n : 1);
    load_block(in, iq, n);
    return ((BPSK *)h)->decode_pulseShape(&in, re, im);
}
void ref_bpsk_timing(void *h, double *re, double *im, size_t n) { ((BPSK *)h)->symbol_timing_synchronization(re, im, n); }
size_t ref_bpsk_phase(void *h, double *re, double *im, size_t n) { return ((BPSK *)h)->phase_shift_correction(re, im, n); }
void ref_bpsk_ambiguity(void *h, double *re, size_t n) { ((BPSK *)h)->phase_ambiguity_correction(re, nullptr, n); }
// state dump in the order of orc_psk's leading doubles (sdr_oracle.h)
void ref_bpsk_state(void *h, double *s, int32_t *cc)
{
    BPSK *b = (BPSK *)h;
    int k = 0;
    s[k++] = b->NcoControlVariables.delay;
    s[k++] = b->Register_mu;
    s[k++] = b->Register_Q;
    s[k++] = b->Register_I;
    s[k++] = b->SymbolTimingSynchronizationVariables.muDelay;
    const BPSK::InterpData *ip[2] = {&b->Interp_Q_Channel, &b->Interp_I_Channel};
    for (int c = 0; c < 2; c++) {
        s[k++] = ip[c]->delay1; s[k++] = ip[c]->delay2; s[k++] = ip[c]->delay3;
        s[k++] = ip[c]->delay4; s[k++] = ip[c]->delay5;
    }
    s[k++] = b->GardnerDetectorVariables.delay1; s[k++] = b->GardnerDetectorVariables.delay2;
    s[k++] = b->GardnerDetectorVariables.delay3; s[k++] = b->GardnerDetectorVariables.delay4;
    s[k++] = b->LoopFilterVariables.delay_x; s[k++] = b->LoopFilterVariables.delay_y;
    s[k++] = b->PhaseCorrectionVariables.dds_delay;
    s[k++] = b->PhaseLoopFilterVariables.delay_x; s[k++] = b->PhaseLoopFilterVariables.delay_y;
    cc[0] = b->m_phase_ambiguity_cc.value();
    cc[1] = cc[2] = cc[3] = 5;
}

// ---- QPSK ----
void *ref_qpsk_new(const double *uw64, size_t uw_len)
{
    return uw64 ? new QPSK(uw64, uw_len) : new QPSK(QPSK::DEFAULT_UNIQUE_WORD, uw_len);
}
void ref_qpsk_free(void *h) { delete (QPSK *)h; }
void ref_qpsk_modulate(void *h, uint8_t *sym, size_t len, double *out_re, double *out_im, size_t len_mod)
{
    ((QPSK *)h)->modulate(sym, len, out_re, out_im, len_mod);
}
size_t ref_qpsk_demodulate(void *h, const double *iq, size_t n, double *out_re, double *out_im)
{
    BBP_Block in(n ? n : 1), out(n ? n : 1);
    load_block(in, iq, n);
    ((QPSK *)h)->demodulate(&in, &out);
    for (size_t i = 0; i < out.n_points; i++) {
        out_re[i] = out.points[i].real();
        if (out_im) out_im[i] = out.points[i].imag();
    }
    return out.n_points;
}
// stage-wise driver: which = 0 decode (iq,n -> re,im; returns n/4), 1 timing, 2 phase, 3 ambiguity
size_t ref_qpsk_stage(void *h, int which, const double *iq, double *re, double *im, size_t n)
{
    QPSK *q = (QPSK *)h;
    if (which == 0) {
        BBP_Block in(n ? n : 1);
        load_block(in, iq, n);
        q->m_length = n / 4;
        q->m_RealData = re;
        q->m_ImagData = im;
        return q->decode_pulseShape(&in);
    }
    q->m_length = n;
    q->m_RealData = re;
    q->m_ImagData = im;
    if (which == 1) q->symbol_timing_synchronization();
    else if (which == 2) q->phase_shift_correction();
    else q->phase_ambiguity_correction();
    return q->m_length;
}
void ref_qpsk_state(void *h, double *s, int32_t *cc)
{
    QPSK *b = (QPSK *)h;
    int k = 0;
    s[k++] = b->NcoControlVariables.delay;
    s[k++] = b->Register_mu;
    s[k++] = b->Register_Q;
    s[k++] = b->Register_I;
    s[k++] = b->SymbolTimingSynchronizationVariables.muDelay;
    const QPSK::InterpData *ip[2] = {&b->Interp_Q_Channel, &b->Interp_I_Channel};
    for (int c = 0; c < 2; c++) {
        s[k++] = ip[c]->delay1; s[k++] = ip[c]->delay2; s[k++] = ip[c]->delay3;
        s[k++] = ip[c]->delay4; s[k++] = ip[c]->delay5;
    }
    s[k++] = b->GardnerDetectorVariables.delay1; s[k++] = b->GardnerDetectorVariables.delay2;
    s[k++] = b->GardnerDetectorVariables.delay3; s[k++] = b->GardnerDetectorVariables.delay4;
    s[k++] = b->LoopFilterVariables.delay_x; s[k++] = b->LoopFilterVariables.delay_y;
    s[k++] = b->PhaseCorrectionVariables.dds_delay;
    s[k++] = b->PhaseLoopFilterVariables.delay_x; s[k++] = b->PhaseLoopFilterVariables.delay_y;
    cc[0] = b->m_phase_ambiguity_cc_000.value();
    cc[1] = b->m_phase_ambiguity_cc_090.value();
    cc[2] = b->m_phase_ambiguity_cc_180.value();
    cc[3] = b->m_phase_ambiguity_cc_270.value();
}

// ---- crc8 / PacketFramer over a RingBuffer, driven like RtlSdr_Receiver::deserializeData
// (rtlsdr_receiver.cpp:2197-2230; that loop is Qt application code and is restated here as the harness)
uint8_t ref_crc8(uint8_t *message, size_t length)
{
    crc8 c;
    return c.getCRC(message, length);
}

int ref_frame(uint8_t *data, size_t length, int crc_enable, uint8_t *bits_out)
{
    PacketFramer<uint8_t> framer;
    framer.checkCRC(crc_enable != 0);
    BBP_Block block(PacketFramer<uint8_t>::MAX_PKT_SIZE_BITS);
    block.n_points = 0;
    int r = framer.serialize(data, length, &block);
    if (r > 0)
        for (int i = 0; i < r; i++) bits_out[i] = (uint8_t)block.points[i].real();
    return r;
}

void ref_deframe(const uint8_t *bits, size_t n, int crc_enable, size_t max_events, int64_t *ev_pos, int32_t *ev_kind,
                 uint8_t *ev_len, uint8_t *ev_crc, uint8_t *payload, uint64_t *summary)
{
    PacketFramer<uint8_t> framer;
    framer.checkCRC(crc_enable != 0);
    RingBuffer<uint8_t> rb((int)n + 1);
    rb.append(bits, n);
    uint8_t pkt[256];
    uint64_t nev = 0, zero_returns = 0;
    size_t front = 0; // bits removed so far
    while (rb.count() > PacketFramer<uint8_t>::MAX_PKT_SIZE_BITS) {
        bool vp = false, vc = false, vh = false;
        int bad = 0;
        size_t before = rb.count();
        int r = framer.deserialize(rb, pkt, 250, vp, vc, vh, bad);
        front += before - rb.count();
        if (r > 0 || r == -1) {
            uint8_t len = PacketFramer<uint8_t>::ToByte(rb, PacketFramer<uint8_t>::LENGTH_POS);
            uint8_t crc = PacketFramer<uint8_t>::ToByte(rb, PacketFramer<uint8_t>::CRC_POS);
            if (nev < max_events) {
                ev_pos[nev] = (int64_t)front;
                ev_kind[nev] = r > 0 ? 1 : (len > 250 ? 2 : 3);
                ev_len[nev] = len;
                ev_crc[nev] = (r > 0 || len <= 250) ? crc : 0;
                if (r > 0) memcpy(payload + 250 * nev, pkt, 250);
            }
            nev++;
            size_t k = r > 0 ? (size_t)r * 8 : 1;
            rb.remove(k);
            front += k;
        } else {
            zero_returns++;
        }
    }
    summary[0] = nev;
    summary[1] = front;
    summary[2] = zero_returns;
}

// ---- the receiver's ones/zeros branch (rtlsdr_receiver.cpp:2063-2074) over the reference's confidence_counter ----
int32_t ref_ones_zeros_confidence(const double *re, const double *im, size_t n, int32_t max, int32_t low, int32_t high,
                                  int32_t start)
{
    confidence_counter cc(max, low, high); // starts at max / 2
    cc.decrement(max);
    cc.increment(start);
    for (size_t i = 0; i < n; i++) {
        if (sqrt(re[i] * re[i] + im[i] * im[i]) >= AM<RADIO_DATA_TYPE>::THRESHOLD) cc.increment(1);
        else cc.decrement(1);
    }
    return cc.value();
}

// ---- DigitalSignalProcessing primitives ----
void ref_dc_block(double *in, double *out, size_t n, double R) { DC_Filter f(R); f.update(in, out, n); }
void *ref_lowpass1_new(double cutoff, double dt) { return new LowPassFilter(cutoff, dt); }
void ref_lowpass1_free(void *h) { delete (LowPassFilter *)h; }
void ref_lowpass1(void *h, const double *in, double *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = ((LowPassFilter *)h)->update(in[i]);
}
// The delay line of NotchFilter::filter is function-static: one per process, never reset.
void ref_notch(double notch_bw, double center_freq, double *block, size_t n) { NotchFilter f(notch_bw, center_freq); f.filter(block, n); }
void ref_integrate(double sample_rate, double *a, size_t n) { FirstOrderDigitalIntegration f(sample_rate); f.integrate(a, n); }
double ref_normalize(double *block, size_t n, double prev_max) { return normalize(block, n, prev_max); }
void *ref_fifth_new() { return new fifth_order_filter(); }
void ref_fifth_free(void *h) { delete (fifth_order_filter *)h; }
size_t ref_fifth_decimate(void *h, double *data, size_t n, size_t by) { return ((fifth_order_filter *)h)->decimate(data, n, by); }
// BBP_Block overload: the block holds `cap` points, of which n_points are valid; returns block->n_points
size_t ref_fifth_decimate_block(void *h, double *iq, size_t cap, size_t n_points, size_t by)
{
    BBP_Block blk(cap);
    for (size_t i = 0; i < cap; i++) blk.points[i] = Complex(iq[2 * i], iq[2 * i + 1]);
    blk.n_points = n_points;
    ((fifth_order_filter *)h)->decimate(&blk, by);
    for (size_t i = 0; i < cap; i++) iq[2 * i] = blk.points[i].real(), iq[2 * i + 1] = blk.points[i].imag();
    return blk.n_points;
}
void ref_fft(double *iq, size_t n)
{
    Complex_Array a(n);
    for (size_t i = 0; i < n; i++) a[i] = Complex(iq[2 * i], iq[2 * i + 1]);
    fft(a, n);
    for (size_t i = 0; i < n; i++) iq[2 * i] = a[i].real(), iq[2 * i + 1] = a[i].imag();
}

} // extern "C"
