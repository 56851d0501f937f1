// api.hip -- the C ABI of libsdrx.so (include/sdrx.h): device layer, host (drop-in) layer with
// pinned staging + async copies, and the reference's legacy extern "C" GPU symbols.
#include "sdrx_internal.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>
#include <vector>

namespace sdrx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

double am_threshold_sq()
{
    static const double thr2 = [] {
        double s = 80.0 * 80.0;
        for (;;) { // IEEE sqrt is monotone: walk down to the first double whose rounded root is below 80
            double below = std::nextafter(s, 0.0);
            if (std::sqrt(below) >= 80.0) s = below;
            else break;
        }
        return s;
    }();
    return thr2;
}

thread_local LastKernel t_last_kernel;

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

static int kernel_family(const char *v) { return !v ? 0 : (strcmp(v, "tile") == 0 ? 1 : (strcmp(v, "systolic") == 0 ? 2 : 0)); }

Tuning &tuning()
{
    static Tuning t = [] { // the environment is read here, once; never in a launch path
        Tuning d, x;
        x.fir_r = env_int("SDRX_FIR_R", d.fir_r);
        x.fir_kernel = kernel_family(getenv("SDRX_FIR_KERNEL"));
        x.fir_waves = env_int("SDRX_FIR_WAVES", d.fir_waves);
        x.fir_lds_pad_kb = env_int("SDRX_FIR_LDS_PAD_KB", d.fir_lds_pad_kb);
        x.fir_nt = env_int("SDRX_FIR_NT", d.fir_nt);
        x.fir_lds_soft_kb = env_int("SDRX_FIR_LDS_SOFT_KB", d.fir_lds_soft_kb);
        x.fir_ct = env_int("SDRX_FIR_CT", d.fir_ct);
        x.fir_amv = env_int("SDRX_FIR_AMV", d.fir_amv);
        x.amv_stats = env_int("SDRX_AMV_STATS", d.amv_stats);
        x.amv_nb = env_int("SDRX_AMV_NB", d.amv_nb);
        x.amv_dbg = env_int("SDRX_AMV_DBG", d.amv_dbg);
        x.amv_bytes = env_int("SDRX_AMV_BYTES", d.amv_bytes);
        x.amv_tpw = env_int("SDRX_AMV_TPW", d.amv_tpw);
        x.fir_fill = env_int("SDRX_FIR_FILL", d.fir_fill);
        x.psk_split = env_int("SDRX_PSK_SPLIT", d.psk_split);
        x.psk_overlap = env_int("SDRX_PSK_OVERLAP", d.psk_overlap);
        x.psk_pieces = env_int("SDRX_PSK_PIECES", d.psk_pieces);
        x.fsk_rows = env_int("SDRX_FSK_ROWS", d.fsk_rows);
        x.iir_cw = env_int("SDRX_IIR_CW", d.iir_cw);
        x.iir_pair = env_int("SDRX_IIR_PAIR", d.iir_pair);
        x.fft_nt = env_int("SDRX_FFT_NT", d.fft_nt);
        x.fft_tile = env_int("SDRX_FFT_TILE", d.fft_tile);
        x.fft_ct = env_int("SDRX_FFT_CT", d.fft_ct);
        return x;
    }();
    return t;
}

// Pinned host staging + device mirror, grown on demand, owned by one host-layer object.
struct Staging {
    void *host = nullptr;
    void *dev = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SDRX_OK;
        release();
        size_t want = bytes + bytes / 4 + 4096;
        SDRX_HIP(hipHostMalloc(&host, want, hipHostMallocDefault));
        SDRX_HIP(hipMalloc(&dev, want));
        cap = want;
        return SDRX_OK;
    }
    void release()
    {
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        host = dev = nullptr;
        cap = 0;
    }
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// memcpy between caller memory and pinned staging; large blocks are split over a few threads (one
// core moves about 10 GB/s, well under what the PCIe link then carries)
static void staged_copy(void *dst, const void *src, size_t bytes)
{
    const size_t piece = (size_t)8 << 20;
    unsigned nt = (unsigned)(bytes / piece);
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned cap = hw ? (hw < 8 ? hw : 8) : 4;
    if (nt > cap) nt = cap;
    if (nt < 2) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 63) & ~(size_t)63;
    for (unsigned t = 0; t < nt; t++) {
        const size_t lo = (size_t)t * per, hi = (t + 1 == nt) ? bytes : ((size_t)(t + 1) * per < bytes ? (size_t)(t + 1) * per : bytes);
        if (lo >= hi) break;
        th.emplace_back([=] { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (auto &x : th) x.join();
}

} // namespace sdrx

using namespace sdrx;

static int noalloc_setup(unsigned char **out); // below, beside the static scratch array

// =========================================================================== library

extern "C" int sdrx_init(int device)
{
    SDRX_HIP(hipSetDevice(device));
    SDRX_HIP(hipFree(nullptr));
    (void)am_threshold_sq();
    (void)tuning();
    // Everything a later call would look up on first use, resolved now: a caller that runs sdrx_init() first can make its
    // first sdrx_psk_demod_noalloc_dev call under stream capture (no symbol lookup inside the capture).  A lookup that
    // fails here is reported by the entry point that needs it, not by sdrx_init: the rest of the library does not use them.
    (void)psk_mf_taps_dev();
    (void)fft_warm_tables(); // (sdrx_fft_dev under capture: its tables exist)
    (void)noalloc_setup(nullptr); // the static scratch's address on this device
    (void)hipGetLastError();
    g_err[0] = 0;
    return SDRX_OK;
}

extern "C" int sdrx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *sdrx_last_error(void) { return g_err; }
extern "C" const char *sdrx_version(void) { return "sdrx 0.2 (gfx950)"; }

extern "C" int sdrx_tune(const char *name, const char *value)
{
    if (!name) return SDRX_EINVAL;
    Tuning &t = tuning();
    const Tuning d;
    const bool dflt = !value || !*value;
    const int v = dflt ? 0 : atoi(value);
    if (strcmp(name, "SDRX_FIR_R") == 0) t.fir_r = dflt ? d.fir_r : v;
    else if (strcmp(name, "SDRX_FIR_KERNEL") == 0) t.fir_kernel = dflt ? 0 : kernel_family(value);
    else if (strcmp(name, "SDRX_FIR_WAVES") == 0) t.fir_waves = dflt ? d.fir_waves : v;
    else if (strcmp(name, "SDRX_FIR_LDS_PAD_KB") == 0) t.fir_lds_pad_kb = dflt ? d.fir_lds_pad_kb : v;
    else if (strcmp(name, "SDRX_FIR_NT") == 0) t.fir_nt = dflt ? d.fir_nt : v;
    else if (strcmp(name, "SDRX_FIR_LDS_SOFT_KB") == 0) t.fir_lds_soft_kb = dflt ? d.fir_lds_soft_kb : v;
    else if (strcmp(name, "SDRX_FIR_CT") == 0) t.fir_ct = dflt ? d.fir_ct : v;
    else if (strcmp(name, "SDRX_FIR_AMV") == 0) t.fir_amv = dflt ? d.fir_amv : v;
    else if (strcmp(name, "SDRX_AMV_STATS") == 0) t.amv_stats = dflt ? d.amv_stats : v;
    else if (strcmp(name, "SDRX_AMV_NB") == 0) t.amv_nb = dflt ? d.amv_nb : v;
    else if (strcmp(name, "SDRX_AMV_DBG") == 0) t.amv_dbg = dflt ? d.amv_dbg : v;
    else if (strcmp(name, "SDRX_AMV_BYTES") == 0) t.amv_bytes = dflt ? d.amv_bytes : v;
    else if (strcmp(name, "SDRX_AMV_TPW") == 0) t.amv_tpw = dflt ? d.amv_tpw : v;
    else if (strcmp(name, "SDRX_FIR_FILL") == 0) t.fir_fill = dflt ? d.fir_fill : v;
    else if (strcmp(name, "SDRX_PSK_SPLIT") == 0) t.psk_split = dflt ? d.psk_split : v;
    else if (strcmp(name, "SDRX_PSK_OVERLAP") == 0) t.psk_overlap = dflt ? d.psk_overlap : v;
    else if (strcmp(name, "SDRX_PSK_PIECES") == 0) t.psk_pieces = dflt ? d.psk_pieces : v;
    else if (strcmp(name, "SDRX_FSK_ROWS") == 0) t.fsk_rows = dflt ? d.fsk_rows : v;
    else if (strcmp(name, "SDRX_IIR_CW") == 0) t.iir_cw = dflt ? d.iir_cw : v;
    else if (strcmp(name, "SDRX_IIR_PAIR") == 0) t.iir_pair = dflt ? d.iir_pair : v;
    else if (strcmp(name, "SDRX_FFT_NT") == 0) t.fft_nt = dflt ? d.fft_nt : v;
    else if (strcmp(name, "SDRX_FFT_TILE") == 0) t.fft_tile = dflt ? d.fft_tile : v;
    else if (strcmp(name, "SDRX_FFT_CT") == 0) t.fft_ct = dflt ? d.fft_ct : v;
    else {
        set_error("sdrx_tune: unknown switch %s", name);
        return SDRX_EINVAL;
    }
    return SDRX_OK;
}

extern "C" const char *sdrx_last_kernel(void)
{
    static thread_local char buf[1536];
    static const char *const lay[] = {"REAL", "IQ_SPLIT", "IQ_INTERLEAVED", "IQ_U8"};
    static const char *const epi[] = {"EPI_STORE", "EPI_AM", "EPI_DISC"};
    const LastKernel &l = t_last_kernel;
    size_t at = 0;
    buf[0] = 0;
    for (int i = 0; i < l.n && at < sizeof(buf) - 1; i++) {
        const KernelNote &k = l.k[i];
        char one[160];
        if (strcmp(k.family, "fir_systolic_kernel") == 0)
            snprintf(one, sizeof(one), "fir_systolic_kernel<D=%d, %s, %s%s>", k.a, lay[k.b & 3], epi[k.c % 3], k.d ? ", fma" : "");
        else if (strcmp(k.family, "fir_tile_kernel") == 0)
            snprintf(one, sizeof(one), "fir_tile_kernel<R=%d, %s, %s%s>", k.a, lay[k.b & 3], epi[k.c % 3], k.d ? ", fma" : "");
        else if (strcmp(k.family, "fir_tile_kernel_ct129x4") == 0)
            snprintf(one, sizeof(one), "fir_tile_kernel<R=%d, %s, %s, T=129, D=4 compile-time>", k.a, lay[k.b & 3], epi[k.c % 3]);
        else if (strcmp(k.family, "fir_tile_kernel_ct64x1") == 0)
            snprintf(one, sizeof(one), "fir_tile_kernel<R=%d, %s, %s, T=64, D=1 compile-time>", k.a, lay[k.b & 3], epi[k.c % 3]);
        else if (strncmp(k.family, "fir_amv_", 8) == 0)
        {
            const bool mm = strncmp(k.family, "fir_amv_mfma", 12) == 0, i8 = strncmp(k.family, "fir_amv_i8", 10) == 0;
            const bool by = strncmp(k.family, "fir_amv_i8_bytes", 16) == 0;
            snprintf(one, sizeof(one), "%s<%s=%d, %s, EPI_AM verified, %s%s>", by ? "fir_amv_i8_bytes_kernel" : (i8 ? "fir_amv_i8_kernel" : (mm ? "fir_amv_mfma_kernel" : "fir_amv_kernel")),
                     mm || i8 ? "NB" : "R", k.a, lay[k.b & 3],
                     i8 ? "int8 mfma 16x16x64 on 8-bit integer tiles (fixed-point taps), fp32 mfma 16x16x4 on the others" : (mm ? "fp32 mfma 16x16x4" : "fp32 fma"),
                     strstr(k.family, "_ct64") ? ", T=64 compile-time" : "");
        }
        else if (strncmp(k.family, "psk_loops", 9) == 0)
            snprintf(one, sizeof(one), "%s<%s>", k.family, k.a == SDRX_BPSK ? "BPSK" : "QPSK");
        else if (strcmp(k.family, "psk_ambiguity_kernel") == 0)
            snprintf(one, sizeof(one), "psk_ambiguity_kernel<%d>", k.b);
        else
            snprintf(one, sizeof(one), "%s", k.family);
        at += (size_t)snprintf(buf + at, sizeof(buf) - at, "%s%s", i ? " -> " : "", one);
    }
    return buf;
}

// =========================================================================== 1. device layer

extern "C" size_t sdrx_fir_state_doubles(int num_taps) { return num_taps > 0 ? (size_t)(2 * (num_taps - 1) + 2) : 2; }
extern "C" int sdrx_fir_dev(const sdrx_fir_job *job) { return launch_fir(job, EPI_STORE); }
extern "C" int sdrx_fir_am_dev(const sdrx_fir_job *job) { return launch_fir(job, EPI_AM); }
extern "C" int sdrx_fir_fmdisc_dev(const sdrx_fir_job *job) { return launch_fir(job, EPI_DISC); }

extern "C" int sdrx_fir_am_verify_stats(uint64_t out[4], int reset)
{
    unsigned long long *d = f32::amv_stats_dev();
    if (!d) return SDRX_EHIP;
    SDRX_HIP(hipDeviceSynchronize());
    if (out) SDRX_HIP(hipMemcpy(out, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) { // (hipMemset on device memory may return before the fill has run, and it runs on the null stream, which
        // the callers' non-blocking streams do not wait for: wait here)
        SDRX_HIP(hipMemsetAsync(d, 0, 4 * sizeof(uint64_t), nullptr));
        SDRX_HIP(hipStreamSynchronize(nullptr));
    }
    return SDRX_OK;
}

extern "C" int sdrx_rtlfm_dev(const uint8_t *iq_u8, size_t n, size_t nchan, size_t in_pitch, int downsample, int rotate90,
                              const int32_t *state_in, int32_t *state_out, int16_t *pcm, size_t out_pitch, void *stream)
{
    return launch_rtlfm(iq_u8, n, nchan, in_pitch, downsample, rotate90, state_in, state_out, pcm, out_pitch,
                        (hipStream_t)stream);
}

extern "C" int sdrx_rtlfm_full_dev(const sdrx_rtlfm_params *params, const uint8_t *iq_u8, size_t n, size_t nchan,
                                   size_t in_pitch, const int32_t *state_in, int32_t *state_out, int32_t *post_state,
                                   int64_t *work, int16_t *pcm, size_t out_pitch, void *stream)
{
    return launch_rtlfm_full(params, iq_u8, n, nchan, in_pitch, state_in, state_out, post_state, work, pcm, out_pitch,
                             (hipStream_t)stream);
}

extern "C" int sdrx_rtlfm_low_pass_real_dev(int16_t *pcm, size_t nchan, size_t out_pitch, int rate_out, int rate_out2,
                                            int32_t *post_state, int32_t *lpr_state, void *stream)
{
    return launch_rtlfm_low_pass_real(pcm, nchan, out_pitch, rate_out, rate_out2, post_state, lpr_state, (hipStream_t)stream);
}

extern "C" int sdrx_deframe_dev(const uint8_t *bits, size_t n, size_t nchan, size_t in_pitch, int crc_enable,
                                sdrx_frame_event *events, uint8_t *payloads, size_t max_events, uint64_t *summary, void *stream)
{
    return launch_deframe(bits, n, nchan, in_pitch, crc_enable, events, payloads, max_events, summary, (hipStream_t)stream);
}

extern "C" int sdrx_am_demod_dev(const double *re, const double *im, uint8_t *out, size_t n, size_t ds, void *stream)
{
    return launch_am_demod(re, im, out, n, ds, (hipStream_t)stream);
}
extern "C" int sdrx_ones_zeros_dev(const double *re, const double *im, size_t n, int32_t *count, void *stream)
{
    return launch_ones_zeros(re, im, n, count, (hipStream_t)stream);
}
extern "C" int sdrx_ones_zeros_confidence_dev(const double *re, const double *im, size_t n, size_t nchan, size_t pitch,
                                              int32_t max, int32_t *counters, void *stream)
{
    return launch_ones_zeros_confidence(re, im, n, nchan, pitch, max, counters, (hipStream_t)stream);
}
extern "C" int sdrx_fm_demod_dev(const double *re, const double *w1, const double *w2, uint8_t *out, size_t n,
                                 size_t ds, void *stream)
{
    return launch_fm_demod(re, w1, w2, out, n, ds, (hipStream_t)stream);
}

extern "C" int sdrx_am_modulate_dev(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, double *out_real,
                                    size_t len_mod, size_t out_pitch, void *stream)
{
    return launch_am_modulate(bits, len, nchan, in_pitch, out_real, len_mod, out_pitch, (hipStream_t)stream);
}
extern "C" int sdrx_fm_modulate_dev(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, const double *w_mark,
                                    const double *w_space, size_t spb, double *out_real, size_t out_pitch, void *stream)
{
    return launch_fm_modulate(bits, len, nchan, in_pitch, w_mark, w_space, spb, out_real, out_pitch, (hipStream_t)stream);
}
extern "C" int sdrx_fm_convolve_dev(const double *x, size_t len1, const double *w, size_t len2, double *out, size_t ds,
                                    void *stream)
{
    return launch_fm_convolve(x, len1, w, len2, out, ds, (hipStream_t)stream);
}

extern "C" void sdrx_fm_tables(double sample_rate, double mark, double space, size_t n, int for_modulator, double *w_mark,
                               double *w_space)
{
    const double PI = 3.141592653589793238460; // Common_Deffinitions.h:14
    const double delta_t = 1.0 / sample_rate;  // FM.h:46, FM.h:125
    if (for_modulator) {                       // FM.h:47-57: RootRaised * cos(2 PI f delta_t i)
        const double root_raised = sqrt(2) / sqrt(n);
        for (size_t i = 0; i < n; i++) {
            w_mark[i] = root_raised * cos(2.0 * PI * mark * delta_t * i);
            w_space[i] = root_raised * cos(2.0 * PI * space * delta_t * i);
        }
    } else { // FM.h:127-131 (and fm.cu:62-68): sqrt(2/ds) * cos(2 PI f i delta_t) -- another operation order
        for (size_t i = 0; i < n; i++) {
            w_mark[i] = sqrt(2.0 / n) * cos(2.0 * PI * mark * i * delta_t);
            w_space[i] = sqrt(2.0 / n) * cos(2.0 * PI * space * i * delta_t);
        }
    }
}

extern "C" const double *sdrx_rrc_coeff(void) { return RRC_COEFF; }

// ---- PSK state (host arithmetic only: constructor logic of BPSK.cpp:52-90 / QPSK.cpp:66-120)
static void psk_state_common(sdrx_psk_state *s, size_t uw_len)
{
    memset(s, 0, sizeof(*s));
    s->g1 = s->g2 = s->g3 = s->g4 = 1.0;          // BPSK.cpp:68-71
    for (int i = 0; i < 4; i++) s->cc[i] = 10 / 2; // confidence_counter(10,5,5) starts at max/2
    s->uw_size = (int32_t)uw_len;
}

extern "C" void sdrx_psk_state_init_bpsk(sdrx_psk_state *s, uint64_t unique_word, size_t uw_len)
{
    psk_state_common(s, uw_len);
    if (uw_len > 64) return; // BPSK.cpp:79
    for (int i = (int)uw_len - 1; i >= 0; i--) {
        uint64_t mask = 1ULL << i;
        s->uw[0][i] = ((unique_word & mask) > 0) ? 1 : 0;
        s->uw[1][i] = ((unique_word & mask) > 0) ? 0 : 1;
    }
}

static const double QPSK_DEFAULT_UW[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                           0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                           0, 0, 0, 0, 3, 0, 2, 1, 0, 3, 2, 1, 2, 0, 1, 2, 3, 2, 2, 3}; // QPSK.cpp:20-28

extern "C" const double *sdrx_qpsk_default_unique_word(void) { return QPSK_DEFAULT_UW; }

extern "C" void sdrx_psk_state_init_qpsk(sdrx_psk_state *s, const double *uw64, size_t uw_len)
{
    const double *dflt = QPSK_DEFAULT_UW;
    static const double l000[4] = {0, 1, 2, 3}, l090[4] = {2, 0, 3, 1}, l180[4] = {3, 2, 1, 0}, l270[4] = {1, 3, 0, 2};
    psk_state_common(s, uw_len);
    if (!uw64) uw64 = dflt;
    if (uw_len > 64) return; // QPSK.cpp:96
    for (int i = 63; i >= 64 - (int)uw_len && i >= 0; i--) {
        if (uw64[i] >= 0 && uw64[i] <= 3) {
            const int v = (int)uw64[i];
            s->uw[0][i] = l000[v];
            s->uw[1][i] = l270[v]; // uw_090 <- LUT_REAL_270, QPSK.cpp:107
            s->uw[2][i] = l180[v];
            s->uw[3][i] = l090[v]; // uw_270 <- LUT_REAL__90, QPSK.cpp:109
        }
    }
}

static const size_t MF_T = 129, MF_STATE = 2 * (MF_T - 1) + 2;

extern "C" size_t sdrx_psk_work_bytes(size_t n, size_t nchan)
{
    // matched-filter output | two state-row areas no kernel touches any more (the matched filter is a one-shot FIR job;
    // kept so that the size callers allocate, and the layout behind it, stay what they were) | the carrier loop's
    // call-locals (block pipeline)
    return align256(nchan * (n / 4) * 2 * sizeof(double)) + 2 * align256(nchan * MF_STATE * sizeof(double)) +
           align256(nchan * 4 * sizeof(double));
}

// One piece of decode_pulseShape: outputs [0, n_out) of the block that starts at iq, n_before samples of the same
// channels in front of it (0: the call's start, zero history: BPSK.cpp:267-268 re-pads with zeros on every call).
static int mf_launch(const double *iq, size_t n, size_t nchan, size_t in_pitch, double *mf_iq, size_t mf_pitch,
                     double *zero_state, double *dummy_state, hipStream_t stream, int *offs_buf = nullptr,
                     unsigned offs_cap = 0, size_t n_before = 0, size_t n_out = (size_t)-1)
{
    const double *taps = psk_mf_taps_dev();
    if (!taps) {
        set_error("matched-filter taps lookup failed");
        return SDRX_EHIP;
    }
    if (n_out == (size_t)-1) n_out = n / 4; // memcpy of number_of_points/4 values: BPSK.cpp:292-293
    if (n_out == 0) return SDRX_OK;
    (void)zero_state; // a job without state rows starts from a zero history and keeps none (BPSK.cpp:267-268 re-pads
    (void)dummy_state; // with zeros on every call): no rows to clear before, none written after
    sdrx_fir_job j;
    memset(&j, 0, sizeof(j));
    j.layout = SDRX_IQ_INTERLEAVED;
    j.in0 = iq;
    j.n = n;
    j.n_before = n_before;
    j.nchan = nchan;
    j.in_pitch = in_pitch;
    j.taps = taps;
    j.num_taps = (int)MF_T;
    j.decim = 4; // decimate(.., 4): BPSK.cpp:289-290
    j.state_in = nullptr;
    j.state_out = nullptr;
    j.state_pitch = MF_STATE;
    j.out0 = mf_iq;
    j.out1 = nullptr;
    j.out_pitch = mf_pitch;
    j.n_out = n_out;
    j.stream = stream;
    return launch_fir(&j, EPI_STORE, offs_buf, offs_cap);
}

// ---- block pipeline of the demodulator.  The matched filter is bound by fp64 VALU throughput on every SIMD; the two
// loops are 2 waves per 64 channels, each bound by its own instruction latency (a bank of 2^14 channels keeps a quarter
// of the chip's wave slots busy, whatever it does).  Run one after the other they leave most of the machine idle in
// turn.  A call is therefore cut in time into PSK_PIECES pieces: the matched filter of piece k+1 (caller's stream;
// its history is the samples in front of it: sdrx_fir_job.n_before) runs beside the loops of piece k (a side stream
// of the library, ordered by events); loop state carries from piece to piece exactly as from call to call.  Same
// kernels, same arithmetic, same order per channel: results are bit-identical to the one-stream form.
static const int PSK_PIECES_MAX = 16;
static const size_t PSK_PIPES_MAX = 64; // caller streams that get a side stream of their own; calls from further ones run on one stream
struct PskPipe {
    hipStream_t side = nullptr;
    hipEvent_t ev[PSK_PIECES_MAX + 1] = {};
    std::mutex mu; // one call's fork / join at a time (two host threads feeding ONE stream is the caller's race, not a crash here)
};
static std::mutex g_pipe_mu;
static std::map<std::pair<int, hipStream_t>, std::unique_ptr<PskPipe>> g_pipe;

// The side stream and events of (current device, caller's stream): created on the first pipelined call from that stream,
// kept until the process ends.  EVERY caller stream has its own: a call never queues behind another stream's call, and the
// only waits the library adds are  side <- event recorded earlier on s  and, at the end,  s <- event recorded earlier on side,
// all enqueued by one host thread in that order.  (Why that cannot deadlock, and what the round-2 hang was: DESIGN.md
// section 4, "block pipeline".)  nullptr: no pipe (the caller then runs the one-stream form).
static PskPipe *psk_pipe_for(hipStream_t s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pipe_mu);
    auto key = std::make_pair(dev, s);
    auto it = g_pipe.find(key);
    if (it != g_pipe.end()) return it->second.get();
    if (g_pipe.size() >= PSK_PIPES_MAX) return nullptr;
    std::unique_ptr<PskPipe> p(new PskPipe);
    bool ok = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess;
    int made = 0;
    for (; ok && made <= PSK_PIECES_MAX; made++) ok = hipEventCreateWithFlags(&p->ev[made], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        for (int k = 0; k < made; k++) (void)hipEventDestroy(p->ev[k]);
        if (p->side) (void)hipStreamDestroy(p->side);
        (void)hipGetLastError();
        return nullptr;
    }
    PskPipe *raw = p.get();
    g_pipe[key] = std::move(p);
    return raw;
}

static int psk_demod_common(int kind, const double *iq, size_t n, size_t nchan, size_t in_pitch, sdrx_psk_state *states,
                            uint8_t *out, size_t out_pitch, unsigned char *w, hipStream_t s, int *offs_buf, unsigned offs_cap,
                            bool may_overlap)
{
    const size_t m = n / 4;
    double *mf = (double *)w;
    double *zst = (double *)(w + align256(nchan * m * 2 * sizeof(double)));
    double *dst = (double *)((unsigned char *)zst + align256(nchan * MF_STATE * sizeof(double)));
    double *carry = (double *)((unsigned char *)dst + align256(nchan * MF_STATE * sizeof(double)));
    KernelChain kc;
    const Tuning &tune = tuning();
    const int want = tune.psk_overlap;
    // Pays only where the loops dominate (few channels, long blocks): there the matched filter hides behind them
    // (round 2's final kernels: 2^13 channels x 2^13 samples 0.79 -> 0.76 ms, 2^12 x 2^14 1.20 -> 1.11, 2^10 x 2^16 3.77 -> 3.69).  Where the two take similar time (2^14 x 2^12) the loop waves,
    // sharing their SIMDs with matched-filter waves, slow down by as much as the overlap saves (0.74 -> 0.89 ms;
    // profiles/r02_cfg3_overlap_trace.json), s_setprio and a high-priority queue notwithstanding: off by default there.
    // Giving each stage CUs of its own (CU-masked queues) is worse still: profiles/r02_cfg3_cumask.json.
    const bool overlap = may_overlap && want != 0 && m >= 512 && (want > 0 || (nchan <= (1u << 13) && m >= 2048));
    // A stream under capture, or one beyond the PSK_PIPES_MAX that have a side stream: the one-stream form (same results).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = overlap && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone);
    PskPipe *pp = nullptr;
    if (!overlap || capturing || !(pp = psk_pipe_for(s))) {
        int rc = mf_launch(iq, n, nchan, in_pitch, mf, m, zst, dst, s, offs_buf, offs_cap);
        if (rc) return rc;
        rc = launch_psk_loops(kind, mf, m, nchan, m, states, out, out_pitch, 1, 1, s);
        if (rc) return rc;
        return launch_psk_ambiguity(kind, out, (m + 1) / 2, nchan, out_pitch, states, s);
    }
    std::lock_guard<std::mutex> lk(pp->mu);
    const int pieces = tune.psk_pieces < 2 ? 2 : (tune.psk_pieces > PSK_PIECES_MAX ? PSK_PIECES_MAX : tune.psk_pieces);
    const size_t m_piece = (((m + pieces - 1) / pieces) + 31) & ~(size_t)31; // whole ring blocks (32 samples), even: /2 keeps 0, 2, ...
    hipStream_t s_mf = s, s_lp = pp->side;
    int k = 0;
    for (size_t m0 = 0; m0 < m; m0 += m_piece, k++) {
        const bool last = m0 + m_piece >= m;
        const size_t mk = last ? m - m0 : m_piece;
        int rc = mf_launch(iq + 2 * 4 * m0, last ? n - 4 * m0 : 4 * mk, nchan, in_pitch, mf + 2 * m0, m, zst, dst, s_mf, offs_buf,
                           offs_cap, 4 * m0, mk);
        if (rc) return rc;
        SDRX_HIP(hipEventRecord(pp->ev[k], s_mf));
        SDRX_HIP(hipStreamWaitEvent(s_lp, pp->ev[k], 0));
        rc = launch_psk_loops(kind, mf + 2 * m0, mk, nchan, m, states, out + m0 / 2, out_pitch, 1, 1, s_lp, carry,
                              (m0 ? 1 : 0) | (last ? 0 : 2));
        if (rc) return rc;
    }
    int rc = launch_psk_ambiguity(kind, out, (m + 1) / 2, nchan, out_pitch, states, s_lp);
    if (rc) return rc;
    SDRX_HIP(hipEventRecord(pp->ev[PSK_PIECES_MAX], s_lp));
    SDRX_HIP(hipStreamWaitEvent(s, pp->ev[PSK_PIECES_MAX], 0)); // everything behind the call on the caller's stream sees its results
    return SDRX_OK;
}

extern "C" int sdrx_psk_demod_dev(int kind, const double *iq, size_t n, size_t nchan, size_t in_pitch,
                                  sdrx_psk_state *states, uint8_t *out, size_t out_pitch, void *work, void *stream)
{
    if (!iq || !states || !out || !work) {
        set_error("sdrx_psk_demod_dev: null pointer");
        return SDRX_EINVAL;
    }
    return psk_demod_common(kind, iq, n, nchan, in_pitch, states, out, out_pitch, (unsigned char *)work, (hipStream_t)stream,
                            nullptr, 0, true);
}

// ---- the one entry point that allocates nothing (no counterpart in the reference; see include/sdrx.h)
//
// sdrx_psk_demod_dev with its `work` argument replaced by a file-scope device array.  Layout of the array:
//   [0, NOALLOC_OFFS_BYTES)   the matched filter's window-offset table (written by a kernel on the call's stream)
//   [NOALLOC_OFFS_BYTES, ...) the workspace sdrx_psk_demod_dev takes: matched-filter output, zero and dummy state rows
// `scratch` below is a host-side stand-in for the symbol, not a device address: every use goes through the
// per-device pointer hipGetSymbolAddress returns (each GPU holds its own copy of the array at its own address).
static __device__ __align__(256) unsigned char scratch[SDRX_NOALLOC_SCRATCH_BYTES];
static const size_t NOALLOC_OFFS_INTS = 1024, NOALLOC_OFFS_BYTES = NOALLOC_OFFS_INTS * sizeof(int);
static const int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// Once per device, with that device current: from sdrx_init(), else from the entry point's first call there.
static int noalloc_setup(unsigned char **out)
{
    int dev = 0;
    SDRX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        set_error("device %d beyond the %d the library keeps a scratch pointer for", dev, NOALLOC_MAX_DEVICES);
        return SDRX_EUNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        unsigned char *p = nullptr;
        SDRX_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) {
            set_error("hipGetSymbolAddress(scratch) returned a null address on device %d", dev);
            return SDRX_EHIP;
        }
        g_scratch_ptr[dev] = p;
    }
    if (out) *out = g_scratch_ptr[dev];
    return SDRX_OK;
}

extern "C" size_t sdrx_psk_demod_noalloc_max_bytes(void) { return SDRX_NOALLOC_SCRATCH_BYTES - NOALLOC_OFFS_BYTES; }

extern "C" int sdrx_psk_demod_noalloc_dev(int kind, const double *iq, size_t n, size_t nchan, size_t in_pitch,
                                          sdrx_psk_state *states, uint8_t *out, size_t out_pitch, void *stream)
{
    if (!iq || !states || !out) {
        set_error("sdrx_psk_demod_noalloc_dev: null pointer");
        return SDRX_EINVAL;
    }
    if (kind != SDRX_BPSK && kind != SDRX_QPSK) {
        set_error("sdrx_psk_demod_noalloc_dev: kind %d", kind);
        return SDRX_EINVAL;
    }
    // nchan * (n / 4) * 16 must not wrap before it is compared with the limit
    const size_t limit = sdrx_psk_demod_noalloc_max_bytes();
    if (nchan > limit || n / 4 > limit || (nchan && (n / 4) > limit / nchan) || sdrx_psk_work_bytes(n, nchan) > limit) {
        set_error("sdrx_psk_demod_noalloc_dev: %zu channels x %zu samples need a workspace beyond the static scratch's "
                  "%zu bytes (use sdrx_psk_demod_dev with a caller workspace)", nchan, n, limit);
        return SDRX_EUNSUPPORTED;
    }
    unsigned char *base = nullptr;
    int rc = noalloc_setup(&base);
    if (rc) return rc;
    // nothing in the array has to start at zero (the matched filter takes its zero history from no rows at all): it
    // keeps the last call's contents and every part is overwritten before it is read.  One stream, no
    // block pipeline: its side stream and events would be created on first use.
    return psk_demod_common(kind, iq, n, nchan, in_pitch, states, out, out_pitch, base + NOALLOC_OFFS_BYTES,
                            (hipStream_t)stream, (int *)base, (unsigned)NOALLOC_OFFS_INTS, false);
}

extern "C" int sdrx_psk_matched_filter_dev(const double *iq, size_t n, size_t nchan, size_t in_pitch, double *mf_iq,
                                           size_t mf_pitch, void *stream)
{
    if (!iq || !mf_iq) return SDRX_EINVAL;
    return mf_launch(iq, n, nchan, in_pitch, mf_iq, mf_pitch, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int sdrx_psk_loops_dev(int kind, double *mf_iq, size_t m, size_t nchan, size_t mf_pitch,
                                  sdrx_psk_state *states, uint8_t *decisions, size_t out_pitch, int run_timing,
                                  int run_phase, void *stream)
{
    return launch_psk_loops(kind, mf_iq, m, nchan, mf_pitch, states, decisions, out_pitch, run_timing, run_phase,
                            (hipStream_t)stream);
}

extern "C" int sdrx_psk_ambiguity_dev(int kind, uint8_t *decisions, size_t m, size_t nchan, size_t out_pitch,
                                      sdrx_psk_state *states, void *stream)
{
    return launch_psk_ambiguity(kind, decisions, m, nchan, out_pitch, states, (hipStream_t)stream);
}

extern "C" int sdrx_psk_modulate_dev(int kind, const uint8_t *sym, size_t len, double *out_re, double *out_im,
                                     size_t zero_prefix, void *stream)
{
    return launch_psk_modulate(kind, sym, len, out_re, out_im, zero_prefix, (hipStream_t)stream);
}

// =========================================================================== 2. host layer

extern "C" int sdrx_fir_design(int type, int two_freq, int num_taps, double Fs, double Fx, double Fu, int max_taps,
                               double *taps)
{
    const double lambda = M_PI * Fx / (Fs / 2); // filt.cpp:52
    if (!two_freq) {                            // filt.cpp:54-67
        if (Fs <= 0) return -1;
        if (Fx <= 0 || Fx >= Fs / 2) return -2;
        if (num_taps <= 0 || num_taps > max_taps) return -3;
        if (type != SDRX_LPF && type != SDRX_HPF) return -5;
        for (int n = 0; n < num_taps; n++) { // filt.cpp:116-120, 131-135
            const double mm = n - (num_taps - 1.0) / 2.0;
            if (type == SDRX_LPF) taps[n] = (mm == 0.0) ? lambda / M_PI : sin(mm * lambda) / (mm * M_PI);
            else taps[n] = (mm == 0.0) ? 1.0 - lambda / M_PI : -sin(mm * lambda) / (mm * M_PI);
        }
        return 0;
    }
    const double phi = M_PI * Fu / (Fs / 2); // filt.cpp:83
    if (Fs <= 0) return -10;                 // filt.cpp:85-99
    if (Fx >= Fu) return -11;
    if (Fx <= 0 || Fx >= Fs / 2) return -12;
    if (Fu <= 0 || Fu >= Fs / 2) return -13;
    if (num_taps <= 0 || num_taps > max_taps) return -14;
    if (type != SDRX_BPF) return -16;
    for (int n = 0; n < num_taps; n++) { // filt.cpp:146-151
        const double mm = n - (num_taps - 1.0) / 2.0;
        taps[n] = (mm == 0.0) ? (phi - lambda) / M_PI : (sin(mm * phi) - sin(mm * lambda)) / (mm * M_PI);
    }
    return 0;
}

struct sdrx_fir {
    int err = 0;
    int T = 0;
    std::vector<double> taps;
    double *d_taps = nullptr;
    double *d_state[2] = {nullptr, nullptr};
    int cur = 0;
    hipStream_t stream = nullptr;
    Staging in, out;
};

static int fir_alloc_device(sdrx_fir *f)
{
    const size_t sd = sdrx_fir_state_doubles(f->T);
    SDRX_HIP(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    SDRX_HIP(hipMalloc((void **)&f->d_taps, f->T * sizeof(double)));
    SDRX_HIP(hipMemcpy(f->d_taps, f->taps.data(), f->T * sizeof(double), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; i++) {
        SDRX_HIP(hipMalloc((void **)&f->d_state[i], sd * sizeof(double)));
        // on the object's own stream, then waited for (a hipMemset would run on the null stream, which a non-blocking stream
        // does not wait for, and may return before the fill has run: the first launch could then read the rows too early)
        SDRX_HIP(hipMemsetAsync(f->d_state[i], 0, sd * sizeof(double), f->stream));
    }
    SDRX_HIP(hipStreamSynchronize(f->stream));
    return SDRX_OK;
}

extern "C" sdrx_fir_t *sdrx_fir_create(int type, int two_freq, int num_taps, double Fs, double Fx, double Fu,
                                       int max_taps)
{
    sdrx_fir *f = new sdrx_fir();
    f->T = num_taps;
    f->taps.assign(num_taps > 0 ? (size_t)num_taps : 1, 0.0);
    f->err = sdrx_fir_design(type, two_freq, num_taps, Fs, Fx, Fu, max_taps, f->taps.data());
    if (f->err == 0 && fir_alloc_device(f) != SDRX_OK) f->err = -4; // "memory allocation ... failed", filt.h:94
    return f;
}

extern "C" void sdrx_fir_destroy(sdrx_fir_t *f)
{
    if (!f) return;
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->d_taps) (void)hipFree(f->d_taps);
    for (int i = 0; i < 2; i++)
        if (f->d_state[i]) (void)hipFree(f->d_state[i]);
    f->in.release();
    f->out.release();
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

extern "C" int sdrx_fir_error_flag(const sdrx_fir_t *f) { return f ? f->err : -4; }

extern "C" void sdrx_fir_get_taps(const sdrx_fir_t *f, double *taps)
{
    if (!f || f->err != 0) return; // filt.cpp:161
    for (int i = 0; i < f->T; i++) taps[i] = f->taps[i];
}

extern "C" void sdrx_fir_reset(sdrx_fir_t *f)
{
    if (!f || f->err != 0) return; // filt.cpp:241
    // on the object's own stream (behind whatever it still runs), and done when this returns: sdrx_fir_execute_batch reads
    // the rows from another object's stream
    for (int i = 0; i < 2; i++) (void)hipMemsetAsync(f->d_state[i], 0, sdrx_fir_state_doubles(f->T) * sizeof(double), f->stream);
    (void)hipStreamSynchronize(f->stream);
}

extern "C" int sdrx_fir_execute(sdrx_fir_t *f, double *buf, size_t n)
{
    if (!f || !buf) return SDRX_EINVAL;
    if (f->err != 0) { // do_sample returns 0 once the error flag is set: filt.cpp:254
        for (size_t i = 0; i < n; i++) buf[i] = 0.0;
        return SDRX_OK;
    }
    if (n == 0) return SDRX_OK;
    const size_t bytes = n * sizeof(double);
    int rc = f->in.ensure(bytes);
    if (rc) return rc;
    rc = f->out.ensure(bytes);
    if (rc) return rc;
    staged_copy(f->in.host, buf, bytes);
    SDRX_HIP(hipMemcpyAsync(f->in.dev, f->in.host, bytes, hipMemcpyHostToDevice, f->stream));
    sdrx_fir_job j;
    memset(&j, 0, sizeof(j));
    j.layout = SDRX_REAL;
    j.in0 = f->in.dev;
    j.n = n;
    j.nchan = 1;
    j.in_pitch = n;
    j.taps = f->d_taps;
    j.num_taps = f->T;
    j.decim = 1;
    j.state_in = f->d_state[f->cur];
    j.state_out = f->d_state[f->cur ^ 1];
    j.state_pitch = sdrx_fir_state_doubles(f->T);
    j.out0 = f->out.dev;
    j.out_pitch = n;
    j.stream = f->stream;
    rc = launch_fir(&j, EPI_STORE);
    if (rc) return rc;
    f->cur ^= 1;
    SDRX_HIP(hipMemcpyAsync(f->out.host, f->out.dev, bytes, hipMemcpyDeviceToHost, f->stream));
    SDRX_HIP(hipStreamSynchronize(f->stream));
    staged_copy(buf, f->out.host, bytes);
    return SDRX_OK;
}

// ---- Filter::execute on many Filter objects at once: one staging block, one upload, ONE launch per filter design
// (the objects keep their own shift registers: rows are gathered into a batch state before the launch and scattered
// back after it by a copy kernel driven by a pointer table)
__global__ void __launch_bounds__(128) fir_state_rows_kernel(double *__restrict__ batch, double *const *__restrict__ table,
                                                             size_t row, size_t pitch, int scatter)
{
    double *obj = table[blockIdx.x];
    double *b = batch + (size_t)blockIdx.x * pitch;
    for (size_t k = threadIdx.x; k < row; k += blockDim.x) {
        if (scatter) obj[k] = b[k];
        else b[k] = obj[k];
    }
}

extern "C" int sdrx_fir_execute_batch(sdrx_fir_t *const *h, double *const *bufs, size_t n, size_t nchan)
{
    if ((!h || !bufs) && nchan) return SDRX_EINVAL;
    for (size_t c = 0; c < nchan; c++)
        if (!h[c] || (!bufs[c] && n)) {
            set_error("sdrx_fir_execute_batch: null filter or buffer at channel %zu", c);
            return SDRX_EINVAL;
        }
    if (n == 0 || nchan == 0) return SDRX_OK;
    std::vector<char> done(nchan, 0);
    for (size_t c = 0; c < nchan; c++) { // an object with its error flag set returns zeros (filt.cpp:254)
        if (h[c]->err != 0) {
            for (size_t i = 0; i < n; i++) bufs[c][i] = 0.0;
            done[c] = 1;
        }
        for (size_t d = 0; d < c; d++)
            if (h[d] == h[c]) {
                set_error("sdrx_fir_execute_batch: filter object listed twice (channels %zu and %zu)", d, c);
                return SDRX_EINVAL;
            }
    }
    std::vector<size_t> grp;
    for (size_t lead = 0; lead < nchan; lead++) {
        if (done[lead]) continue;
        sdrx_fir *f0 = h[lead];
        grp.clear();
        for (size_t c = lead; c < nchan; c++) // every object of the same design (same taps, bit for bit) goes in this launch
            if (!done[c] && h[c]->T == f0->T && memcmp(h[c]->taps.data(), f0->taps.data(), (size_t)f0->T * sizeof(double)) == 0) {
                grp.push_back(c);
                done[c] = 1;
            }
        const size_t g = grp.size(), pitch = sdrx_fir_state_doubles(f0->T);
        const size_t samp_bytes = align256(g * n * sizeof(double)), tab_bytes = align256(2 * g * sizeof(double *)),
                     st_bytes = align256(g * pitch * sizeof(double));
        int rc = f0->in.ensure(samp_bytes + tab_bytes + 2 * st_bytes);
        if (rc) return rc;
        rc = f0->out.ensure(samp_bytes);
        if (rc) return rc;
        char *hin = (char *)f0->in.host, *din = (char *)f0->in.dev;
        double **tab = (double **)(hin + samp_bytes); // [0, g): state_in rows of the objects; [g, 2g): their state_out rows
        for (size_t k = 0; k < g; k++) {
            sdrx_fir *f = h[grp[k]];
            staged_copy(hin + k * n * sizeof(double), bufs[grp[k]], n * sizeof(double));
            tab[k] = f->d_state[f->cur];
            tab[g + k] = f->d_state[f->cur ^ 1];
        }
        hipStream_t st = f0->stream;
        SDRX_HIP(hipMemcpyAsync(din, hin, samp_bytes + tab_bytes, hipMemcpyHostToDevice, st));
        double **dtab = (double **)(din + samp_bytes);
        double *bst_in = (double *)(din + samp_bytes + tab_bytes), *bst_out = (double *)(din + samp_bytes + tab_bytes + st_bytes);
        hipLaunchKernelGGL(fir_state_rows_kernel, dim3((unsigned)g), dim3(128), 0, st, bst_in, dtab, pitch, pitch, 0);
        sdrx_fir_job j;
        memset(&j, 0, sizeof(j));
        j.layout = SDRX_REAL;
        j.in0 = din;
        j.n = n;
        j.nchan = g;
        j.in_pitch = n;
        j.taps = f0->d_taps;
        j.num_taps = f0->T;
        j.decim = 1;
        j.state_in = bst_in;
        j.state_out = bst_out;
        j.state_pitch = pitch;
        j.out0 = f0->out.dev;
        j.out_pitch = n;
        j.stream = st;
        rc = launch_fir(&j, EPI_STORE);
        if (rc) return rc;
        hipLaunchKernelGGL(fir_state_rows_kernel, dim3((unsigned)g), dim3(128), 0, st, bst_out, dtab + g, pitch, pitch, 1);
        SDRX_HIP(hipGetLastError());
        SDRX_HIP(hipMemcpyAsync(f0->out.host, f0->out.dev, g * n * sizeof(double), hipMemcpyDeviceToHost, st));
        SDRX_HIP(hipStreamSynchronize(st));
        for (size_t k = 0; k < g; k++) {
            h[grp[k]]->cur ^= 1;
            staged_copy(bufs[grp[k]], (char *)f0->out.host + k * n * sizeof(double), n * sizeof(double));
        }
    }
    return SDRX_OK;
}

extern "C" double sdrx_fir_do_sample(sdrx_fir_t *f, double x)
{
    double v = x;
    if (sdrx_fir_execute(f, &v, 1) != SDRX_OK) return 0.0;
    return v;
}

// ---- a ring of chunk launches of ONE stream, captured once in a HIP graph (device layer; see sdrx.h)
struct sdrx_fir_ring {
    sdrx_fir_job job; // chunk 0
    FirEpilogue epi = EPI_STORE;
    size_t chunks = 0, in_stride = 0, out_stride = 0;
    int mode = 1;            // 0: chunk launches issued one by one, 1: one launch over the whole (contiguous) ring, 2: HIP graph
    int parity = 0;          // which of the two state buffers holds the stream's state now
    hipStream_t cap = nullptr;
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t exec[2] = {nullptr, nullptr};
};

static int ring_issue(const sdrx_fir_ring *r, int parity, hipStream_t st)
{
    for (size_t k = 0; k < r->chunks; k++) {
        sdrx_fir_job j = r->job;
        j.in0 = (const char *)r->job.in0 + k * r->in_stride;
        if (r->job.in1) j.in1 = (const char *)r->job.in1 + k * r->in_stride;
        j.out0 = (char *)r->job.out0 + k * r->out_stride;
        if (r->job.out1) j.out1 = (char *)r->job.out1 + k * r->out_stride;
        const bool flip = ((k + (size_t)parity) & 1) != 0;
        j.state_in = flip ? (const void *)r->job.state_out : r->job.state_in;
        j.state_out = flip ? (void *)r->job.state_in : r->job.state_out;
        j.stream = st;
        int rc = launch_fir(&j, r->epi);
        if (rc) return rc;
    }
    return SDRX_OK;
}

extern "C" void sdrx_fir_ring_destroy(sdrx_fir_ring_t *r)
{
    if (!r) return;
    for (int p = 0; p < 2; p++) {
        if (r->exec[p]) (void)hipGraphExecDestroy(r->exec[p]);
        if (r->graph[p]) (void)hipGraphDestroy(r->graph[p]);
    }
    if (r->cap) (void)hipStreamDestroy(r->cap);
    delete r;
}

extern "C" sdrx_fir_ring_t *sdrx_fir_ring_create(const sdrx_fir_job *chunk0, int epilogue, size_t chunks, size_t in_stride,
                                                 size_t out_stride, int use_graph)
{
    if (!chunk0 || chunks == 0 || epilogue < 0 || epilogue > 2 || !chunk0->state_in || !chunk0->state_out) {
        set_error("sdrx_fir_ring_create: bad argument");
        return nullptr;
    }
    sdrx_fir_ring *r = new sdrx_fir_ring();
    r->job = *chunk0;
    r->epi = (FirEpilogue)epilogue;
    r->chunks = chunks;
    r->in_stride = in_stride;
    r->out_stride = out_stride;
    // A ring whose chunks abut in memory IS one longer block of the same stream (Filter::execute carries its shift
    // register, filt.cpp:256-259; the decimation phase continues when the chunk length is a multiple of D; rotate_90
    // restarts every 4 samples): one launch over chunks * n samples gives the chunked results bit for bit, without
    // the ~5.6 us the GPU spends between two dependent launches (a graph does not remove that: measured 5.57 us per
    // chunk replayed against 5.81 us issued one by one).
    const size_t in_elem = chunk0->layout == SDRX_IQ_U8 ? 2 : (chunk0->dtype == SDRX_F32 ? sizeof(float) : sizeof(double)) *
                                                                  (chunk0->layout == SDRX_IQ_INTERLEAVED ? 2 : 1);
    const size_t D = chunk0->decim ? chunk0->decim : 1, n_out1 = chunk0->n_out ? chunk0->n_out : (chunk0->n + D - 1) / D;
    const size_t fp = chunk0->dtype == SDRX_F32 ? sizeof(float) : sizeof(double);
    const size_t out_elem = epilogue == EPI_AM ? 1 : ((epilogue == EPI_STORE && chunk0->layout != SDRX_REAL && !chunk0->out1) ? 2 * fp : fp);
    const bool contiguous = chunk0->nchan == 1 && in_stride == chunk0->n * in_elem && out_stride == n_out1 * out_elem &&
                            chunk0->n % D == 0 && chunk0->n_out == 0 && (chunk0->layout != SDRX_IQ_U8 || chunk0->n % 4 == 0);
    r->mode = use_graph == 0 ? 0 : ((use_graph == 1 && contiguous) ? 1 : 2);
    if (r->mode == 1) { // one job over the ring
        r->job.n = chunk0->n * chunks;
        r->job.in_pitch = r->job.n;
        r->job.out_pitch = n_out1 * chunks;
        r->chunks = 1;
        return r;
    }
    if (r->mode == 0) return r;
    // One launch outside the capture, on throw-away state rows: whatever the launcher creates on the first use of a
    // tile geometry (its window-offset table, the kernel's LDS attribute) exists before the capture begins.  It writes
    // chunk 0's output slot, which the first replay overwrites.
    const size_t st_bytes = chunk0->nchan * chunk0->state_pitch * (chunk0->dtype == SDRX_F32 ? sizeof(float) : sizeof(double));
    void *tmp = nullptr;
    bool ok = hipMalloc(&tmp, 2 * st_bytes + 256) == hipSuccess && hipMemsetAsync(tmp, 0, 2 * st_bytes + 256, nullptr) == hipSuccess; // (the launch below is on the null stream too)
    if (ok) {
        sdrx_fir_job w = *chunk0;
        w.state_in = tmp;
        w.state_out = (char *)tmp + st_bytes;
        w.stream = nullptr;
        ok = launch_fir(&w, r->epi) == SDRX_OK && hipDeviceSynchronize() == hipSuccess;
    }
    if (tmp) (void)hipFree(tmp);
    ok = ok && hipStreamCreateWithFlags(&r->cap, hipStreamNonBlocking) == hipSuccess;
    // odd ring lengths leave the state in the other buffer after a replay: one graph per starting parity
    for (int p = 0; ok && p < ((chunks & 1) ? 2 : 1); p++) {
        ok = hipStreamBeginCapture(r->cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
        const int rc = ok ? ring_issue(r, p, r->cap) : SDRX_EHIP;
        ok = hipStreamEndCapture(r->cap, &r->graph[p]) == hipSuccess && rc == SDRX_OK && ok;
        ok = ok && hipGraphInstantiate(&r->exec[p], r->graph[p], nullptr, nullptr, 0) == hipSuccess;
    }
    if (!ok) {
        if (!*sdrx_last_error()) set_error("sdrx_fir_ring_create: graph capture failed");
        sdrx_fir_ring_destroy(r);
        return nullptr;
    }
    return r;
}

extern "C" int sdrx_fir_ring_launch(sdrx_fir_ring_t *r, void *stream)
{
    if (!r) return SDRX_EINVAL;
    const int p = (r->chunks & 1) ? r->parity : 0;
    if (r->mode == 2) SDRX_HIP(hipGraphLaunch(r->exec[p], (hipStream_t)stream));
    else {
        int rc = ring_issue(r, r->parity, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (r->chunks & 1) r->parity ^= 1;
    return SDRX_OK;
}

extern "C" const void *sdrx_fir_ring_state(const sdrx_fir_ring_t *r)
{
    if (!r) return nullptr;
    // after an even number of launches in all the state is where chunk 0 reads it from
    return r->parity ? (const void *)r->job.state_out : r->job.state_in;
}

// ---- AM / FM on host buffers
struct HostCtx { // one per thread: a stream and staging for the stateless host-layer calls
    hipStream_t stream = nullptr;
    Staging a, b, c;
    int ready()
    {
        if (!stream) SDRX_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return SDRX_OK;
    }
};
static thread_local HostCtx t_ctx;

extern "C" size_t sdrx_am_demodulate(const double *re, const double *im, uint8_t *out, size_t n, size_t ds)
{
    if (!re || !im || !out || ds == 0 || n == 0) return 0;
    HostCtx &c = t_ctx;
    const size_t bytes = n * sizeof(double), n_out = (n + ds - 1) / ds;
    if (c.ready() || c.a.ensure(bytes) || c.b.ensure(bytes) || c.c.ensure(n_out + 8)) {
        printf("[HIP][sdrx_am_demodulate] Error! %s\n", g_err); // print-and-continue: am.cu:143-146
        return 0;
    }
    staged_copy(c.a.host, re, bytes);
    staged_copy(c.b.host, im, bytes);
    bool ok = hipMemcpyAsync(c.a.dev, c.a.host, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              hipMemcpyAsync(c.b.dev, c.b.host, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              launch_am_demod((double *)c.a.dev, (double *)c.b.dev, (uint8_t *)c.c.dev, n, ds, c.stream) == SDRX_OK &&
              hipMemcpyAsync(c.c.host, c.c.dev, n_out, hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
              hipStreamSynchronize(c.stream) == hipSuccess;
    if (!ok) {
        printf("[HIP][sdrx_am_demodulate] Error! %s\n", g_err);
        return 0;
    }
    staged_copy(out, c.c.host, n_out);
    return n_out; // out_index of AM.h:76
}

extern "C" void sdrx_am_modulate(const uint8_t *bits, size_t len, double *out_real, size_t len_mod)
{
    // AM.h:41-54 + raw_upsample (Resample.h:15-27) on the device: bits up, the written prefix back.  Outputs past
    // len * floor(len_mod / len) are not written by the reference and are not touched here.
    if (!bits || !out_real || len == 0) return;
    const size_t rep = len_mod / len;
    if (rep == 0) return;
    const size_t n_write = len * rep;
    HostCtx &c = t_ctx;
    if (c.ready() || c.a.ensure(n_write * sizeof(double)) || c.c.ensure(len + 8)) {
        printf("[HIP][sdrx_am_modulate] Error! %s\n", g_err);
        return;
    }
    memcpy(c.c.host, bits, len);
    bool ok = hipMemcpyAsync(c.c.dev, c.c.host, len, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              launch_am_modulate((const uint8_t *)c.c.dev, len, 1, len, (double *)c.a.dev, len_mod, n_write, c.stream) == SDRX_OK &&
              hipMemcpyAsync(c.a.host, c.a.dev, n_write * sizeof(double), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
              hipStreamSynchronize(c.stream) == hipSuccess;
    if (!ok) {
        printf("[HIP][sdrx_am_modulate] Error! %s\n", g_err);
        return;
    }
    staged_copy(out_real, c.a.host, n_write * sizeof(double));
}

static void fm_waves(double rate, double mark, double space, size_t ds, std::vector<double> &w1,
                     std::vector<double> &w2)
{
    w1.resize(ds ? ds : 1);
    w2.resize(ds ? ds : 1);
    sdrx_fm_tables(rate, mark, space, ds, 0, w1.data(), w2.data()); // FM.h:127-131, as fm.cu:62-68 does
}

extern "C" size_t sdrx_fm_demodulate(double rx_rate, double mark, double space, const double *re, uint8_t *out,
                                     size_t n, size_t ds)
{
    if (!re || !out || ds == 0) return 0;
    const size_t nbits = n / ds;
    if (nbits == 0) return 0;
    HostCtx &c = t_ctx;
    std::vector<double> w1, w2;
    fm_waves(rx_rate, mark, space, ds, w1, w2);
    const size_t bytes = n * sizeof(double), wbytes = ds * sizeof(double);
    if (c.ready() || c.a.ensure(bytes) || c.b.ensure(2 * wbytes) || c.c.ensure(nbits + 8)) {
        printf("[HIP][sdrx_fm_demodulate] Error! %s\n", g_err);
        return 0;
    }
    staged_copy(c.a.host, re, bytes);
    memcpy(c.b.host, w1.data(), wbytes);
    memcpy((char *)c.b.host + wbytes, w2.data(), wbytes);
    bool ok = hipMemcpyAsync(c.a.dev, c.a.host, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              hipMemcpyAsync(c.b.dev, c.b.host, 2 * wbytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              launch_fm_demod((double *)c.a.dev, (double *)c.b.dev, (double *)c.b.dev + ds, (uint8_t *)c.c.dev, n, ds,
                              c.stream) == SDRX_OK &&
              hipMemcpyAsync(c.c.host, c.c.dev, nbits, hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
              hipStreamSynchronize(c.stream) == hipSuccess;
    if (!ok) {
        printf("[HIP][sdrx_fm_demodulate] Error! %s\n", g_err);
        return 0;
    }
    staged_copy(out, c.c.host, nbits);
    return nbits; // FM.h:147
}

extern "C" void sdrx_fm_modulate(double tx_rate, double mark, double space, const uint8_t *bits, size_t len,
                                 double *out_real, size_t spb)
{
    // FM.h:44-74: the two cosine tables are host arithmetic (2*spb libm cos calls, as in the reference); the per-bit
    // copies run on the device.
    if (!bits || !out_real || spb == 0 || len == 0) return;
    HostCtx &c = t_ctx;
    const size_t n = len * spb, wbytes = spb * sizeof(double);
    if (c.ready() || c.a.ensure(n * sizeof(double)) || c.b.ensure(2 * wbytes) || c.c.ensure(len + 8)) {
        printf("[HIP][sdrx_fm_modulate] Error! %s\n", g_err);
        return;
    }
    sdrx_fm_tables(tx_rate, mark, space, spb, 1, (double *)c.b.host, (double *)c.b.host + spb);
    memcpy(c.c.host, bits, len);
    bool ok = hipMemcpyAsync(c.b.dev, c.b.host, 2 * wbytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              hipMemcpyAsync(c.c.dev, c.c.host, len, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              launch_fm_modulate((const uint8_t *)c.c.dev, len, 1, len, (double *)c.b.dev, (double *)c.b.dev + spb, spb,
                                 (double *)c.a.dev, n, c.stream) == SDRX_OK &&
              hipMemcpyAsync(c.a.host, c.a.dev, n * sizeof(double), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
              hipStreamSynchronize(c.stream) == hipSuccess;
    if (!ok) {
        printf("[HIP][sdrx_fm_modulate] Error! %s\n", g_err);
        return;
    }
    staged_copy(out_real, c.a.host, n * sizeof(double));
}

extern "C" void sdrx_fm_convolve(const double *input1, size_t length1, const double *input2, size_t length2,
                                 double *output, size_t length_output, size_t ds)
{
    // FM<T>::convolve (FM.h:83-108, repaired: 7th parameter ds).  lengthOutput is not used by the reference either:
    // it writes output[0 .. number of i = ds-1, 2ds-1, ... < length1).
    (void)length_output;
    if (!input1 || !input2 || !output || ds == 0 || length1 < ds || length2 == 0) return;
    const size_t n_out = (length1 - (ds - 1) + ds - 1) / ds;
    HostCtx &c = t_ctx;
    if (c.ready() || c.a.ensure(length1 * sizeof(double)) || c.b.ensure(length2 * sizeof(double)) ||
        c.c.ensure(n_out * sizeof(double))) {
        printf("[HIP][sdrx_fm_convolve] Error! %s\n", g_err);
        return;
    }
    staged_copy(c.a.host, input1, length1 * sizeof(double));
    memcpy(c.b.host, input2, length2 * sizeof(double));
    memcpy(c.c.host, output, n_out * sizeof(double)); // the sums start from the outputs' prior values
    bool ok = hipMemcpyAsync(c.a.dev, c.a.host, length1 * sizeof(double), hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              hipMemcpyAsync(c.b.dev, c.b.host, length2 * sizeof(double), hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              hipMemcpyAsync(c.c.dev, c.c.host, n_out * sizeof(double), hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              launch_fm_convolve((const double *)c.a.dev, length1, (const double *)c.b.dev, length2, (double *)c.c.dev, ds,
                                 c.stream) == SDRX_OK &&
              hipMemcpyAsync(c.c.host, c.c.dev, n_out * sizeof(double), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
              hipStreamSynchronize(c.stream) == hipSuccess;
    if (!ok) {
        printf("[HIP][sdrx_fm_convolve] Error! %s\n", g_err);
        return;
    }
    memcpy(output, c.c.host, n_out * sizeof(double));
}

// ---- FIR + demod chains on host buffers
// One stream through FIR (+ demod) from host buffers.  Long calls are cut into chunks of CHAIN_CHUNK
// samples (a multiple of the decimation factor), two staging slots: while the GPU copies in, filters
// and copies out chunk k, the host fills the pinned buffer of chunk k+1 and drains chunk k-1 --
// pinned memory stays bounded however long the call is.  The FIR state carries from chunk to chunk
// exactly as it does from call to call.
static const size_t CHAIN_CHUNK = (size_t)1 << 22;

struct ChainSlot {
    Staging in0, in1, out;
    hipEvent_t done = nullptr;
    size_t n = 0, n_out = 0, out_off = 0; // what is in flight in this slot
    bool busy = false;
};

struct sdrx_chain {
    int T = 0;
    unsigned D = 1;
    double *d_taps = nullptr;
    double *d_state[2] = {nullptr, nullptr};
    int cur = 0;
    hipStream_t stream = nullptr;
    ChainSlot slot[2];
};

extern "C" sdrx_chain_t *sdrx_chain_create(const double *taps, int num_taps, unsigned decim)
{
    if (!taps || num_taps <= 0 || decim == 0) return nullptr;
    sdrx_chain *c = new sdrx_chain();
    c->T = num_taps;
    c->D = decim;
    const size_t sd = sdrx_fir_state_doubles(num_taps);
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&c->d_taps, num_taps * sizeof(double)) == hipSuccess &&
              hipMemcpy(c->d_taps, taps, num_taps * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    for (int i = 0; ok && i < 2; i++)
        ok = hipMalloc((void **)&c->d_state[i], sd * sizeof(double)) == hipSuccess &&
             hipMemsetAsync(c->d_state[i], 0, sd * sizeof(double), c->stream) == hipSuccess && // (ordered before the chain's first launch)
             hipEventCreateWithFlags(&c->slot[i].done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) {
        set_error("sdrx_chain_create: device allocation failed");
        sdrx_chain_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void sdrx_chain_destroy(sdrx_chain_t *c)
{
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_taps) (void)hipFree(c->d_taps);
    for (int i = 0; i < 2; i++) {
        if (c->d_state[i]) (void)hipFree(c->d_state[i]);
        c->slot[i].in0.release();
        c->slot[i].in1.release();
        c->slot[i].out.release();
        if (c->slot[i].done) (void)hipEventDestroy(c->slot[i].done);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void sdrx_chain_reset(sdrx_chain_t *c)
{
    if (!c) return;
    for (int i = 0; i < 2; i++) (void)hipMemsetAsync(c->d_state[i], 0, sdrx_fir_state_doubles(c->T) * sizeof(double), c->stream);
    (void)hipStreamSynchronize(c->stream);
}

// re/im: split fp64 rails; or u8 != nullptr: raw interleaved bytes (SDRX_IQ_U8) with bias / flags
static size_t chain_run(sdrx_chain *c, const double *re, const double *im, const uint8_t *u8, double u8_bias, int u8_flags,
                        size_t n, void *out, size_t out_elem, FirEpilogue epi)
{
    if (!c || !out || n == 0 || (!u8 && (!re || !im))) return 0;
    // whole decimation groups, so that chunk outputs abut, and whole groups of 4 samples, so that rotate_90
    // (which restarts at every CALL, Baseband_Stream.cpp:103-126) sees the same phase in every chunk
    size_t group = c->D;
    while (group % 4) group += c->D; // lcm(D, 4)
    const size_t chunk = CHAIN_CHUNK < group ? group : CHAIN_CHUNK - CHAIN_CHUNK % group;
    const size_t in_elem = u8 ? 2 : sizeof(double);
    size_t done_out = 0;
    bool failed = false;
    auto drain = [&](ChainSlot &sl) { // wait for the slot's chunk and hand its outputs to the caller
        if (!sl.busy) return;
        if (hipEventSynchronize(sl.done) != hipSuccess) failed = true;
        if (!failed) staged_copy((char *)out + sl.out_off * out_elem, sl.out.host, sl.n_out * out_elem);
        done_out += sl.n_out;
        sl.busy = false;
    };
    size_t k = 0;
    for (size_t off = 0; off < n && !failed; off += chunk, k++) {
        ChainSlot &sl = c->slot[k & 1];
        drain(sl);
        const size_t m = (n - off) < chunk ? (n - off) : chunk, m_out = (m + c->D - 1) / c->D;
        if (sl.in0.ensure(m * in_elem) || (!u8 && sl.in1.ensure(m * in_elem)) || sl.out.ensure(m_out * out_elem + 8)) {
            failed = true;
            break;
        }
        if (u8) {
            staged_copy(sl.in0.host, u8 + 2 * off, m * 2);
        } else {
            staged_copy(sl.in0.host, re + off, m * in_elem);
            staged_copy(sl.in1.host, im + off, m * in_elem);
        }
        if (hipMemcpyAsync(sl.in0.dev, sl.in0.host, m * in_elem, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            (!u8 && hipMemcpyAsync(sl.in1.dev, sl.in1.host, m * in_elem, hipMemcpyHostToDevice, c->stream) != hipSuccess)) {
            failed = true;
            break;
        }
        sdrx_fir_job j;
        memset(&j, 0, sizeof(j));
        j.layout = u8 ? SDRX_IQ_U8 : SDRX_IQ_SPLIT;
        j.in0 = sl.in0.dev;
        j.in1 = u8 ? nullptr : sl.in1.dev;
        j.u8_bias = u8_bias;
        j.u8_flags = u8_flags;
        j.n = m;
        j.nchan = 1;
        j.in_pitch = m;
        j.taps = c->d_taps;
        j.num_taps = c->T;
        j.decim = c->D;
        j.state_in = c->d_state[c->cur];
        j.state_out = c->d_state[c->cur ^ 1];
        j.state_pitch = sdrx_fir_state_doubles(c->T);
        j.out0 = sl.out.dev;
        j.out_pitch = m_out;
        j.stream = c->stream;
        if (launch_fir(&j, epi) != SDRX_OK) {
            failed = true;
            break;
        }
        c->cur ^= 1;
        if (hipMemcpyAsync(sl.out.host, sl.out.dev, m_out * out_elem, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipEventRecord(sl.done, c->stream) != hipSuccess) {
            failed = true;
            break;
        }
        sl.n = m;
        sl.n_out = m_out;
        sl.out_off = off / c->D;
        sl.busy = true;
    }
    drain(c->slot[k & 1]); // oldest first
    drain(c->slot[(k + 1) & 1]);
    if (failed) {
        (void)hipStreamSynchronize(c->stream);
        c->slot[0].busy = c->slot[1].busy = false;
        return 0;
    }
    return done_out;
}

extern "C" size_t sdrx_chain_fir_am(sdrx_chain_t *c, const double *re, const double *im, size_t n, uint8_t *bits)
{
    return chain_run(c, re, im, nullptr, 0.0, 0, n, bits, 1, EPI_AM);
}

extern "C" size_t sdrx_chain_fir_fmdisc(sdrx_chain_t *c, const double *re, const double *im, size_t n, double *angle)
{
    return chain_run(c, re, im, nullptr, 0.0, 0, n, angle, sizeof(double), EPI_DISC);
}

extern "C" size_t sdrx_chain_fir_am_u8(sdrx_chain_t *c, const uint8_t *iq_u8, size_t n, double bias, int flags, uint8_t *bits)
{
    return chain_run(c, nullptr, nullptr, iq_u8, bias, flags, n, bits, 1, EPI_AM);
}

extern "C" size_t sdrx_chain_fir_fmdisc_u8(sdrx_chain_t *c, const uint8_t *iq_u8, size_t n, double bias, int flags,
                                           double *angle)
{
    return chain_run(c, nullptr, nullptr, iq_u8, bias, flags, n, angle, sizeof(double), EPI_DISC);
}

// ---- PSK objects
struct sdrx_psk {
    int kind = 0;
    sdrx_psk_state *d_state = nullptr;
    hipStream_t stream = nullptr;
    Staging in, mid, out, hits;
};

extern "C" sdrx_psk_t *sdrx_psk_create(int kind, uint64_t bpsk_uw, const double *qpsk_uw64, size_t uw_len)
{
    if (kind != SDRX_BPSK && kind != SDRX_QPSK) return nullptr;
    sdrx_psk *p = new sdrx_psk();
    p->kind = kind;
    sdrx_psk_state st;
    if (kind == SDRX_BPSK) sdrx_psk_state_init_bpsk(&st, bpsk_uw, uw_len);
    else sdrx_psk_state_init_qpsk(&st, qpsk_uw64, uw_len);
    bool ok = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&p->d_state, sizeof(st)) == hipSuccess &&
              hipMemcpy(p->d_state, &st, sizeof(st), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        set_error("sdrx_psk_create: device allocation failed");
        sdrx_psk_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void sdrx_psk_destroy(sdrx_psk_t *p)
{
    if (!p) return;
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->d_state) (void)hipFree(p->d_state);
    p->in.release();
    p->mid.release();
    p->out.release();
    p->hits.release();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" void sdrx_psk_get_state(sdrx_psk_t *p, sdrx_psk_state *out)
{
    if (!p || !out) return;
    (void)hipStreamSynchronize(p->stream);
    (void)hipMemcpy(out, p->d_state, sizeof(*out), hipMemcpyDeviceToHost);
}

extern "C" void sdrx_psk_set_state(sdrx_psk_t *p, const sdrx_psk_state *in)
{
    if (!p || !in) return;
    (void)hipStreamSynchronize(p->stream);
    (void)hipMemcpy(p->d_state, in, sizeof(*in), hipMemcpyHostToDevice);
}

static const size_t PSK_MAX_HITS = 4096;

static size_t psk_fetch_hits(sdrx_psk *p, uint64_t *hits, size_t max_hits)
{
    sdrx_psk_state st;
    if (hipMemcpy(&st, p->d_state, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    size_t nh = (size_t)st.n_hits;
    if (hits && max_hits) {
        size_t k = nh < max_hits ? nh : max_hits;
        if (k > PSK_MAX_HITS) k = PSK_MAX_HITS;
        memcpy(hits, p->hits.host, k * sizeof(uint64_t));
    }
    return nh;
}

extern "C" size_t sdrx_psk_demodulate(sdrx_psk_t *p, const double *iq, size_t n, double *out, uint64_t *hits,
                                      size_t max_hits)
{
    if (!p || !iq || !out) return 0;
    const size_t m = n / 4, nd = (m + 1) / 2; // decode: n/4 (BPSK.cpp:299); decimate by 2: ceil (Resample.h:51)
    if (m == 0) return 0;
    const size_t in_bytes = n * 2 * sizeof(double);
    if (p->in.ensure(in_bytes) || p->mid.ensure(sdrx_psk_work_bytes(n, 1)) || p->out.ensure(nd + 8) ||
        p->hits.ensure(PSK_MAX_HITS * sizeof(uint64_t)))
        return 0;
    staged_copy(p->in.host, iq, in_bytes);
    if (hipMemcpyAsync(p->in.dev, p->in.host, in_bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess) return 0;
    unsigned char *w = (unsigned char *)p->mid.dev;
    double *mf = (double *)w;
    double *zst = (double *)(w + align256(m * 2 * sizeof(double)));
    double *dst = (double *)((unsigned char *)zst + align256(MF_STATE * sizeof(double)));
    if (mf_launch((double *)p->in.dev, n, 1, n, mf, m, zst, dst, p->stream) != SDRX_OK) return 0;
    if (launch_psk_loops(p->kind, mf, m, 1, m, p->d_state, (uint8_t *)p->out.dev, nd, 1, 1, p->stream) != SDRX_OK)
        return 0;
    if (launch_psk_ambiguity_hits(p->kind, (uint8_t *)p->out.dev, nd, 1, nd, p->d_state,
                                  (unsigned long long *)p->hits.dev, PSK_MAX_HITS, p->stream) != SDRX_OK)
        return 0;
    if (hipMemcpyAsync(p->out.host, p->out.dev, nd, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipMemcpyAsync(p->hits.host, p->hits.dev, PSK_MAX_HITS * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream) !=
            hipSuccess ||
        hipStreamSynchronize(p->stream) != hipSuccess)
        return 0;
    const uint8_t *d = (const uint8_t *)p->out.host;
    for (size_t i = 0; i < nd; i++) out[i] = (double)d[i]; // output block real part: BPSK.cpp:167-172
    (void)psk_fetch_hits(p, hits, max_hits);
    return nd;
}

extern "C" size_t sdrx_psk_demodulate_plot(sdrx_psk_t *p, const double *iq, size_t n, uint8_t *out_bits, double *plot_re,
                                           double *plot_im, uint64_t *hits, size_t max_hits)
{
    // BPSK::demodulate(Complex_Array&, uint8_t*, n, downsample, plot bufs), BPSK.cpp:178-211, after its
    // decimateComplexArray: matched filter, timing loop (its output also goes to the plot buffers), carrier loop + /2,
    // ambiguity pass, bits = real > 0.5.  One upload, one synchronisation.
    if (!p || !iq || !out_bits) return 0;
    const size_t m = n / 4, nd = (m + 1) / 2;
    if (m == 0) return 0;
    const size_t in_bytes = n * 2 * sizeof(double), mf_bytes = m * 2 * sizeof(double);
    if (p->in.ensure(in_bytes) || p->mid.ensure(sdrx_psk_work_bytes(n, 1)) || p->out.ensure(nd + 8) ||
        p->hits.ensure(PSK_MAX_HITS * sizeof(uint64_t)))
        return 0;
    staged_copy(p->in.host, iq, in_bytes);
    if (hipMemcpyAsync(p->in.dev, p->in.host, in_bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess) return 0;
    unsigned char *w = (unsigned char *)p->mid.dev;
    double *mf = (double *)w;
    double *zst = (double *)(w + align256(mf_bytes));
    double *dst = (double *)((unsigned char *)zst + align256(MF_STATE * sizeof(double)));
    if (mf_launch((double *)p->in.dev, n, 1, n, mf, m, zst, dst, p->stream) != SDRX_OK) return 0;
    if (launch_psk_loops(p->kind, mf, m, 1, m, p->d_state, (uint8_t *)p->out.dev, nd, 1, 0, p->stream) != SDRX_OK) return 0;
    if (hipMemcpyAsync(p->mid.host, mf, mf_bytes, hipMemcpyDeviceToHost, p->stream) != hipSuccess) return 0;
    if (launch_psk_loops(p->kind, mf, m, 1, m, p->d_state, (uint8_t *)p->out.dev, nd, 0, 1, p->stream) != SDRX_OK) return 0;
    if (launch_psk_ambiguity_hits(p->kind, (uint8_t *)p->out.dev, nd, 1, nd, p->d_state,
                                  (unsigned long long *)p->hits.dev, PSK_MAX_HITS, p->stream) != SDRX_OK)
        return 0;
    if (hipMemcpyAsync(p->out.host, p->out.dev, nd, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipMemcpyAsync(p->hits.host, p->hits.dev, PSK_MAX_HITS * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream) !=
            hipSuccess ||
        hipStreamSynchronize(p->stream) != hipSuccess)
        return 0;
    const double *h = (const double *)p->mid.host;
    if (plot_re)
        for (size_t i = 0; i < m; i++) plot_re[i] = h[2 * i];
    if (plot_im)
        for (size_t i = 0; i < m; i++) plot_im[i] = h[2 * i + 1];
    memcpy(out_bits, p->out.host, nd); // decisions are 0 / 1 (0..3 for QPSK): `real > 0.5 ? 1 : 0` of BPSK.cpp:203 for BPSK
    if (p->kind == SDRX_QPSK)
        for (size_t i = 0; i < nd; i++) out_bits[i] = out_bits[i] > 0 ? 1 : 0;
    (void)psk_fetch_hits(p, hits, max_hits);
    return nd;
}

extern "C" size_t sdrx_psk_decode_pulse_shape(sdrx_psk_t *p, const double *iq, size_t n, double *re, double *im)
{
    if (!p || !iq || !re || !im) return 0;
    const size_t m = n / 4;
    if (m == 0) return 0;
    const size_t in_bytes = n * 2 * sizeof(double);
    if (p->in.ensure(in_bytes) || p->mid.ensure(sdrx_psk_work_bytes(n, 1))) return 0;
    staged_copy(p->in.host, iq, in_bytes);
    if (hipMemcpyAsync(p->in.dev, p->in.host, in_bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess) return 0;
    unsigned char *w = (unsigned char *)p->mid.dev;
    double *mf = (double *)w;
    double *zst = (double *)(w + align256(m * 2 * sizeof(double)));
    double *dst = (double *)((unsigned char *)zst + align256(MF_STATE * sizeof(double)));
    if (mf_launch((double *)p->in.dev, n, 1, n, mf, m, zst, dst, p->stream) != SDRX_OK) return 0;
    if (hipMemcpyAsync(p->mid.host, mf, m * 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipStreamSynchronize(p->stream) != hipSuccess)
        return 0;
    const double *h = (const double *)p->mid.host;
    for (size_t i = 0; i < m; i++) { re[i] = h[2 * i]; im[i] = h[2 * i + 1]; }
    return m;
}

static int psk_stage_loops(sdrx_psk *p, double *re, double *im, size_t n, int timing, int phase)
{
    const size_t bytes = n * 2 * sizeof(double), nd = (n + 1) / 2;
    if (p->mid.ensure(bytes) || p->out.ensure(nd + 8)) return SDRX_ENOMEM;
    double *h = (double *)p->mid.host;
    for (size_t i = 0; i < n; i++) { h[2 * i] = re[i]; h[2 * i + 1] = im[i]; }
    SDRX_HIP(hipMemcpyAsync(p->mid.dev, p->mid.host, bytes, hipMemcpyHostToDevice, p->stream));
    int rc = launch_psk_loops(p->kind, (double *)p->mid.dev, n, 1, n, p->d_state, (uint8_t *)p->out.dev, nd, timing,
                              phase, p->stream);
    if (rc) return rc;
    if (timing) SDRX_HIP(hipMemcpyAsync(p->mid.host, p->mid.dev, bytes, hipMemcpyDeviceToHost, p->stream));
    if (phase) SDRX_HIP(hipMemcpyAsync(p->out.host, p->out.dev, nd, hipMemcpyDeviceToHost, p->stream));
    SDRX_HIP(hipStreamSynchronize(p->stream));
    return SDRX_OK;
}

extern "C" void sdrx_psk_timing(sdrx_psk_t *p, double *re, double *im, size_t n)
{
    if (!p || !re || !im || n == 0) return;
    if (psk_stage_loops(p, re, im, n, 1, 0) != SDRX_OK) return;
    const double *h = (const double *)p->mid.host;
    for (size_t i = 0; i < n; i++) { re[i] = h[2 * i]; im[i] = h[2 * i + 1]; }
}

extern "C" size_t sdrx_psk_phase_correction(sdrx_psk_t *p, double *re, double *im, size_t n)
{
    if (!p || !re || !im || n == 0) return 0;
    if (psk_stage_loops(p, re, im, n, 0, 1) != SDRX_OK) return 0;
    const size_t nd = (n + 1) / 2;
    const uint8_t *d = (const uint8_t *)p->out.host;
    for (size_t i = 0; i < nd; i++) re[i] = (double)d[i];
    if (nd < n) re[nd] = 0.0;                    // decimate's tail clear touches only block[dst]: Resample.h:46-49
    for (size_t i = 0; i < n; i++) im[i] = 0.0; // BPSK.cpp:410
    return nd;
}

extern "C" size_t sdrx_psk_ambiguity(sdrx_psk_t *p, double *re, size_t n, uint64_t *hits, size_t max_hits)
{
    if (!p || !re || n == 0) return 0;
    if (p->out.ensure(n + 8) || p->hits.ensure(PSK_MAX_HITS * sizeof(uint64_t))) return 0;
    uint8_t *h = (uint8_t *)p->out.host;
    for (size_t i = 0; i < n; i++) h[i] = (uint8_t)(int)re[i];
    if (hipMemcpyAsync(p->out.dev, p->out.host, n, hipMemcpyHostToDevice, p->stream) != hipSuccess) return 0;
    if (launch_psk_ambiguity_hits(p->kind, (uint8_t *)p->out.dev, n, 1, n, p->d_state,
                                  (unsigned long long *)p->hits.dev, PSK_MAX_HITS, p->stream) != SDRX_OK)
        return 0;
    if (hipMemcpyAsync(p->out.host, p->out.dev, n, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipMemcpyAsync(p->hits.host, p->hits.dev, PSK_MAX_HITS * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream) !=
            hipSuccess ||
        hipStreamSynchronize(p->stream) != hipSuccess)
        return 0;
    for (size_t i = 0; i < n; i++) re[i] = (double)h[i];
    return psk_fetch_hits(p, hits, max_hits);
}

extern "C" int sdrx_psk_modulate(int kind, const uint8_t *sym, size_t len, double *out_real, double *out_imag,
                                 size_t len_mod)
{
    if (!sym || !out_real || (kind == SDRX_QPSK && !out_imag)) return SDRX_EINVAL;
    // size checks print and return like the reference: BPSK.cpp:125-129, QPSK.cpp:158-162
    if (kind == SDRX_BPSK && len_mod < 8 * len) {
        printf("ERROR:: QPSK modulation output arrays are not large enough!!! [%zu vs %zu\n", len_mod, 8 * len);
        return SDRX_OK;
    }
    if (kind == SDRX_QPSK && len_mod != 8 * len) {
        printf("ERROR:: QPSK modulation output arrays are not large enough!!!\n");
        return SDRX_OK;
    }
    if (len == 0) return SDRX_OK;
    HostCtx &c = t_ctx;
    const size_t n = 8 * len, bytes = n * sizeof(double);
    int rc = c.ready();
    if (rc || (rc = c.a.ensure(bytes)) || (rc = c.b.ensure(bytes)) || (rc = c.c.ensure(len + 8))) return rc;
    staged_copy(c.a.host, out_real, bytes); // prior contents take part beyond the byte-count memset (BPSK.cpp:131)
    if (kind == SDRX_QPSK) staged_copy(c.b.host, out_imag, bytes);
    memcpy(c.c.host, sym, len);
    SDRX_HIP(hipMemcpyAsync(c.a.dev, c.a.host, bytes, hipMemcpyHostToDevice, c.stream));
    if (kind == SDRX_QPSK) SDRX_HIP(hipMemcpyAsync(c.b.dev, c.b.host, bytes, hipMemcpyHostToDevice, c.stream));
    SDRX_HIP(hipMemcpyAsync(c.c.dev, c.c.host, len, hipMemcpyHostToDevice, c.stream));
    rc = launch_psk_modulate(kind, (const uint8_t *)c.c.dev, len, (double *)c.a.dev,
                             kind == SDRX_QPSK ? (double *)c.b.dev : nullptr, len_mod / 8, c.stream);
    if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(c.a.host, c.a.dev, bytes, hipMemcpyDeviceToHost, c.stream));
    if (kind == SDRX_QPSK) SDRX_HIP(hipMemcpyAsync(c.b.host, c.b.dev, bytes, hipMemcpyDeviceToHost, c.stream));
    SDRX_HIP(hipStreamSynchronize(c.stream));
    staged_copy(out_real, c.a.host, bytes);
    if (kind == SDRX_QPSK) staged_copy(out_imag, c.b.host, bytes);
    return SDRX_OK;
}

// =========================================================================== 3. legacy layer

namespace {
struct Radio {
    hipStream_t stream = nullptr;
    void *h_re = nullptr, *h_im = nullptr, *h_out = nullptr; // pinned
    double *d_re = nullptr, *d_im = nullptr;
    uint8_t *d_out = nullptr;
    double *d_w1 = nullptr, *d_w2 = nullptr; // FM only
    size_t cap = 0;
};
std::vector<Radio> g_am, g_fm;
double g_fm_mark = 19000, g_fm_space = 20000;
size_t g_fm_rate = 0, g_fm_spb = 0;
// The receiver runs one demodulating thread per RADIO_ID (rtlsdr_receiver.cpp:191-194): demodulations hold this lock
// shared while they use the mark / space tables, whoever rebuilds the tables (a new samples-per-bit) holds it alone.
std::shared_mutex g_fm_mu;

bool radio_alloc(Radio &r, size_t max_points)
{
    r.cap = max_points;
    return hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) == hipSuccess &&
           hipHostMalloc(&r.h_re, max_points * sizeof(double), hipHostMallocDefault) == hipSuccess &&
           hipHostMalloc(&r.h_im, max_points * sizeof(double), hipHostMallocDefault) == hipSuccess &&
           hipHostMalloc(&r.h_out, max_points + 8, hipHostMallocDefault) == hipSuccess &&
           hipMalloc((void **)&r.d_re, max_points * sizeof(double)) == hipSuccess &&
           hipMalloc((void **)&r.d_im, max_points * sizeof(double)) == hipSuccess &&
           hipMalloc((void **)&r.d_out, max_points + 8) == hipSuccess;
}

void radio_free(Radio &r)
{
    if (r.stream) (void)hipStreamSynchronize(r.stream);
    if (r.h_re) (void)hipHostFree(r.h_re);
    if (r.h_im) (void)hipHostFree(r.h_im);
    if (r.h_out) (void)hipHostFree(r.h_out);
    if (r.d_re) (void)hipFree(r.d_re);
    if (r.d_im) (void)hipFree(r.d_im);
    if (r.d_out) (void)hipFree(r.d_out);
    if (r.d_w1) (void)hipFree(r.d_w1);
    if (r.d_w2) (void)hipFree(r.d_w2);
    if (r.stream) (void)hipStreamDestroy(r.stream);
    r = Radio();
}

bool fm_upload_waves(size_t spb, size_t rate)
{
    std::vector<double> w1, w2;
    fm_waves((double)rate, g_fm_mark, g_fm_space, spb, w1, w2); // fm.cu:62-68
    for (Radio &r : g_fm) {
        (void)hipStreamSynchronize(r.stream);
        if (r.d_w1) (void)hipFree(r.d_w1);
        if (r.d_w2) (void)hipFree(r.d_w2);
        r.d_w1 = r.d_w2 = nullptr;
        if (hipMalloc((void **)&r.d_w1, (spb ? spb : 1) * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&r.d_w2, (spb ? spb : 1) * sizeof(double)) != hipSuccess ||
            hipMemcpy(r.d_w1, w1.data(), spb * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(r.d_w2, w2.data(), spb * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            return false;
    }
    g_fm_spb = spb;
    g_fm_rate = rate;
    return true;
}
} // namespace

extern "C" void am_gpu_initialize(size_t MAX_NUM_POINTS, size_t num_of_radios)
{
    g_am.resize(num_of_radios);
    for (Radio &r : g_am)
        if (!radio_alloc(r, MAX_NUM_POINTS)) printf("[HIP][am_gpu_initialize] Error!\n");
}

extern "C" void am_gpu_free(void)
{
    for (Radio &r : g_am) radio_free(r);
    g_am.clear();
}

extern "C" double am_gpu_demodulation(int RADIO_ID, double *real, double *imag, uint8_t *output,
                                      size_t number_of_points, size_t ds)
{
    using clk = std::chrono::high_resolution_clock;
    if (RADIO_ID < 0 || (size_t)RADIO_ID >= g_am.size() || number_of_points > g_am[RADIO_ID].cap || ds == 0) {
        printf("[HIP][am_gpu_demodulation] Error!\n");
        return 0.0;
    }
    Radio &r = g_am[RADIO_ID];
    const size_t bytes = number_of_points * sizeof(double);
    staged_copy(r.h_re, real, bytes);
    staged_copy(r.h_im, imag, bytes);
    bool ok = hipMemcpyAsync(r.d_re, r.h_re, bytes, hipMemcpyHostToDevice, r.stream) == hipSuccess &&
              hipMemcpyAsync(r.d_im, r.h_im, bytes, hipMemcpyHostToDevice, r.stream) == hipSuccess &&
              hipStreamSynchronize(r.stream) == hipSuccess;
    clk::time_point t1 = clk::now(); // kernel wall seconds, as am.cu:130-136
    ok = ok && launch_am_demod(r.d_re, r.d_im, r.d_out, number_of_points, ds, r.stream) == SDRX_OK &&
         hipStreamSynchronize(r.stream) == hipSuccess;
    clk::time_point t2 = clk::now();
    const size_t nout = number_of_points / ds; // am.cu:140 pulls back number_of_points/ds bytes
    ok = ok && hipMemcpyAsync(r.h_out, r.d_out, nout, hipMemcpyDeviceToHost, r.stream) == hipSuccess &&
         hipStreamSynchronize(r.stream) == hipSuccess;
    if (!ok) printf("[HIP][am_gpu_demodulation] Error!\n"); // am.cu:143-146
    else staged_copy(output, r.h_out, nout);
    return std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
}

extern "C" void fm_gpu_initialize(size_t num_of_radios, size_t MaxBufferSize, size_t rx_sample_rate,
                                  size_t samples_per_bit, double MARK_FREQ, double SPACE_FREQ)
{
    g_fm.resize(num_of_radios);
    g_fm_mark = MARK_FREQ;
    g_fm_space = SPACE_FREQ;
    bool ok = true;
    for (Radio &r : g_fm) ok = radio_alloc(r, MaxBufferSize) && ok;
    ok = ok && fm_upload_waves(samples_per_bit, rx_sample_rate);
    if (!ok) printf("[HIP][fm_gpu_initialize] Error!\n");
}

extern "C" void fm_gpu_free(void)
{
    for (Radio &r : g_fm) radio_free(r);
    g_fm.clear();
}

extern "C" void fm_gpu_set_samples_per_bit(size_t samples_per_bit, size_t rx_sample_rate)
{
    // The receiver declares and calls this with ONE argument (rtlsdr_receiver.cpp:103,1720), so the
    // second may be garbage: anything that is not a plausible rate falls back to the init-time rate.
    std::unique_lock<std::shared_mutex> lk(g_fm_mu);
    size_t rate = (rx_sample_rate == 0 || rx_sample_rate > 1000000000000ULL) ? g_fm_rate : rx_sample_rate;
    if (!fm_upload_waves(samples_per_bit, rate)) printf("[HIP][fm_gpu_set_samples_per_bit] Error!\n");
}

extern "C" double fm_gpu_demodulation(int RADIO_ID, double *real, double *imag, uint8_t *output,
                                      size_t number_of_points, size_t ds)
{
    using clk = std::chrono::high_resolution_clock;
    (void)imag; // FM<T>::demodulate reads the real rail only (FM.h:134-135)
    if (RADIO_ID < 0 || (size_t)RADIO_ID >= g_fm.size() || number_of_points > g_fm[RADIO_ID].cap || ds == 0) {
        printf("[HIP][fm_gpu_demodulation] Error!\n");
        return 0.0;
    }
    std::shared_lock<std::shared_mutex> rd(g_fm_mu);
    while (ds != g_fm_spb) { // kernel is launched with WaveSize = ds (fm.cu:149): rebuild the tables, alone
        rd.unlock();
        {
            std::unique_lock<std::shared_mutex> wr(g_fm_mu);
            if (ds != g_fm_spb && !fm_upload_waves(ds, g_fm_rate)) {
                printf("[HIP][fm_gpu_demodulation] Error!\n");
                return 0.0;
            }
        }
        rd.lock();
    }
    Radio &r = g_fm[RADIO_ID];
    const size_t bytes = number_of_points * sizeof(double);
    staged_copy(r.h_re, real, bytes);
    bool ok = hipMemcpyAsync(r.d_re, r.h_re, bytes, hipMemcpyHostToDevice, r.stream) == hipSuccess &&
              hipStreamSynchronize(r.stream) == hipSuccess;
    clk::time_point t1 = clk::now();
    ok = ok && launch_fm_demod(r.d_re, r.d_w1, r.d_w2, r.d_out, number_of_points, ds, r.stream) == SDRX_OK &&
         hipStreamSynchronize(r.stream) == hipSuccess;
    clk::time_point t2 = clk::now();
    const size_t nout = number_of_points / ds;
    ok = ok && hipMemcpyAsync(r.h_out, r.d_out, nout, hipMemcpyDeviceToHost, r.stream) == hipSuccess &&
         hipStreamSynchronize(r.stream) == hipSuccess;
    if (!ok) printf("[HIP][fm_gpu_demodulation] Error!\n"); // fm.cu:159-162
    else staged_copy(output, r.h_out, nout);
    return std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
}

extern "C" int32_t ones_zeros_demodulation(double *real, double *imag, size_t number_of_points)
{
    HostCtx &c = t_ctx;
    const size_t bytes = number_of_points * sizeof(double);
    if (number_of_points == 0) return 0;
    if (c.ready() || c.a.ensure(bytes) || c.b.ensure(bytes) || c.c.ensure(64)) {
        printf("[HIP][ones_zeros_demodulation] Error!\n");
        return 0;
    }
    staged_copy(c.a.host, real, bytes);
    staged_copy(c.b.host, imag, bytes);
    bool ok = hipMemcpyAsync(c.a.dev, c.a.host, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              hipMemcpyAsync(c.b.dev, c.b.host, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
              launch_ones_zeros((double *)c.a.dev, (double *)c.b.dev, number_of_points, (int32_t *)c.c.dev,
                                c.stream) == SDRX_OK &&
              hipMemcpyAsync(c.c.host, c.c.dev, sizeof(int32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
              hipStreamSynchronize(c.stream) == hipSuccess;
    if (!ok) {
        printf("[HIP][ones_zeros_demodulation] Error!\n"); // ones_zeros.cu:46-49
        return 0;
    }
    return *(int32_t *)c.c.host;
}
