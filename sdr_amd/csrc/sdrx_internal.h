// sdrx_internal.h -- shared between the translation units of libsdrx.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/sdrx.h"

namespace sdrx {

void set_error(const char *fmt, ...);

#define SDRX_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            sdrx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SDRX_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

// Smallest double s with sqrt_rn(s) >= 80 (AM<T>::THRESHOLD, AM.h:23).  The kernels compare the
// squared magnitude against it, which equals `sqrt(re*re+im*im) >= 80` of AM.h:73 for every input
// because IEEE sqrt is monotone; found on the host by stepping down from 6400 with nextafter.
double am_threshold_sq();

// 32-bit magic for q = u / d, exact for u < 2^20, 1 <= d < 2^11  (q = (u * magic) >> 32)
inline unsigned magic_u32(unsigned d) { return (unsigned)((0x100000000ULL + d - 1) / d); }

// Experiment switches.  Read from the environment ONCE, when the library is first used;
// sdrx_tune() changes one afterwards (tests, experiments).  No launch path calls getenv.
struct Tuning {
    int fir_r = 0;            // SDRX_FIR_R: outputs per lane of the workgroup-tile kernel (0 = the launcher's choice)
    int fir_kernel = 0;       // SDRX_FIR_KERNEL: 0 auto, 1 "tile", 2 "systolic"
    int fir_waves = 1;        // SDRX_FIR_WAVES: waves per workgroup of the systolic kernel
    int fir_lds_pad_kb = 0;   // SDRX_FIR_LDS_PAD_KB: extra LDS per systolic workgroup (lowers the occupancy)
    int fir_nt = 256;         // SDRX_FIR_NT: lanes per workgroup tile
    int fir_lds_soft_kb = 64; // SDRX_FIR_LDS_SOFT_KB: preferred LDS per workgroup tile
    int fir_fill = 1;         // SDRX_FIR_FILL: 0 = interior tiles stage through the general (bounds-tested) path too
    int fir_amv = 1;          // SDRX_FIR_AMV: 0 = sdrx_fir_am_dev never takes the verified fp32 kernels; 1 auto; 2 the packed-fma form; 3 the fp32 matrix-core form; 4 the int8 matrix-core form (where the job has am_work)
    int amv_nb = 2;           // SDRX_AMV_NB: blocks of 256 outputs per wave of fir_amv_mfma_kernel (2 or 4)
    int amv_dbg = 0;          // SDRX_AMV_DBG: timing experiments (wrong results)
    int amv_bytes = 1;        // SDRX_AMV_BYTES: 0 = raw-byte calls stay on fir_amv_i8_kernel alone (no fir_amv_i8_bytes_kernel)
    int amv_tpw = 0;          // SDRX_AMV_TPW: tiles per workgroup of fir_amv_i8_bytes_kernel (0 = by call size)
    int amv_stats = 0;        // SDRX_AMV_STATS: 1 = fir_amv_kernel counts its recomputed outputs (sdrx_fir_am_verify_stats)
    int fir_ct = 1;           // SDRX_FIR_CT: 0 = never use the compile-time (T = 129, D = 4) instance of the tile kernel
    int psk_split = -1;       // SDRX_PSK_SPLIT: -1 auto, 0 one-wave loops, 1 two-wave loops, 2 three-wave loops
    int fsk_rows = 0;         // SDRX_FSK_ROWS: bits per wave of the FSK correlator (0 auto, 64, 16)
    int psk_pieces = 4;       // SDRX_PSK_PIECES: pieces a call is cut into by the block pipeline (2..16)
    int psk_overlap = -1;     // SDRX_PSK_OVERLAP: -1 auto, 0 one stream, 1 block pipeline on two streams
    int iir_cw = 0;           // SDRX_IIR_CW: rows per wave of the recurrences (0 = by bank size)
    int iir_pair = -1;        // SDRX_IIR_PAIR: -1 auto, 0 one wave per row group, 1 the two-wave form of the recurrences
    int fft_nt = 0;           // SDRX_FFT_NT: threads per fft workgroup (0 = by length; 256, 512, 1024)
    int fft_ct = 1;           // SDRX_FFT_CT: 0 = never use the compile-time-length instances of the fft kernel
    int fft_tile = 0;         // SDRX_FFT_TILE: points per fft workgroup tile = transforms x length (0 = by length)
};
Tuning &tuning();

// What the launchers picked last on this thread (sdrx_last_kernel(): bench.py's roofline.kernel).  An entry point
// that launches a chain of kernels opens a KernelChain; its launches are then listed in order.
struct KernelNote {
    const char *family; // "fir_systolic_kernel", "fir_tile_kernel", "fir_mfma_kernel<NB=4>", "psk_loops_split_kernel", ...
    int a, b, c, d;     // family-specific template arguments
};
struct LastKernel {
    KernelNote k[12];
    int n = 0;
    bool chain = false;
};
extern thread_local LastKernel t_last_kernel;
inline void note_kernel(const char *family, int a = 0, int b = 0, int c = 0, int d = 0)
{
    LastKernel &l = t_last_kernel;
    if (!l.chain) l.n = 0;
    if (l.n < 12) l.k[l.n++] = KernelNote{family, a, b, c, d};
}
struct KernelChain {
    KernelChain() { t_last_kernel.n = 0; t_last_kernel.chain = true; }
    ~KernelChain() { t_last_kernel.chain = false; }
};

// launchers implemented in fir_kernels.hip
enum FirEpilogue { EPI_STORE = 0, EPI_AM = 1, EPI_DISC = 2 };
// offs_buf (optional): device memory for the workgroup-tile kernel's window-offset table (offs_cap ints), filled by a
// kernel on the job's stream -- the launch then allocates nothing even for a geometry not seen before.
int launch_fir(const sdrx_fir_job *job, FirEpilogue epi, int *offs_buf = nullptr, unsigned offs_cap = 0);
int launch_fir_mfma(const sdrx_fir_job *job, size_t n_out); // fir_mfma.hip: fast mode 2, fp64 only
namespace f32 {
// fir_kernels_f32.hip: dense FIR + AM decision, filtered in fp32 with every decision verified against a rigorous bound and
// recomputed in the reference's fp64 arithmetic when inside it (bits identical to the exact kernel by construction)
int launch_fir_amv(const sdrx_fir_job *job, unsigned long long *stats);
unsigned long long *amv_stats_dev(); // this device's four debug counters (nullptr on failure)
}

// demod_kernels.hip
int launch_am_demod(const double *re, const double *im, uint8_t *out, size_t n, size_t ds, hipStream_t s);
int launch_ones_zeros(const double *re, const double *im, size_t n, int32_t *count, hipStream_t s);
int launch_ones_zeros_confidence(const double *re, const double *im, size_t n, size_t nchan, size_t pitch, int32_t cmax,
                                 int32_t *counters, hipStream_t s);
int launch_fm_demod(const double *re, const double *w1, const double *w2, uint8_t *out, size_t n, size_t ds,
                    hipStream_t s);

int launch_rtlfm(const uint8_t *iq, size_t n, size_t nchan, size_t in_pitch, int ds, int rotate90,
                 const int32_t *st_in, int32_t *st_out, int16_t *pcm, size_t out_pitch, hipStream_t s);

int launch_rtlfm_full(const sdrx_rtlfm_params *p, const uint8_t *iq, size_t n, size_t nchan, size_t in_pitch,
                      const int32_t *st_in, int32_t *st_out, int32_t *post_state, int64_t *work, int16_t *pcm,
                      size_t out_pitch, hipStream_t s);

int launch_rtlfm_low_pass_real(int16_t *pcm, size_t nchan, size_t out_pitch, int rate_out, int rate_out2,
                               int32_t *post_state, int32_t *lpr_state, hipStream_t s);

// mod_kernels.hip
int launch_am_modulate(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, double *out, size_t len_mod,
                       size_t out_pitch, hipStream_t s);
int launch_fm_modulate(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, const double *w_mark,
                       const double *w_space, size_t spb, double *out, size_t out_pitch, hipStream_t s);
int launch_fm_convolve(const double *x, size_t len1, const double *w, size_t len2, double *out, size_t ds, hipStream_t s);

// framer_kernels.hip
int launch_deframe(const uint8_t *bits, size_t n, size_t nchan, size_t in_pitch, int crc_enable, sdrx_frame_event *events,
                   uint8_t *payloads, size_t max_events, uint64_t *summary, hipStream_t s);

// psk_kernels.hip
// carry / piece: one call cut into several launches over consecutive pieces (psk_kernels.hip: PIECE_NOT_FIRST = 1,
// PIECE_NOT_LAST = 2); carry = nchan x 4 doubles.  A whole call in one launch: carry = nullptr, piece = 0.
int launch_psk_loops(int kind, double *mf_iq, size_t m, size_t nchan, size_t mf_pitch, sdrx_psk_state *states,
                     uint8_t *decisions, size_t out_pitch, int run_timing, int run_phase, hipStream_t s,
                     double *carry = nullptr, int piece = 0);
int launch_psk_ambiguity(int kind, uint8_t *decisions, size_t m, size_t nchan, size_t out_pitch,
                         sdrx_psk_state *states, hipStream_t s);
int launch_psk_ambiguity_hits(int kind, uint8_t *decisions, size_t m, size_t nchan, size_t out_pitch,
                              sdrx_psk_state *states, unsigned long long *hits, size_t max_hits, hipStream_t s);
int launch_psk_modulate(int kind, const uint8_t *sym, size_t len, double *out_re, double *out_im, size_t zero_prefix,
                        hipStream_t s);
int fft_warm_tables(); // dsp_kernels.hip: every twiddle table (64 ... 4096 points) of the current device, for sdrx_init()
const double *psk_mf_taps_dev(); // RRC reversed (h[k] = RRC[128-k]), device resident, created on first use
extern const double RRC_COEFF[129];

} // namespace sdrx
