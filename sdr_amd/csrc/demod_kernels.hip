// demod_kernels.hip -- stand-alone AM / ones-zeros / FSK demodulators for gfx950.
//
//   am_demod_kernel    : AM<T>::demodulate (AM.h:63-77) == kernel_am_demod (am.cu:72-81)
//   ones_zeros_kernel  : the receiver's ones/zeros count (rtlsdr_receiver.cpp:2063-2074), which
//                        ones_zeros.cu:28-52 was meant to compute
//   fsk_corr_kernel    : FM<T>::demodulate (FM.h:83-148 with the SURVEY section 8c repair)
#include "sdrx_internal.h"
#include <cstdlib>
#include <map>
#include <mutex>

#include "../../include/sdrx_math.h"

namespace sdrx {

typedef double v2d __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------- AM

// ds == 1: coalesced 16-byte loads (two samples per lane and load, 1 KB per wave instruction), four per rail in
// flight; the two decisions of a load leave as one 16-bit store.  (The first version gave every lane 8 consecutive
// samples and one 8-byte store: its loads touched 32 lines each and it ran at 0.70 of the HBM peak.)
constexpr int AM_K = 4;
__global__ void __launch_bounds__(256) am_demod_dense_kernel(const double *__restrict__ re,
                                                             const double *__restrict__ im,
                                                             uint8_t *__restrict__ out, size_t n, double thr2)
{
    const size_t pair0 = (size_t)blockIdx.x * (256 * AM_K) + threadIdx.x; // index of a pair of samples
    const v2d *r2 = (const v2d *)re, *i2 = (const v2d *)im;
    const size_t npair = n / 2;
    v2d a[AM_K], b[AM_K];
#pragma unroll
    for (int k = 0; k < AM_K; k++) {
        const size_t p = pair0 + (size_t)k * 256;
        if (p < npair) a[k] = __builtin_nontemporal_load(r2 + p), b[k] = __builtin_nontemporal_load(i2 + p); // read once
    }
#pragma unroll
    for (int k = 0; k < AM_K; k++) {
        const size_t p = pair0 + (size_t)k * 256;
        if (p < npair) {
            const double e0 = a[k].x * a[k].x + b[k].x * b[k].x; // AM.h:73, sqrt folded into thr2
            const double e1 = a[k].y * a[k].y + b[k].y * b[k].y;
            *(unsigned short *)(out + 2 * p) = (unsigned short)((e0 >= thr2 ? 1u : 0u) | (e1 >= thr2 ? 0x100u : 0u));
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) // an odd last sample
        out[n - 1] = (re[n - 1] * re[n - 1] + im[n - 1] * im[n - 1]) >= thr2 ? 1 : 0;
}

// ds > 1: one kept sample per lane (AM.h:68: i = 0, ds, 2ds, ...)
__global__ void __launch_bounds__(256) am_demod_strided_kernel(const double *__restrict__ re,
                                                               const double *__restrict__ im,
                                                               uint8_t *__restrict__ out, size_t n_out, size_t ds,
                                                               double thr2)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    size_t i = k * ds;
    double a = re[i], b = im[i];
    out[k] = (a * a + b * b) >= thr2 ? 1 : 0;
}

int launch_am_demod(const double *re, const double *im, uint8_t *out, size_t n, size_t ds, hipStream_t s)
{
    if (!re || !im || !out || ds == 0) {
        set_error("am_demod: null pointer or ds == 0");
        return SDRX_EINVAL;
    }
    if (n == 0) return SDRX_OK;
    const double thr2 = am_threshold_sq();
    const bool aligned = ((((size_t)re) | ((size_t)im)) & 15) == 0 && (((size_t)out) & 1) == 0;
    if (ds == 1 && aligned) {
        const size_t blocks = (n / 2 + 256 * AM_K - 1) / (256 * AM_K);
        hipLaunchKernelGGL(am_demod_dense_kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, s, re, im,
                           out, n, thr2);
    } else {
        size_t n_out = (n + ds - 1) / ds;
        hipLaunchKernelGGL(am_demod_strided_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, re, im,
                           out, n_out, ds, thr2);
    }
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

// -------------------------------------------------------------------------------- ones / zeros

__global__ void __launch_bounds__(256) ones_zeros_kernel(const double *__restrict__ re,
                                                         const double *__restrict__ im, size_t n, double thr2,
                                                         int32_t *__restrict__ count)
{
    int local = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((((size_t)re) | ((size_t)im)) & 15) == 0) { // two samples per lane and load (16 bytes), read once: streamed
        const v2d *r2 = (const v2d *)re, *i2 = (const v2d *)im;
        const size_t npair = n / 2;
        for (size_t p = t0; p < npair; p += stride) {
            const v2d a = __builtin_nontemporal_load(r2 + p), b = __builtin_nontemporal_load(i2 + p);
            local += (a.x * a.x + b.x * b.x) >= thr2 ? 1 : 0; // rtlsdr_receiver.cpp:2066
            local += (a.y * a.y + b.y * b.y) >= thr2 ? 1 : 0;
        }
        if ((n & 1) && t0 == 0) local += (re[n - 1] * re[n - 1] + im[n - 1] * im[n - 1]) >= thr2 ? 1 : 0;
    } else {
        for (size_t i = t0; i < n; i += stride) {
            const double a = __builtin_nontemporal_load(re + i), b = __builtin_nontemporal_load(im + i);
            local += (a * a + b * b) >= thr2 ? 1 : 0;
        }
    }
    // wave-level butterfly (64 lanes), then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

int launch_ones_zeros(const double *re, const double *im, size_t n, int32_t *count, hipStream_t s)
{
    if (!re || !im || !count) {
        set_error("ones_zeros: null pointer");
        return SDRX_EINVAL;
    }
    SDRX_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (n == 0) return SDRX_OK;
    size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ones_zeros_kernel, dim3((unsigned)blocks), dim3(256), 0, s, re, im, n, am_threshold_sq(),
                       count);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

// -------------------------------------------------------------------- ones / zeros confidence walk
//
// The receiver's CPU branch (rtlsdr_receiver.cpp:2063-2074) does not count: it walks a confidence_counter,
// increment(1) for a sample with envelope >= 80, decrement(1) otherwise, clamped to [0, max] after EVERY step
// (confidence_counter.h:50-72) -- order matters once the counter touches a rail.  A step is the map
// x -> min(H, max(L, x + A)) with (A, L, H) = (+-1, 0, max), and such maps are closed under composition:
//   (g2 o g1) = (A1 + A2, clamp(L1 + A2, L2, H2), clamp(H1 + A2, L2, H2)),
// so a stream's walk is an ordered reduction.  Per wave: 64 coalesced loads, the 64 decisions of each as a
// ballot mask parked in lane j, so lane l ends up with the decisions of samples [64 l, 64 l + 64) of the
// wave's 4096; it composes them in order, the wave reduces in lane order by shuffles, the workgroup in wave
// order through LDS.  One workgroup per channel.
struct SatMap {
    long long A, L, H;
};
constexpr long long SAT_INF = 1ll << 60;

__device__ __forceinline__ long long clampll(long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ SatMap sat_then(const SatMap &g1, const SatMap &g2) // g1 first, then g2
{
    return {g1.A + g2.A, clampll(g1.L + g2.A, g2.L, g2.H), clampll(g1.H + g2.A, g2.L, g2.H)};
}

constexpr int OZC_WAVES = 16;

// STREAM: nontemporal loads (a bank of streams: 0.87 vs 0.92 ms for 512 x 2^19; a single stream is slower with them)
// WIDE: 16-byte loads, two samples per lane (rows 16-byte aligned): a wave's 64 loads cover 8192 samples; the two
// decisions of a load are two ballot masks (even and odd samples), parked together in lane j, which walks its 128
// samples as E bit 0, O bit 0, E bit 1, ...
// seg_len != 0: a stream is cut into gridDim.y segments, one workgroup each; a segment's composed map goes to
// oz_parts[channel * gridDim.y + segment] (a file-scope device array: one per GPU) and ones_zeros_fold_kernel applies the
// segments' maps in order -- for a few long streams (one radio's block), where one workgroup per stream leaves the chip
// idle.  Calls that use the array are ordered on the GPU by an event (launch_ones_zeros_confidence).
constexpr unsigned OZC_MAX_PARTS = 16384, OZC_MAX_SEGS = 64;
static __device__ SatMap oz_parts[OZC_MAX_PARTS];

template <bool STREAM, bool WIDE>
__global__ void __launch_bounds__(64 * OZC_WAVES) ones_zeros_confidence_kernel(const double *__restrict__ re_all,
                                                                                const double *__restrict__ im_all, size_t n_all,
                                                                                size_t pitch, double thr2, int32_t cmax,
                                                                                int32_t *__restrict__ counters, size_t seg_len)
{
    __shared__ SatMap part[OZC_WAVES];
    constexpr size_t PER_LANE = WIDE ? 128 : 64, PER_WAVE = 64 * PER_LANE;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *re = re_all + (size_t)blockIdx.x * pitch, *im = im_all + (size_t)blockIdx.x * pitch;
    const size_t first = seg_len ? (size_t)blockIdx.y * seg_len : 0;
    const size_t n = seg_len ? (first + seg_len < n_all ? first + seg_len : n_all) : n_all; // end of this workgroup's samples
    SatMap run = {0, -SAT_INF, SAT_INF}; // (thread 0) the walk so far
    for (size_t base = first; base < n; base += PER_WAVE * OZC_WAVES) {
        const size_t wbase = base + (size_t)wave * PER_WAVE;
        unsigned long long mine = 0, mine_odd = 0;
#pragma unroll 8
        for (int j = 0; j < 64; j++) {
            if constexpr (WIDE) {
                const size_t i = wbase + 128 * (size_t)j + 2 * (size_t)lane;
                bool one0 = false, one1 = false;
                if (i + 1 < n) {
                    const v2d a = STREAM ? __builtin_nontemporal_load((const v2d *)(re + i)) : *(const v2d *)(re + i);
                    const v2d b = STREAM ? __builtin_nontemporal_load((const v2d *)(im + i)) : *(const v2d *)(im + i);
                    one0 = (a.x * a.x + b.x * b.x) >= thr2; // rtlsdr_receiver.cpp:2066 (sqrt folded into thr2)
                    one1 = (a.y * a.y + b.y * b.y) >= thr2;
                } else if (i < n) {
                    const double a = re[i], b = im[i];
                    one0 = (a * a + b * b) >= thr2;
                }
                const unsigned long long m0 = __ballot(one0), m1 = __ballot(one1);
                if (lane == (unsigned)j) { mine = m0; mine_odd = m1; }
            } else {
                const size_t i = wbase + 64 * (size_t)j + lane;
                bool one = false;
                if (i < n) {
                    const double a = STREAM ? __builtin_nontemporal_load(re + i) : re[i];
                    const double b = STREAM ? __builtin_nontemporal_load(im + i) : im[i];
                    one = (a * a + b * b) >= thr2; // rtlsdr_receiver.cpp:2066 (sqrt folded into thr2)
                }
                const unsigned long long m = __ballot(one);
                if (lane == (unsigned)j) mine = m;
            }
        }
        // this lane's run: samples wbase + PER_LANE lane + [0, cnt)
        const size_t seg = wbase + PER_LANE * (size_t)lane;
        const int cnt = seg >= n ? 0 : (n - seg < PER_LANE ? (int)(n - seg) : (int)PER_LANE);
        SatMap g = {0, -SAT_INF, SAT_INF};
        for (int b = 0; b < cnt; b++) {
            const unsigned long long bits = WIDE ? ((b & 1) ? mine_odd : mine) : mine;
            const long long a = ((bits >> (WIDE ? (b >> 1) : b)) & 1ull) ? 1 : -1;
            g.A += a;
            g.L = clampll(g.L + a, 0, cmax);
            g.H = clampll(g.H + a, 0, cmax);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { // lane i (earlier) absorbs lane i + off (later)
            SatMap o = {__shfl_down(g.A, off), __shfl_down(g.L, off), __shfl_down(g.H, off)};
            g = sat_then(g, o);
        }
        if (lane == 0) part[wave] = g;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int w = 0; w < OZC_WAVES; w++) run = sat_then(run, part[w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (seg_len) {
            oz_parts[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = run;
        } else {
            long long x = counters[blockIdx.x];
            x = clampll(x + run.A, run.L, run.H);
            counters[blockIdx.x] = (int32_t)x;
        }
    }
}

// one wave per stream: the segments' maps in order, then the carried counter through the composed map
__global__ void __launch_bounds__(64) ones_zeros_fold_kernel(int32_t *__restrict__ counters, unsigned segs)
{
    const unsigned lane = threadIdx.x;
    SatMap g = {0, -SAT_INF, SAT_INF};
    if (lane < segs) g = oz_parts[(size_t)blockIdx.x * segs + lane];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { // lane i (earlier) absorbs lane i + off (later); identity maps beyond segs
        SatMap o = {__shfl_down(g.A, off), __shfl_down(g.L, off), __shfl_down(g.H, off)};
        if (lane + off < 64) g = sat_then(g, o);
    }
    if (lane == 0) {
        long long x = counters[blockIdx.x];
        counters[blockIdx.x] = (int32_t)clampll(x + g.A, g.L, g.H);
    }
}

struct OzOrder { // the last call that used oz_parts on a device: the next one waits for it on the GPU
    hipEvent_t ev = nullptr;
    bool used = false;
};
static std::mutex g_oz_mu;
static std::map<int, OzOrder> g_oz;

int launch_ones_zeros_confidence(const double *re, const double *im, size_t n, size_t nchan, size_t pitch, int32_t cmax,
                                 int32_t *counters, hipStream_t s)
{
    if (nchan == 0 || n == 0) return SDRX_OK;
    if (!re || !im || !counters || cmax < 0 || (nchan > 1 && pitch < n)) {
        set_error("ones_zeros_confidence: bad argument");
        return SDRX_EINVAL;
    }
    if (nchan > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    const bool wide = ((((size_t)re) | ((size_t)im)) & 15) == 0 && (nchan == 1 || pitch % 2 == 0);
    const dim3 wg(64 * OZC_WAVES);
    const double thr2 = am_threshold_sq();
    // few long streams: segments of >= one workgroup pass (131072 samples) each, about 1024 workgroups in all
    const size_t pass = (size_t)8192 * OZC_WAVES;
    size_t segs = 1;
    if (nchan < 256 && n >= 2 * pass) {
        segs = (n + pass - 1) / pass;
        const size_t want = 1024 / nchan;
        if (segs > want) segs = want;
        if (segs > OZC_MAX_SEGS) segs = OZC_MAX_SEGS;
        if (segs * nchan > OZC_MAX_PARTS) segs = 1;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (segs > 1 && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) segs = 1; // (events)
    size_t seg_len = 0;
    if (segs > 1) {
        seg_len = ((n + segs - 1) / segs + 8191) & ~(size_t)8191;
        segs = (n + seg_len - 1) / seg_len;
    }
    const dim3 grid((unsigned)nchan, (unsigned)segs);
    auto launch = [&]() {
        if (nchan >= 64) {
            if (wide) hipLaunchKernelGGL((ones_zeros_confidence_kernel<true, true>), grid, wg, 0, s, re, im, n, pitch, thr2, cmax, counters, seg_len);
            else hipLaunchKernelGGL((ones_zeros_confidence_kernel<true, false>), grid, wg, 0, s, re, im, n, pitch, thr2, cmax, counters, seg_len);
        } else {
            if (wide) hipLaunchKernelGGL((ones_zeros_confidence_kernel<false, true>), grid, wg, 0, s, re, im, n, pitch, thr2, cmax, counters, seg_len);
            else hipLaunchKernelGGL((ones_zeros_confidence_kernel<false, false>), grid, wg, 0, s, re, im, n, pitch, thr2, cmax, counters, seg_len);
        }
    };
    if (segs <= 1) {
        launch();
        SDRX_HIP(hipGetLastError());
        return SDRX_OK;
    }
    int dev = 0;
    SDRX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_oz_mu);
    OzOrder &o = g_oz[dev];
    if (!o.ev) SDRX_HIP(hipEventCreateWithFlags(&o.ev, hipEventDisableTiming));
    if (o.used) SDRX_HIP(hipStreamWaitEvent(s, o.ev, 0)); // the previous user of oz_parts on this GPU has finished
    launch();
    hipLaunchKernelGGL(ones_zeros_fold_kernel, dim3((unsigned)nchan), dim3(64), 0, s, counters, (unsigned)segs);
    SDRX_HIP(hipGetLastError());
    SDRX_HIP(hipEventRecord(o.ev, s));
    o.used = true;
    return SDRX_OK;
}

// -------------------------------------------------------------------------------- FSK correlator
//
// bit b: op1 = sum_{p=ds-1..0} x[b*ds+p]*w1[p], op2 likewise with w2 (each accumulator starts at 0
// and takes the window's LAST sample first: FM.h:94-100), bit = op1 > op2.  One lane per bit keeps
// that order; the window is streamed through LDS in chunks of KC samples per bit so that global
// reads are runs of KC*8 contiguous bytes, and lane blocks sit at an odd pitch (conflict-free).
// w1[p] / w2[p] are the same for every lane: scalar loads.
constexpr int FSK_NT = 128;
constexpr int FSK_KC = 32;

__global__ void __launch_bounds__(FSK_NT) fsk_corr_kernel(const double *__restrict__ x,
                                                          const double *__restrict__ w1,
                                                          const double *__restrict__ w2,
                                                          uint8_t *__restrict__ out, size_t nbits, unsigned ds)
{
    __shared__ double lds[FSK_NT * (FSK_KC + 1)];
    const unsigned tid = threadIdx.x;
    const size_t bit0 = (size_t)blockIdx.x * FSK_NT;
    const size_t bits_here = (nbits - bit0) < (size_t)FSK_NT ? (nbits - bit0) : (size_t)FSK_NT;
    const unsigned kc_full = ds < (unsigned)FSK_KC ? ds : (unsigned)FSK_KC;
    const unsigned kpad = kc_full | 1u;
    const unsigned nchunks = (ds + kc_full - 1) / kc_full;
    double op1 = 0.0, op2 = 0.0;
    for (int chunk = (int)nchunks - 1; chunk >= 0; chunk--) {
        const unsigned c_lo = (unsigned)chunk * kc_full;
        const unsigned kc = (ds - c_lo) < kc_full ? (ds - c_lo) : kc_full;
        const unsigned total = (unsigned)bits_here * kc;
        const unsigned magic = (unsigned)((0x100000000ULL + kc - 1) / kc); // exact e / kc for e < 2^20
        for (unsigned e = tid; e < total; e += FSK_NT) {
            unsigned b = magic ? __umulhi(e, magic) : e /* kc == 1: the magic wraps to 0 */, k = e - b * kc;
            lds[b * kpad + k] = x[(bit0 + b) * ds + c_lo + k];
        }
        __syncthreads();
        if (tid < bits_here) {
            const double *row = lds + tid * kpad;
            for (int k = (int)kc - 1; k >= 0; k--) {
                double v = row[k];
                op1 += v * w1[c_lo + k];
                op2 += v * w2[c_lo + k];
            }
        }
        __syncthreads();
    }
    if (tid < bits_here) out[bit0 + tid] = (op1 > op2) ? 1 : 0; // FM.h:139 (repaired)
}

// The same correlator for ds >= 16, one WAVE per 64 bits and no workgroup barrier: the wave walks the
// windows from their last sample down in chunks of 16 samples per bit.  A chunk is 64 runs of 128
// contiguous bytes, loaded 4 runs per instruction into registers one chunk AHEAD of the arithmetic,
// transposed through the wave's own LDS slice (pitch 17 doubles: conflict-free both ways), and each
// lane then adds its bit's 16 products in the reference's order.  Taps come by scalar loads.
// ROWS = 64 bits per wave (chunks of 16 samples) or ROWS = 16 (chunks of 64 samples: a load instruction is one bit's 512
// contiguous bytes, only 16 lanes do the arithmetic).  Windows that are not multiples of the 128-byte line make every
// line the target of two consecutive 16-sample chunks (FETCH_SIZE 1.52 x the input at ds = 100); with 64-sample chunks
// it is five lines per four, and the loads are four times longer runs: 0.54 -> 0.73 of the HBM peak at ds = 250.
constexpr int FSKW_KC = 16;

template <int ROWS>
__global__ void __launch_bounds__(256) fsk_corr_wave_kernel(const double *__restrict__ x, const double *__restrict__ w1,
                                                            const double *__restrict__ w2, uint8_t *__restrict__ out,
                                                            size_t nbits, unsigned ds)
{
    constexpr int KC = 1024 / ROWS, PITCH = KC + 1;
    constexpr int LPR = KC < 64 ? KC : 64; // lanes per row in a load instruction
    constexpr int RPI = 64 / LPR;          // rows per load instruction
    constexpr int NI = ROWS / RPI;         // load instructions per chunk (16)
    __shared__ double smem[4 * ROWS * PITCH];
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *lds = smem + wave * ROWS * PITCH;
    const size_t bit0 = ((size_t)blockIdx.x * 4 + wave) * ROWS;
    if (bit0 >= nbits) return;
    const unsigned nchunks = (ds + KC - 1) / KC;
    // lane's element i of a chunk: row RPI i + lane / LPR, sample lane % LPR of the chunk
    const unsigned k_ld = lane % LPR, row_ld = lane / LPR;
    const double *xrow[NI]; // row bases of the rows this lane loads from
    bool row_ok[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const size_t b = bit0 + RPI * i + row_ld;
        row_ok[i] = b < nbits;
        xrow[i] = x + (row_ok[i] ? b : bit0) * (size_t)ds + k_ld;
    }
    double regs[NI];
    auto load_chunk = [&](unsigned chunk) {
        const unsigned c_lo = chunk * KC;
        const bool k_ok = c_lo + k_ld < ds; // only the window's top chunk can be partial
#pragma unroll
        for (int i = 0; i < NI; i++) regs[i] = (k_ok && row_ok[i]) ? xrow[i][c_lo] : 0.0;
    };
    load_chunk(nchunks - 1);
    double op1 = 0.0, op2 = 0.0;
    for (int chunk = (int)nchunks - 1; chunk >= 0; chunk--) {
#pragma unroll
        for (int i = 0; i < NI; i++) lds[(RPI * i + row_ld) * PITCH + k_ld] = regs[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (chunk > 0) load_chunk((unsigned)chunk - 1); // in flight during this chunk's arithmetic
        const unsigned c_lo = (unsigned)chunk * KC;
        if (ROWS == 64 || lane < ROWS) {
            const double *row = lds + lane * PITCH;
            if (c_lo + KC <= ds) {
                if constexpr (KC == 16) {
                    double v[KC];
#pragma unroll
                    for (int k = 0; k < KC; k++) v[k] = row[k];
#pragma unroll
                    for (int k = KC - 1; k >= 0; k--) { // last sample first: FM.h:94-100
                        op1 += v[k] * w1[c_lo + k];
                        op2 += v[k] * w2[c_lo + k];
                    }
                } else {
#pragma unroll 16
                    for (int k = KC - 1; k >= 0; k--) {
                        const double v = row[k];
                        op1 += v * w1[c_lo + k];
                        op2 += v * w2[c_lo + k];
                    }
                }
            } else {
                for (int k = (int)(ds - c_lo) - 1; k >= 0; k--) {
                    const double v = row[k];
                    op1 += v * w1[c_lo + k];
                    op2 += v * w2[c_lo + k];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // these reads stay above the next chunk's writes
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < ROWS && bit0 + lane < nbits) out[bit0 + lane] = (op1 > op2) ? 1 : 0; // FM.h:139 (repaired)
}

// Measured and left as is (round 2, profiles/r02_fsk_study.json): windows that are not a multiple of the 128-byte line
// (ds = 100) make every line the target of two consecutive chunks, FETCH_SIZE = 1.52 x the input, 0.62 of the HBM peak
// against 0.76-0.79 for ds = 16 / 64 / 128.  A form that loads aligned blocks into a two-block LDS ring per bit (every
// line fetched once, FETCH = 1.0-1.16 x) was bit-identical but needs twice the LDS per wave: 8 instead of 16 waves per
// CU, and its dependent add chains no longer hide: 0.51 (ds = 100), 0.56 (ds = 128).  Removed.
int launch_fm_demod(const double *re, const double *w1, const double *w2, uint8_t *out, size_t n, size_t ds,
                    hipStream_t s)
{
    if (!re || !w1 || !w2 || !out || ds == 0) {
        set_error("fm_demod: null pointer or ds == 0");
        return SDRX_EINVAL;
    }
    if (ds > 0x7fffffffu) return SDRX_EUNSUPPORTED;
    size_t nbits = n / ds; // FM.h:114
    if (nbits == 0) return SDRX_OK;
    if (ds >= (size_t)FSKW_KC) {
        const int rows_tune = tuning().fsk_rows; // 0 auto | 64 | 16
        // measured (2^28 samples, fraction of the HBM peak, 64 / 16 bits per wave): ds = 64 0.78 / 0.71, 100 0.62 / 0.66,
        // 128 0.77 / 0.78, 250 0.54 / 0.73, 333 0.55 / 0.74, 500 0.56 / 0.70, 1000 0.63 / 0.65, 2000 0.61 / 0.46; a
        // single radio's block with long bits (1.2 M samples, ds = 2000: 600 bits) 155 / 114 us
        const bool mid = ds >= 96 && ds <= 1200, tiny = nbits < 4096;
        const int rows = ds >= 64 && (rows_tune == 16 || (rows_tune == 0 && (mid || tiny))) ? 16 : 64;
        const size_t per_wg = 4 * (size_t)rows, wblocks = (nbits + per_wg - 1) / per_wg;
        if (wblocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
        if (rows == 64)
            hipLaunchKernelGGL(fsk_corr_wave_kernel<64>, dim3((unsigned)wblocks), dim3(256), 0, s, re, w1, w2, out, nbits,
                               (unsigned)ds);
        else
            hipLaunchKernelGGL(fsk_corr_wave_kernel<16>, dim3((unsigned)wblocks), dim3(256), 0, s, re, w1, w2, out, nbits,
                               (unsigned)ds);
        SDRX_HIP(hipGetLastError());
        note_kernel("fsk_corr_wave_kernel", rows, 0, 0, 0);
        return SDRX_OK;
    }
    size_t blocks = (nbits + FSK_NT - 1) / FSK_NT;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(fsk_corr_kernel, dim3((unsigned)blocks), dim3(FSK_NT), 0, s, re, w1, w2, out, nbits,
                       (unsigned)ds);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

// ------------------------------------------------------------------------ rtl_fm int16 chain
//
// rtlsdr_callback's conversion (rtl_fm.c:808-822), low_pass (rtl_fm.c:303-323) and fm_demod with
// polar_discriminant (rtl_fm.c:422-435, 518-546), fused: one thread per decimated sample.  A thread
// sums its `ds` raw byte pairs (the first one of a call continues the carried partial sum), the
// sums are narrowed to int16 like `lowpassed[]`, the previous decimated sample comes from the
// neighbouring lane (across waves through LDS, across workgroups by recomputing one sum), and
// pcm = (int)(atan2(cj, cr) / 3.14159 * 2^14).  All integer steps are exact; atan2 is include/sdrx_math.h.
__device__ __forceinline__ void rtlfm_sum(const uint8_t *__restrict__ x, long long first, long long last, int rotate90,
                                          int &sr, int &sj)
{
    // Without rotate_90 the window's sums are plain byte sums: four bytes (two samples) per load, I and Q
    // picked apart by a byte dot product, 127 per sample taken off at the end (integer arithmetic: exact in any
    // order).  Half-filled dwords at either end are read as the 16 bits that belong to the window.
    if (!rotate90 && ((size_t)x & 1) == 0 && last > first) {
        const unsigned off = (unsigned)(((size_t)x & 3) >> 1); // the row starts on the second half of a dword
        const uint8_t *xa = x - 2 * off;
        long long b0 = 2 * (first + off), b1 = 2 * (last + off); // the window's bytes relative to xa
        unsigned si = 0, sq = 0;
        if (b0 & 2) {
            const unsigned v = (unsigned)*(const unsigned short *)(xa + b0);
            si += v & 0xffu, sq += v >> 8;
            b0 += 2;
        }
        if ((b1 & 2) && b1 > b0) {
            const unsigned v = (unsigned)*(const unsigned short *)(xa + b1 - 2);
            si += v & 0xffu, sq += v >> 8;
            b1 -= 2;
        }
        const unsigned *w = (const unsigned *)xa;
        for (long long d = b0 >> 2; d < (b1 >> 2); d++) {
            const unsigned v = w[d];
            si = __builtin_amdgcn_udot4(v, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(v, 0x01000100u, sq, false);
        }
        const int cnt = (int)(last - first);
        sr += (int)si - 127 * cnt;
        sj += (int)sq - 127 * cnt;
        return;
    }
    for (long long i = first; i < last; i++) { // sample index within the call
        unsigned a = x[2 * i], b = x[2 * i + 1];
        if (rotate90) {
            const unsigned ph = (unsigned)i & 3u, a0 = a, b0 = b;
            a = ph == 0 ? a0 : (ph == 1 ? 255u - b0 : (ph == 2 ? 255u - a0 : b0));
            b = ph == 0 ? b0 : (ph == 1 ? a0 : (ph == 2 ? 255u - b0 : 255u - a0));
        }
        sr += (int)a - 127;
        sj += (int)b - 127;
    }
}

// lowpassed[] pair number mm of a call (rtl_fm.c:303-323): samples [mm*ds - prev_index, (mm+1)*ds - prev_index) of
// the call; pair 0 continues the carried partial sum; narrowed to int16 like lowpassed[] (rtl_fm.c:117,313-314)
struct RtlfmCarry {
    int now_r, now_j, prev_index;
};
__device__ __forceinline__ void rtlfm_lowpassed(const uint8_t *__restrict__ x, long long mm, int ds, int rotate90,
                                                const RtlfmCarry &c, int &r, int &j)
{
    int sr = mm == 0 ? c.now_r : 0, sj = mm == 0 ? c.now_j : 0;
    long long first = mm * ds - c.prev_index, last = first + ds;
    rtlfm_sum(x, first < 0 ? 0 : first, last, rotate90, sr, sj);
    r = (int)(int16_t)sr;
    j = (int)(int16_t)sj;
}

// Power squelch (rtl_fm.c:650-667, full_demod :752-761).  First half: t = sum, p = sum of squares of the call's
// lowpassed[] values, exact in 64-bit integers in any order: rtlfm_kernel adds its workgroups' shares to acc[2 chan],
// acc[2 chan + 1] (zeroed by the launcher) while it demodulates as if the channel were open; rtlfm_post_kernel, which
// runs when every share is in, takes the decision and gives a squelched channel what fm_demod makes of a zeroed
// lowpassed[]: a row of zeros (atan2(0, 0)) and pre_r = pre_j = 0.
// second half (rtl_fm.c:661-667 and full_demod :752-761): sr = (int)sqrt((p - err) / len) < squelch_level, evaluated
// as !(v >= sq_thr) with sq_thr the smallest double whose rounded square root reaches the level (NaN / negative v
// convert to INT_MIN on the CPU: squelched)
__device__ __forceinline__ bool rtlfm_squelched(const unsigned long long *acc, size_t chan, long long m_total, double sq_thr)
{
    const long long t = (long long)acc[2 * chan], p = (long long)acc[2 * chan + 1];
    const double len = (double)(2 * m_total);
    const double dc = (double)t / len;
    const double err = (double)(t * 2) * dc - dc * dc * len;
    const double v = ((double)p - err) / len;
    return !(v >= sq_thr);
}

__global__ void __launch_bounds__(256) rtlfm_kernel(const uint8_t *__restrict__ iq, size_t n, size_t nchan,
                                                    size_t in_pitch, int ds,
                                                    int rotate90, const int32_t *__restrict__ st_in,
                                                    int32_t *__restrict__ st_out, int16_t *__restrict__ pcm,
                                                    size_t out_pitch, unsigned long long *__restrict__ acc)
{
    __shared__ int xr[256], xj[256];
    __shared__ long long red_t[4], red_p[4];
    const size_t chan = (size_t)blockIdx.z * 65535u + blockIdx.y; // grid.y is limited to 65535
    if (chan >= nchan) return;                                     // a whole workgroup
    const uint8_t *x = iq + 2 * chan * in_pitch;
    const int32_t *si = st_in + 5 * chan;
    const RtlfmCarry c = {si[0], si[1], si[2]};
    const long long m_total = ((long long)c.prev_index + (long long)n) / ds; // outputs of this call
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    int ar = 0, aj = 0;
    if (m < m_total) rtlfm_lowpassed(x, m, ds, rotate90, c, ar, aj);
    xr[threadIdx.x] = ar;
    xj[threadIdx.x] = aj;
    if (acc != nullptr) { // (uniform) this workgroup's share of the channel's power sums
        long long t = (long long)ar + aj, p = (long long)ar * ar + (long long)aj * aj;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            t += __shfl_xor(t, off);
            p += __shfl_xor(p, off);
        }
        if ((threadIdx.x & 63) == 0) red_t[threadIdx.x >> 6] = t, red_p[threadIdx.x >> 6] = p;
    }
    __syncthreads();
    if (acc != nullptr && threadIdx.x == 0) {
        const long long t = red_t[0] + red_t[1] + red_t[2] + red_t[3], p = red_p[0] + red_p[1] + red_p[2] + red_p[3];
        if (t | p) {
            atomicAdd(acc + 2 * chan, (unsigned long long)t);
            atomicAdd(acc + 2 * chan + 1, (unsigned long long)p);
        }
    }
    if (m < m_total) {
        int br = 0, bj = 0;
        if (threadIdx.x > 0) { br = xr[threadIdx.x - 1]; bj = xj[threadIdx.x - 1]; }
        else if (m == 0) { br = si[3]; bj = si[4]; }              // pre_r, pre_j (rtl_fm.c:523)
        else rtlfm_lowpassed(x, m - 1, ds, rotate90, c, br, bj); // first thread of a later workgroup
        bj = -bj;                                                  // multiply(a, conj b): rtl_fm.c:422-426,432
        const int cr = ar * br - aj * bj, cj = aj * br + ar * bj;
        const double angle = sdrx_atan2((double)cj, (double)cr);
        pcm[chan * out_pitch + (size_t)m] = (int16_t)(int)(angle / 3.14159 * (1 << 14)); // rtl_fm.c:434
        if (m == m_total - 1) { st_out[5 * chan + 3] = ar; st_out[5 * chan + 4] = aj; } // rtl_fm.c:543-544
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { // the samples after the last whole window stay in the partial sum
        int sr = m_total == 0 ? c.now_r : 0, sj = m_total == 0 ? c.now_j : 0;
        long long first = m_total * ds - c.prev_index;
        rtlfm_sum(x, first < 0 ? 0 : first, (long long)n, rotate90, sr, sj);
        st_out[5 * chan + 0] = sr;
        st_out[5 * chan + 1] = sj;
        st_out[5 * chan + 2] = (int32_t)(((long long)c.prev_index + (long long)n) % ds);
        if (m_total == 0) { st_out[5 * chan + 3] = si[3]; st_out[5 * chan + 4] = si[4]; }
    }
}

// After fm_demod (full_demod, rtl_fm.c:768-775), one workgroup per channel, in place on the channel's pcm row:
//   low_pass_simple (:325-339): result[k] = (int16)(sum of `step` results), windows over ceil(len / step) * step
//     values (the reference does not stop at len either), then result[W + 1] = result[W]; result_len = len / step;
//   deemph_filter (:598-617): avg += (d +- a/2) / a, serial, one lane (avg is a function static in rtl_fm: per channel
//     here, post_state[4 chan]);
//   dc_block_filter (:619-632): 64-bit sum, avg = (sum / len + 9 dc_avg) / 10, subtracted from every result.
constexpr int DEEMPH_CHUNK = 4096;

__global__ void __launch_bounds__(256) rtlfm_post_kernel(int16_t *__restrict__ pcm_all, size_t out_pitch, size_t n, int ds,
                                                         int step, int deemph_a, int dc_block,
                                                         const int32_t *__restrict__ st_in, int32_t *__restrict__ post_state,
                                                         const unsigned long long *__restrict__ acc, double sq_thr,
                                                         int32_t *__restrict__ st_out)
{
    __shared__ long long red[4];
    __shared__ __align__(16) int16_t stage[DEEMPH_CHUNK];
    const size_t chan = blockIdx.x;
    const unsigned tid = threadIdx.x, nt = blockDim.x; // 256, or 64 when the serial de-emphasis sets the pace
    int16_t *res = pcm_all + chan * out_pitch;
    int32_t *ps = post_state + 4 * chan;
    int len = (int)(((long long)st_in[5 * chan + 2] + (long long)n) / ds); // result_len after fm_demod
    if (acc != nullptr) { // the squelch decision (every thread the same), now that the channel's sums are complete
        const bool mute = rtlfm_squelched(acc, chan, len, sq_thr);
        if (mute) { // lowpassed[] zeroed before fm_demod (:755-757)
            for (int i = (int)tid; i < len; i += nt) res[i] = 0;
            if (tid == 0 && len > 0) st_out[5 * chan + 3] = 0, st_out[5 * chan + 4] = 0; // rtl_fm.c:543-544
        }
        if (tid == 0) ps[2] = mute ? ps[2] + 1 : 0; // squelch_hits (:754,:759)
        __syncthreads();
    }
    if (step > 1) {
        const int W = (len + step - 1) / step;
        for (int k0 = 0; k0 < W; k0 += nt) { // output k lands below every value a later window reads
            const int k = k0 + (int)tid;
            int sum = 0;
            if (k < W)
                for (int i2 = 0; i2 < step; i2++) sum += (int)res[(size_t)k * step + i2];
            __syncthreads();
            if (k < W) res[k] = (int16_t)sum;
        }
        __syncthreads();
        if (tid == 0) res[W + 1] = res[W];
        len = len / step;
        __syncthreads();
    }
    if (deemph_a > 0) { // the workgroup moves the row through LDS in chunks; one lane runs the recurrence there
        int avg = ps[0];
        const int half = deemph_a / 2;
        // which wave runs the recurrence: a SIMD issues one scalar instruction per turn whichever of its waves it is
        // from, so the serial waves of the workgroups that share a CU should sit on different SIMDs
        const unsigned serial_wave = blockIdx.x & (nt / 64 - 1);
        // magic = floor(2^32 / a) + 1 (also for a power of two: the product must lie strictly above x / a for x > 0 and
        // strictly below for x < 0): (x * magic) >> 32 (signed) == floor(x / a) for x >= 0, == trunc(x / a) - 1 for x < 0,
        // given |x| < 2^20 and 3 <= a < 2^11 (magic < 2^31, |x| * (magic * a - 2^32) < 2^32); checked over the whole
        // domain by tests/test_oracle_golden.py::test_deemph_magic_division_is_exact_on_its_whole_domain
        const unsigned magic = deemph_a >= 3 && deemph_a < 2048 ? (unsigned)(0x100000000ULL / (unsigned)deemph_a) + 1u : 0u;
        for (int i0 = 0; i0 < len; i0 += DEEMPH_CHUNK) {
            const int cn = len - i0 < DEEMPH_CHUNK ? len - i0 : DEEMPH_CHUNK;
            for (int i = (int)tid; i < cn; i += nt) stage[i] = res[i0 + i];
            __syncthreads();
            if ((tid >> 6) == serial_wave) {
                // The recurrence is one chain of 32-bit integer operations: every lane of the first wave holds the same
                // values (readfirstlane), so the chain runs on the scalar unit -- one result per cycle or two instead
                // of a vector instruction's four-cycle issue -- and lane 0 stores.  8 samples per 16-byte LDS access;
                // the next block is read before the current block's chain runs.
                uint4 *blk = reinterpret_cast<uint4 *>(stage);
                const int nblk = (cn + 7) / 8;
                int av = __builtin_amdgcn_readfirstlane(avg);
                uint4 nxt = blk[0];
                for (int b = 0; b < nblk; b++) {
                    const uint4 cur = nxt;
                    if (b + 1 < nblk) nxt = blk[b + 1];
                    unsigned w[4] = {(unsigned)__builtin_amdgcn_readfirstlane((int)cur.x), (unsigned)__builtin_amdgcn_readfirstlane((int)cur.y),
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)cur.z), (unsigned)__builtin_amdgcn_readfirstlane((int)cur.w)};
                    const int lim = cn - 8 * b < 8 ? cn - 8 * b : 8;
                    // avg never leaves the range of the int16 inputs once inside it (|step| <= |d|), so there
                    // |d +- a/2| < 2^20 and the magic multiply is the exact quotient; otherwise divide
                    if (magic && (unsigned)(av + 32768) < 65536u) {
                        // x = d +- a/2 with the sign of d; trunc(x / a) = mulhi(x, magic) for x >= 0 and mulhi(x, magic) + 1
                        // for x < 0 (the product lies less than 1/a below x / a).  A short last block runs all 8 steps
                        // and takes the avg of its last real sample back out of the packed results (int16 range).
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int v = (int)(int16_t)(w[k >> 1] >> (16 * (k & 1)));
                            const bool pos = v > av; // d > 0
                            const int x = (v - av) + (pos ? half : -half);
                            av = (pos ? av : av + 1) + __mulhi(x, (int)magic);
                            w[k >> 1] = (w[k >> 1] & ~(0xffffu << (16 * (k & 1)))) | (((unsigned)av & 0xffffu) << (16 * (k & 1)));
                        }
                        if (lim < 8) av = (int)(int16_t)(w[(lim - 1) >> 1] >> (16 * ((lim - 1) & 1)));
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) { // (unrolled: w[] must stay in registers)
                            const int v = (int)(int16_t)(w[k >> 1] >> (16 * (k & 1)));
                            const int d = v - av;
                            av += k < lim ? (d > 0 ? d + half : d - half) / deemph_a : 0;
                            w[k >> 1] = (w[k >> 1] & ~(0xffffu << (16 * (k & 1)))) | (((unsigned)av & 0xffffu) << (16 * (k & 1)));
                        }
                    }
                    if ((tid & 63) == 0) blk[b] = make_uint4(w[0], w[1], w[2], w[3]);
                }
                avg = av;
            }
            __syncthreads();
            for (int i = (int)tid; i < cn; i += nt) res[i0 + i] = stage[i];
            __syncthreads();
        }
        if (tid == 64 * serial_wave) ps[0] = avg;
        __syncthreads();
    }
    if (dc_block && len > 0) {
        long long sum = 0;
        for (int i = (int)tid; i < len; i += nt) sum += res[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if ((tid & 63) == 0) red[tid >> 6] = sum;
        __syncthreads();
        sum = red[0];
        for (unsigned w2 = 1; w2 < nt / 64; w2++) sum += red[w2];
        int avg = (int)(sum / len);
        avg = (avg + ps[1] * 9) / 10;
        __syncthreads(); // every thread has read ps[1]
        for (int i = (int)tid; i < len; i += nt) res[i] = (int16_t)((int)res[i] - avg);
        if (tid == 0) ps[1] = avg;
    }
    if (tid == 0) ps[3] = len;
}

// low_pass_real (rtl_fm.c:341-361), the last stage of full_demod when rate_out2 > 0 (:788-789): a boxcar over a
// rational rate change.  Sample i is added to now_lpr, prev_lpr_index += slow, and once it reaches fast the window's
// sum / (fast / slow) is emitted.  Which samples close a window depends only on (prev_lpr_index, slow, fast):
// output k closes at i_k = ceil(((k+1) fast - prev0) / slow) - 1, so windows are independent integer sums: one lane
// per output.  In place: output k lands at index k <= i_k, below everything later chunks read, so the row is walked in
// chunks (loads, barrier, stores).  One workgroup per channel.  lpr[2 chan] = {now_lpr, prev_lpr_index}, carried.
__global__ void __launch_bounds__(256) rtlfm_low_pass_real_kernel(int16_t *__restrict__ pcm_all, size_t out_pitch, int fast,
                                                                   int slow, int32_t *__restrict__ post_state,
                                                                   int32_t *__restrict__ lpr_all)
{
    __shared__ long long red[4];
    const size_t chan = blockIdx.x;
    const unsigned tid = threadIdx.x;
    int16_t *res = pcm_all + chan * out_pitch;
    int32_t *ps = post_state + 4 * chan, *lpr = lpr_all + 2 * chan;
    const long long len = ps[3], prev0 = lpr[1], F = fast, S = slow;
    const int now0 = lpr[0], div = fast / slow;
    const long long K = len > 0 ? (prev0 + len * S) / F : 0; // windows closed by this call
    auto close_of = [&](long long k) { return ((k + 1) * F - prev0 + S - 1) / S - 1; }; // i_k (prev0 < fast: numerator > 0)
    for (long long k0 = 0; k0 < K; k0 += 256) {
        const long long k = k0 + tid;
        int sum = 0;
        if (k < K) {
            const long long lo = k == 0 ? 0 : close_of(k - 1) + 1, hi = close_of(k);
            for (long long i = lo; i <= hi; i++) sum += (int)res[i];
            if (k == 0) sum += now0;
        }
        __syncthreads();
        if (k < K) res[k] = (int16_t)(sum / div);
    }
    // what is left after the last closed window stays in now_lpr (indices >= K: none of them was overwritten)
    const long long lo = K == 0 ? 0 : close_of(K - 1) + 1;
    long long tail = 0;
    for (long long i = lo + tid; i < len; i += 256) tail += res[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tail += __shfl_xor(tail, off);
    if ((tid & 63) == 0) red[tid >> 6] = tail;
    __syncthreads();
    if (tid == 0) {
        tail = red[0] + red[1] + red[2] + red[3];
        lpr[0] = (int32_t)(tail + (K == 0 ? now0 : 0));
        lpr[1] = (int32_t)(prev0 + len * S - K * F);
        ps[3] = (int32_t)K; // result_len = i2
    }
}

int launch_rtlfm_low_pass_real(int16_t *pcm, size_t nchan, size_t out_pitch, int rate_out, int rate_out2,
                               int32_t *post_state, int32_t *lpr_state, hipStream_t s)
{
    if (!pcm || !post_state || !lpr_state) {
        set_error("sdrx_rtlfm_low_pass_real_dev: null pointer");
        return SDRX_EINVAL;
    }
    // slow > fast divides by fast / slow == 0 in the reference; rates beyond 2^30 overflow its int arithmetic
    if (rate_out2 <= 0 || rate_out < rate_out2 || rate_out > (1 << 30)) {
        set_error("sdrx_rtlfm_low_pass_real_dev: needs 0 < rate_out2 <= rate_out <= 2^30 (got %d, %d)", rate_out2, rate_out);
        return SDRX_EINVAL;
    }
    if (nchan == 0) return SDRX_OK;
    if (nchan > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(rtlfm_low_pass_real_kernel, dim3((unsigned)nchan), dim3(256), 0, s, pcm, out_pitch, rate_out, rate_out2,
                       post_state, lpr_state);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

static int rtlfm_grid(size_t n, size_t nchan, int ds, dim3 &grid)
{
    const size_t blocks = (n / (size_t)ds + 1 + 255) / 256; // prev_index < ds adds at most one output
    if (blocks > 0x7fffffffULL || nchan > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    const unsigned gy = nchan < 65535 ? (unsigned)nchan : 65535u, gz = (unsigned)((nchan + 65534) / 65535);
    grid = dim3((unsigned)blocks, gy, gz);
    return SDRX_OK;
}

int launch_rtlfm(const uint8_t *iq, size_t n, size_t nchan, size_t in_pitch, int ds, int rotate90,
                 const int32_t *st_in, int32_t *st_out, int16_t *pcm, size_t out_pitch, hipStream_t s)
{
    if (!iq || !st_in || !st_out || !pcm || ds < 1 || st_in == st_out) {
        set_error("sdrx_rtlfm_dev: bad argument (downsample %d)", ds);
        return SDRX_EINVAL;
    }
    if (nchan == 0) return SDRX_OK;
    dim3 grid;
    int rc = rtlfm_grid(n, nchan, ds, grid);
    if (rc != SDRX_OK) return rc;
    hipLaunchKernelGGL(rtlfm_kernel, grid, dim3(256), 0, s, iq, n, nchan, in_pitch, ds, rotate90, st_in, st_out, pcm,
                       out_pitch, (unsigned long long *)nullptr);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

// smallest double v with (int)sqrt(v) >= level, i.e. sqrt_rn(v) >= level (IEEE sqrt is monotone)
static double squelch_threshold(int level)
{
    double v = (double)level * (double)level;
    while (v > 0.0 && sqrt(nextafter(v, 0.0)) >= (double)level) v = nextafter(v, 0.0);
    return v;
}

int launch_rtlfm_full(const sdrx_rtlfm_params *p, const uint8_t *iq, size_t n, size_t nchan, size_t in_pitch,
                      const int32_t *st_in, int32_t *st_out, int32_t *post_state, int64_t *work, int16_t *pcm,
                      size_t out_pitch, hipStream_t s)
{
    if (!p || !iq || !st_in || !st_out || !pcm || !post_state || p->downsample < 1 || st_in == st_out || p->squelch_level < 0 ||
        (p->squelch_level && !work)) {
        set_error("sdrx_rtlfm_full_dev: bad argument");
        return SDRX_EINVAL;
    }
    if (nchan == 0) return SDRX_OK;
    const int ds = p->downsample, step = p->post_downsample;
    if (step > 1) { // low_pass_simple reads whole windows and writes result[W + 1]
        const size_t m_max = n / (size_t)ds + 1, W = (m_max + step - 1) / step;
        if (out_pitch < W * step || out_pitch < W + 2) {
            set_error("sdrx_rtlfm_full_dev: out_pitch %zu too small for post_downsample %d (needs %zu)", out_pitch, step,
                      W * step > W + 2 ? W * step : W + 2);
            return SDRX_EINVAL;
        }
    }
    if (n / (size_t)ds + 1 > 0x3fffffffULL) return SDRX_EUNSUPPORTED;
    dim3 grid;
    int rc = rtlfm_grid(n, nchan, ds, grid);
    if (rc != SDRX_OK) return rc;
    unsigned long long *acc = nullptr;
    double thr = 0.0;
    if (p->squelch_level) {
        SDRX_HIP(hipMemsetAsync(work, 0, nchan * 2 * sizeof(int64_t), s));
        acc = (unsigned long long *)work;
        thr = squelch_threshold(p->squelch_level);
    }
    hipLaunchKernelGGL(rtlfm_kernel, grid, dim3(256), 0, s, iq, n, nchan, in_pitch, ds, p->rotate90, st_in, st_out, pcm,
                       out_pitch, acc);
    hipLaunchKernelGGL(rtlfm_post_kernel, dim3((unsigned)nchan), dim3(256), 0, s, pcm, out_pitch, n, ds, step, p->deemph_a,
                       p->dc_block, st_in, post_state, (const unsigned long long *)acc, thr, st_out);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
