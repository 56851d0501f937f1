// demod_kernels.hip -- stand-alone AM / ones-zeros / FSK demodulators for gfx950.
//
//   am_demod_kernel    : AM<T>::demodulate (AM.h:63-77) == kernel_am_demod (am.cu:72-81)
//   ones_zeros_kernel  : the receiver's ones/zeros count (rtlsdr_receiver.cpp:2063-2074), which
//                        ones_zeros.cu:28-52 was meant to compute
//   fsk_corr_kernel    : FM<T>::demodulate (FM.h:83-148 with the SURVEY section 8c repair)
#include "sdrx_internal.h"

#include "../../include/sdrx_math.h"

namespace sdrx {

// ---------------------------------------------------------------------------------------- AM

// ds == 1: 8 consecutive samples per lane (4 x 16-byte loads per rail, one 8-byte store).
__global__ void __launch_bounds__(256) am_demod_dense_kernel(const double *__restrict__ re,
                                                             const double *__restrict__ im,
                                                             uint8_t *__restrict__ out, size_t n, double thr2)
{
    size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (base >= n) return;
    if (base + 8 <= n) {
        const double2 *r2 = (const double2 *)(re + base);
        const double2 *i2 = (const double2 *)(im + base);
        unsigned long long packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double2 a = r2[k], b = i2[k];
            double e0 = a.x * a.x + b.x * b.x; // AM.h:73, sqrt folded into thr2
            double e1 = a.y * a.y + b.y * b.y;
            packed |= (unsigned long long)(e0 >= thr2) << (16 * k);
            packed |= (unsigned long long)(e1 >= thr2) << (16 * k + 8);
        }
        *(unsigned long long *)(out + base) = packed;
    } else {
        for (size_t i = base; i < n; i++) out[i] = (re[i] * re[i] + im[i] * im[i]) >= thr2 ? 1 : 0;
    }
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
    const bool aligned = ((((size_t)re) | ((size_t)im)) & 15) == 0 && (((size_t)out) & 7) == 0;
    if (ds == 1 && aligned) {
        size_t threads = (n + 7) / 8;
        hipLaunchKernelGGL(am_demod_dense_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, re, im,
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
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double a = re[i], b = im[i];
        local += (a * a + b * b) >= thr2 ? 1 : 0; // rtlsdr_receiver.cpp:2066
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

__global__ void __launch_bounds__(64 * OZC_WAVES) ones_zeros_confidence_kernel(const double *__restrict__ re_all,
                                                                                const double *__restrict__ im_all, size_t n,
                                                                                size_t pitch, double thr2, int32_t cmax,
                                                                                int32_t *__restrict__ counters)
{
    __shared__ SatMap part[OZC_WAVES];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *re = re_all + (size_t)blockIdx.x * pitch, *im = im_all + (size_t)blockIdx.x * pitch;
    SatMap run = {0, -SAT_INF, SAT_INF}; // (thread 0) the walk so far
    for (size_t base = 0; base < n; base += 4096 * OZC_WAVES) {
        const size_t wbase = base + (size_t)wave * 4096;
        unsigned long long mine = 0;
#pragma unroll 8
        for (int j = 0; j < 64; j++) {
            const size_t i = wbase + 64 * (size_t)j + lane;
            bool one = false;
            if (i < n) {
                const double a = re[i], b = im[i];
                one = (a * a + b * b) >= thr2; // rtlsdr_receiver.cpp:2066 (sqrt folded into thr2)
            }
            const unsigned long long m = __ballot(one);
            if (lane == (unsigned)j) mine = m;
        }
        // this lane's run: samples wbase + 64 lane + [0, cnt)
        const size_t seg = wbase + 64 * (size_t)lane;
        const int cnt = seg >= n ? 0 : (n - seg < 64 ? (int)(n - seg) : 64);
        SatMap g = {0, -SAT_INF, SAT_INF};
        for (int b = 0; b < cnt; b++) {
            const long long a = ((mine >> b) & 1ull) ? 1 : -1;
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
        long long x = counters[blockIdx.x];
        x = clampll(x + run.A, run.L, run.H);
        counters[blockIdx.x] = (int32_t)x;
    }
}

int launch_ones_zeros_confidence(const double *re, const double *im, size_t n, size_t nchan, size_t pitch, int32_t cmax,
                                 int32_t *counters, hipStream_t s)
{
    if (nchan == 0 || n == 0) return SDRX_OK;
    if (!re || !im || !counters || cmax < 0 || (nchan > 1 && pitch < n)) {
        set_error("ones_zeros_confidence: bad argument");
        return SDRX_EINVAL;
    }
    if (nchan > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(ones_zeros_confidence_kernel, dim3((unsigned)nchan), dim3(64 * OZC_WAVES), 0, s, re, im, n, pitch,
                       am_threshold_sq(), cmax, counters);
    SDRX_HIP(hipGetLastError());
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
constexpr int FSKW_KC = 16;

__global__ void __launch_bounds__(256) fsk_corr_wave_kernel(const double *__restrict__ x, const double *__restrict__ w1,
                                                            const double *__restrict__ w2, uint8_t *__restrict__ out,
                                                            size_t nbits, unsigned ds)
{
    constexpr int KC = FSKW_KC, PITCH = KC + 1;
    __shared__ double smem[4 * 64 * PITCH];
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *lds = smem + wave * 64 * PITCH;
    const size_t bit0 = ((size_t)blockIdx.x * 4 + wave) * 64;
    if (bit0 >= nbits) return;
    const unsigned nchunks = (ds + KC - 1) / KC;
    // lane's element i of a chunk: row 4 i + lane / 16, sample lane % 16 of the chunk
    const unsigned k_ld = lane & 15, row_ld = lane >> 4;
    const double *xrow[KC]; // row bases of the 16 rows this lane loads from
    bool row_ok[KC];
#pragma unroll
    for (int i = 0; i < KC; i++) {
        const size_t b = bit0 + 4 * i + row_ld;
        row_ok[i] = b < nbits;
        xrow[i] = x + (row_ok[i] ? b : bit0) * (size_t)ds + k_ld;
    }
    double regs[KC];
    auto load_chunk = [&](unsigned chunk) {
        const unsigned c_lo = chunk * KC;
        const bool k_ok = c_lo + k_ld < ds; // only the window's top chunk can be partial
#pragma unroll
        for (int i = 0; i < KC; i++) regs[i] = (k_ok && row_ok[i]) ? xrow[i][c_lo] : 0.0;
    };
    load_chunk(nchunks - 1);
    double op1 = 0.0, op2 = 0.0;
    for (int chunk = (int)nchunks - 1; chunk >= 0; chunk--) {
#pragma unroll
        for (int i = 0; i < KC; i++) lds[(4 * i + row_ld) * PITCH + k_ld] = regs[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (chunk > 0) load_chunk((unsigned)chunk - 1); // in flight during this chunk's arithmetic
        const unsigned c_lo = (unsigned)chunk * KC;
        const double *row = lds + lane * PITCH;
        if (c_lo + KC <= ds) {
            double v[KC];
#pragma unroll
            for (int k = 0; k < KC; k++) v[k] = row[k];
#pragma unroll
            for (int k = KC - 1; k >= 0; k--) { // last sample first: FM.h:94-100
                op1 += v[k] * w1[c_lo + k];
                op2 += v[k] * w2[c_lo + k];
            }
        } else {
            for (int k = (int)(ds - c_lo) - 1; k >= 0; k--) {
                const double v = row[k];
                op1 += v * w1[c_lo + k];
                op2 += v * w2[c_lo + k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // these reads stay above the next chunk's writes
        __builtin_amdgcn_wave_barrier();
    }
    if (bit0 + lane < nbits) out[bit0 + lane] = (op1 > op2) ? 1 : 0; // FM.h:139 (repaired)
}

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
        const size_t wblocks = (nbits + 255) / 256;
        if (wblocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
        hipLaunchKernelGGL(fsk_corr_wave_kernel, dim3((unsigned)wblocks), dim3(256), 0, s, re, w1, w2, out, nbits,
                           (unsigned)ds);
        SDRX_HIP(hipGetLastError());
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

__global__ void __launch_bounds__(256) rtlfm_kernel(const uint8_t *__restrict__ iq, size_t n, size_t nchan,
                                                    size_t in_pitch, int ds,
                                                    int rotate90, const int32_t *__restrict__ st_in,
                                                    int32_t *__restrict__ st_out, int16_t *__restrict__ pcm,
                                                    size_t out_pitch)
{
    __shared__ int xr[256], xj[256];
    const size_t chan = (size_t)blockIdx.z * 65535u + blockIdx.y; // grid.y is limited to 65535
    if (chan >= nchan) return;                                     // a whole workgroup
    const uint8_t *x = iq + 2 * chan * in_pitch;
    const int32_t *si = st_in + 5 * chan;
    const int now_r = si[0], now_j = si[1], prev_index = si[2];
    const long long m_total = ((long long)prev_index + (long long)n) / ds; // outputs of this call
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    // output m covers samples [m*ds - prev_index, (m+1)*ds - prev_index); output 0 also the carried partial sum
    auto lowpassed = [&](long long mm, int &r, int &j) {
        int sr = mm == 0 ? now_r : 0, sj = mm == 0 ? now_j : 0;
        long long first = mm * ds - prev_index, last = first + ds;
        rtlfm_sum(x, first < 0 ? 0 : first, last, rotate90, sr, sj);
        r = (int)(int16_t)sr; // lowpassed[] is int16_t (rtl_fm.c:117,313-314)
        j = (int)(int16_t)sj;
    };
    int ar = 0, aj = 0;
    if (m < m_total) lowpassed(m, ar, aj);
    xr[threadIdx.x] = ar;
    xj[threadIdx.x] = aj;
    __syncthreads();
    if (m < m_total) {
        int br, bj;
        if (threadIdx.x > 0) { br = xr[threadIdx.x - 1]; bj = xj[threadIdx.x - 1]; }
        else if (m == 0) { br = si[3]; bj = si[4]; }              // pre_r, pre_j (rtl_fm.c:523)
        else lowpassed(m - 1, br, bj);                             // first thread of a later workgroup
        bj = -bj;                                                  // multiply(a, conj b): rtl_fm.c:422-426,432
        const int cr = ar * br - aj * bj, cj = aj * br + ar * bj;
        const double angle = sdrx_atan2((double)cj, (double)cr);
        pcm[chan * out_pitch + (size_t)m] = (int16_t)(int)(angle / 3.14159 * (1 << 14)); // rtl_fm.c:434
        if (m == m_total - 1) { st_out[5 * chan + 3] = ar; st_out[5 * chan + 4] = aj; } // rtl_fm.c:543-544
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { // the samples after the last whole window stay in the partial sum
        int sr = m_total == 0 ? now_r : 0, sj = m_total == 0 ? now_j : 0;
        long long first = m_total * ds - prev_index;
        rtlfm_sum(x, first < 0 ? 0 : first, (long long)n, rotate90, sr, sj);
        st_out[5 * chan + 0] = sr;
        st_out[5 * chan + 1] = sj;
        st_out[5 * chan + 2] = (int32_t)(((long long)prev_index + (long long)n) % ds);
        if (m_total == 0) { st_out[5 * chan + 3] = si[3]; st_out[5 * chan + 4] = si[4]; }
    }
}

int launch_rtlfm(const uint8_t *iq, size_t n, size_t nchan, size_t in_pitch, int ds, int rotate90,
                 const int32_t *st_in, int32_t *st_out, int16_t *pcm, size_t out_pitch, hipStream_t s)
{
    if (!iq || !st_in || !st_out || !pcm || ds < 1 || st_in == st_out) {
        set_error("sdrx_rtlfm_dev: bad argument (downsample %d)", ds);
        return SDRX_EINVAL;
    }
    if (nchan == 0) return SDRX_OK;
    const size_t blocks = (n / (size_t)ds + 1 + 255) / 256; // prev_index < ds adds at most one output
    if (blocks > 0x7fffffffULL || nchan > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    const unsigned gy = nchan < 65535 ? (unsigned)nchan : 65535u, gz = (unsigned)((nchan + 65534) / 65535);
    hipLaunchKernelGGL(rtlfm_kernel, dim3((unsigned)blocks, gy, gz), dim3(256), 0, s, iq, n, nchan, in_pitch, ds,
                       rotate90, st_in, st_out, pcm, out_pitch);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
