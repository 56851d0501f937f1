// fir_mfma.hip -- fp64 matrix-core (MFMA) Toeplitz formulation of a dense FIR: fast mode 2 of sdrx_fir_dev.
//
// Replaces, as an opt-in fast mode, Filter::execute (RadioNode/third-party/FIR-filter-class/filt.cpp:248-273)
// for long dense filters (BASELINE config 5).  Not the reference's arithmetic (it fuses multiply and
// add); bit-identical to fast mode 1, the fma chain.
#include "sdrx_internal.h"

namespace sdrx {

// ---- fp64 MFMA Toeplitz kernel: fast mode 2, dense (D = 1) long filters, one rail per launch ----
// v_mfma_f64_16x16x4_f64 computes D[i][j] = fma(A[i][3], B[3][j], ... fma(A[i][0], B[0][j], C[i][j])):
// on gfx950 it is bit for bit a sequential fma chain over its four k-steps (measured against every
// order and against separate roundings), at the fp64 vector FMA rate but with one instruction
// issued per 64 cycles.  One wave filters NB blocks of 256 consecutive outputs of one channel.
// Block b, MFMA column j, row i is output m = m0 + 256 b + 16 j + i; iteration `it` (Delta = 4 it - 16)
// gives every row its next four taps, k = Delta + i + kk, kk ascending:
//     A[i][kk]   = h[Delta + i + kk]                   (a Toeplitz fragment of the zero-padded taps, from LDS;
//                                                        the same for all blocks and all columns)
//     B_b[kk][j] = x[m0 + 256 b + 16 j - Delta - kk]   (one 8-byte load per lane: a sliding 2 KB window)
// so each output accumulates fma(x[m-k], h[k], acc) for k = 0 .. T-1 ascending, exactly the chain of
// fast mode 1 (padding taps contribute fma(x, 0, acc) = acc for finite x).  Lane layouts (probed):
// A: lane = 16 kk + i; B: lane = 16 kk + j; D: lane = 16 (i % 4) + j, register i / 4.
typedef double v4d __attribute__((ext_vector_type(4)));

struct MfmaGeom {
    const double *in;  // rail base of channel 0
    size_t n, in_pitch; // samples per channel; doubles between channels
    unsigned xs;        // doubles between consecutive samples of the rail (2 for interleaved I/Q)
    int T;
    const double *hist_in; // this rail's carried history of channel 0 (x[-1], x[-2], ...)
    double *hist_out;
    size_t st_pitch;
    double *out;
    size_t out_pitch;
    unsigned os;
    unsigned nchan, tiles; // workgroup tiles of 1024*NB outputs per channel
    unsigned wgs_per_xcd;
    unsigned long long total_tiles;
};

template <int NB>
__global__ void __launch_bounds__(256) fir_mfma_kernel(const MfmaGeom g, const double *__restrict__ taps)
{
    // LDS: hp[16 + k] = h[k] with zeros on both sides, then the workgroup's input span, one pad
    // double after every 16 samples (the 16 columns of a B fragment are 16 samples apart: pitch 17
    // doubles spreads them over all banks)
    extern __shared__ __align__(16) unsigned char smem[];
    const int T = g.T;
    const int H = T - 1;
    const int NIT = (T + 16 + 3) / 4;
    const int HP = 4 * NIT + 20;
    double *hp = (double *)smem;
    double *xs = hp + HP;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; // one contiguous run of tiles per XCD (see above)
    const unsigned long long w = (unsigned long long)xcd * g.wgs_per_xcd + slot;
    if (slot >= g.wgs_per_xcd || w >= g.total_tiles) return; // whole workgroups
    const unsigned chan = (unsigned)(w / g.tiles), tile = (unsigned)(w - (unsigned long long)chan * g.tiles);
    const double *x = g.in + (size_t)chan * g.in_pitch;
    const double *hist = g.hist_in + (size_t)chan * g.st_pitch;
    const long long mw = (long long)tile * (1024 * NB);      // first output of the workgroup (4 waves x NB blocks x 256)
    const long long sbase = mw + 16 - 4 * (long long)(NIT - 1) - 3; // lowest sample any lane touches
    const int count = 1024 * NB + 4 * (NIT - 1) + 3 - 16 + 1;       // ... up to sample mw + 1024 NB

    for (int i = (int)tid; i < HP; i += 256) {
        const int k = i - 16;
        hp[i] = (k >= 0 && k < T) ? taps[k] : 0.0;
    }
    if (sbase >= 0 && (size_t)(sbase + count) <= g.n) {
        const double *xp = x + (size_t)sbase * g.xs;
        for (int e = (int)tid; e < count; e += 256) xs[e + (e >> 4)] = xp[(size_t)e * g.xs];
    } else { // the stream's first tiles (carried history) and its last one (zeros past the end)
        for (int e = (int)tid; e < count; e += 256) {
            const long long sidx = sbase + e;
            double v = 0.0;
            if (sidx >= 0) {
                if ((size_t)sidx < g.n) v = x[(size_t)sidx * g.xs];
            } else if (-sidx - 1 < (long long)H) {
                v = hist[-sidx - 1];
            }
            xs[e + (e >> 4)] = v;
        }
    }
    __syncthreads();

    // this lane's B element of block 0 at iteration 0: sample m0 + 16 j + 16 - kk, m0 = mw + 1024 wave
    int e0 = (int)(1024 * NB * 0 + 1024 * wave) + 16 * (int)(lane & 15) + 16 - (int)(lane >> 4) + (int)(mw - sbase);
    const double *ap = hp + (lane & 15) + (lane >> 4);
    v4d acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
    constexpr int BSTEP = 256 + 16; // LDS doubles between the blocks' elements (256 samples and their pads)
    // iterations per trip: their 5 U LDS reads are issued ahead of the trip's 4 U matrix instructions.  Measured on
    // config 5 (T = 1024: 260 iterations), fraction of the 78.6 TFLOP/s peak: U = 2 0.69, 4 0.77, 8 0.84, 10 0.86, 12 0.86,
    // 16 0.86 (blocks per wave NB = 3 / 5 / 6 instead of 4: 0.76 / 0.75 / 0.67-0.77)
#ifndef SDRX_MFMA_U
#define SDRX_MFMA_U 10
#endif
    constexpr int U = SDRX_MFMA_U;
    int it = 0;
    for (; it + U <= NIT; it += U) {
        double a[U], bv[U][NB];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a[u] = ap[4 * (it + u)];
            const int e = e0 - 4 * u;
            const double *bp = xs + e + (e >> 4);
#pragma unroll
            for (int b = 0; b < NB; b++) bv[u][b] = bp[b * BSTEP];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int b = 0; b < NB; b++)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bv[u][b], acc[b], 0, 0, 0);
        e0 -= 4 * U;
    }
    for (; it < NIT; it++, e0 -= 4) {
        const double a = ap[4 * it];
        const double *bp = xs + e0 + (e0 >> 4);
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[b * BSTEP], acc[b], 0, 0, 0);
    }

    double *out = g.out + (size_t)chan * g.out_pitch;
    const long long m0 = mw + 1024 * (long long)wave;
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long m = m0 + 256 * b + 16 * (long long)(lane & 15) + 4 * r + (long long)(lane >> 4);
            if ((size_t)m < g.n) out[(size_t)m * g.os] = acc[b][r];
        }
    }
    if (tile == g.tiles - 1) { // carry the shift register: the last T-1 inputs of the call (filt.cpp:256-259)
        double *ho = g.hist_out + (size_t)chan * g.st_pitch;
        for (int k = (int)tid; k < H; k += 256) {
            const long long i = (long long)g.n - 1 - k;
            ho[k] = i >= 0 ? x[(size_t)i * g.xs] : ((-i - 1 < (long long)H) ? hist[-i - 1] : 0.0);
        }
    }
}

// fast mode 2 (fp64 MFMA): one launch per rail
int launch_fir_mfma(const sdrx_fir_job *job, size_t n_out)
{
#ifndef SDRX_MFMA_NB
#define SDRX_MFMA_NB 4
#endif
    constexpr int NB = SDRX_MFMA_NB;
    const int T = job->num_taps;
    const int H = T - 1;
    hipStream_t stream = (hipStream_t)job->stream;
    if (n_out != job->n || job->nchan > 0xffffffffULL) {
        set_error("sdrx_fir_dev fast mode 2: n_out must equal n");
        return SDRX_EINVAL;
    }
    const int rails = job->layout == SDRX_REAL ? 1 : 2;
    for (int rail = 0; rail < rails; rail++) {
        MfmaGeom g;
        memset(&g, 0, sizeof(g));
        g.n = job->n;
        g.T = T;
        g.st_pitch = job->state_pitch;
        g.hist_in = (const double *)job->state_in + (size_t)rail * H;
        g.hist_out = (double *)job->state_out + (size_t)rail * H;
        g.nchan = (unsigned)job->nchan;
        if (job->layout == SDRX_IQ_INTERLEAVED) {
            g.in = (const double *)job->in0 + rail;
            g.in_pitch = 2 * job->in_pitch;
            g.xs = 2;
        } else {
            g.in = (const double *)(rail == 0 ? job->in0 : job->in1);
            g.in_pitch = job->in_pitch;
            g.xs = 1;
        }
        if (rails == 2 && !job->out1) { // interleaved complex output
            g.out = (double *)job->out0 + rail;
            g.out_pitch = 2 * job->out_pitch;
            g.os = 2;
        } else {
            g.out = (double *)(rail == 0 ? job->out0 : job->out1);
            g.out_pitch = job->out_pitch;
            g.os = 1;
        }
        const size_t tiles = (job->n + 1024 * NB - 1) / (1024 * NB); // a workgroup = 4 waves x NB blocks x 256 outputs
        g.tiles = (unsigned)tiles;
        g.total_tiles = (unsigned long long)tiles * job->nchan;
        g.wgs_per_xcd = (unsigned)((g.total_tiles + 7) / 8);
        if (tiles > 0x7fffffffULL || g.total_tiles > 0x0fffffffULL) {
            set_error("grid too large: tiles=%zu nchan=%zu", tiles, job->nchan);
            return SDRX_EUNSUPPORTED;
        }
        const int NIT = (T + 16 + 3) / 4;
        const size_t count = (size_t)1024 * NB + 4 * (size_t)(NIT - 1) + 3 - 16 + 1;
        const size_t lds = ((size_t)(4 * NIT + 20) + count + count / 16 + 2) * sizeof(double);
        if (lds > 160 * 1024) {
            set_error("fast mode 2: %d taps do not fit LDS", T);
            return SDRX_EUNSUPPORTED;
        }
        if (lds > 48 * 1024)
            SDRX_HIP(hipFuncSetAttribute((const void *)fir_mfma_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fir_mfma_kernel<NB>, dim3(8u * g.wgs_per_xcd, 1, 1), dim3(256, 1, 1), lds, stream, g, (const double *)job->taps);
        SDRX_HIP(hipGetLastError());
    }
    // a real-rail call leaves the second history untouched; the discriminator's carried sample always is
    const size_t tail0 = rails == 1 ? (size_t)H : (size_t)2 * H, tail_n = (size_t)2 * H + 2 - tail0;
    SDRX_HIP(hipMemcpy2DAsync((double *)job->state_out + tail0, job->state_pitch * sizeof(double),
                              (const double *)job->state_in + tail0,
                              job->state_pitch * sizeof(double), tail_n * sizeof(double), job->nchan,
                              hipMemcpyDeviceToDevice, stream));
    return SDRX_OK;
}

} // namespace sdrx
