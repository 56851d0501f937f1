// dsp_kernels.hip -- the reference's DigitalSignalProcessing primitives as batched kernels
// (SURVEY section 8f rank 3).  Reference: RadioNode/src/libraries/DigitalSignalProcessing/
//   DC_Filter.cpp:12-19, LowPassFilter.cpp:7-16, NotchFilter.cpp:7-37,
//   FirstOrderDigitalIntegration.cpp:12-18, Normalize.cpp:7-43, fifth_order_filter.cpp:65-88, FFT.cpp:7-28.
// None of them has a caller in the reference; they are here so that a bank of channels can run
// them where the samples already are.  Same contract as the rest of the library: inside one
// channel every operation happens in the reference's order with one rounding each
// (-ffp-contract=off); parallelism is across channels (the recurrences), across outputs
// (normalize, the fifth-order decimator) or across butterflies of one stage (fft).
#include "sdrx_internal.h"

#include <cstdlib>
#include <mutex>

#include "../../include/sdrx_math.h"

namespace sdrx {
namespace {

// read-once / write-once rows: nontemporal accesses (build with -DSDRX_DSP_PLAIN for ordinary ones)
#ifndef SDRX_DSP_PLAIN
#define DSP_LOAD(p) __builtin_nontemporal_load(p)
#define DSP_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define DSP_LOAD(p) (*(p))
#define DSP_STORE(v, p) (*(p) = (v))
#endif

struct IirCoef {
    double c[5];
};

enum { K_DC = SDRX_IIR_DC_BLOCK, K_LP = SDRX_IIR_LOWPASS, K_NOTCH = SDRX_IIR_NOTCH, K_INT = SDRX_IIR_INTEGRATE };
constexpr int K_DCA = 100; // the DC blocker called in place (an instance of its own: no per-sample select on the call's kind)

// samples that need their index looked at before the recurrence is in its steady form
template <int KIND> constexpr int iir_prologue() { return KIND == K_NOTCH ? 2 : (KIND == K_LP ? 0 : 1); }

template <int KIND> __device__ __forceinline__ double iir_first(double x, double (&s)[4])
{
    // DC_Filter: output[0] = input[0]; integrate: array[0] stays; notch: samples 0 and 1 stay
    // and do not enter the delay line.
    if (KIND == K_DC || KIND == K_DCA) s[0] = x, s[1] = x;
    if (KIND == K_INT) s[0] = x;
    return x;
}

template <int KIND> __device__ __forceinline__ double iir_step(double x, double (&s)[4], const IirCoef &k)
{
    if (KIND == K_DC || KIND == K_DCA) { // y(n) = x(n) - x(n-1) + R*y(n-1); in place (K_DCA) input[i-1] has already been
        const double xp = KIND == K_DCA ? s[1] : s[0]; // overwritten by output[i-1] when the reference reads it (DC_Filter.cpp:18)
        double y = x - xp + k.c[0] * s[1];
        s[0] = x, s[1] = y;
        return y;
    }
    if (KIND == K_LP) { // output += (input - output) * ePow
        s[0] += (x - s[0]) * k.c[0];
        return s[0];
    }
    if (KIND == K_NOTCH) { // s = x_1, x_2, y_1, y_2
        double y = k.c[0] * x + k.c[1] * s[0] + k.c[2] * s[1] + k.c[3] * s[2] + k.c[4] * s[3];
        s[1] = s[0], s[0] = x, s[3] = s[2], s[2] = y;
        return y;
    }
    double y = s[0] + k.c[0] * x; // array[i] = array[i-1] + tStep*array[i]
    s[0] = y;
    return y;
}

// One channel per lane: the recurrences are serial in time.  A wave owns CW rows and walks them in
// chunks of IC samples (four 128-byte lines per row).  Global accesses are row-coalesced -- an
// instruction covers 64 / IC rows x IC consecutive samples -- and a wave-private LDS tile (pitch IC + 1: the
// per-lane reads are conflict-free) turns them into each lane's own run; the next chunk's loads are
// in flight while the current one is computed.  No s_barrier: nothing is shared between waves.
// IC = samples per chunk.  Measured, DC blocker, 16 rows per wave (ms): 16384 rows x 8192: 16 0.53, 32 0.46, 64 0.47;
// 131072 x 1024: 0.52 / 0.49 / 0.46; 65536 x 2048: 0.56 / 0.53 / 0.45; 1024 x 131072 (4 rows per wave): 4.55 / 3.72 / 3.49.
// (64 rows per wave with 64-sample chunks spills: 2-5 ms; the launcher never picks it.)
#ifndef SDRX_IIR_IC
#define SDRX_IIR_IC 64
#endif
constexpr int IC = SDRX_IIR_IC, IPITCH = IC + 1, RPI = 64 / IC; // RPI: rows covered by one load instruction

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int KIND, bool GUARD>
__device__ __forceinline__ void iir_chunk(double (&v)[IC], double (&s)[4], const IirCoef &k, size_t s0, size_t n)
{
    constexpr int P = iir_prologue<KIND>();
#pragma unroll
    for (int j = 0; j < IC; j++) {
        if (GUARD && s0 + j >= n) break;
        if (j < P && s0 == 0) v[j] = iir_first<KIND>(v[j], s);
        else v[j] = iir_step<KIND>(v[j], s, k);
    }
}

// CW = rows per wave (64, 16 or 4).  The chain of one row is latency-bound (2-5 dependent fp64 operations
// per sample), so a bank with few rows is spread over more waves: with CW < 64 the lanes l and l + CW
// compute the same row redundantly and only the first copy writes.
template <int KIND, int CW>
__global__ __launch_bounds__(64) void iir_kernel(const double *x, double *y, size_t n, size_t nchan, size_t pitch, IirCoef k,
                                                 double *state)
{
    constexpr int NI = (CW + RPI - 1) / RPI; // load instructions per chunk: each covers RPI rows x IC samples
    __shared__ double tile[CW * IPITCH];
    const unsigned l = threadIdx.x, sub = l / IC, col = l % IC, cl = l % CW;
    const size_t c0 = (size_t)blockIdx.x * CW, c = c0 + cl;
    double s[4] = {0, 0, 0, 0};
    const bool carried = (KIND == K_LP || KIND == K_NOTCH) && state != nullptr && c < nchan;
    if (carried)
        for (int j = 0; j < 4; j++) s[j] = state[c * 4 + j];
    // load/store role of this lane: rows c0 + 4i + sub (i < NI), sample s0 + col
    const size_t row0 = (c0 + sub) * pitch + col;
    // a wave whose CW rows all exist moves its full chunks through wave-uniform row pointers plus one lane offset: no
    // per-access predicate, no per-lane 64-bit address (the kernel is bound by the instructions a wave issues)
    const bool rows_full = RPI == 1 && c0 + CW <= nchan;
    const double *xb = x + c0 * pitch;
    double *yb = y + c0 * pitch;
    auto load = [&](double (&nxt)[NI], size_t s0) {
        if (rows_full && s0 + IC <= n) {
            const double *p = xb + s0;
#pragma unroll
            for (int i = 0; i < NI; i++) nxt[i] = DSP_LOAD(p + (size_t)i * pitch + l);
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; i++)
            nxt[i] = (RPI * i + sub < CW && c0 + RPI * i + sub < nchan && s0 + col < n) ? DSP_LOAD(x + row0 + (size_t)(RPI * i) * pitch + s0) : 0.0;
    };
    // one chunk: its samples (loaded two chunks ago into nxt) go to the tile, the chunk two ahead is requested into the
    // same registers -- with one wave per SIMD the loads of ONE chunk in flight (CW x 512 bytes) do not cover the HBM latency
    auto chunk = [&](double (&nxt)[NI], size_t s0) {
#pragma unroll
        for (int i = 0; i < NI; i++)
            if (RPI * i + sub < CW) tile[(RPI * i + sub) * IPITCH + col] = nxt[i];
        wave_lds_sync();
        if (s0 + 2 * IC < n) load(nxt, s0 + 2 * IC);
        double v[IC];
#pragma unroll
        for (int j = 0; j < IC; j++) v[j] = tile[cl * IPITCH + j];
        if (s0 + IC <= n) iir_chunk<KIND, false>(v, s, k, s0, n);
        else iir_chunk<KIND, true>(v, s, k, s0, n);
        wave_lds_sync();
        if (l < CW) {
#pragma unroll
            for (int j = 0; j < IC; j++) tile[cl * IPITCH + j] = v[j];
        }
        wave_lds_sync();
        if (rows_full && s0 + IC <= n) {
            double *p = yb + s0;
#pragma unroll
            for (int i = 0; i < NI; i++) DSP_STORE(tile[i * IPITCH + l], p + (size_t)i * pitch + l);
        } else {
#pragma unroll
            for (int i = 0; i < NI; i++)
                if (RPI * i + sub < CW && c0 + RPI * i + sub < nchan && s0 + col < n) DSP_STORE(tile[(RPI * i + sub) * IPITCH + col], y + row0 + (size_t)(RPI * i) * pitch + s0);
        }
        wave_lds_sync();
    };
    double nxt_a[NI], nxt_b[NI];
    load(nxt_a, 0);
    if (IC < n) load(nxt_b, IC);
    for (size_t s0 = 0; s0 < n; s0 += 2 * IC) {
        chunk(nxt_a, s0);
        if (s0 + IC < n) chunk(nxt_b, s0 + IC);
    }
    if (carried && l < CW)
        for (int j = 0; j < 4; j++) state[c * 4 + j] = s[j];
}

// Two-wave form for banks that give a SIMD no more than one wave (a lone wave issues an instruction per ~8 cycles, and
// the chain is a quarter of what a chunk costs it): the two waves of a workgroup share CW rows and take the chunks in
// turn.  In round r the wave r % 2 runs the chain of chunk r (state handed over through LDS) while the other one
// writes back chunk r - 1, which it ran the round before, and stages chunk r + 1, its next; one s_barrier per round,
// the same count for both waves.  Same arithmetic in the same order per row: bit-identical to iir_kernel.
template <int KIND, int CW>
__global__ __launch_bounds__(128) void iir_pair_kernel(const double *x, double *y, size_t n, size_t nchan, size_t pitch, IirCoef k,
                                                       double *state)
{
    static_assert(RPI == 1, "one row per load instruction");
    constexpr int NI = CW;
    __shared__ double tiles[2][CW * IPITCH];
    __shared__ double hand[CW][4];
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63, cl = l % CW;
    double *tile = tiles[w];
    const size_t c0 = (size_t)blockIdx.x * CW, c = c0 + cl;
    double s[4] = {0, 0, 0, 0};
    const bool carried = (KIND == K_LP || KIND == K_NOTCH) && state != nullptr && c < nchan;
    if (carried)
        for (int j = 0; j < 4; j++) s[j] = state[c * 4 + j];
    const bool rows_full = c0 + CW <= nchan;
    const double *xb = x + c0 * pitch;
    double *yb = y + c0 * pitch;
    const size_t nchunks = (n + IC - 1) / IC;
    double nxt[NI], v[IC];
    auto load = [&](size_t ch) {
        const size_t s0 = ch * IC;
        if (ch >= nchunks) return;
        if (rows_full && s0 + IC <= n) {
            const double *p = xb + s0;
#pragma unroll
            for (int i = 0; i < NI; i++) nxt[i] = DSP_LOAD(p + (size_t)i * pitch + l);
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) nxt[i] = (c0 + i < nchan && s0 + l < n) ? DSP_LOAD(xb + (size_t)i * pitch + s0 + l) : 0.0;
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NI; i++) tile[i * IPITCH + l] = nxt[i];
        wave_lds_sync();
    };
    if (w == 0) { // wave 0 runs chunk 0 in round 0: staged here; wave 1 stages chunk 1 during round 0
        load(0);
        if (nchunks > 0) stage();
        load(2);
    } else {
        load(1);
    }
    for (size_t r = 0; r < nchunks + 1; r++) {
        if ((r & 1) == w) {
            if (r < nchunks) {
                const size_t s0 = r * IC;
                if (r > 0)
                    for (int j = 0; j < 4; j++) s[j] = hand[cl][j];
#pragma unroll
                for (int j = 0; j < IC; j++) v[j] = tile[cl * IPITCH + j];
                if (s0 + IC <= n) iir_chunk<KIND, false>(v, s, k, s0, n);
                else iir_chunk<KIND, true>(v, s, k, s0, n);
                if (l < CW)
                    for (int j = 0; j < 4; j++) hand[cl][j] = s[j];
            }
        } else {
            if (r >= 1) { // chunk r - 1: this wave ran it in the round before, its results are in v
                const size_t s0 = (r - 1) * IC;
                wave_lds_sync();
                if (l < CW) {
#pragma unroll
                    for (int j = 0; j < IC; j++) tile[cl * IPITCH + j] = v[j];
                }
                wave_lds_sync();
                if (rows_full && s0 + IC <= n) {
                    double *p = yb + s0;
#pragma unroll
                    for (int i = 0; i < NI; i++) DSP_STORE(tile[i * IPITCH + l], p + (size_t)i * pitch + l);
                } else {
#pragma unroll
                    for (int i = 0; i < NI; i++)
                        if (c0 + i < nchan && s0 + l < n) DSP_STORE(tile[i * IPITCH + l], yb + (size_t)i * pitch + s0 + l);
                }
                wave_lds_sync();
            }
            if (r + 1 < nchunks) { // chunk r + 1 waits in nxt
                stage();
                load(r + 3);
            }
        }
        __syncthreads();
    }
    if (carried && w == 0 && l < CW) {
        if (nchunks > 0)
            for (int j = 0; j < 4; j++) s[j] = hand[cl][j];
        for (int j = 0; j < 4; j++) state[c * 4 + j] = s[j];
    }
}

template <int KIND>
void launch_iir(int cw, bool pair, const double *x, double *y, size_t n, size_t nchan, size_t pitch, const IirCoef &k, double *state,
                hipStream_t s)
{
    dim3 grid((unsigned)((nchan + cw - 1) / cw)), block(64);
    if (cw != 64 && pair) {
        if (cw == 16) iir_pair_kernel<KIND, 16><<<grid, dim3(128), 0, s>>>(x, y, n, nchan, pitch, k, state);
        else iir_pair_kernel<KIND, 4><<<grid, dim3(128), 0, s>>>(x, y, n, nchan, pitch, k, state);
        return;
    }
    if (cw == 64) iir_kernel<KIND, 64><<<grid, block, 0, s>>>(x, y, n, nchan, pitch, k, state);
    else if (cw == 16) iir_kernel<KIND, 16><<<grid, block, 0, s>>>(x, y, n, nchan, pitch, k, state);
    else iir_kernel<KIND, 4><<<grid, block, 0, s>>>(x, y, n, nchan, pitch, k, state);
}

// ---- normalize (Normalize.cpp:7-22): size_t running maximum, then (x / max)*2 - 1 ----
// The running maximum only ever takes the values (size_t)|x[i]| of samples that exceed it, so
// its final value is max((size_t)prev_max, max_i (size_t)|x[i]|) in any order: a reduction.
// Non-negative doubles order like their bit patterns, hence the integer atomic.
__global__ void norm_init_kernel(double *mx, size_t nchan)
{
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchan) mx[c] = (double)(unsigned long long)mx[c]; // size_t max = prev_max
}

constexpr int NORM_PER_THREAD = 8;

__global__ __launch_bounds__(256) void norm_max_kernel(const double *x, size_t n, size_t nchan, size_t pitch, double *mx)
{
    const size_t c = (size_t)blockIdx.y + (size_t)blockIdx.z * gridDim.y;
    if (c >= nchan) return;
    const double *xp = x + c * pitch;
    size_t i0 = (size_t)blockIdx.x * (256 * NORM_PER_THREAD) + threadIdx.x;
    double m = 0.0;
    for (int j = 0; j < NORM_PER_THREAD; j++) {
        size_t i = i0 + (size_t)j * 256;
        if (i < n) {
            double a = trunc(fabs(xp[i]));
            if (a > m) m = a; // false for NaN, like `max < abs(x)`
        }
    }
    unsigned long long b = (unsigned long long)__double_as_longlong(m);
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(b, o);
        b = t > b ? t : b;
    }
    if ((threadIdx.x & 63) == 0 && b != 0) atomicMax(reinterpret_cast<unsigned long long *>(mx + c), b);
}

__global__ __launch_bounds__(256) void norm_scale_kernel(double *x, size_t n, size_t nchan, size_t pitch, const double *mx)
{
    const size_t c = (size_t)blockIdx.y + (size_t)blockIdx.z * gridDim.y;
    if (c >= nchan) return;
    double *xp = x + c * pitch;
    const double m = mx[c];
    size_t i0 = (size_t)blockIdx.x * (256 * NORM_PER_THREAD) + threadIdx.x;
    for (int j = 0; j < NORM_PER_THREAD; j++) {
        size_t i = i0 + (size_t)j * 256;
        if (i < n) DSP_STORE((DSP_LOAD(xp + i) / m) * 2.0 - 1.0, xp + i);
    }
}

// ---- fifth_order_filter::decimate(double*, size_t, size_t) (fifth_order_filter.cpp:65-88) ----
// Per pass: data[0] = F(state); then for i = 2, 4, ... < n the window slides by one (even) sample pair
// and data[i/2] = F(window).  With pair(j) = (data[2j-2], data[2j]) for j >= 2, pair(1) = (new data[0],
// data[2]), pair(0) = (r_e, r_f), pair(-1) = (r_c, r_d) at pass entry, output k = F(pair(k-2), pair(k-1),
// pair(k)): independent outputs.  In place: output k lands below every sample a later output reads, so one
// workgroup walks the row in chunks, loads before a barrier, stores after it.
struct Pair {
    double e, f;
};

template <int STRIDE>
__device__ __forceinline__ Pair fifth_pair(const double *d, long j, const double (&s)[6], double d0n)
{
    if (j == -1) return {s[2], s[3]};
    if (j == 0) return {s[4], s[5]};
    if (j == 1) return {d0n, d[2 * STRIDE]};
    return {d[(2 * j - 2) * STRIDE], d[2 * j * STRIDE]};
}

__device__ __forceinline__ double fifth_out(Pair ab, Pair cd, Pair ef)
{
    const double a = 1, b = 5, c = 10, d = 11;
    return ((a * (ab.e + ef.f)) + ((ab.f + ef.e) * b) + ((cd.e + cd.f) * c)) / d;
}

// BLOCK = false: the double* overload (one rail, every pass over the same n).  BLOCK = true: the BBP_Block
// overload (:28-63): the real and the imaginary rail of interleaved complex points (each its own six state
// doubles; one workgroup takes both, so a point is loaded and stored whole), and n_points shrinks to dst + 1 after every pass (3 if the pass had no iteration: the
// reference's `dst = 2` start value).
// With E[j] = data[2j] (E[0] = the new data[0]) output k >= 3 is F over E[k-3 .. k].  A chunk of FIFTH_CHUNK outputs
// needs the samples 2(k0-3) .. 2(k0+FIFTH_CHUNK-1): loaded as 16-byte pairs (whole lines, coalesced; the odd samples
// are never used but share the lines), the even halves kept in LDS, one barrier per chunk (two LDS buffers).
constexpr int FIFTH_CHUNK = 1024, FIFTH_PER = FIFTH_CHUNK / 256;

template <bool BLOCK>
__global__ __launch_bounds__(256) void fifth_kernel(double *data, size_t n, size_t pitch, size_t by, double *state)
{
    constexpr int STRIDE = BLOCK ? 2 : 1, RAILS = BLOCK ? 2 : 1; // BLOCK: both rails of a point travel together
    __shared__ double ev[2][RAILS][FIFTH_CHUNK + 4]; // E[k0-3 .. k0+FIFTH_CHUNK-1] of the chunk at [0 .. FIFTH_CHUNK+2]
    const size_t c = blockIdx.x;
    double *d = data + c * pitch * STRIDE;
    double *sp = state + c * (6 * RAILS);
    // 16-byte accesses need the row's first sample on a 16-byte boundary
    const bool vec = (reinterpret_cast<size_t>(d) & 15) == 0;
    double s[RAILS][6];
    for (int r = 0; r < RAILS; r++)
        for (int j = 0; j < 6; j++) s[r][j] = sp[6 * r + j];
    size_t np = n;
    int buf = 0;
    for (size_t dec = 2; dec <= by; dec *= 2) {
        const long K = (long)((np - 1) / 2); // iterations i = 2k < np, k = 1..K
        double d0n[RAILS], ns[RAILS][6], head[RAILS];
        const bool heads = threadIdx.x >= 1 && threadIdx.x <= 3 && (long)threadIdx.x <= K; // outputs 1 .. 3 reach into
        for (int r = 0; r < RAILS; r++) {                                                   // the carried state
            d0n[r] = fifth_out({s[r][0], s[r][1]}, {s[r][2], s[r][3]}, {s[r][4], s[r][5]});
            for (int j = 0; j < 6; j++) ns[r][j] = s[r][j];
            if (K >= 1) { // the window of the last iteration, read before anything is overwritten
                Pair p0 = fifth_pair<STRIDE>(d + r, K - 2, s[r], d0n[r]), p1 = fifth_pair<STRIDE>(d + r, K - 1, s[r], d0n[r]),
                     p2 = fifth_pair<STRIDE>(d + r, K, s[r], d0n[r]);
                ns[r][0] = p0.e, ns[r][1] = p0.f, ns[r][2] = p1.e, ns[r][3] = p1.f, ns[r][4] = p2.e, ns[r][5] = p2.f;
            }
            head[r] = 0;
            if (heads) {
                const long k = threadIdx.x;
                head[r] = fifth_out(fifth_pair<STRIDE>(d + r, k - 2, s[r], d0n[r]), fifth_pair<STRIDE>(d + r, k - 1, s[r], d0n[r]),
                                    fifth_pair<STRIDE>(d + r, k, s[r], d0n[r]));
            }
        }
        __syncthreads(); // every thread has read what it needs of the row as the pass found it
        for (long k0 = 4; k0 <= K || k0 == 4; k0 += FIFTH_CHUNK, buf ^= 1) {
            // E[k0-3 + i], i < FIFTH_CHUNK + 3, wherever an output of this chunk needs it (k <= K reads up to E[K])
            const long e0 = k0 - 3, need = (K - e0 + 1 < FIFTH_CHUNK + 3 ? K - e0 + 1 : FIFTH_CHUNK + 3);
            for (long i = threadIdx.x; i < need; i += 256) {
                const double *src = d + (size_t)(2 * (e0 + i)) * STRIDE;
                if (BLOCK) {
                    double2 v = vec ? *reinterpret_cast<const double2 *>(src) : make_double2(src[0], src[1]);
                    ev[buf][0][i] = v.x, ev[buf][RAILS - 1][i] = v.y;
                } else { // (scalar for the row's last sample when n is odd: its pair would end past the row)
                    ev[buf][0][i] = vec && (size_t)(2 * (e0 + i)) + 1 < n ? reinterpret_cast<const double2 *>(src)->x : *src;
                }
            }
            __syncthreads();
            if (k0 == 4 && heads)
                for (int r = 0; r < RAILS; r++) d[threadIdx.x * STRIDE + r] = head[r];
#pragma unroll
            for (int q = 0; q < FIFTH_PER; q++) {
                const long i = threadIdx.x + 256 * q, k = k0 + i;
                if (k <= K) { // E[k-3 .. k] = ev[i .. i+3]
                    double o[RAILS];
                    for (int r = 0; r < RAILS; r++) {
                        const double e0v = ev[buf][r][i], e1v = ev[buf][r][i + 1], e2v = ev[buf][r][i + 2], e3v = ev[buf][r][i + 3];
                        o[r] = fifth_out({e0v, e1v}, {e1v, e2v}, {e2v, e3v});
                    }
                    if (BLOCK && vec) *reinterpret_cast<double2 *>(d + k * STRIDE) = make_double2(o[0], o[RAILS - 1]);
                    else
                        for (int r = 0; r < RAILS; r++) d[k * STRIDE + r] = o[r];
                }
            }
            if (k0 > K) break; // (the one pass through the loop that a short row makes for its head outputs)
        }
        for (int r = 0; r < RAILS; r++) {
            if (threadIdx.x == 0) d[r] = d0n[r];
            for (int j = 0; j < 6; j++) s[r][j] = ns[r][j];
        }
        if (BLOCK) np = (size_t)(K >= 1 ? K : 2) + 1;
        __syncthreads(); // the next pass reads what this one wrote
    }
    if (threadIdx.x == 0)
        for (int r = 0; r < RAILS; r++)
            for (int j = 0; j < 6; j++) sp[6 * r + j] = s[r][j];
}

// ---- fft (FFT.cpp:7-28): recursive radix-2 DIT == bit-reversal + log2 N butterfly stages with the
// same arithmetic per butterfly: t = polar(1, -2*PI*k/len) * odd; (even + t, even - t).
// (-2*PI*k)/len == (-2*PI*(k*N/len))/N bit for bit (powers of two), so one table of N/2 twiddles serves
// every stage; it is computed once per (device, N) and kept.  One workgroup per transform, data and
// twiddles in LDS; three stages at a time on 8 elements in registers (which butterflies meet in one
// thread changes nothing in their arithmetic), so the data crosses LDS ceil(log2 N / 3) times.
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void fft_twiddle_kernel(v2d *tw, unsigned N)
{
    unsigned j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N / 2) return;
    double theta = -2 * 3.141592653589793238460 * (double)j / (double)N, sn, cs;
    sdrx_sincos(theta, &sn, &cs);
    tw[j] = v2d{cs, sn}; // std::polar(1.0, theta)
}

// LDS layouts (slot = 16 bytes).  A 16-byte read is served in four groups of 16 lanes on banks = slot mod 16, a 16-byte
// write in eight groups of 8 lanes on slot mod 8 (MI355X LDS).  Data: the index with its higher octal digits folded into
// the lowest by xor -- a bijection that needs no spare slots and leaves the bit-reversed scatter, every stage group's
// strided reads and writes, and the final row read (nearly) conflict-free.  Twiddles: the same with hexadecimal digits,
// so that the multiples of 2^m a stage reads spread over all 16 banks (they sat on one: 21 LDS cycles per read instead of
// 4; profiles/probes/fft_lds_sim.py models every access).  Both folds are linear over xor: fold(a + b) = fold(a) ^ fold(b) where a
// and b share no bits, which lets the wave-uniform part of an index be folded on the scalar unit.
__device__ __forceinline__ unsigned fft_pad(unsigned i) { return i ^ ((i >> 3) & 7) ^ ((i >> 6) & 7) ^ ((i >> 9) & 7); }
__device__ __forceinline__ unsigned fft_twz(unsigned j) { return j ^ ((j >> 4) & 15) ^ ((j >> 8) & 15); }
static_assert(SDRX_FFT_MAX <= 4096, "fft_pad folds four octal digits, fft_twz three hexadecimal ones");

// A workgroup's tile holds B = 2^log2b transforms of N points side by side (index b * N + i); a thread's group of 2^R
// elements lies inside one transform, and the twiddle index depends on the position inside it only.
template <int R, int NT>
__device__ __forceinline__ void fft_group(v2d *x, const v2d *tw, unsigned st, unsigned log2n, unsigned total)
{
    constexpr unsigned E = 1u << R;
    const unsigned half = 1u << st, per = (1u << log2n) >> R; // groups per transform
    for (unsigned t = threadIdx.x; t < total / E; t += NT) {
        const unsigned tb = t >> (log2n - R), tt = t & (per - 1);
        const unsigned lo = tt & (half - 1), base = (tb << log2n) + ((tt >> st) << (st + R)) + lo;
        const unsigned fb = fft_pad(base); // bits st .. st + R - 1 of base are clear
        v2d v[E];
#pragma unroll
        for (unsigned q = 0; q < E; q++) v[q] = x[fb ^ fft_pad(q << st)];
#pragma unroll
        for (unsigned r = 0; r < (unsigned)R; r++) { // stage st + r pairs the elements whose q differs in bit r
            const unsigned shift = log2n - 1 - st - r;
            const unsigned fl = fft_twz(lo << shift); // lo < 2^st; the q part of the index lies above it
#pragma unroll
            for (unsigned q = 0; q < E; q++) {
                if (q & (1u << r)) continue;
                const v2d w = tw[fl ^ fft_twz((q & ((1u << r) - 1)) << (st + shift))], o = v[q | (1u << r)], e = v[q];
                const double tr = w.x * o.x - w.y * o.y, ti = w.x * o.y + w.y * o.x;
                v[q] = v2d{e.x + tr, e.y + ti};
                v[q | (1u << r)] = v2d{e.x - tr, e.y - ti};
            }
        }
#pragma unroll
        for (unsigned q = 0; q < E; q++) x[fb ^ fft_pad(q << st)] = v[q];
    }
    __syncthreads();
}

// Persistent workgroups of NT threads: each walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (a tile = B
// consecutive rows, so that short transforms still give every thread a group; ONE: B = 1 known at compile time); the
// twiddles are staged once, and the NEXT tile's samples are loaded into registers (at most SDRX_FFT_MAX / NT per thread)
// before the current one's stages run, so that the load latency is not a phase of its own.  L2 != 0: log2 N known at
// compile time (every stride, shift and folded constant becomes an immediate); L2 == 0: any length.  Whole-transform
// tiles stream their rows with nontemporal accesses (measured: 2-4 % faster there, 7 % slower on tiles of short rows).
// NOREV: the rows are already in bit-reversed order (one block of a longer transform whose global permutation has been
// done: sdrx_fft_dev above SDRX_FFT_MAX points).
template <int NT, bool ONE, int L2, bool NOREV = false>
__global__ __launch_bounds__(NT) void fft_kernel(double *iq, const v2d *__restrict__ tw_g, unsigned log2n_, unsigned log2b_,
                                                 size_t pitch, unsigned nchan)
{
    constexpr unsigned PER = SDRX_FFT_MAX / NT;
    extern __shared__ v2d fft_lds[];
    const unsigned log2n = L2 ? (unsigned)L2 : log2n_, log2b = ONE ? 0 : log2b_;
    const unsigned N = 1u << log2n, total = N << log2b, ntiles = (nchan + (1u << log2b) - 1) >> log2b;
    v2d *tw = fft_lds, *x = fft_lds + N / 2;
    for (unsigned j = threadIdx.x; j < N / 2; j += NT) tw[fft_twz(j)] = tw_g[j];
    v2d nxt[PER];
    auto fetch = [&](unsigned tile) {
        const v2d *g = reinterpret_cast<const v2d *>(iq) + (size_t)(tile << log2b) * pitch;
#pragma unroll
        for (unsigned q = 0; q < PER; q++) {
            const unsigned i = threadIdx.x + NT * q;
            if (ONE) {
                if (i < N) nxt[q] = DSP_LOAD(g + i);
            } else if (i < total && (tile << log2b) + (i >> log2n) < nchan)
                nxt[q] = g[(size_t)(i >> log2n) * pitch + (i & (N - 1))];
        }
    };
    // element q of this thread is index i = threadIdx.x + NT * q of the tile.  Where NT <= N the two terms share no
    // bits, neither before nor after the bit reversal inside the transform, so the fold of each is taken apart: the
    // thread's own once, the q part on the scalar unit.
    const bool split = NT <= N;
    const unsigned own_fwd = fft_pad(threadIdx.x), own_rev = fft_pad(__brev(threadIdx.x) >> (32 - log2n));
    auto slot_rev = [&](unsigned q) { // where element q goes: bit-reversed inside its transform
        const unsigned hi = NT * q;
        if constexpr (NOREV) return fft_pad(threadIdx.x + hi);
        if (split) return own_rev ^ fft_pad((hi & ~(N - 1)) + (__brev(hi & (N - 1)) >> (32 - log2n)));
        const unsigned i = threadIdx.x + hi;
        return fft_pad((i & ~(N - 1)) + (__brev(i & (N - 1)) >> (32 - log2n)));
    };
    if (blockIdx.x < ntiles) fetch(blockIdx.x);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads(); // the previous tile's last reads of x are done (and, the first time, tw is complete)
#pragma unroll
        for (unsigned q = 0; q < PER; q++)
            if (threadIdx.x + NT * q < total) x[slot_rev(q)] = nxt[q];
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        __syncthreads();
        if (L2) {
#pragma unroll
            for (unsigned st = 0; st + 3 <= (unsigned)L2; st += 3) fft_group<3, NT>(x, tw, st, L2, total);
            if (L2 % 3 == 2) fft_group<2, NT>(x, tw, L2 - 2, L2, total);
            if (L2 % 3 == 1) fft_group<1, NT>(x, tw, L2 - 1, L2, total);
        } else {
            unsigned st = 0;
            for (; st + 3 <= log2n; st += 3) fft_group<3, NT>(x, tw, st, log2n, total);
            if (log2n - st == 2) fft_group<2, NT>(x, tw, st, log2n, total);
            if (log2n - st == 1) fft_group<1, NT>(x, tw, st, log2n, total);
        }
        v2d *g = reinterpret_cast<v2d *>(iq) + (size_t)(tile << log2b) * pitch;
        for (unsigned hi = 0; hi < total; hi += NT) {
            const unsigned i = threadIdx.x + hi;
            if (i >= total) break;
            const v2d v = x[own_fwd ^ fft_pad(hi)]; // hi has no bit below NT: fft_pad(i) taken apart
            if (ONE) DSP_STORE(v, g + i);
            else if ((tile << log2b) + (i >> log2n) < nchan) g[(size_t)(i >> log2n) * pitch + (i & (N - 1))] = v;
        }
    }
}

constexpr int FFT_MAX_LOG2 = 12;
static_assert((1 << FFT_MAX_LOG2) == SDRX_FFT_MAX, "table size");
struct FftTable {
    v2d *p = nullptr;
    hipEvent_t filled = nullptr; // recorded behind the kernel that fills the table
    bool done = false;           // that event has been seen complete: nothing to wait for any more
};
FftTable g_fft_tw[64][FFT_MAX_LOG2 + 1];
bool g_fft_attr[64];
std::mutex g_fft_mu;

// Twiddle table of one length on the current device.  The first call for a length allocates it and fills it on `s`
// (hipMalloc: not possible while `s` is being captured -- sdrx_init() makes all twelve tables, 64 KB, so that a caller
// that ran it never gets here); calls on other streams wait for the fill through an event, never through a stream sync.
int fft_twiddles(unsigned log2n, hipStream_t s, const v2d **out)
{
    int dev = 0;
    SDRX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        set_error("sdrx_fft_dev: device index %d", dev);
        return SDRX_EUNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_fft_mu);
    if (!g_fft_attr[dev]) {
        const void *inst[] = {(const void *)fft_kernel<256, true, 0>,   (const void *)fft_kernel<512, true, 0>,
                              (const void *)fft_kernel<1024, true, 0>,  (const void *)fft_kernel<256, false, 0>,
                              (const void *)fft_kernel<512, false, 0>,  (const void *)fft_kernel<1024, false, 0>,
                              (const void *)fft_kernel<256, true, 10>,  (const void *)fft_kernel<256, true, 11>,
                              (const void *)fft_kernel<1024, true, 12>, (const void *)fft_kernel<1024, false, 6>,
                              (const void *)fft_kernel<1024, false, 7>, (const void *)fft_kernel<1024, false, 8>,
                              (const void *)fft_kernel<1024, false, 9>, (const void *)fft_kernel<1024, true, 12, true>};
        for (const void *f : inst) SDRX_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_fft_attr[dev] = true;
    }
    FftTable &t = g_fft_tw[dev][log2n];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (!t.p) {
        if (capturing) {
            (void)hipGetLastError();
            set_error("sdrx_fft_dev: the first call for %u points on a device allocates its twiddle table, which a stream capture "
                      "forbids: call sdrx_init() (or make one uncaptured call) first", 1u << log2n);
            return SDRX_EUNSUPPORTED;
        }
        const unsigned N = 1u << log2n;
        v2d *p = nullptr;
        hipEvent_t ev = nullptr;
        SDRX_HIP(hipMalloc(&p, sizeof(v2d) * (N / 2 ? N / 2 : 1)));
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipFree(p);
            set_error("sdrx_fft_dev: event creation failed");
            return SDRX_EHIP;
        }
        fft_twiddle_kernel<<<(N / 2 + 255) / 256, 256, 0, s>>>(p, N);
        SDRX_HIP(hipGetLastError());
        SDRX_HIP(hipEventRecord(ev, s));
        t.p = p;
        t.filled = ev;
    } else if (!t.done) {
        if (hipEventQuery(t.filled) == hipSuccess) t.done = true;
        else {
            (void)hipGetLastError();
            if (capturing) {
                set_error("sdrx_fft_dev: the twiddle table of %u points is still being filled on another stream", 1u << log2n);
                return SDRX_EUNSUPPORTED;
            }
            SDRX_HIP(hipStreamWaitEvent(s, t.filled, 0));
        }
    }
    *out = t.p;
    return SDRX_OK;
}

// ---- longer transforms (any power of two above SDRX_FFT_MAX): the recursion of FFT.cpp:9-29 is the bit-reversal permutation
// followed by log2 N butterfly stages.  After the permutation the first 12 stages stay inside blocks of 4096 consecutive
// points (each block IS a 4096-point transform without its own permutation: the LDS kernel, NOREV); the remaining stages pair
// points 2^s >= 4096 apart and run from global memory, three at a time on 8 points per thread (every access coalesced: a
// thread's points lie 2^st apart, neighbouring threads' next to them), their twiddles std::polar(1, -2 PI k / len) evaluated
// on the fly with the expression, and therefore the bits, of the table (sdrx_sincos).
__global__ void __launch_bounds__(256) fft_bitrev_kernel(v2d *x, unsigned log2n)
{
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= (1u << log2n)) return;
    const unsigned r = __brev(p) >> (32 - log2n);
    if (p < r) {
        const v2d a = x[p], b = x[r];
        x[p] = b;
        x[r] = a;
    }
}

template <int R> __global__ void __launch_bounds__(256) fft_global_stages_kernel(v2d *x, unsigned st, unsigned log2n)
{
    constexpr unsigned E = 1u << R;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ((1u << log2n) >> R)) return;
    const unsigned half = 1u << st, lo = t & (half - 1);
    const size_t base = ((size_t)(t >> st) << (st + R)) + lo;
    v2d v[E];
#pragma unroll
    for (unsigned q = 0; q < E; q++) v[q] = x[base + ((size_t)q << st)];
#pragma unroll
    for (unsigned r = 0; r < (unsigned)R; r++) { // stage st + r: len = 2^(st + r + 1), k = position inside the half
        const double len = (double)(2ull << (st + r));
#pragma unroll
        for (unsigned ql = 0; ql < (1u << r); ql++) {
            const unsigned k = lo + (ql << st);
            const double theta = -2 * 3.141592653589793238460 * (double)k / len;
            double sn, cs;
            sdrx_sincos(theta, &sn, &cs);
#pragma unroll
            for (unsigned qh = 0; qh < (E >> (r + 1)); qh++) {
                const unsigned q = (qh << (r + 1)) | ql;
                const v2d o = v[q | (1u << r)], e = v[q];
                const double tr = cs * o.x - sn * o.y, ti = cs * o.y + sn * o.x;
                v[q] = v2d{e.x + tr, e.y + ti};
                v[q | (1u << r)] = v2d{e.x - tr, e.y - ti};
            }
        }
    }
#pragma unroll
    for (unsigned q = 0; q < E; q++) x[base + ((size_t)q << st)] = v[q];
}

int check_rows(const void *p, size_t n, size_t nchan, size_t pitch, const char *who)
{
    if (nchan == 0 || n == 0) return 1; // nothing to do
    if (!p) {
        set_error("%s: null buffer", who);
        return SDRX_EINVAL;
    }
    if (nchan > 1 && pitch < n) {
        set_error("%s: pitch %zu < n %zu", who, pitch, n);
        return SDRX_EINVAL;
    }
    if (nchan >= (1ull << 31)) {
        set_error("%s: too many channels", who);
        return SDRX_EUNSUPPORTED;
    }
    return 0;
}

void fold_channels(size_t nchan, unsigned &gy, unsigned &gz)
{
    gy = (unsigned)(nchan < 65535 ? nchan : 65535);
    gz = (unsigned)((nchan + gy - 1) / gy);
}

} // namespace

// sdrx_init(): every table of the current device, on the null stream
int fft_warm_tables()
{
    for (unsigned l = 1; l <= (unsigned)FFT_MAX_LOG2; l++) {
        const v2d *tw = nullptr;
        const int rc = fft_twiddles(l, nullptr, &tw);
        if (rc != SDRX_OK) return rc;
    }
    return SDRX_OK;
}
} // namespace sdrx

using namespace sdrx;

extern "C" int sdrx_iir_dev(int kind, const double *coef, const double *x, double *y, size_t n, size_t nchan, size_t pitch,
                            double *state, void *stream)
{
    int r = check_rows(x, n, nchan, pitch, "sdrx_iir_dev");
    if (r) return r < 0 ? r : SDRX_OK;
    if (!y) {
        set_error("sdrx_iir_dev: null output");
        return SDRX_EINVAL;
    }
    static const int ncoef[4] = {1, 1, 5, 1};
    if (kind < 0 || kind > 3 || !coef) {
        set_error("sdrx_iir_dev: kind %d / coefficients", kind);
        return SDRX_EINVAL;
    }
    IirCoef k = {{0, 0, 0, 0, 0}};
    for (int j = 0; j < ncoef[kind]; j++) k.c[j] = coef[j];
    const bool aliased = kind == K_DC && (const double *)y == x; // aliased call: the reference's own in-place behaviour
    // rows per wave and one- or two-wave form, by bank size (1024 SIMDs; DC blocker / notch in ms over 2^27 samples,
    // profiles/r02_iir_pair_probe.txt): from 2049 rows 16 rows per wave -- 4096 rows 0.92 / 1.30 with 4 rows per wave, 0.69 /
    // 0.88 with 16 and two waves; 8192 rows 0.93 / 1.29 -> 0.45 / 0.51; 12288 rows x 2^13 0.71 / 0.98 -> 0.37 / 0.40 with one
    // wave -- and the two-wave form while its second wave still finds a SIMD of its own (<= 512 groups of rows:
    // 1024 rows 2.60 / 4.60 -> 1.71 / 3.14); rows of at most two chunks never (nothing to take in turn)
    int cw = nchan > 2048 ? 16 : 4;
    if (const int v = tuning().iir_cw; v == 64 || v == 16 || v == 4) cw = v;
    bool pair = cw != 64 && (nchan + cw - 1) / cw <= 512;
    if (n <= 2 * (size_t)IC) pair = false;
    if (const int v = tuning().iir_pair; v == 0 || v == 1) pair = v == 1;
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
    case K_DC:
        if (aliased) launch_iir<K_DCA>(cw, pair, x, y, n, nchan, pitch, k, state, s);
        else launch_iir<K_DC>(cw, pair, x, y, n, nchan, pitch, k, state, s);
        break;
    case K_LP: launch_iir<K_LP>(cw, pair, x, y, n, nchan, pitch, k, state, s); break;
    case K_NOTCH: launch_iir<K_NOTCH>(cw, pair, x, y, n, nchan, pitch, k, state, s); break;
    default: launch_iir<K_INT>(cw, pair, x, y, n, nchan, pitch, k, state, s); break;
    }
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

extern "C" double sdrx_lowpass1_coef(double cutoff_frequency, double delta_time)
{
    return 1.0 - exp(-delta_time * cutoff_frequency); // LowPassFilter.cpp:9
}

extern "C" void sdrx_notch_design(double notch_bw, double center_freq, double *coef5)
{
    // NotchFilter.cpp:7-16 (its own 3.14159)
    double R = 1.0 * notch_bw;
    double K = (1 - 2 * R * cos(2.0 * 3.14159 * center_freq) + R * R) / (2 - 2 * cos(2 * 3.14159 * center_freq));
    coef5[0] = K;
    coef5[1] = -2.0 * K * cos(2.0 * 3.14159 * center_freq);
    coef5[2] = K;
    coef5[3] = 2.0 * R * cos(2.0 * 3.14159 * center_freq);
    coef5[4] = -(R * R);
}

extern "C" int sdrx_normalize_dev(double *x, size_t n, size_t nchan, size_t pitch, double *max_inout, void *stream)
{
    int r = check_rows(x, n, nchan, pitch, "sdrx_normalize_dev");
    if (r < 0) return r;
    if (nchan == 0) return SDRX_OK;
    if (!max_inout) {
        set_error("sdrx_normalize_dev: null max_inout");
        return SDRX_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    norm_init_kernel<<<(unsigned)((nchan + 255) / 256), 256, 0, s>>>(max_inout, nchan);
    if (n) {
        unsigned gy, gz;
        fold_channels(nchan, gy, gz);
        dim3 grid((unsigned)((n + 256 * NORM_PER_THREAD - 1) / (256 * NORM_PER_THREAD)), gy, gz);
        norm_max_kernel<<<grid, 256, 0, s>>>(x, n, nchan, pitch, max_inout);
        norm_scale_kernel<<<grid, 256, 0, s>>>(x, n, nchan, pitch, max_inout);
    }
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

extern "C" int sdrx_fifth_order_decimate_dev(double *data, size_t n, size_t nchan, size_t pitch, size_t decimate_by,
                                             double *state6, void *stream)
{
    int r = check_rows(data, n, nchan, pitch, "sdrx_fifth_order_decimate_dev");
    if (r) return r < 0 ? r : SDRX_OK;
    if (!state6) {
        set_error("sdrx_fifth_order_decimate_dev: null state");
        return SDRX_EINVAL;
    }
    if (decimate_by < 2) return SDRX_OK; // no pass runs (fifth_order_filter.cpp:70)
    fifth_kernel<false><<<(unsigned)nchan, 256, 0, (hipStream_t)stream>>>(data, n, pitch, decimate_by, state6);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

extern "C" int sdrx_fifth_order_decimate_block_dev(double *iq, size_t n_points, size_t nchan, size_t pitch, size_t decimate_by,
                                                   double *state12, size_t *n_points_out, void *stream)
{
    // n_points after the passes (host arithmetic: fifth_order_filter.cpp:31-62)
    size_t np = n_points, most = n_points;
    for (size_t dec = 2; dec <= decimate_by && n_points; dec *= 2) {
        size_t K = (np - 1) / 2;
        np = (K >= 1 ? K : 2) + 1;
        if (np > most) most = np;
    }
    if (n_points_out) *n_points_out = np;
    int r = check_rows(iq, n_points, nchan, pitch, "sdrx_fifth_order_decimate_block_dev");
    if (r) return r < 0 ? r : SDRX_OK;
    if (!state12) {
        set_error("sdrx_fifth_order_decimate_block_dev: null state");
        return SDRX_EINVAL;
    }
    if (pitch < most) { // a pass without an iteration leaves n_points = 3: the next pass reads point 2
        set_error("sdrx_fifth_order_decimate_block_dev: rows must hold %zu points", most);
        return SDRX_EINVAL;
    }
    if (decimate_by < 2) return SDRX_OK;
    fifth_kernel<true><<<(unsigned)nchan, 256, 0, (hipStream_t)stream>>>(iq, n_points, pitch, decimate_by, state12);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

extern "C" int sdrx_fft_dev(double *iq, size_t n, size_t nchan, size_t pitch, void *stream)
{
    int r = check_rows(iq, n, nchan, pitch, "sdrx_fft_dev");
    if (r) return r < 0 ? r : SDRX_OK;
    if (n & (n - 1) || n > ((size_t)1 << SDRX_FFT_MAX_LOG2_ANY)) {
        set_error("sdrx_fft_dev: length %zu is not a power of two <= 2^%d", n, SDRX_FFT_MAX_LOG2_ANY);
        return SDRX_EUNSUPPORTED;
    }
    if (n == 1) return SDRX_OK;
    unsigned log2n = 0;
    while ((1ull << log2n) < n) log2n++;
    if (n > SDRX_FFT_MAX) { // permutation, 4096-point blocks in LDS, the upper stages from global memory (see above)
        hipStream_t s = (hipStream_t)stream;
        const v2d *tw = nullptr;
        int rc = fft_twiddles(FFT_MAX_LOG2, s, &tw);
        if (rc != SDRX_OK) return rc;
        const size_t blocks = n >> FFT_MAX_LOG2;
        if (blocks > 0xffffffffull) return SDRX_EUNSUPPORTED;
        const size_t lds = (SDRX_FFT_MAX / 2 + SDRX_FFT_MAX) * sizeof(v2d);
        const unsigned grid = (unsigned)(blocks < 256 ? blocks : 256); // one workgroup per CU (LDS), each looping over its blocks
        for (size_t c = 0; c < nchan; c++) {
            v2d *row = reinterpret_cast<v2d *>(iq) + c * pitch;
            fft_bitrev_kernel<<<(unsigned)(n / 256), 256, 0, s>>>(row, log2n);
            fft_kernel<1024, true, 12, true><<<grid, 1024, lds, s>>>((double *)row, tw, FFT_MAX_LOG2, 0, SDRX_FFT_MAX, (unsigned)blocks);
            unsigned st = FFT_MAX_LOG2;
            for (; st + 3 <= log2n; st += 3) fft_global_stages_kernel<3><<<(unsigned)((n / 8 + 255) / 256), 256, 0, s>>>(row, st, log2n);
            if (log2n - st == 2) fft_global_stages_kernel<2><<<(unsigned)((n / 4 + 255) / 256), 256, 0, s>>>(row, st, log2n);
            if (log2n - st == 1) fft_global_stages_kernel<1><<<(unsigned)((n / 2 + 255) / 256), 256, 0, s>>>(row, st, log2n);
        }
        SDRX_HIP(hipGetLastError());
        return SDRX_OK;
    }
    const v2d *tw = nullptr;
    int rc = fft_twiddles(log2n, (hipStream_t)stream, &tw);
    if (rc != SDRX_OK) return rc;
    // tile: B = 2^log2b transforms side by side per workgroup, so that short lengths fill the workgroup's threads
    const Tuning &tn = tuning();
    // measured (profiles/probes/fftprobe.py, 2^26 points): lengths >= 1024 run best one transform per workgroup (256 threads; 1024
    // for 4096 points, whose LDS admits one workgroup per CU), shorter ones as 4096-point tiles under 1024 threads
    size_t tile = tn.fft_tile > 0 ? (size_t)tn.fft_tile : (n >= 1024 ? n : 4096);
    if (tile > SDRX_FFT_MAX) tile = SDRX_FFT_MAX;
    unsigned log2b = 0;
    while ((n << (log2b + 1)) <= tile && (1ull << (log2b + 1)) <= nchan) log2b++;
    const size_t total = n << log2b, ntiles = (nchan + (1ull << log2b) - 1) >> log2b;
    int nt = tn.fft_nt == 256 || tn.fft_nt == 512 || tn.fft_nt == 1024 ? tn.fft_nt : (total >= 4096 ? 1024 : 256);
    const size_t lds = (n / 2 + total) * sizeof(v2d);
    // as many workgroups as are resident at once (LDS, or 2048 threads per CU), each looping over its tiles
    size_t per_cu = (160 * 1024) / (lds + 256);
    const size_t by_threads = 2048 / (size_t)nt;
    per_cu = per_cu < 1 ? 1 : (per_cu > by_threads ? by_threads : per_cu);
    const unsigned grid = (unsigned)(ntiles < per_cu * 256 ? ntiles : per_cu * 256);
    hipStream_t s = (hipStream_t)stream;
#define SDRX_FFT_LAUNCH(NT_, ONE_, L2_) \
    fft_kernel<NT_, ONE_, L2_><<<grid, NT_, lds, s>>>(iq, tw, log2n, log2b, pitch, (unsigned)nchan)
    const bool ct = tn.fft_ct != 0; // compile-time lengths: the default shapes of 64 ... 4096 points
    if (log2b == 0) {
        if (ct && nt == 256 && log2n == 10) SDRX_FFT_LAUNCH(256, true, 10);
        else if (ct && nt == 256 && log2n == 11) SDRX_FFT_LAUNCH(256, true, 11);
        else if (ct && nt == 1024 && log2n == 12) SDRX_FFT_LAUNCH(1024, true, 12);
        else if (nt == 256) SDRX_FFT_LAUNCH(256, true, 0);
        else if (nt == 512) SDRX_FFT_LAUNCH(512, true, 0);
        else SDRX_FFT_LAUNCH(1024, true, 0);
    } else {
        if (ct && nt == 1024 && log2n == 6) SDRX_FFT_LAUNCH(1024, false, 6);
        else if (ct && nt == 1024 && log2n == 7) SDRX_FFT_LAUNCH(1024, false, 7);
        else if (ct && nt == 1024 && log2n == 8) SDRX_FFT_LAUNCH(1024, false, 8);
        else if (ct && nt == 1024 && log2n == 9) SDRX_FFT_LAUNCH(1024, false, 9);
        else if (nt == 256) SDRX_FFT_LAUNCH(256, false, 0);
        else if (nt == 512) SDRX_FFT_LAUNCH(512, false, 0);
        else SDRX_FFT_LAUNCH(1024, false, 0);
    }
#undef SDRX_FFT_LAUNCH
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}
