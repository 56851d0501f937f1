// fir_kernels.hip -- time-tiled, register-blocked FIR for gfx950 with fused demod epilogues.
//
// Replaces Filter::do_sample / Filter::execute (RadioNode/third-party/FIR-filter-class/filt.cpp:248-273)
// and, fused behind it, decimate<T> (Resample.h:33-52), AM<T>::demodulate (AM.h:63-77) and the
// phase-difference discriminator (rtl_fm.c:422-435,518-546).  The same kernel is the PSK matched
// filter (BPSK.cpp:279-290: T = 129, D = 4, zero history).
//
// Parallelisation: the reference is one serial loop per stream; here every OUTPUT keeps the
// reference's summation (acc = 0; acc += x[i-k]*h[k], k ascending, one rounding per mul and per
// add) and outputs are spread over lanes, tiles and channels.  A 256-thread workgroup stages the
// tile's input span (its T-1 sample overlap-save halo comes from the previous tile or from the
// carried history) into LDS with one coalesced pass, then each lane walks its window newest-first,
// feeding R register accumulators per rail from each LDS read; the tap for a step is the same for
// every lane, so taps travel through scalar loads, not LDS.  LDS layout: lane blocks of Sp samples
// at an odd pitch Spad, which makes the per-step ds_read (b128 for I/Q, b64 for one rail)
// bank-conflict free for any decimation factor.
#include "sdrx_internal.h"

#include "../../include/sdrx_math.h"

#include <array>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#ifndef SDRX_STAGE_UNR
#define SDRX_STAGE_UNR 12
#endif
#ifndef SDRX_STEADY_UB
#define SDRX_STEADY_UB 4
#endif

namespace sdrx {

struct FirGeom {
    const void *in0, *in1;
    size_t n, in_pitch;
    const double *taps;
    int T;
    unsigned D;
    const double *st_in;
    double *st_out;
    size_t st_pitch;
    void *out0, *out1;
    size_t out_pitch, n_out;
    unsigned tiles;        // per channel
    unsigned out_per_tile; // NT*R - OV
    unsigned S;            // R*D: sample distance between lanes
    unsigned Sp;           // staged samples per lane block = min(S, Wn)
    unsigned Spad;         // Sp | 1: LDS elements between lane blocks
    unsigned magicSp;      // 0 => plain division
    unsigned n_staged;
    const int *offs;       // Wn byte offsets into a lane's LDS window, newest sample first
    double thr2;
};

template <int RAILS> struct Elem;
template <> struct Elem<1> { typedef double type; };
template <> struct Elem<2> { typedef double2 type; };

// x[c][i] for any i: history for i < 0, zero past the end.
template <int LAYOUT>
__device__ __forceinline__ typename Elem<(LAYOUT == SDRX_REAL ? 1 : 2)>::type
fetch_sample(const FirGeom &g, size_t chan, long long i)
{
    typename Elem<(LAYOUT == SDRX_REAL ? 1 : 2)>::type v;
    const int H = g.T - 1;
    if (i >= 0 && (size_t)i < g.n) {
        size_t idx = chan * g.in_pitch + (size_t)i;
        if constexpr (LAYOUT == SDRX_REAL) {
            v = ((const double *)g.in0)[idx];
        } else if constexpr (LAYOUT == SDRX_IQ_SPLIT) {
            v.x = ((const double *)g.in0)[idx];
            v.y = ((const double *)g.in1)[idx];
        } else {
            v = ((const double2 *)g.in0)[idx];
        }
    } else if (i < 0 && -i - 1 < (long long)H) {
        const double *st = g.st_in + chan * g.st_pitch;
        if constexpr (LAYOUT == SDRX_REAL) {
            v = st[-i - 1];
        } else {
            v.x = st[-i - 1];
            v.y = st[H + (-i - 1)];
        }
    } else {
        if constexpr (LAYOUT == SDRX_REAL) v = 0.0;
        else { v.x = 0.0; v.y = 0.0; }
    }
    return v;
}

// f(integral_constant<int, I>) for I in [LO, HI): loop bounds that depend on I stay compile-time
template <int LO, int HI, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        static_for<LO + 1, HI>(f);
    }
}

template <bool FMA> __device__ __forceinline__ double mac(double acc, double x, double h)
{
    if constexpr (FMA) return __builtin_fma(x, h, acc);
    else return acc + x * h; // compiled with -ffp-contract=off: two roundings, like filt.cpp:262
}

template <int R, int LAYOUT, int EPI, bool FMA>
__global__ void __launch_bounds__(256) fir_tile_kernel(const FirGeom g)
{
    constexpr int RAILS = (LAYOUT == SDRX_REAL) ? 1 : 2;
    constexpr int OV = (EPI == EPI_DISC) ? 1 : 0;
    typedef typename Elem<RAILS>::type elem_t;
    extern __shared__ __align__(16) unsigned char smem[];
    elem_t *lds = (elem_t *)smem;

    const unsigned tid = threadIdx.x;
    const unsigned NT = blockDim.x;
    const unsigned tile = blockIdx.x;
    const size_t chan = blockIdx.y;
    const int T = g.T;
    const int D = (int)g.D;
    const int Wn = T + (R - 1) * D;

    // first output of the tile (may be -1 for the discriminator's overlap output)
    const long long m_tile0 = (long long)tile * g.out_per_tile - OV;
    const long long b0 = m_tile0 * D - (T - 1);

    // ---- stage the tile's input span into LDS (coalesced global reads) ----
    // Several loads per lane are issued before the first LDS write so that a CU keeps
    // tens of KB in flight; interior tiles (no history, no end of stream) skip the bounds tests.
    {
        constexpr int UNR = SDRX_STAGE_UNR; // all of a tile's loads are in flight before the first LDS write
        const long long span_last = b0 + (long long)((g.n_staged - 1) / g.Sp) * g.S + (g.n_staged - 1) % g.Sp;
        const bool interior = (b0 >= 0) && ((size_t)span_last < g.n);
        for (unsigned u0 = tid; u0 < g.n_staged; u0 += NT * UNR) {
            elem_t v[UNR];
            unsigned pos[UNR];
#pragma unroll
            for (int e = 0; e < UNR; e++) {
                unsigned u = u0 + (unsigned)e * NT;
                unsigned uc = u < g.n_staged ? u : g.n_staged - 1;
                unsigned blk = g.magicSp ? __umulhi(uc, g.magicSp) : (uc / g.Sp);
                unsigned k = uc - blk * g.Sp;
                long long i = b0 + (long long)blk * g.S + k;
                pos[e] = blk * g.Spad + k;
                if (interior) {
                    size_t idx = chan * g.in_pitch + (size_t)i;
                    if constexpr (LAYOUT == SDRX_REAL) {
                        v[e] = ((const double *)g.in0)[idx];
                    } else if constexpr (LAYOUT == SDRX_IQ_SPLIT) {
                        v[e].x = ((const double *)g.in0)[idx];
                        v[e].y = ((const double *)g.in1)[idx];
                    } else {
                        v[e] = ((const double2 *)g.in0)[idx];
                    }
                } else {
                    v[e] = fetch_sample<LAYOUT>(g, chan, i);
                }
            }
#pragma unroll
            for (int e = 0; e < UNR; e++)
                if (u0 + (unsigned)e * NT < g.n_staged) lds[pos[e]] = v[e];
        }
    }
    __syncthreads();

    // ---- per-lane windows: R outputs per rail, newest sample first ----
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; r++) { acc0[r] = 0.0; acc1[r] = 0.0; }

    const elem_t *lbase = lds + (size_t)tid * g.Spad;
    const double *__restrict__ taps = g.taps;
    const int *__restrict__ offs = g.offs; // same for every lane: scalar loads, no per-step address arithmetic

    // accumulator r meets tap j - (R-1-r)*D at window step j: one base pointer per accumulator,
    // so that a block of steps reads a contiguous run of taps per accumulator (one scalar load)
    const double *tp[R];
#pragma unroll
    for (int r = 0; r < R; r++) tp[r] = (const double *)((const char *)taps - (long long)(R - 1 - r) * D * 8);

    auto step = [&](int j, bool guarded) {
        elem_t x = *(const elem_t *)((const char *)lbase + offs[j]);
#pragma unroll
        for (int r = 0; r < R; r++) {
            int k = j - (R - 1 - r) * D;
            if (!guarded || (k >= 0 && k < T)) {
                double h = taps[k];
                if constexpr (RAILS == 1) {
                    acc0[r] = mac<FMA>(acc0[r], x, h);
                } else {
                    acc0[r] = mac<FMA>(acc0[r], x.x, h);
                    acc1[r] = mac<FMA>(acc1[r], x.y, h);
                }
            }
        }
    };

    // Window step j multiplies sample (newest - j) into accumulator r with tap k = j - (R-1-r)*D.
    // Head ramp, phase p (steps [p*D, (p+1)*D)): accumulators r >= R-1-p are live; tail ramp, phase p
    // (steps [T + p*D, T + (p+1)*D)): r <= R-2-p.  Written out per phase, the live set is a
    // compile-time range: no per-accumulator tests, and a phase's taps are D apart in one run.
    auto ramp = [&](int j, auto lo, auto hi) {
        elem_t x = *(const elem_t *)((const char *)lbase + offs[j]);
#pragma unroll
        for (int r = decltype(lo)::value; r <= decltype(hi)::value; r++) {
            const double h = tp[r][j];
            if constexpr (RAILS == 1) {
                acc0[r] = mac<FMA>(acc0[r], x, h);
            } else {
                acc0[r] = mac<FMA>(acc0[r], x.x, h);
                acc1[r] = mac<FMA>(acc1[r], x.y, h);
            }
        }
    };
    int j = 0;
    const bool structured = (R - 1) * D <= T; // otherwise (a filter shorter than the lane's spread): tested steps
    if (structured) {
        static_for<0, R - 1>([&](auto p) {
            for (int d = 0; d < D; d++, j++) ramp(j, std::integral_constant<int, R - 1 - decltype(p)::value>{},
                                                  std::integral_constant<int, R - 1>{});
        });
    } else {
        const int jA = (R - 1) * D < T ? (R - 1) * D : T;
        for (; j < jA; j++) step(j, true);
    }
    // steady state (all accumulators live) in blocks of UB steps: UB LDS reads back to back, then
    // UB*R*RAILS multiply-adds
    constexpr int UB = SDRX_STEADY_UB;
    for (; j + UB <= T; j += UB) {
        elem_t xs[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) xs[u] = *(const elem_t *)((const char *)lbase + offs[j + u]);
#pragma unroll
        for (int u = 0; u < UB; u++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double h = tp[r][j + u];
                if constexpr (RAILS == 1) {
                    acc0[r] = mac<FMA>(acc0[r], xs[u], h);
                } else {
                    acc0[r] = mac<FMA>(acc0[r], xs[u].x, h);
                    acc1[r] = mac<FMA>(acc1[r], xs[u].y, h);
                }
            }
        }
    }
    for (; j < T; j++) step(j, false);
    if (structured) {
        static_for<0, R - 1>([&](auto p) {
            for (int d = 0; d < D; d++, j++) ramp(j, std::integral_constant<int, 0>{},
                                                  std::integral_constant<int, R - 2 - decltype(p)::value>{});
        });
    } else {
        for (; j < Wn; j++) step(j, true);
    }

    const long long m0 = m_tile0 + (long long)tid * R;
    const bool last_tile = (tile == g.tiles - 1);
    double *st_out = g.st_out + chan * g.st_pitch;
    const double *st_in = g.st_in + chan * g.st_pitch;
    const int H = T - 1;

    if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            long long m = m0 + r;
            if (m >= 0 && (size_t)m < g.n_out) {
                size_t o = chan * g.out_pitch + (size_t)m;
                if constexpr (RAILS == 1) {
                    ((double *)g.out0)[o] = acc0[r];
                } else if (g.out1) {
                    ((double *)g.out0)[o] = acc0[r];
                    ((double *)g.out1)[o] = acc1[r];
                } else {
                    ((double2 *)g.out0)[o] = make_double2(acc0[r], acc1[r]);
                }
            }
        }
    } else if constexpr (EPI == EPI_AM) {
        uint8_t *out = (uint8_t *)g.out0 + chan * g.out_pitch;
        unsigned long long packed = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            double e = acc0[r] * acc0[r] + acc1[r] * acc1[r]; // AM.h:73 (the sqrt is folded into thr2)
            packed |= (unsigned long long)(e >= g.thr2 ? 1u : 0u) << (8 * r);
        }
        if (R == 8 && m0 >= 0 && (size_t)(m0 + 7) < g.n_out && (((size_t)(out + m0)) & 7) == 0) {
            *(unsigned long long *)(out + m0) = packed;
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                long long m = m0 + r;
                if (m >= 0 && (size_t)m < g.n_out) out[m] = (uint8_t)((packed >> (8 * r)) & 0xff);
            }
        }
    } else { // EPI_DISC
        // previous decimated sample for r = 0 comes from the lane below; across waves through LDS
        if (tile == 0 && tid == 0) { // output -1 of the call is the carried pre_r/pre_j (rtl_fm.c:523)
            acc0[0] = st_in[2 * H];
            acc1[0] = st_in[2 * H + 1];
        }
        __syncthreads(); // every lane is done reading the staged tile
        double *xch = (double *)smem;
        const unsigned lane = tid & 63, wave = tid >> 6;
        if (lane == 63) { xch[2 * wave] = acc0[R - 1]; xch[2 * wave + 1] = acc1[R - 1]; }
        __syncthreads();
        double pr = __shfl_up(acc0[R - 1], 1);
        double pj = __shfl_up(acc1[R - 1], 1);
        if (lane == 0 && wave > 0) { pr = xch[2 * (wave - 1)]; pj = xch[2 * (wave - 1) + 1]; }
        double *out = (double *)g.out0 + chan * g.out_pitch;
#pragma unroll
        for (int r = 0; r < R; r++) {
            long long m = m0 + r;
            double ar = acc0[r], aj = acc1[r];
            // multiply(a, conj b): rtl_fm.c:422-426,432
            double br = pr, bj = -pj;
            double cr = ar * br - aj * bj;
            double cj = aj * br + ar * bj;
            bool is_overlap = (tid == 0 && r == 0);
            if (!is_overlap && m >= 0 && (size_t)m < g.n_out) {
                out[m] = sdrx_atan2(cj, cr);
                if ((size_t)m == g.n_out - 1) { st_out[2 * H] = ar; st_out[2 * H + 1] = aj; } // rtl_fm.c:543-544
            }
            pr = ar;
            pj = aj;
        }
    }

    // ---- carry the shift register: the last T-1 inputs of the call (filt.cpp:256-259) ----
    if (last_tile) {
        for (int k = (int)tid; k < H; k += (int)NT) {
            long long i = (long long)g.n - 1 - k;
            elem_t v = fetch_sample<LAYOUT>(g, chan, i);
            if constexpr (RAILS == 1) {
                st_out[k] = v;
                st_out[H + k] = st_in[H + k];
            } else {
                st_out[k] = v.x;
                st_out[H + k] = v.y;
            }
        }
        if (EPI != EPI_DISC && tid == 0) { st_out[2 * H] = st_in[2 * H]; st_out[2 * H + 1] = st_in[2 * H + 1]; }
    }
}

// ------------------------------------------------------------------------------------ host side

typedef void (*fir_fn)(const FirGeom);

template <int LAYOUT, int EPI, bool FMA> static fir_fn pick_r(int R)
{
    switch (R) {
    case 1: return fir_tile_kernel<1, LAYOUT, EPI, FMA>;
    case 2: return fir_tile_kernel<2, LAYOUT, EPI, FMA>;
    case 4: return fir_tile_kernel<4, LAYOUT, EPI, FMA>;
    default: return fir_tile_kernel<8, LAYOUT, EPI, FMA>;
    }
}

template <int LAYOUT, bool FMA> static fir_fn pick_epi(int epi, int R)
{
    if constexpr (LAYOUT == SDRX_REAL) {
        return pick_r<LAYOUT, EPI_STORE, FMA>(R);
    } else {
        switch (epi) {
        case EPI_AM: return pick_r<LAYOUT, EPI_AM, FMA>(R);
        case EPI_DISC: return pick_r<LAYOUT, EPI_DISC, FMA>(R);
        default: return pick_r<LAYOUT, EPI_STORE, FMA>(R);
        }
    }
}

static fir_fn pick_kernel(int layout, int epi, bool fma, int R)
{
    if (layout == SDRX_REAL) return fma ? pick_epi<SDRX_REAL, true>(epi, R) : pick_epi<SDRX_REAL, false>(epi, R);
    if (layout == SDRX_IQ_SPLIT)
        return fma ? pick_epi<SDRX_IQ_SPLIT, true>(epi, R) : pick_epi<SDRX_IQ_SPLIT, false>(epi, R);
    return fma ? pick_epi<SDRX_IQ_INTERLEAVED, true>(epi, R) : pick_epi<SDRX_IQ_INTERLEAVED, false>(epi, R);
}

// Per-lane LDS window offsets for a tile geometry.  Tiny (Wn ints), created the first time a
// geometry is used on a device and kept until the process ends; later calls allocate nothing.
static const int *window_offsets_dev(unsigned Wn, unsigned Sp, unsigned Spad, unsigned elem)
{
    static std::mutex mu;
    static std::map<std::array<unsigned, 5>, int *> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    const std::array<unsigned, 5> key = {(unsigned)dev, Wn, Sp, Spad, elem};
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<int> h(Wn);
    for (unsigned j = 0; j < Wn; j++) {
        const unsigned c = Wn - 1 - j; // step j reads the sample c places after the window's oldest
        h[j] = (int)(((c / Sp) * Spad + c % Sp) * elem);
    }
    int *d = nullptr;
    if (hipMalloc((void **)&d, Wn * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), Wn * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    cache[key] = d;
    return d;
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

int launch_fir(const sdrx_fir_job *job, FirEpilogue epi)
{
    if (!job || !job->in0 || !job->taps || !job->out0 || !job->state_in || !job->state_out) {
        set_error("sdrx_fir*_dev: null pointer in job");
        return SDRX_EINVAL;
    }
    if (job->num_taps <= 0 || job->decim == 0 || job->state_in == job->state_out) {
        set_error("sdrx_fir*_dev: num_taps=%d decim=%u, state_in %s state_out", job->num_taps, job->decim,
                  job->state_in == job->state_out ? "==" : "!=");
        return SDRX_EINVAL;
    }
    const int layout = job->layout;
    if (layout != SDRX_REAL && layout != SDRX_IQ_SPLIT && layout != SDRX_IQ_INTERLEAVED) return SDRX_EINVAL;
    if (layout == SDRX_IQ_SPLIT && !job->in1) return SDRX_EINVAL;
    if (layout == SDRX_REAL && epi != EPI_STORE) {
        set_error("AM / discriminator epilogues need I/Q input");
        return SDRX_EINVAL;
    }
    const int T = job->num_taps;
    const unsigned D = job->decim;
    if (job->state_pitch < (size_t)(2 * (T - 1) + 2)) {
        set_error("state_pitch %zu < %d", job->state_pitch, 2 * (T - 1) + 2);
        return SDRX_EINVAL;
    }
    hipStream_t stream = (hipStream_t)job->stream;
    const size_t n_out = job->n_out ? job->n_out : (job->n + D - 1) / D;
    const size_t state_bytes = (size_t)(2 * (T - 1) + 2) * sizeof(double);
    if (job->nchan == 0) return SDRX_OK;
    if (n_out == 0 || job->n == 0) { // nothing to filter: the state is carried unchanged
        SDRX_HIP(hipMemcpy2DAsync(job->state_out, job->state_pitch * sizeof(double), job->state_in,
                                  job->state_pitch * sizeof(double), state_bytes, job->nchan,
                                  hipMemcpyDeviceToDevice, stream));
        return SDRX_OK;
    }

    const int rails = layout == SDRX_REAL ? 1 : 2;
    const size_t elem = (size_t)rails * sizeof(double);
    const int OV = epi == EPI_DISC ? 1 : 0;
    // register blocking: more outputs per lane when windows overlap heavily (small D)
    int R = D == 1 ? 8 : (D <= 4 ? 4 : (D <= 8 ? 2 : 1));
    R = env_int("SDRX_FIR_R", R);
    if (R != 1 && R != 2 && R != 4 && R != 8) R = 1;
    unsigned NT = (unsigned)env_int("SDRX_FIR_NT", 256);
    if (NT != 64 && NT != 128 && NT != 256) NT = 256;
    const size_t lds_soft = 64 * 1024, lds_hard = 160 * 1024;
    FirGeom g;
    size_t lds_bytes = 0;
    for (;;) {
        const unsigned Wn = (unsigned)T + (unsigned)(R - 1) * D;
        g.S = (unsigned)R * D;
        g.Sp = g.S < Wn ? g.S : Wn;
        g.Spad = g.Sp | 1u;
        g.n_staged = (NT - 1) * g.Sp + Wn;
        const unsigned last = g.n_staged - 1;
        lds_bytes = ((size_t)(last / g.Sp) * g.Spad + last % g.Sp + 1) * elem;
        if (lds_bytes < 64) lds_bytes = 64;
        if (lds_bytes <= lds_soft) break;
        if (R > 1) { R >>= 1; continue; }
        if (NT > 64) { NT >>= 1; continue; }
        if (lds_bytes <= lds_hard) break;
        set_error("FIR tile does not fit LDS: taps=%d decim=%u needs %zu bytes", T, D, lds_bytes);
        return SDRX_EUNSUPPORTED;
    }
    g.offs = window_offsets_dev((unsigned)T + (unsigned)(R - 1) * D, g.Sp, g.Spad, (unsigned)elem);
    if (!g.offs) {
        set_error("window offset table upload failed");
        return SDRX_EHIP;
    }
    g.magicSp = (g.Sp < 2048 && g.n_staged < (1u << 20)) ? magic_u32(g.Sp) : 0;
    g.in0 = job->in0;
    g.in1 = job->in1;
    g.n = job->n;
    g.in_pitch = job->in_pitch;
    g.taps = job->taps;
    g.T = T;
    g.D = D;
    g.st_in = job->state_in;
    g.st_out = job->state_out;
    g.st_pitch = job->state_pitch;
    g.out0 = job->out0;
    g.out1 = job->out1;
    g.out_pitch = job->out_pitch;
    g.n_out = n_out;
    g.out_per_tile = NT * (unsigned)R - (unsigned)OV;
    const size_t tiles = (n_out + g.out_per_tile - 1) / g.out_per_tile;
    if (tiles > 0x7fffffffULL || job->nchan > 65535) {
        set_error("grid too large: tiles=%zu nchan=%zu", tiles, job->nchan);
        return SDRX_EUNSUPPORTED;
    }
    g.tiles = (unsigned)tiles;
    g.thr2 = am_threshold_sq();

    fir_fn fn = pick_kernel(layout, (int)epi, job->fast_fma != 0, R);
    if (lds_bytes > 48 * 1024)
        SDRX_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid((unsigned)tiles, (unsigned)job->nchan, 1), block(NT, 1, 1);
    hipLaunchKernelGGL(fn, grid, block, lds_bytes, stream, g);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
