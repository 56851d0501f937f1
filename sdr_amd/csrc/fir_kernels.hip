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
    double thr2;
    unsigned span;           // sample distance from a tile's first to its last staged sample
    // systolic kernel only
    unsigned wave_lds_bytes; // LDS bytes of one wave's slice
    unsigned wgs_per_xcd;    // workgroups each XCD takes (a contiguous run of the stream)
    unsigned long long total_tiles; // tiles * channels
};

template <int RAILS> struct Elem;
template <> struct Elem<1> { typedef double type; };
// a first-class vector (not HIP's double2 struct): copies are plain loads/stores, so arrays of it stay in registers
typedef double v2d __attribute__((ext_vector_type(2)));
template <> struct Elem<2> { typedef v2d type; };

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
            v = ((const v2d *)g.in0)[idx];
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

// One lane's R windows: acc[r] = sum_k h[k] * x[newest_r - k], k ascending (filt.cpp:260-263).
// lbase = the lane's oldest staged sample in LDS; offs[j] = byte offset of the sample j places
// before the window's newest one.
template <int R, int RAILS, bool FMA>
__device__ __forceinline__ void fir_window(const typename Elem<RAILS>::type *lbase, const FirGeom &g,
                                           const double *__restrict__ taps, const int *__restrict__ offs,
                                           double (&acc0)[R], double (&acc1)[R])
{
    typedef typename Elem<RAILS>::type elem_t;
    const int T = g.T;
    const int D = (int)g.D;
    const int Wn = T + (R - 1) * D;
#pragma unroll
    for (int r = 0; r < R; r++) { acc0[r] = 0.0; acc1[r] = 0.0; }

    // taps and offs are the same for every lane: scalar loads, no per-step address arithmetic
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
    // Head ramp, phase p (steps [p*D, (p+1)*D)): accumulators r >= R-1-p are live; tail ramp, phase p
    // (steps [T + p*D, T + (p+1)*D)): r <= R-2-p.  Written out per phase, the live set is a
    // compile-time range: no per-accumulator tests.
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
}

// Epilogue of one lane: outputs m0 .. m0+R-1 of channel `chan`.  pr/pj (discriminator only) = the
// decimated sample before acc[0]; `overlap0` marks acc[0] as the tile's overlap output (computed
// only to serve as that previous sample, never stored).
template <int R, int RAILS, int EPI>
__device__ __forceinline__ void fir_epilogue(const FirGeom &g, size_t chan, long long m0, double (&acc0)[R],
                                             double (&acc1)[R], double pr, double pj, bool overlap0)
{
    const int H = g.T - 1;
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
                    ((v2d *)g.out0)[o] = v2d{acc0[r], acc1[r]};
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
        double *st_out = g.st_out + chan * g.st_pitch;
        double *out = (double *)g.out0 + chan * g.out_pitch;
#pragma unroll
        for (int r = 0; r < R; r++) {
            long long m = m0 + r;
            double ar = acc0[r], aj = acc1[r];
            // multiply(a, conj b): rtl_fm.c:422-426,432
            double br = pr, bj = -pj;
            double cr = ar * br - aj * bj;
            double cj = aj * br + ar * bj;
            if (!(overlap0 && r == 0) && m >= 0 && (size_t)m < g.n_out) {
                out[m] = sdrx_atan2(cj, cr);
                if ((size_t)m == g.n_out - 1) { st_out[2 * H] = ar; st_out[2 * H + 1] = aj; } // rtl_fm.c:543-544
            }
            pr = ar;
            pj = aj;
        }
    }
}

// Carry the shift register: the last T-1 inputs of the call (filt.cpp:256-259), by NT lanes.
template <int LAYOUT, int EPI>
__device__ __forceinline__ void fir_carry_state(const FirGeom &g, size_t chan, unsigned tid, unsigned NT)
{
    constexpr int RAILS = (LAYOUT == SDRX_REAL) ? 1 : 2;
    typedef typename Elem<RAILS>::type elem_t;
    const int H = g.T - 1;
    double *st_out = g.st_out + chan * g.st_pitch;
    const double *st_in = g.st_in + chan * g.st_pitch;
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

// ---- workgroup tiles: 256 (or fewer) lanes share one staged span; for long filters, where the
// T-1 sample halo would dwarf a single wave's span ----
template <int R, int LAYOUT, int EPI, bool FMA>
__global__ void __launch_bounds__(256) fir_tile_kernel(const FirGeom g, const double *__restrict__ taps,
                                                       const int *__restrict__ offs)
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

    // first output of the tile (may be -1 for the discriminator's overlap output)
    const long long m_tile0 = (long long)tile * g.out_per_tile - OV;
    const long long b0 = m_tile0 * D - (T - 1);

    // ---- stage the tile's input span into LDS (coalesced global reads) ----
    // Several loads per lane are issued before the first LDS write so that a CU keeps
    // tens of KB in flight; interior tiles (no history, no end of stream) skip the bounds tests.
    {
        constexpr int UNR = SDRX_STAGE_UNR;
        const long long span_last = b0 + (long long)g.span;
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
                        v[e] = ((const v2d *)g.in0)[idx];
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

    double acc0[R], acc1[R];
    fir_window<R, RAILS, FMA>(lds + (size_t)tid * g.Spad, g, taps, offs, acc0, acc1);

    const long long m0 = m_tile0 + (long long)tid * R;
    double pr = 0.0, pj = 0.0;
    if constexpr (EPI == EPI_DISC) {
        // previous decimated sample for r = 0 comes from the lane below; across waves through LDS
        if (tile == 0 && tid == 0) { // output -1 of the call is the carried pre_r/pre_j (rtl_fm.c:523)
            const double *st_in = g.st_in + chan * g.st_pitch;
            acc0[0] = st_in[2 * (T - 1)];
            acc1[0] = st_in[2 * (T - 1) + 1];
        }
        __syncthreads(); // every lane is done reading the staged tile
        double *xch = (double *)smem;
        const unsigned lane = tid & 63, wave = tid >> 6;
        if (lane == 63) { xch[2 * wave] = acc0[R - 1]; xch[2 * wave + 1] = acc1[R - 1]; }
        __syncthreads();
        pr = __shfl_up(acc0[R - 1], 1);
        pj = __shfl_up(acc1[R - 1], 1);
        if (lane == 0 && wave > 0) { pr = xch[2 * (wave - 1)]; pj = xch[2 * (wave - 1) + 1]; }
    }
    fir_epilogue<R, RAILS, EPI>(g, chan, m0, acc0, acc1, pr, pj, tid == 0);
    if (tile == g.tiles - 1) fir_carry_state<LAYOUT, EPI>(g, chan, tid, NT);
}

// ---- systolic wave tiles, for a decimating filter whose D is one of a few compile-time values ----
// Lane l keeps its OWN D samples x[base + D*l .. base + D*l + D-1] in registers (they pass through
// LDS once, to turn the coalesced load order into per-lane runs).  The accumulator of an output is
// what travels: it takes taps 0..D-1 in the lane that holds the output's newest samples, is shifted
// one lane down (DPP wave_shl:1, no LDS) to meet taps D..2D-1 and that lane's samples, and so on
// for S = ceil(T/D) stages.  Every output still sums k = 0..T-1 in ascending order with one
// rounding per multiply and per add, so results equal the other kernels' bit for bit; the main
// loop has no memory instruction but the scalar tap loads.  Outputs that start in the last S-1
// lanes leave the wave unfinished: tiles overlap by S-1 lane groups (the host picks this kernel
// only while that is a small part of 64).
__device__ __forceinline__ double dpp_from_lane_above(double v) // lane l <- lane l+1; lane 63 <- 0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true); // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_lane_below(double v) // lane l <- lane l-1; lane 0 <- 0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true); // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int DT, int LAYOUT, int EPI, bool FMA>
__global__ void __launch_bounds__(256) fir_systolic_kernel(const FirGeom g, const double *__restrict__ taps,
                                                           const int *__restrict__)
{
    constexpr int RAILS = (LAYOUT == SDRX_REAL) ? 1 : 2;
    constexpr int OV = (EPI == EPI_DISC) ? 1 : 0;
    constexpr unsigned PITCH = (unsigned)DT | 1u; // LDS elements between lane runs: odd -> conflict-free b128/b64 reads
    typedef typename Elem<RAILS>::type elem_t;
    extern __shared__ __align__(16) unsigned char smem[];

    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned waves = blockDim.x >> 6;
    elem_t *lds = (elem_t *)(smem + (size_t)wave * g.wave_lds_bytes);
    const int T = g.T;
    const int S = (T + DT - 1) / DT;            // stages
    const unsigned V = 64u - (unsigned)(S - 1); // lanes that end up with a finished output

    // Workgroups are dealt to the 8 XCDs round-robin; give each XCD one contiguous run of the
    // stream instead, so that the S-1 lane groups two neighbouring tiles share meet in one L2.
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned wg = xcd * g.wgs_per_xcd + slot;
    const unsigned long long w0 = (unsigned long long)wg * waves + wave;
    if (slot >= g.wgs_per_xcd || w0 >= g.total_tiles) return; // whole waves: no barrier below
    const unsigned chan = (unsigned)(w0 / g.tiles), tile = (unsigned)(w0 - (unsigned long long)chan * g.tiles);

    // first sample of the tile: lane 0 ends with output tile*out_per_tile - OV, which started S-1 lanes up
    const long long b0 = ((long long)tile * g.out_per_tile - OV) * DT - (long long)DT * S + 1;
    const bool interior = (b0 >= 0) && ((size_t)(b0 + 64 * DT - 1) < g.n);
    if (interior) {
        const size_t first = (size_t)chan * g.in_pitch + (size_t)b0;
        elem_t regs[DT];
        static_for<0, DT>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            if constexpr (LAYOUT == SDRX_REAL) {
                regs[e] = ((const double *)g.in0 + first)[e * 64 + lane];
            } else if constexpr (LAYOUT == SDRX_IQ_SPLIT) {
                regs[e].x = ((const double *)g.in0 + first)[e * 64 + lane];
                regs[e].y = ((const double *)g.in1 + first)[e * 64 + lane];
            } else {
                regs[e] = ((const v2d *)g.in0 + first)[e * 64 + lane];
            }
        });
        static_for<0, DT>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const unsigned u = (unsigned)e * 64 + lane;
            lds[(u / (unsigned)DT) * PITCH + u % (unsigned)DT] = regs[e];
        });
    } else { // the first / last tile of a stream: history, zeros past the end
        for (unsigned u = lane; u < 64u * DT; u += 64)
            lds[(u / (unsigned)DT) * PITCH + u % (unsigned)DT] = fetch_sample<LAYOUT>(g, chan, b0 + (long long)u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the slice is this wave's alone: no s_barrier
    __builtin_amdgcn_wave_barrier();
    elem_t X[DT]; // this lane's run, oldest first
    static_for<0, DT>([&](auto dc) { X[decltype(dc)::value] = lds[lane * PITCH + decltype(dc)::value]; });

    double a0 = 0.0, a1 = 0.0;
    const double *__restrict__ h = taps;
    for (int s = 0; s < S - 1; s++, h += DT) { // full stages: taps s*DT .. s*DT+DT-1 meet X[DT-1] .. X[0]
        static_for<0, DT>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (RAILS == 1) {
                a0 = mac<FMA>(a0, X[DT - 1 - d], h[d]);
            } else {
                a0 = mac<FMA>(a0, X[DT - 1 - d].x, h[d]);
                a1 = mac<FMA>(a1, X[DT - 1 - d].y, h[d]);
            }
        });
        a0 = dpp_from_lane_above(a0);
        if constexpr (RAILS == 2) a1 = dpp_from_lane_above(a1);
    }
    const int left = T - (S - 1) * DT; // taps of the last stage, 1..DT
    static_for<0, DT>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if (d < left) {
            if constexpr (RAILS == 1) {
                a0 = mac<FMA>(a0, X[DT - 1 - d], h[d]);
            } else {
                a0 = mac<FMA>(a0, X[DT - 1 - d].x, h[d]);
                a1 = mac<FMA>(a1, X[DT - 1 - d].y, h[d]);
            }
        }
    });

    double acc0[1] = {a0}, acc1[1] = {a1};
    const long long m0 = (long long)tile * g.out_per_tile - OV + (long long)lane;
    double pr = 0.0, pj = 0.0;
    if constexpr (EPI == EPI_DISC) {
        if (tile == 0 && lane == 0) { // output -1 of the call is the carried pre_r/pre_j (rtl_fm.c:523)
            const double *st_in = g.st_in + (size_t)chan * g.st_pitch;
            acc0[0] = st_in[2 * (T - 1)];
            acc1[0] = st_in[2 * (T - 1) + 1];
        }
        pr = dpp_from_lane_below(acc0[0]); // lane 0 needs none: its output is the overlap
        pj = dpp_from_lane_below(acc1[0]);
    }
    if (lane < V) fir_epilogue<1, RAILS, EPI>(g, chan, m0, acc0, acc1, pr, pj, lane == 0);
    if (tile == g.tiles - 1) fir_carry_state<LAYOUT, EPI>(g, chan, lane, 64);
}

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
    constexpr int U = 4;            // iterations per trip: their loads are issued ahead of the matrix instructions
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

#ifdef SDRX_EXPERIMENT /* ISA studies: one instantiation, no host side */
template __global__ void fir_systolic_kernel<10, SDRX_IQ_INTERLEAVED, EPI_DISC, false>(const FirGeom, const double *,
                                                                                         const int *);
} // namespace sdrx
#else
// ------------------------------------------------------------------------------------ host side

typedef void (*fir_fn)(const FirGeom, const double *, const int *);

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

// decimation factors the systolic kernel is built for
#define SDRX_SYSTOLIC_DS(X) X(5) X(8) X(10) X(12)

template <int LAYOUT, int EPI, bool FMA> static fir_fn pick_sys_d(unsigned D)
{
    switch (D) {
#define SDRX_CASE(d) \
    case d: return fir_systolic_kernel<d, LAYOUT, EPI, FMA>;
        SDRX_SYSTOLIC_DS(SDRX_CASE)
#undef SDRX_CASE
    default: return nullptr;
    }
}

template <int LAYOUT, bool FMA> static fir_fn pick_sys_epi(int epi, unsigned D)
{
    if constexpr (LAYOUT == SDRX_REAL) {
        return pick_sys_d<LAYOUT, EPI_STORE, FMA>(D);
    } else {
        switch (epi) {
        case EPI_AM: return pick_sys_d<LAYOUT, EPI_AM, FMA>(D);
        case EPI_DISC: return pick_sys_d<LAYOUT, EPI_DISC, FMA>(D);
        default: return pick_sys_d<LAYOUT, EPI_STORE, FMA>(D);
        }
    }
}

static fir_fn pick_systolic(int layout, int epi, bool fma, unsigned D)
{
    if (layout == SDRX_REAL) return fma ? pick_sys_epi<SDRX_REAL, true>(epi, D) : pick_sys_epi<SDRX_REAL, false>(epi, D);
    if (layout == SDRX_IQ_SPLIT)
        return fma ? pick_sys_epi<SDRX_IQ_SPLIT, true>(epi, D) : pick_sys_epi<SDRX_IQ_SPLIT, false>(epi, D);
    return fma ? pick_sys_epi<SDRX_IQ_INTERLEAVED, true>(epi, D) : pick_sys_epi<SDRX_IQ_INTERLEAVED, false>(epi, D);
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

// fast mode 2 (fp64 MFMA): one launch per rail
static int launch_fir_mfma(const sdrx_fir_job *job, size_t n_out)
{
    constexpr int NB = 4;
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
        g.hist_in = job->state_in + (size_t)rail * H;
        g.hist_out = job->state_out + (size_t)rail * H;
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
        hipLaunchKernelGGL(fir_mfma_kernel<NB>, dim3(8u * g.wgs_per_xcd, 1, 1), dim3(256, 1, 1), lds, stream, g, job->taps);
        SDRX_HIP(hipGetLastError());
    }
    // a real-rail call leaves the second history untouched; the discriminator's carried sample always is
    const size_t tail0 = rails == 1 ? (size_t)H : (size_t)2 * H, tail_n = (size_t)2 * H + 2 - tail0;
    SDRX_HIP(hipMemcpy2DAsync(job->state_out + tail0, job->state_pitch * sizeof(double), job->state_in + tail0,
                              job->state_pitch * sizeof(double), tail_n * sizeof(double), job->nchan,
                              hipMemcpyDeviceToDevice, stream));
    return SDRX_OK;
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

    if (job->fast_fma == 2 && D == 1 && epi == EPI_STORE && T >= 32) return launch_fir_mfma(job, n_out);

    const int rails = layout == SDRX_REAL ? 1 : 2;
    const size_t elem = (size_t)rails * sizeof(double);
    const int OV = epi == EPI_DISC ? 1 : 0;
    // register blocking: more outputs per lane when windows overlap heavily (small D)
    const int R_pref = D == 1 ? 8 : (D <= 4 ? 4 : (D <= 8 ? 2 : 1));
    int R = env_int("SDRX_FIR_R", R_pref);
    if (R != 1 && R != 2 && R != 4 && R != 8) R = 1;
    const char *force = getenv("SDRX_FIR_KERNEL"); // "tile" | "systolic": experiments only
    const size_t lds_hard = 160 * 1024;

    FirGeom g;
    memset(&g, 0, sizeof(g));
    g.in0 = job->in0;
    g.in1 = job->in1;
    g.n = job->n;
    g.in_pitch = job->in_pitch;
    g.T = T;
    g.D = D;
    g.st_in = job->state_in;
    g.st_out = job->state_out;
    g.st_pitch = job->state_pitch;
    g.out0 = job->out0;
    g.out1 = job->out1;
    g.out_pitch = job->out_pitch;
    g.n_out = n_out;
    g.thr2 = am_threshold_sq();
    if (job->nchan > 65535) {
        set_error("grid too large: nchan=%zu", job->nchan);
        return SDRX_EUNSUPPORTED;
    }

    // 1. systolic wave tiles: D is one of the built-in factors and few lanes are lost to the S-1 stage shifts
    if (!force || strcmp(force, "systolic") == 0) {
        const int S = (T + (int)D - 1) / (int)D;
        const int lost = S - 1 + OV;
        fir_fn fn = pick_systolic(layout, (int)epi, job->fast_fma != 0, D);
        if (fn && lost <= 12) {
            unsigned wpw = (unsigned)env_int("SDRX_FIR_WAVES", 2); // waves per workgroup (2: 14 waves fit a CU's LDS, measured best)
            if (wpw != 1 && wpw != 2 && wpw != 4) wpw = 4;
            g.out_per_tile = 64u - (unsigned)lost;
            const size_t tiles = (n_out + g.out_per_tile - 1) / g.out_per_tile;
            g.tiles = (unsigned)tiles;
            g.total_tiles = (unsigned long long)tiles * job->nchan;
            g.wave_lds_bytes = (unsigned)((64 * (D | 1u) * elem + 15) & ~(size_t)15);
            const unsigned long long wgs = (g.total_tiles + wpw - 1) / wpw;
            g.wgs_per_xcd = (unsigned)((wgs + 7) / 8);
            if (tiles <= 0x7fffffffULL && wgs <= 0x0fffffffULL) {
                hipLaunchKernelGGL(fn, dim3(8u * g.wgs_per_xcd, 1, 1), dim3(64 * wpw, 1, 1),
                                   (size_t)wpw * g.wave_lds_bytes, stream, g, job->taps, (const int *)nullptr);
                SDRX_HIP(hipGetLastError());
                return SDRX_OK;
            }
        }
    }

    // 2. workgroup tiles
    unsigned NT = (unsigned)env_int("SDRX_FIR_NT", 256);
    if (NT != 64 && NT != 128 && NT != 256) NT = 256;
    const size_t lds_soft = 64 * 1024;
    size_t lds_bytes = 0;
    for (;;) {
        const unsigned Wn = (unsigned)T + (unsigned)(R - 1) * D;
        g.S = (unsigned)R * D;
        g.Sp = g.S < Wn ? g.S : Wn;
        g.Spad = g.Sp | 1u;
        g.n_staged = (NT - 1) * g.Sp + Wn;
        const unsigned last = g.n_staged - 1;
        g.span = (last / g.Sp) * g.S + last % g.Sp;
        lds_bytes = ((size_t)(last / g.Sp) * g.Spad + last % g.Sp + 1) * elem;
        if (lds_bytes < 64) lds_bytes = 64;
        if (lds_bytes <= lds_soft) break;
        if (R > 1) { R >>= 1; continue; }
        if (NT > 64) { NT >>= 1; continue; }
        if (lds_bytes <= lds_hard) break;
        set_error("FIR tile does not fit LDS: taps=%d decim=%u needs %zu bytes", T, D, lds_bytes);
        return SDRX_EUNSUPPORTED;
    }
    g.magicSp = (g.Sp < 2048 && g.n_staged < (1u << 20)) ? magic_u32(g.Sp) : 0;
    g.out_per_tile = NT * (unsigned)R - (unsigned)OV;
    const size_t tiles = (n_out + g.out_per_tile - 1) / g.out_per_tile;
    if (tiles > 0x7fffffffULL) {
        set_error("grid too large: tiles=%zu", tiles);
        return SDRX_EUNSUPPORTED;
    }
    g.tiles = (unsigned)tiles;
    const int *offs = window_offsets_dev((unsigned)T + (unsigned)(R - 1) * D, g.Sp, g.Spad, (unsigned)elem);
    if (!offs) {
        set_error("window offset table upload failed");
        return SDRX_EHIP;
    }
    fir_fn fn = pick_kernel(layout, (int)epi, job->fast_fma != 0, R);
    if (lds_bytes > 48 * 1024)
        SDRX_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(fn, dim3(g.tiles, (unsigned)job->nchan, 1), dim3(NT, 1, 1), lds_bytes, stream, g, job->taps, offs);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
#endif /* SDRX_EXPERIMENT */
