// psk_kernels.hip -- BPSK / QPSK loops, unique-word ambiguity pass and pulse shaper for gfx950.
//
//   psk_loops_kernel     : symbol_timing_synchronization (BPSK.cpp:302-389 / QPSK.cpp:264-351) and
//                          phase_shift_correction + decimate-by-2 (BPSK.cpp:393-444 / QPSK.cpp:355-414)
//   psk_ambiguity_kernel : phase_ambiguity_correction (BPSK.cpp:446-487 / QPSK.cpp:416-509)
//   psk_modulate_kernel  : encode_LUT + upsample_fill_w_zeros + encode_pulseShape
//                          (BPSK.cpp:112-146 / QPSK.cpp:144-188, Resample.cpp:24-36)
// The matched filter (decode_pulseShape) is the generic FIR tile kernel of fir_kernels.hip.
//
// The two loops are sample-recursive with sign / floor / compare nonlinearities, so they cannot be
// tiled in time without changing results.  They run one CHANNEL per lane with the whole loop state
// in registers and exactly the reference's operation order (no FMA contraction); throughput comes
// from the number of independent channels.  cos/sin are include/sdrx_math.h on both sides.
#include "sdrx_internal.h"
#include <type_traits>

#include <cstdlib>
#include <mutex>

#include "../../include/sdrx_math.h"

namespace sdrx {

// MATLAB rcosine(1/8,1,'sqrt',0.5,8): values of BPSK.cpp:18-50 / QPSK.cpp:32-64.  ONE literal list: the host table
// (BPSK::RRC_COEFF through sdrx_rrc_coeff()) and the device table below are both made from it.
#define SDRX_RRC_VALUES \
    -0.000882663055055987, -0.000746201590523265, -0.000336789968798916, 0.000225038893553655, \
    0.000757880681389978,  0.001077943504359050,  0.001062303725777150,  0.000695521249813576, \
    8.24465490986367e-05,  -0.000579554041955506, -0.001062303725777150, -0.001183702054816000, \
    -0.000874477709296126, -0.000211083163638599, 0.000599193283619645,  0.001279536876191750, \
    0.001573979573701230,  0.001338143762729140,  0.000599193283619642,  -0.000438236415412023, \
    -0.001446863119017230, -0.002078893324143190, -0.002084671221261930, -0.001408617284422990, \
    -0.000227352605090172, 0.001088699270085450,  0.002084671221261940,  0.002371289907799650, \
    0.001768388256576610,  0.000391988582678128,  -0.001351122110122730, -0.002869993360503870, \
    -0.003572683794274230, -0.003072685906611650, -0.001351122110122720, 0.001186107048403510, \
    0.003789403406949890,  0.005575980755162950,  0.005816670050933640,  0.004213278581633850, \
    0.001071805138282270,  -0.002702696087965530, -0.005816670050933650, -0.006968745846090550, \
    -0.005305164769729840, -0.000815536975401244, 0.005468827588591990,  0.011573823048484000, \
    0.015005271935951800,  0.013430574641186900,  0.005468827588591980,  -0.008615615450369140, \
    -0.026525823848649200, -0.044008760276746200, -0.055452254485567400, -0.054931750238658300, \
    -0.037513179839879400, -0.000532386899804849, 0.055452254485567400,  0.126288206185736000, \
    0.204577471545948000,  0.280728910881642000,  0.344536138081295000,  0.386975375596443000, \
    0.401855774335913000,  0.386975375596443000,  0.344536138081295000,  0.280728910881642000, \
    0.204577471545948000,  0.126288206185736000,  0.055452254485567400,  -0.000532386899804849, \
    -0.037513179839879400, -0.054931750238658300, -0.055452254485567400, -0.044008760276746200, \
    -0.026525823848649200, -0.008615615450369140, 0.005468827588591980,  0.013430574641186900, \
    0.015005271935951800,  0.011573823048484000,  0.005468827588591990,  -0.000815536975401244, \
    -0.005305164769729840, -0.006968745846090550, -0.005816670050933650, -0.002702696087965530, \
    0.001071805138282270,  0.004213278581633850,  0.005816670050933640,  0.005575980755162950, \
    0.003789403406949890,  0.001186107048403510,  -0.001351122110122720, -0.003072685906611650, \
    -0.003572683794274230, -0.002869993360503870, -0.001351122110122730, 0.000391988582678128, \
    0.001768388256576610,  0.002371289907799650,  0.002084671221261940,  0.001088699270085450, \
    -0.000227352605090172, -0.001408617284422990, -0.002084671221261930, -0.002078893324143190, \
    -0.001446863119017230, -0.000438236415412023, 0.000599193283619642,  0.001338143762729140, \
    0.001573979573701230,  0.001279536876191750,  0.000599193283619645,  -0.000211083163638599, \
    -0.000874477709296126, -0.001183702054816000, -0.001062303725777150, -0.000579554041955506, \
    8.24465490986367e-05,  0.000695521249813576,  0.001062303725777150,  0.001077943504359050, \
    0.000757880681389978,  0.000225038893553655,  -0.000336789968798916, -0.000746201590523265, \
    -0.000882663055055987

const double RRC_COEFF[129] = {SDRX_RRC_VALUES};


#define SDRX_SGN(v) (((v) == 0.0) ? 0.0 : (((v) < 0.0) ? -1.0 : 1.0)) /* BPSK.h:16 */

// What a speculative block of the loops watches instead of branching: a block takes the straight-line common case of
// SIGN_FUNC and of sincos, and is repeated with the general forms if anything it met was outside that case (silence, a
// stream's zero padding, a loop that has run away): psk_loops_split_kernel, psk_loops_tri_kernel.
struct SpecWatch {
    double zmin = 1.0;  // smallest magnitude a sign was taken of in the carrier loop
    double amax = 0.0;  // largest magnitude handed to sincos
    bool odd = false;   // the timing loop took the sign of a zero or a NaN
    __device__ __forceinline__ bool raised() const { return odd || !(zmin > 0.0) || !(amax <= 1.5e6); }
};

// SIGN_FUNC (BPSK.h:16).  For every v that is neither a zero nor a NaN it is 1.0 with v's sign: one bit-field insert instead
// of two compares and two selects on the loops' serial chain.
//  * carrier loop (NAN_MATTERS = false): 8 (sgn(ti) tr - sgn(tr) ti) is a NaN whenever tr or ti is, whatever the signs are
//    taken to be, so only zeros have to be caught: the smallest |v| of the block (one v_min_f64 per sign, no scalar work).
//  * timing loop (NAN_MATTERS = true): the detector multiplies OLDER samples by the new sample's sign, so a NaN's sign
//    must be +1 as SIGN_FUNC has it: the caller tests the product of the two rails' new samples for zero / NaN once per step.
template <bool SPECULATE, bool NAN_MATTERS> __device__ __forceinline__ double sgn_of(double v, SpecWatch &w)
{
    if constexpr (SPECULATE) {
        if constexpr (!NAN_MATTERS) w.zmin = __builtin_fmin(w.zmin, __builtin_fabs(v));
        return __builtin_copysign(1.0, v);
    } else {
        return SDRX_SGN(v);
    }
}

// ------------------------------------------------------------------------------------- loops

constexpr int SINCOS_TAB_DOUBLES = 2 * SDRX_SINCOS_TAB_N;
static __device__ const double d_sincos_tab[SINCOS_TAB_DOUBLES] = {SDRX_SINCOS_TAB_VALUES};

// the workgroup's copy of sdrx_sincos_tab's table (4 KB); the caller puts a barrier behind it
__device__ __forceinline__ void sincos_table_to_lds(double *tab)
{
    for (unsigned i = threadIdx.x; i < SINCOS_TAB_DOUBLES; i += blockDim.x) tab[i] = d_sincos_tab[i];
}

struct Interp {
    double d1, d2, d3, d4, d5;
};

__device__ __forceinline__ double interp_filter(Interp &d, double in, double mu)
{
    const double K = -0.5; // BPSK.cpp:319-344
    double V1 = d.d1 - K * in;
    double V2 = V1 + d.d2;
    double V3 = V2 - d.d3;
    double V4 = K * in - d.d1 + d.d4;
    double V5 = V4 + d.d2;
    double V6 = d.d3 + V5;
    double V7 = V3 * mu + V6;
    double V8 = d.d5 + mu * V7;
    d.d3 = d.d2;
    d.d2 = d.d1;
    d.d1 = K * in;
    d.d5 = d.d4;
    d.d4 = in;
    return V8;
}

// Loop state of one channel, in registers.
struct TimingState {
    double nco, reg_mu, reg_Q, reg_I, mu_delay;
    Interp iq, ii;
    double g1, g2, g3, g4, lf_x, lf_y;
};
struct PhaseState {
    double dds_delay, plf_x, plf_y;
    double tr, ti, temp; // locals of phase_shift_correction: restart at 0 every call
};

// One sample through symbol_timing_synchronization (BPSK.cpp:307-316): (re, im) -> (reg_Q, reg_I).
__device__ __forceinline__ void timing_step(TimingState &t, double &re, double &im)
{
    const double strobe = (t.nco < 0) ? 1.0 : 0.0;
    t.reg_mu = (strobe >= 1) ? t.mu_delay * 2 : t.reg_mu;
    const double mu = t.reg_mu;
    const double Q = interp_filter(t.iq, re, mu); // "Q" is fed the real rail (BPSK.cpp:311)
    const double I = interp_filter(t.ii, im, mu);
    // GardnerDetector, BPSK.cpp:346-363
    const double eg = (t.g3 * (-SDRX_SGN(Q) + SDRX_SGN(t.g4)) + (t.g1 * (-SDRX_SGN(I) + SDRX_SGN(t.g2))));
    const double e = (strobe >= 0.5) ? eg : 0.0;
    t.g2 = t.g1;
    t.g1 = I;
    t.g4 = t.g3;
    t.g3 = Q;
    const double K1 = -0.00587880145347487, K2 = -2.35152058138995e-05; // LoopFilter, BPSK.cpp:365-373
    const double v = t.lf_y + K2 * e + K1 * (e - t.lf_x);
    t.lf_x = e;
    t.lf_y = v;
    const double eta = t.nco - floor(t.nco); // NcoControl, BPSK.cpp:375-380
    t.nco = eta - 0.5 - v;
    t.mu_delay = eta;
    t.reg_Q = (strobe >= 1) ? Q : t.reg_Q;
    t.reg_I = (strobe >= 1) ? I : t.reg_I;
    re = t.reg_Q;
    im = t.reg_I;
}

// The output lookup on a de-rotated sample (tr, ti).
template <int KIND> __device__ __forceinline__ int psk_decision(double tr, double ti)
{
    if constexpr (KIND == SDRX_BPSK) {
        return (tr < 0.0) ? 1 : 0; // BPSK.cpp:413-421: SIGN_FUNC(tr) < 0 is tr < 0 (a NaN has sign +1: neither)
    } else {
        // QPSK.cpp:396-414 compares the two SIGN_FUNC values; written on the samples themselves: sign < 0 is v < 0,
        // sign > 0 is !(v < 0) && !(v == 0) (which holds for a NaN, whose SIGN_FUNC is +1)
        const bool rn = tr < 0.0, in = ti < 0.0;
        const bool rp = !rn && !(tr == 0.0), ip = !in && !(ti == 0.0);
        return (rn && in) ? 3 : ((rp && in) ? 2 : ((rn && ip) ? 1 : 0));
    }
}

// One sample through phase_shift_correction + the output lookup; returns the decision.
// SPECULATE: only the straight-line common cases of sincos and SIGN_FUNC run and `watch` collects what would leave them;
// the caller repeats the block without SPECULATE if any lane's watch is raised (SpecWatch; sdrx_math.h: sdrx_sincos_tab_core).
// cos / sin: sdrx_sincos_tab (include/sdrx_math.h), its 256-entry table in LDS (`tab`: sincos_table_to_lds).
template <int KIND, bool SPECULATE = false>
__device__ __forceinline__ int phase_step(PhaseState &p, double re, double im, const double *tab, SpecWatch *watch = nullptr)
{
    const double PK1 = 0.0620001240002480, PK2 = 0.000992001984003968; // BPSK.cpp:424-432
    double sn, cs;
    SpecWatch unused;
    SpecWatch &flag = SPECULATE ? *watch : unused;
    auto sincos_of = [&](double a) {
        if constexpr (SPECULATE) {
            (void)sdrx_sincos_tab_core(a, tab, &sn, &cs);
            flag.amax = __builtin_fmax(flag.amax, __builtin_fabs(a)); // (a NaN is not kept: it comes out of both forms as a NaN)
        } else {
            sdrx_sincos_tab(a, tab, &sn, &cs);
        }
    };
    if constexpr (KIND == SDRX_BPSK) { // BPSK.cpp:399-411
        sincos_of(p.dds_delay + p.temp);
        p.dds_delay = p.dds_delay + p.temp;
        p.tr = re * cs - im * sn;
        p.ti = im * cs + re * sn;
        const double dif = sgn_of<SPECULATE, false>(p.ti, flag) * p.tr - sgn_of<SPECULATE, false>(p.tr, flag) * p.ti;
        const double val = 8.0 * dif;
        // (PK1 + PK2) * (8 dif) and (8 (PK1 + PK2)) * dif are the same real product rounded once -- the same double -- as
        // long as 8 dif itself does not overflow; the second form keeps the scaling off the serial chain.  An overflow
        // leaves val = +-inf in plf_x, the next step's loop-filter output is infinite and so is the sincos argument made of
        // it (amax); what a block's last two steps leave behind is covered by the test of plf_x and plf_y where the
        // block's watch is read (carrier_block's callers).
        if constexpr (SPECULATE) p.temp = p.plf_y - PK1 * p.plf_x + (8.0 * (PK1 + PK2)) * dif;
        else p.temp = p.plf_y - PK1 * p.plf_x + (PK1 + PK2) * val;
        p.plf_x = val;
        p.plf_y = p.temp;
        return psk_decision<KIND>(p.tr, p.ti);
    } else { // QPSK.cpp:361-373: loop filter first, from the previous product
        const double dif = sgn_of<SPECULATE, false>(p.ti, flag) * p.tr - sgn_of<SPECULATE, false>(p.tr, flag) * p.ti;
        const double val = 8.0 * dif;
        // (PK1 + PK2) * (8 dif) and (8 (PK1 + PK2)) * dif are the same real product rounded once -- the same double -- as
        // long as 8 dif itself does not overflow; the second form keeps the scaling off the serial chain.  An overflow
        // leaves val = +-inf in plf_x, the next step's loop-filter output is infinite and so is the sincos argument made of
        // it (amax); what a block's last two steps leave behind is covered by the test of plf_x and plf_y where the
        // block's watch is read (carrier_block's callers).
        if constexpr (SPECULATE) p.temp = p.plf_y - PK1 * p.plf_x + (8.0 * (PK1 + PK2)) * dif;
        else p.temp = p.plf_y - PK1 * p.plf_x + (PK1 + PK2) * val;
        p.plf_x = val;
        p.plf_y = p.temp;
        sincos_of(p.dds_delay + p.temp);
        p.dds_delay = p.dds_delay + p.temp;
        p.tr = re * cs - im * sn;
        p.ti = im * cs + re * sn;
        return psk_decision<KIND>(p.tr, p.ti);
    }
}

// One reference call may be cut into several launches over consecutive pieces of the block (the block pipeline of
// sdrx_psk_demod_dev).  The loop structs carry through `states` as they do from call to call; the three locals of
// phase_shift_correction, which restart at 0 only once per CALL, carry through `carry` (4 doubles per channel).
enum { PIECE_NOT_FIRST = 1, PIECE_NOT_LAST = 2 };

// The two loops are each one long dependency chain per sample, but the timing loop of sample i+1
// does not depend on the carrier loop of sample i: the kernel runs them one sample apart in the
// same straight-line block (no branches inside: every conditional above is a select), so the
// compiler interleaves the two chains and a lone wave keeps the VALU busy during each other's latencies.
template <int KIND, bool TIMING, bool PHASE>
__global__ void __launch_bounds__(64) psk_loops_kernel(double2 *__restrict__ mf, size_t m, size_t nchan, size_t pitch,
                                                       sdrx_psk_state *__restrict__ states, uint8_t *__restrict__ dec,
                                                       size_t out_pitch, double *__restrict__ carry, int piece)
{
    __shared__ __align__(16) double tab[SINCOS_TAB_DOUBLES];
    if constexpr (PHASE) {
        sincos_table_to_lds(tab);
        __syncthreads();
    }
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchan) return;
    // latency-bound chains: when a throughput kernel (the next piece's matched filter) shares the SIMD, issue first
    __builtin_amdgcn_s_setprio(3);
    sdrx_psk_state *st = states + c;
    TimingState t = {st->nco_delay, st->reg_mu, st->reg_Q, st->reg_I, st->mu_delay,
                     {st->iq_d[0], st->iq_d[1], st->iq_d[2], st->iq_d[3], st->iq_d[4]},
                     {st->ii_d[0], st->ii_d[1], st->ii_d[2], st->ii_d[3], st->ii_d[4]},
                     st->g1, st->g2, st->g3, st->g4, st->lf_x, st->lf_y};
    PhaseState p = {st->dds_delay, st->plf_x, st->plf_y, 0.0, 0.0, 0.0};
    if (piece & PIECE_NOT_FIRST) { p.tr = carry[4 * c]; p.ti = carry[4 * c + 1]; p.temp = carry[4 * c + 2]; }
    double2 *x = mf + c * pitch;
    uint8_t *d = dec + c * out_pitch;

    // what the carrier loop (or the store, in a timing-only run) does with the timing loop's sample i
    auto finish = [&](double re, double im, size_t i, bool even) {
        if constexpr (PHASE) {
            const int decision = phase_step<KIND>(p, re, im, tab);
            if (even) d[i >> 1] = (uint8_t)decision; // decimate(real, n, 2): keeps 0, 2, 4, ...
        } else {
            x[i] = make_double2(re, im);
        }
    };

    constexpr int PF = 4; // samples fetched ahead per lane: one 64-byte line
    const size_t full = m - m % PF;
    double pr = 0.0, pi = 0.0; // sample i-1 after the timing loop, waiting for its carrier-loop turn
    if (full) {
        double2 nxt[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) nxt[k] = x[k];
        for (size_t i0 = 0; i0 < full; i0 += PF) {
            double2 cur[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) cur[k] = nxt[k];
            if (i0 + PF < full) {
#pragma unroll
                for (int k = 0; k < PF; k++) nxt[k] = x[i0 + PF + k];
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                double re = cur[k].x, im = cur[k].y;
                // sample i0+k-1 through the carrier loop, sample i0+k through the timing loop: independent
                if (k > 0 || i0 > 0) finish(pr, pi, i0 + k - 1, ((k - 1) & 1) == 0);
                if constexpr (TIMING) timing_step(t, re, im);
                pr = re;
                pi = im;
            }
        }
        finish(pr, pi, full - 1, ((PF - 1) & 1) == 0);
    }
    for (size_t i = full; i < m; i++) { // ragged tail, one sample at a time
        double re = x[i].x, im = x[i].y;
        if constexpr (TIMING) timing_step(t, re, im);
        finish(re, im, i, (i & 1) == 0);
    }
    if constexpr (TIMING) {
        st->nco_delay = t.nco; st->reg_mu = t.reg_mu; st->reg_Q = t.reg_Q; st->reg_I = t.reg_I; st->mu_delay = t.mu_delay;
        st->iq_d[0] = t.iq.d1; st->iq_d[1] = t.iq.d2; st->iq_d[2] = t.iq.d3; st->iq_d[3] = t.iq.d4; st->iq_d[4] = t.iq.d5;
        st->ii_d[0] = t.ii.d1; st->ii_d[1] = t.ii.d2; st->ii_d[2] = t.ii.d3; st->ii_d[3] = t.ii.d4; st->ii_d[4] = t.ii.d5;
        st->g1 = t.g1; st->g2 = t.g2; st->g3 = t.g3; st->g4 = t.g4;
        st->lf_x = t.lf_x; st->lf_y = t.lf_y;
    }
    if constexpr (PHASE) {
        st->dds_delay = p.dds_delay; st->plf_x = p.plf_x; st->plf_y = p.plf_y;
        if (piece & PIECE_NOT_LAST) { carry[4 * c] = p.tr; carry[4 * c + 1] = p.ti; carry[4 * c + 2] = p.temp; }
    }
}

// Both loops of a full demodulation on TWO waves of one workgroup (two SIMDs of a CU): wave 0 runs the
// timing loop over block j of SPLIT_KB samples into an LDS ring, wave 1 the carrier loop over block
// j-1 out of it, one s_barrier per block.  With one channel per lane a bank of 2^14 channels is only
// 256 waves on 1024 SIMDs, and each loop alone is issue-bound on its SIMD: splitting them uses a
// second SIMD per 64 channels.  Same per-sample arithmetic as psk_loops_kernel.
// samples per ring block.  Measured on 2^14 channels x 1024 matched-filter samples (loops alone): 8 0.258 ms, 16 0.215,
// 24 0.204, 32 0.2015, 64 0.282 (one barrier per block; the carrier loop starts one block late)
#ifndef SDRX_SPLIT_KB
#define SDRX_SPLIT_KB 32
#endif
constexpr int SPLIT_KB = SDRX_SPLIT_KB;

template <int KIND>
__global__ void __launch_bounds__(128) psk_loops_split_kernel(const double2 *__restrict__ mf, size_t m, size_t nchan,
                                                              size_t pitch, sdrx_psk_state *__restrict__ states,
                                                              uint8_t *__restrict__ dec, size_t out_pitch,
                                                              double *__restrict__ carry, int piece)
{
    __shared__ double2 ring[2][SPLIT_KB][64];
    __shared__ __align__(16) double tab[SINCOS_TAB_DOUBLES];
    sincos_table_to_lds(tab); // (the first barrier of the block loop below comes before the carrier wave's first step)
    __builtin_amdgcn_s_setprio(3); // latency-bound chains: issue ahead of a throughput kernel sharing the SIMD
    const unsigned lane = threadIdx.x & 63;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t c0 = (size_t)blockIdx.x * 64 + lane;
    const bool active = c0 < nchan;
    const size_t c = active ? c0 : nchan - 1; // idle lanes shadow the last channel and store nothing
    sdrx_psk_state *st = states + c;
    const double2 *x = mf + c * pitch;
    uint8_t *d = dec + c * out_pitch;
    const size_t nblocks = (m + SPLIT_KB - 1) / SPLIT_KB;

    TimingState t = {st->nco_delay, st->reg_mu, st->reg_Q, st->reg_I, st->mu_delay,
                     {st->iq_d[0], st->iq_d[1], st->iq_d[2], st->iq_d[3], st->iq_d[4]},
                     {st->ii_d[0], st->ii_d[1], st->ii_d[2], st->ii_d[3], st->ii_d[4]},
                     st->g1, st->g2, st->g3, st->g4, st->lf_x, st->lf_y};
    PhaseState p = {st->dds_delay, st->plf_x, st->plf_y, 0.0, 0.0, 0.0};
    if (piece & PIECE_NOT_FIRST) { p.tr = carry[4 * c]; p.ti = carry[4 * c + 1]; p.temp = carry[4 * c + 2]; }
    // A block that lies wholly inside the call (all but the last) runs without a per-sample test: one straight-line
    // run of SPLIT_KB steps, so the delay lines are renamed instead of moved and the signs the Gardner detector takes
    // of older samples are the ones already computed for them.
    auto timing_block = [&](auto full, const double2 (&cur)[SPLIT_KB], size_t j) {
#pragma unroll
        for (int k = 0; k < SPLIT_KB; k++) {
            double re = cur[k].x, im = cur[k].y;
            if (decltype(full)::value || j * SPLIT_KB + k < m) timing_step(t, re, im); // uniform: m is the same for all lanes
            ring[j & 1][k][lane] = make_double2(re, im);
        }
    };
    auto carrier_block = [&](auto full, auto speculate, size_t jb, SpecWatch &watch) {
        const size_t i0 = jb * SPLIT_KB;
#pragma unroll
        for (int k = 0; k < SPLIT_KB; k++) {
            const double2 v = ring[jb & 1][k][lane];
            if (decltype(full)::value || i0 + k < m) {
                const int decision = phase_step<KIND, decltype(speculate)::value>(p, v.x, v.y, tab, &watch);
                if ((k & 1) == 0 && active) d[(i0 + k) >> 1] = (uint8_t)decision; // SPLIT_KB is even: parity of k
            }
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    double2 nxt[SPLIT_KB];
    if (role == 0) {
#pragma unroll
        for (int k = 0; k < SPLIT_KB; k++) nxt[k] = ((size_t)k < m) ? x[k] : make_double2(0.0, 0.0);
    }
    for (size_t j = 0; j <= nblocks; j++) {
        if (role == 0) {
            if (j < nblocks) {
                double2 cur[SPLIT_KB];
#pragma unroll
                for (int k = 0; k < SPLIT_KB; k++) cur[k] = nxt[k];
                const size_t nb = (j + 1) * SPLIT_KB; // the next block's samples land during this block's arithmetic
                if (nb + SPLIT_KB <= m) {
#pragma unroll
                    for (int k = 0; k < SPLIT_KB; k++) nxt[k] = x[nb + k];
                } else {
#pragma unroll
                    for (int k = 0; k < SPLIT_KB; k++) nxt[k] = (nb + k < m) ? x[nb + k] : make_double2(0.0, 0.0);
                }
                if (nb <= m) timing_block(yes_t(), cur, j);
                else timing_block(no_t(), cur, j);
            }
        } else if (j >= 1) {
            // the block on sincos's common case alone; if any lane's argument was outside it (a loop that has run
            // away: never in normal operation) the block is repeated from the saved state through the full function,
            // which returns the same values for every lane that was inside
            const PhaseState p0 = p;
            SpecWatch watch;
            if (j * SPLIT_KB <= m) carrier_block(yes_t(), yes_t(), j - 1, watch);
            else carrier_block(no_t(), yes_t(), j - 1, watch);
            if (__builtin_amdgcn_ballot_w64(watch.raised() || !(__builtin_fabs(p.plf_x) <= 1.0e300) || !(__builtin_fabs(p.plf_y) <= 1.0e300)) != 0) {
                p = p0;
                carrier_block(no_t(), no_t(), j - 1, watch);
            }
        }
        __syncthreads();
    }
    if (!active) return;
    if (role == 0) {
        st->nco_delay = t.nco; st->reg_mu = t.reg_mu; st->reg_Q = t.reg_Q; st->reg_I = t.reg_I; st->mu_delay = t.mu_delay;
        st->iq_d[0] = t.iq.d1; st->iq_d[1] = t.iq.d2; st->iq_d[2] = t.iq.d3; st->iq_d[3] = t.iq.d4; st->iq_d[4] = t.iq.d5;
        st->ii_d[0] = t.ii.d1; st->ii_d[1] = t.ii.d2; st->ii_d[2] = t.ii.d3; st->ii_d[3] = t.ii.d4; st->ii_d[4] = t.ii.d5;
        st->g1 = t.g1; st->g2 = t.g2; st->g3 = t.g3; st->g4 = t.g4;
        st->lf_x = t.lf_x; st->lf_y = t.lf_y;
    } else {
        st->dds_delay = p.dds_delay; st->plf_x = p.plf_x; st->plf_y = p.plf_y;
        if (piece & PIECE_NOT_LAST) { carry[4 * c] = p.tr; carry[4 * c + 1] = p.ti; carry[4 * c + 2] = p.temp; }
    }
}

// The same on FOUR waves (the four SIMDs of a CU; the kernel keeps the name it had with three).  A lone wave pays 6 to 8
// cycles per instruction of a mostly serial stream (profiles/probes/issue_probe.hip, profiles/r03_tri_prof.txt), so a
// loop's step time is the instruction count of its slowest wave, and the work is cut wherever the recurrences allow:
//   wave 0  InterpolationFilter's V1 ... V6 and its delay-line rotation on both rails (BPSK.cpp:323-328,336-341): functions of
//           the input samples only, not of mu; one ring block ahead, (V3, V6, d5) per rail through ring A
//   wave 1  the rest of the timing loop (V7 = V3*mu + V6; V8 = d5 + mu*V7; detector, loop filter, NCO); (reg_Q, reg_I)
//           through ring B
//   wave 2  the carrier loop; every second de-rotated sample through ring C
//   wave 3  the output lookup and the stores (eight decisions per lane and store)
// The same operations in the same order on the same values: the same bits.  The delay lines live in wave 0 (it stores
// them back), the rest of the timing state in wave 1, the carrier state in wave 2.
constexpr int TRI_KB = 16; // samples per ring block: ring A 96 KB + ring B 32 KB + ring C 16 KB (+ 4 KB sincos table): one workgroup per CU

struct InterpPrefix { double V3, V6, d5; };
__device__ __forceinline__ InterpPrefix interp_prefix(Interp &d, double in)
{
    const double K = -0.5; // BPSK.cpp:319-344, the statements that do not read mu
    const double Kin = K * in;
    const double V1 = d.d1 - Kin;
    const double V2 = V1 + d.d2;
    const double V3 = V2 - d.d3;
    const double V4 = Kin - d.d1 + d.d4;
    const double V5 = V4 + d.d2;
    const double V6 = d.d3 + V5;
    const InterpPrefix r = {V3, V6, d.d5};
    d.d3 = d.d2;
    d.d2 = d.d1;
    d.d1 = Kin;
    d.d5 = d.d4;
    d.d4 = in;
    return r;
}

// The Gardner detector's SIGN_FUNC values of g1 .. g4, carried beside them: each is the sign of an interpolator output
// taken once, when that output was new (the detector of BPSK.cpp:346-363 takes it again one and two samples later).
struct GardnerSigns { double s1, s2, s3, s4; };

// timing_step with the interpolator's prefix already done (q, i: rails "Q" and "I" of this sample).
// SPECULATE: see sgn_of.
template <bool SPECULATE>
__device__ __forceinline__ void timing_step_tail(TimingState &t, GardnerSigns &gs, const InterpPrefix &q, const InterpPrefix &i,
                                                 double &re, double &im, SpecWatch &watch)
{
    const double strobe = (t.nco < 0) ? 1.0 : 0.0;
    t.reg_mu = (strobe >= 1) ? t.mu_delay * 2 : t.reg_mu;
    const double mu = t.reg_mu;
    const double V7q = q.V3 * mu + q.V6, Q = q.d5 + mu * V7q;
    const double V7i = i.V3 * mu + i.V6, I = i.d5 + mu * V7i;
    const double sQ = sgn_of<SPECULATE, true>(Q, watch), sI = sgn_of<SPECULATE, true>(I, watch);
    if constexpr (SPECULATE) watch.odd |= __builtin_amdgcn_class(Q * I, 0x063); // sNaN, qNaN, -0, +0 (a product that underflows: a needless repeat)
    const double eg = (t.g3 * (-sQ + gs.s4) + (t.g1 * (-sI + gs.s2)));
    const double e = (strobe >= 0.5) ? eg : 0.0;
    t.g2 = t.g1;
    t.g1 = I;
    t.g4 = t.g3;
    t.g3 = Q;
    gs.s2 = gs.s1;
    gs.s1 = sI;
    gs.s4 = gs.s3;
    gs.s3 = sQ;
    const double K1 = -0.00587880145347487, K2 = -2.35152058138995e-05;
    const double v = t.lf_y + K2 * e + K1 * (e - t.lf_x);
    t.lf_x = e;
    t.lf_y = v;
    const double eta = t.nco - floor(t.nco);
    t.nco = eta - 0.5 - v;
    t.mu_delay = eta;
    t.reg_Q = (strobe >= 1) ? Q : t.reg_Q;
    t.reg_I = (strobe >= 1) ? I : t.reg_I;
    re = t.reg_Q;
    im = t.reg_I;
}

template <int KIND>
__global__ void __launch_bounds__(256) psk_loops_tri_kernel(const double2 *__restrict__ mf, size_t m, size_t nchan,
                                                            size_t pitch, sdrx_psk_state *__restrict__ states,
                                                            uint8_t *__restrict__ dec, size_t out_pitch,
                                                            double *__restrict__ carry, int piece)
{
    extern __shared__ __align__(16) unsigned char tri_smem[];
    typedef double2 (*ring_a_t)[TRI_KB][3][64]; // [buffer][sample][(V3q,V6q) | (V3i,V6i) | (d5q,d5i)][lane]
    typedef double2 (*ring_b_t)[TRI_KB][64];    // [buffer][sample][lane]: (reg_Q, reg_I)
    // (static: its address is a compile-time constant, so a table read's address is the scaled index itself)
    __shared__ __align__(16) double tab[SINCOS_TAB_DOUBLES];
    ring_a_t ringA = (ring_a_t)tri_smem;
    ring_b_t ringB = (ring_b_t)(tri_smem + sizeof(double2) * 2 * TRI_KB * 3 * 64);
    typedef double2 (*ring_c_t)[TRI_KB / 2][64];  // [buffer][even sample][lane]: (tr, ti)
    ring_c_t ringC = (ring_c_t)(tri_smem + sizeof(double2) * 2 * TRI_KB * 4 * 64);
    sincos_table_to_lds(tab); // (two barriers of the block loop come before the carrier wave's first step)
    __builtin_amdgcn_s_setprio(3);
    const unsigned lane = threadIdx.x & 63;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t c0 = (size_t)blockIdx.x * 64 + lane;
    const bool active = c0 < nchan;
    const size_t c = active ? c0 : nchan - 1; // idle lanes shadow the last channel and store nothing
    sdrx_psk_state *st = states + c;
    const double2 *x = mf + c * pitch;
    uint8_t *d = dec + c * out_pitch;
    const size_t nblocks = (m + TRI_KB - 1) / TRI_KB;

    Interp iq = {st->iq_d[0], st->iq_d[1], st->iq_d[2], st->iq_d[3], st->iq_d[4]};
    Interp ii = {st->ii_d[0], st->ii_d[1], st->ii_d[2], st->ii_d[3], st->ii_d[4]};
    TimingState t = {st->nco_delay, st->reg_mu, st->reg_Q, st->reg_I, st->mu_delay, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
                     st->g1, st->g2, st->g3, st->g4, st->lf_x, st->lf_y};
    GardnerSigns gs = {SDRX_SGN(t.g1), SDRX_SGN(t.g2), SDRX_SGN(t.g3), SDRX_SGN(t.g4)};
    PhaseState p = {st->dds_delay, st->plf_x, st->plf_y, 0.0, 0.0, 0.0};
    if (piece & PIECE_NOT_FIRST) { p.tr = carry[4 * c]; p.ti = carry[4 * c + 1]; p.temp = carry[4 * c + 2]; }

    auto prefix_block = [&](auto full, const double2 (&cur)[TRI_KB], size_t j) {
#pragma unroll
        for (int k = 0; k < TRI_KB; k++) {
            if (decltype(full)::value || j * TRI_KB + k < m) { // uniform: m is the same for all lanes
                const InterpPrefix q = interp_prefix(iq, cur[k].x); // "Q" is fed the real rail (BPSK.cpp:311)
                const InterpPrefix i = interp_prefix(ii, cur[k].y);
                ringA[j & 1][k][0][lane] = make_double2(q.V3, q.V6);
                ringA[j & 1][k][1][lane] = make_double2(i.V3, i.V6);
                ringA[j & 1][k][2][lane] = make_double2(q.d5, i.d5);
            }
        }
    };
    // The ring reads of step k + TRI_PF are issued before step k's arithmetic (and before its ring write, which the
    // compiler must assume to alias them): an LDS round trip is longer than a quarter of a step, and a lone wave has
    // nothing else to run while it waits.  (Reading a whole ring block, stale entries past the call's end included, is
    // harmless: their values are never used.)
    constexpr int TRI_PF = 3;
    auto timing_block = [&](auto full, auto speculate, size_t j, SpecWatch &watch) {
        double2 a[TRI_KB], b[TRI_KB], dd[TRI_KB];
#pragma unroll
        for (int k = 0; k < TRI_PF; k++) { a[k] = ringA[j & 1][k][0][lane]; b[k] = ringA[j & 1][k][1][lane]; dd[k] = ringA[j & 1][k][2][lane]; }
#pragma unroll
        for (int k = 0; k < TRI_KB; k++) {
            if (k + TRI_PF < TRI_KB) {
                a[k + TRI_PF] = ringA[j & 1][k + TRI_PF][0][lane];
                b[k + TRI_PF] = ringA[j & 1][k + TRI_PF][1][lane];
                dd[k + TRI_PF] = ringA[j & 1][k + TRI_PF][2][lane];
            }
            if (decltype(full)::value || j * TRI_KB + k < m) {
                const InterpPrefix q = {a[k].x, a[k].y, dd[k].x}, i = {b[k].x, b[k].y, dd[k].y};
                double re, im;
                timing_step_tail<decltype(speculate)::value>(t, gs, q, i, re, im, watch);
                ringB[j & 1][k][lane] = make_double2(re, im);
            }
        }
    };
    // The de-rotated samples the decisions are taken of (every second one: BPSK.cpp:413-421's decimate-by-2) go back to
    // a fourth wave through ring C: the lookup and, above all, the stores -- 64 lanes, 64 cache lines -- cost the carrier
    // wave a quarter of its time (profiles/r03_tri_prof.txt).
    auto carrier_block = [&](auto full, auto speculate, size_t jb, SpecWatch &watch) {
        const size_t i0 = jb * TRI_KB;
        double2 v[TRI_KB];
#pragma unroll
        for (int k = 0; k < TRI_KB; k++) v[k] = ringB[jb & 1][k][lane];
#pragma unroll
        for (int k = 0; k < TRI_KB; k++) {
            if (decltype(full)::value || i0 + k < m) {
                (void)phase_step<KIND, decltype(speculate)::value>(p, v[k].x, v[k].y, tab, &watch);
                if ((k & 1) == 0) ringC[jb & 1][k >> 1][lane] = make_double2(p.tr, p.ti); // TRI_KB is even: parity of k
            }
        }
    };
    // decisions of block jb: eight per lane, one 8-byte store where the row allows it
    const bool row_aligned = (((size_t)d) & 7) == 0;
    auto decision_block = [&](size_t jb) {
        const size_t i0 = jb * TRI_KB;
        unsigned w[2] = {0u, 0u};
#pragma unroll
        for (int k2 = 0; k2 < TRI_KB / 2; k2++) {
            const double2 v = ringC[jb & 1][k2][lane]; // (entries past the call's end: stale, not stored)
            w[k2 >> 2] |= (unsigned)psk_decision<KIND>(v.x, v.y) << (8 * (k2 & 3));
        }
        if (!active) return;
        uint8_t *o = d + (i0 >> 1);
        if (i0 + TRI_KB <= m && row_aligned) {
            *(uint2 *)o = make_uint2(w[0], w[1]);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < TRI_KB / 2; k2++)
                if (i0 + 2 * k2 < m) o[k2] = (uint8_t)(w[k2 >> 2] >> (8 * (k2 & 3)));
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    double2 nxt[TRI_KB];
    if (role == 0) {
#pragma unroll
        for (int k = 0; k < TRI_KB; k++) nxt[k] = ((size_t)k < m) ? x[k] : make_double2(0.0, 0.0);
    }
    // iteration j: wave 0 on block j, wave 1 on block j-1, wave 2 on block j-2, wave 3 on the decisions of block j-3; one
    // barrier per iteration.  A ring buffer written in iteration j is read in iteration j+1 and written again in j+2: two
    // buffers each.
#ifdef SDRX_TRI_PROFILE // profiles/probes/tri_prof.hip: where each wave's time goes (never defined in the library)
    long long prof_work = 0, prof_wait = 0, prof_replays = 0;
#endif
    for (size_t j = 0; j < nblocks + 3; j++) {
#ifdef SDRX_TRI_PROFILE
        const long long prof_t0 = __builtin_amdgcn_s_memtime();
#endif
        if (role == 0) {
            if (j < nblocks) {
                double2 cur[TRI_KB];
#pragma unroll
                for (int k = 0; k < TRI_KB; k++) cur[k] = nxt[k];
                const size_t nb = (j + 1) * TRI_KB; // the next block's samples land during this block's arithmetic
                if (nb + TRI_KB <= m) {
#pragma unroll
                    for (int k = 0; k < TRI_KB; k++) nxt[k] = x[nb + k];
                } else {
#pragma unroll
                    for (int k = 0; k < TRI_KB; k++) nxt[k] = (nb + k < m) ? x[nb + k] : make_double2(0.0, 0.0);
                }
                if (nb <= m) prefix_block(yes_t(), cur, j);
                else prefix_block(no_t(), cur, j);
            }
        } else if (role == 3) {
            if (j >= 3) decision_block(j - 3);
        } else if (role == 1) {
            if (j >= 1 && j - 1 < nblocks) {
                const TimingState t0 = t; // (speculation on the signs' common case: sgn_of)
                const GardnerSigns gs0 = gs;
                SpecWatch watch;
                if (j * TRI_KB <= m) timing_block(yes_t(), yes_t(), j - 1, watch);
                else timing_block(no_t(), yes_t(), j - 1, watch);
                if (__builtin_amdgcn_ballot_w64(watch.raised()) != 0) {
                    t = t0;
                    gs = gs0;
                    timing_block(no_t(), no_t(), j - 1, watch);
                }
            }
        } else if (role == 2 && j >= 2 && j - 2 < nblocks) {
            const PhaseState p0 = p; // (speculation on the common case of sincos and of the signs: SpecWatch)
            SpecWatch watch;
            if ((j - 1) * TRI_KB <= m) carrier_block(yes_t(), yes_t(), j - 2, watch);
            else carrier_block(no_t(), yes_t(), j - 2, watch);
            if (__builtin_amdgcn_ballot_w64(watch.raised() || !(__builtin_fabs(p.plf_x) <= 1.0e300) || !(__builtin_fabs(p.plf_y) <= 1.0e300)) != 0) {
                p = p0;
                carrier_block(no_t(), no_t(), j - 2, watch);
            }
        }
#ifdef SDRX_TRI_PROFILE
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the block's own LDS traffic counts as work
        const long long prof_t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        prof_work += prof_t1 - prof_t0;
        prof_wait += (long long)__builtin_amdgcn_s_memtime() - prof_t1;
    }
    if (blockIdx.x == 0 && lane == 0)
        printf("role %u: work %lld wait %lld ticks over %zu iterations (%zu samples)\n", role, prof_work, prof_wait, nblocks + 3, m);
    {
#else
        __syncthreads();
#endif
    }
    if (!active) return;
    if (role == 0) {
        st->iq_d[0] = iq.d1; st->iq_d[1] = iq.d2; st->iq_d[2] = iq.d3; st->iq_d[3] = iq.d4; st->iq_d[4] = iq.d5;
        st->ii_d[0] = ii.d1; st->ii_d[1] = ii.d2; st->ii_d[2] = ii.d3; st->ii_d[3] = ii.d4; st->ii_d[4] = ii.d5;
    } else if (role == 1) {
        st->nco_delay = t.nco; st->reg_mu = t.reg_mu; st->reg_Q = t.reg_Q; st->reg_I = t.reg_I; st->mu_delay = t.mu_delay;
        st->g1 = t.g1; st->g2 = t.g2; st->g3 = t.g3; st->g4 = t.g4;
        st->lf_x = t.lf_x; st->lf_y = t.lf_y;
    } else if (role == 2) {
        st->dds_delay = p.dds_delay; st->plf_x = p.plf_x; st->plf_y = p.plf_y;
        if (piece & PIECE_NOT_LAST) { carry[4 * c] = p.tr; carry[4 * c + 1] = p.ti; carry[4 * c + 2] = p.temp; }
    }
}
constexpr size_t TRI_LDS_BYTES = sizeof(double2) * 2 * (TRI_KB * 4 + TRI_KB / 2) * 64; // the rings (dynamic); + the 4 KB table (static)

int launch_psk_loops(int kind, double *mf_iq, size_t m, size_t nchan, size_t mf_pitch, sdrx_psk_state *states,
                     uint8_t *decisions, size_t out_pitch, int run_timing, int run_phase, hipStream_t s, double *carry,
                     int piece)
{
    if (piece && !carry) {
        set_error("psk_loops: a piece of a call needs the carry rows");
        return SDRX_EINVAL;
    }
    if (!mf_iq || !states || (run_phase && !decisions) || (kind != SDRX_BPSK && kind != SDRX_QPSK)) {
        set_error("psk_loops: bad argument");
        return SDRX_EINVAL;
    }
    if (nchan == 0 || m == 0 || (!run_timing && !run_phase)) return SDRX_OK;
    size_t blocks = (nchan + 63) / 64;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    // few channels (fewer waves than SIMDs): give each 64-channel group a second SIMD
    const int split_env = tuning().psk_split; // 0 / 1 force one form (tests, experiments)
    // (the two-wave form holds 64 KB of LDS per workgroup: two per CU, 512 at a time; measured, loops alone, two-wave /
    // one-wave: 2^15 channels 0.125 / 0.181 ms, 2^16 0.128 / 0.119, 2^17 0.146 / 0.115)
    const bool split = split_env >= 0 ? split_env != 0 : blocks <= 512;
    // the four-wave form (148 KB of LDS: one workgroup per CU) while every workgroup has a CU to itself
    const bool tri = split_env >= 0 ? split_env == 2 : blocks <= 256;
    if (run_timing && run_phase && tri) {
        typedef void (*tri_fn)(const double2 *, size_t, size_t, size_t, sdrx_psk_state *, uint8_t *, size_t, double *, int);
        tri_fn fn = kind == SDRX_BPSK ? (tri_fn)psk_loops_tri_kernel<SDRX_BPSK> : (tri_fn)psk_loops_tri_kernel<SDRX_QPSK>;
        SDRX_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRI_LDS_BYTES));
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), TRI_LDS_BYTES, s, (const double2 *)mf_iq, m, nchan, mf_pitch, states,
                           decisions, out_pitch, carry, piece);
        SDRX_HIP(hipGetLastError());
        note_kernel("psk_loops_tri_kernel", kind, 0, 0, 0);
        return SDRX_OK;
    }
    if (run_timing && run_phase && split) {
        if (kind == SDRX_BPSK)
            hipLaunchKernelGGL(psk_loops_split_kernel<SDRX_BPSK>, dim3((unsigned)blocks), dim3(128), 0, s,
                               (const double2 *)mf_iq, m, nchan, mf_pitch, states, decisions, out_pitch, carry, piece);
        else
            hipLaunchKernelGGL(psk_loops_split_kernel<SDRX_QPSK>, dim3((unsigned)blocks), dim3(128), 0, s,
                               (const double2 *)mf_iq, m, nchan, mf_pitch, states, decisions, out_pitch, carry, piece);
        SDRX_HIP(hipGetLastError());
        note_kernel("psk_loops_split_kernel", kind, 0, 0, 0);
        return SDRX_OK;
    }
    typedef void (*fn_t)(double2 *, size_t, size_t, size_t, sdrx_psk_state *, uint8_t *, size_t, double *, int);
    fn_t fn;
    if (kind == SDRX_BPSK)
        fn = run_timing ? (run_phase ? psk_loops_kernel<SDRX_BPSK, true, true> : psk_loops_kernel<SDRX_BPSK, true, false>)
                        : psk_loops_kernel<SDRX_BPSK, false, true>;
    else
        fn = run_timing ? (run_phase ? psk_loops_kernel<SDRX_QPSK, true, true> : psk_loops_kernel<SDRX_QPSK, true, false>)
                        : psk_loops_kernel<SDRX_QPSK, false, true>;
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64), 0, s, (double2 *)mf_iq, m, nchan, mf_pitch, states, decisions,
                       out_pitch, carry, piece);
    SDRX_HIP(hipGetLastError());
    note_kernel("psk_loops_kernel", kind, run_timing, run_phase, 0);
    return SDRX_OK;
}

// --------------------------------------------------------------------------------- ambiguity
//
// One workgroup per channel.  Positions are tested 256 at a time; hits are rare and must update
// the saturating confidence counters in stream order (confidence_counter.cpp:41-61 is not
// commutative at the rails), so lane 0 folds a chunk's hits sequentially, only for chunks that
// have any.
template <int NT>
__global__ void __launch_bounds__(NT) psk_ambiguity_kernel(int kind, uint8_t *__restrict__ dec, size_t m,
                                                            size_t out_pitch, sdrx_psk_state *__restrict__ states,
                                                            unsigned long long *__restrict__ hits, size_t max_hits)
{
    __shared__ uint8_t uwc[4][64];
    __shared__ signed char ev[NT];
    __shared__ int s_cc[4];
    __shared__ int s_lut, s_nh;
    const unsigned tid = threadIdx.x;
    const size_t c = blockIdx.x;
    sdrx_psk_state *st = states + c;
    uint8_t *d = dec + c * out_pitch;
    const int U = st->uw_size > 64 ? 64 : (st->uw_size < 0 ? 0 : st->uw_size);
    // the one-wave form: the channel's decisions pass through LDS in segments of AMB_SEG positions (+ the unique word's
    // reach), the 16..64 compares of every position then read LDS instead of issuing a global byte load each
    constexpr bool STAGED = NT == 64;
    constexpr size_t AMB_SEG = 1920, AMB_REACH = 68; // segment + 64 symbols + the word the last position's compare over-reads
    __shared__ __align__(16) uint8_t sd[STAGED ? 2048 + 16 : 16];
    static_assert(AMB_SEG + AMB_REACH + 16 <= 2048 + 16 && AMB_SEG % 64 == 0, "segment fits the staging row");
    if (tid < 64 && (int)tid < U) {
        const int idx = (kind == SDRX_BPSK) ? (U - 1 - (int)tid) : (63 - (int)tid); // BPSK.cpp:456, QPSK.cpp:428
        for (int k = 0; k < 4; k++) uwc[k][tid] = (uint8_t)st->uw[k][idx];
    }
    if (tid < 4) s_cc[tid] = st->cc[tid];
    if (tid == 0) s_nh = 0;
    __syncthreads();
    // One-wave form: a position's U decisions are compared four bytes at a time.  A 32-bit word of the staged row at
    // any byte offset is two aligned LDS reads joined by v_alignbyte; the unique word's four rotations are packed into
    // words once per channel (uwq[q] = word q of all four).  QPSK's compare is an if / else-if chain per symbol
    // (QPSK.cpp:432-443: a symbol equal to an earlier rotation's entry never counts for a later one), which equals
    // "all U symbols match rotation k" whenever the four rows differ pairwise at every index -- always so for rows
    // built from the LUT rotations; a unique word with out-of-range entries leaves equal rows and takes the
    // symbol-by-symbol loop below.
    __shared__ uint4 uwq[16];
    bool fast = false;
    if constexpr (STAGED) {
        bool clash = false;
        if ((int)tid < U && kind != SDRX_BPSK)
            for (int k = 0; k < 4; k++)
                for (int l = k + 1; l < 4; l++) clash = clash || uwc[k][tid] == uwc[l][tid];
        fast = __builtin_amdgcn_ballot_w64(clash) == 0; // NT == 64: the workgroup is one wave
        if (fast && tid < 16) {
            unsigned w[4] = {0, 0, 0, 0};
            for (int b = 0; b < 4; b++) {
                const int j = 4 * (int)tid + b;
                if (j < U)
                    for (int k = 0; k < 4; k++) w[k] |= (unsigned)uwc[k][j] << (8 * b);
            }
            uwq[tid] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();
    }
    int cc0 = s_cc[0], cc1 = s_cc[1], cc2 = s_cc[2], cc3 = s_cc[3], nh = 0; // one-wave form: the counters as wave-uniform values
    if (m >= (size_t)U) {
        const size_t npos_all = m - (size_t)U; // i < num_of_points - UW_SIZE
      for (size_t seg0 = 0; seg0 < npos_all; seg0 += STAGED ? AMB_SEG : npos_all) {
        const size_t npos = STAGED ? (npos_all - seg0 < AMB_SEG ? npos_all - seg0 : AMB_SEG) : npos_all;
        const uint8_t *src = d; // where position 0 of the segment is read from
        if constexpr (STAGED) {
            __syncthreads(); // the previous segment's reads are done
            const uint8_t *g = d + seg0;
            const size_t cnt = (m - seg0) < (AMB_SEG + AMB_REACH) ? (m - seg0) : (AMB_SEG + AMB_REACH);
            if ((((size_t)g) & 3) == 0) {
                for (size_t w = tid; w < (cnt + 3) / 4; w += NT) {
                    unsigned v = 0;
                    if (4 * w + 4 <= cnt) v = *(const unsigned *)(g + 4 * w);
                    else for (size_t k = 4 * w; k < cnt; k++) v |= (unsigned)g[k] << (8 * (k - 4 * w));
                    *(unsigned *)(sd + 4 * w) = v;
                }
            } else {
                for (size_t k = tid; k < cnt; k += NT) sd[k] = g[k];
            }
            __syncthreads();
            src = sd;
        }
        for (size_t base = 0; base < npos; base += NT) {
            const size_t i = base + tid;
            int code = -1;
            if (STAGED && fast) {
                const unsigned *D = (const unsigned *)sd;
                const unsigned ic = i < npos ? (unsigned)i : 0u;
                const unsigned a = ic >> 2, sh = ic & 3u;
                unsigned lo = D[a];
                unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                const int nq = (U + 3) >> 2;
                for (int q = 0; q < nq; q++) {
                    const unsigned hi = D[a + q + 1]; // at most byte AMB_SEG + 70 of sd[2048 + 16]
                    unsigned w = __builtin_amdgcn_alignbyte(hi, lo, sh); // row bytes ic + 4q .. ic + 4q + 3
                    lo = hi;
                    const int rem = U - 4 * q;
                    if (rem < 4) w &= (1u << (8 * rem)) - 1u;
                    const uint4 u = uwq[q];
                    d0 |= w ^ u.x;
                    d1 |= w ^ u.y;
                    d2 |= w ^ u.z;
                    d3 |= w ^ u.w;
                }
                if (i < npos) {
                    if (kind == SDRX_BPSK) code = d0 == 0 ? 0 : (d1 == 0 ? 2 : -1);
                    else code = d0 == 0 ? 0 : (d1 == 0 ? 1 : (d2 == 0 ? 2 : (d3 == 0 ? 3 : -1)));
                }
            } else if (i < npos) {
                if (kind == SDRX_BPSK) {
                    bool m0 = true, m180 = true;
                    for (int j = 0; j < U && (m0 || m180); j++) {
                        uint8_t v = src[i + j];
                        m0 = m0 && (v == uwc[0][j]);
                        m180 = m180 && (v == uwc[1][j]);
                    }
                    code = m0 ? 0 : (m180 ? 2 : -1);
                } else {
                    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
                    for (int j = 0; j < U; j++) {
                        uint8_t v = src[i + j];
                        if (v == uwc[0][j]) cnt0++;
                        else if (v == uwc[1][j]) cnt1++;
                        else if (v == uwc[2][j]) cnt2++;
                        else if (v == uwc[3][j]) cnt3++;
                    }
                    code = cnt0 == U ? 0 : (cnt1 == U ? 1 : (cnt2 == U ? 2 : (cnt3 == U ? 3 : -1)));
                }
            }
            if constexpr (STAGED) {
                // one wave: the chunk's hits as a ballot mask, folded in stream order by wave-uniform (scalar) code
                unsigned long long hm = __builtin_amdgcn_ballot_w64(code >= 0);
                while (hm) {
                    const int t = __builtin_ctzll(hm);
                    hm &= hm - 1;
                    const int cd = __builtin_amdgcn_readlane(code, t);
                    if (kind == SDRX_BPSK) {
                        if (cd == 0) { if (--cc0 <= 0) cc0 = 0; }   // --cc: BPSK.cpp:468
                        else { if (++cc0 >= 10) cc0 = 10; }          // ++cc: BPSK.cpp:473
                    } else {                                         // QPSK.cpp:446-477
                        if (cd == 0) { if (++cc0 >= 10) cc0 = 10; } else { if (--cc0 <= 0) cc0 = 0; }
                        if (cd == 1) { if (++cc1 >= 10) cc1 = 10; } else { if (--cc1 <= 0) cc1 = 0; }
                        if (cd == 2) { if (++cc2 >= 10) cc2 = 10; } else { if (--cc2 <= 0) cc2 = 0; }
                        if (cd == 3) { if (++cc3 >= 10) cc3 = 10; } else { if (--cc3 <= 0) cc3 = 0; }
                    }
                    if (hits && (size_t)nh < max_hits && tid == 0)
                        hits[c * max_hits + nh] = ((unsigned long long)(seg0 + base + t) << 3) | (unsigned long long)cd;
                    nh++;
                }
            } else {
            ev[tid] = (signed char)code;
            const int any = __syncthreads_or(code >= 0);
            if (any && tid == 0) {
                for (int t = 0; t < NT; t++) {
                    const int cd = ev[t];
                    if (cd < 0) continue;
                    if (kind == SDRX_BPSK) {
                        if (cd == 0) { if (--s_cc[0] <= 0) s_cc[0] = 0; }   // --cc: BPSK.cpp:468
                        else { if (++s_cc[0] >= 10) s_cc[0] = 10; }          // ++cc: BPSK.cpp:473
                    } else {
                        for (int k = 0; k < 4; k++) {                         // QPSK.cpp:446-477
                            if (k == cd) { if (++s_cc[k] >= 10) s_cc[k] = 10; }
                            else { if (--s_cc[k] <= 0) s_cc[k] = 0; }
                        }
                    }
                    if (hits && (size_t)s_nh < max_hits)
                        hits[c * max_hits + s_nh] = ((unsigned long long)(base + t) << 3) | (unsigned long long)cd;
                    s_nh++;
                }
            }
            __syncthreads();
            }
        }
      }
    }
    if constexpr (STAGED) {
        if (tid == 0) { s_cc[0] = cc0; s_cc[1] = cc1; s_cc[2] = cc2; s_cc[3] = cc3; s_nh = nh; }
    }
    if (tid == 0) {
        int lut = -1;
        if (kind == SDRX_BPSK) {
            if (s_cc[0] >= 5) lut = 2; // BPSK.cpp:478 high_confidence() -> LUT_180
        } else {                       // QPSK.cpp:481-507
            if (s_cc[1] >= 5) lut = 1;
            else if (s_cc[2] >= 5) lut = 2;
            else if (s_cc[3] >= 5) lut = 3;
        }
        s_lut = lut;
        for (int k = 0; k < 4; k++) st->cc[k] = s_cc[k];
        st->n_hits = s_nh;
    }
    __syncthreads();
    const int lut = s_lut;
    if (lut < 0) return;
    if constexpr (STAGED) {
        if ((((size_t)d) & 3) == 0) { // four decisions per lane; the maps on packed bytes
            for (size_t w = tid; w < (m + 3) / 4; w += NT) {
                unsigned v = 0;
                if (4 * w + 4 <= m) v = *(const unsigned *)(d + 4 * w);
                else for (size_t k = 4 * w; k < m; k++) v |= (unsigned)d[k] << (8 * (k - 4 * w));
                const unsigned b = 0x01010101u;
                unsigned o;
                if (kind == SDRX_BPSK) o = v ^ b;
                else if (lut == 1) o = ((~v & b) << 1) | ((v >> 1) & b); // {2,0,3,1}: out1 = !in0, out0 = in1
                else if (lut == 2) o = 3u * b - (v & (3u * b));          // {3,2,1,0}
                else o = ((v & b) << 1) | ((~v >> 1) & b);               // {1,3,0,2}: out1 = in0, out0 = !in1
                if (4 * w + 4 <= m) *(unsigned *)(d + 4 * w) = o;
                else for (size_t k = 4 * w; k < m; k++) d[k] = (uint8_t)(o >> (8 * (k - 4 * w)));
            }
            return;
        }
    }
    for (size_t i = tid; i < m; i += NT) {
        const unsigned v = d[i];
        unsigned o;
        if (kind == SDRX_BPSK) o = v ^ 1u;                       // LUT_180 = {1,0}, BPSK.cpp:15
        else if (lut == 1) o = (0x1302u >> (4 * (v & 3))) & 0xf; // LUT_REAL__90 = {2,0,3,1}, QPSK.cpp:17
        else if (lut == 2) o = 3u - (v & 3);                     // LUT_REAL_180 = {3,2,1,0}, QPSK.cpp:18
        else o = (0x2031u >> (4 * (v & 3))) & 0xf;               // LUT_REAL_270 = {1,3,0,2}, QPSK.cpp:19
        d[i] = (uint8_t)o;
    }
}

int launch_psk_ambiguity(int kind, uint8_t *decisions, size_t m, size_t nchan, size_t out_pitch,
                         sdrx_psk_state *states, hipStream_t s)
{
    if (!decisions || !states || (kind != SDRX_BPSK && kind != SDRX_QPSK)) {
        set_error("psk_ambiguity: bad argument");
        return SDRX_EINVAL;
    }
    if (nchan == 0) return SDRX_OK;
    if (nchan > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    // one wave per channel (decisions staged through LDS, four symbols per compare) for short rows and for banks with
    // enough channels to fill the chip with waves; four waves per channel reading global bytes for a few long rows
    if (m <= 2048 || nchan >= 256)
        hipLaunchKernelGGL(psk_ambiguity_kernel<64>, dim3((unsigned)nchan), dim3(64), 0, s, kind, decisions, m, out_pitch,
                           states, (unsigned long long *)nullptr, (size_t)0);
    else
        hipLaunchKernelGGL(psk_ambiguity_kernel<256>, dim3((unsigned)nchan), dim3(256), 0, s, kind, decisions, m, out_pitch,
                           states, (unsigned long long *)nullptr, (size_t)0);
    SDRX_HIP(hipGetLastError());
    note_kernel("psk_ambiguity_kernel", kind, (m <= 2048 || nchan >= 256) ? 64 : 256);
    return SDRX_OK;
}

int launch_psk_ambiguity_hits(int kind, uint8_t *decisions, size_t m, size_t nchan, size_t out_pitch,
                              sdrx_psk_state *states, unsigned long long *hits, size_t max_hits, hipStream_t s)
{
    if (!decisions || !states) return SDRX_EINVAL;
    if (nchan == 0) return SDRX_OK;
    if (m <= 2048)
        hipLaunchKernelGGL(psk_ambiguity_kernel<64>, dim3((unsigned)nchan), dim3(64), 0, s, kind, decisions, m, out_pitch,
                           states, hits, max_hits);
    else
        hipLaunchKernelGGL(psk_ambiguity_kernel<256>, dim3((unsigned)nchan), dim3(256), 0, s, kind, decisions, m, out_pitch,
                           states, hits, max_hits);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

// --------------------------------------------------------------------------------- modulator
//
// out[i] (+)= sum_j up[i-j] * RRC[128-j], j ascending, up = symbols zero-stuffed by 8: only
// j == i (mod 8) meets a symbol, every other product is an exact zero and is skipped.
// (not const: a const namespace-scope variable has internal linkage and is folded away, and then the runtime has
// no symbol for hipGetSymbolAddress to find)
__device__ double d_rrc[129] = {SDRX_RRC_VALUES};

// The matched filter multiplies x[i-k] by RRC[128-k] (BPSK.cpp:283) and the table is symmetric bit for bit (checked
// here at compile time), so the FIR kernel's tap array h[k] IS d_rrc: no reversed copy, nothing to allocate.
constexpr double rrc_ct[129] = {SDRX_RRC_VALUES};
constexpr bool rrc_symmetric()
{
    for (int k = 0; k < 129; k++)
        if (rrc_ct[k] != rrc_ct[128 - k]) return false;
    return true;
}
static_assert(rrc_symmetric(), "RRC_COEFF must equal its own reverse for d_rrc to serve as the matched filter's taps");

const double *psk_mf_taps_dev()
{
    static std::mutex mu;
    static const double *ptr[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!ptr[dev]) { // each GPU holds its own copy of the table at its own address
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(d_rrc)) != hipSuccess) return nullptr;
        ptr[dev] = (const double *)p;
    }
    return ptr[dev];
}


__global__ void __launch_bounds__(256) psk_modulate_kernel(int kind, const uint8_t *__restrict__ sym, size_t len,
                                                           double *__restrict__ out_re,
                                                           double *__restrict__ out_im, size_t zero_prefix)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len * 8) return;
    double ar = (i < zero_prefix) ? 0.0 : out_re[i]; // memset(out, 0, output_length) clears BYTES: BPSK.cpp:131
    double ai = 0.0;
    if (kind == SDRX_QPSK) ai = (i < zero_prefix) ? 0.0 : out_im[i];
    for (size_t j = i & 7; j <= 128 && j <= i; j += 8) { // BPSK.cpp:137-143
        const size_t s = (i - j) >> 3;
        if (s >= len) continue;
        const unsigned v = sym[s];
        const double h = d_rrc[128 - j];
        if (kind == SDRX_BPSK) {
            ar += ((v & 1) ? -1.0 : 1.0) * h; // LUT_REAL = {1,-1}, BPSK.cpp:14
        } else {
            ar += ((v & 1) ? -1.0 : 1.0) * h; // LUT_REAL = {1,-1,1,-1}, QPSK.cpp:14
            ai += ((v & 2) ? -1.0 : 1.0) * h; // LUT_IMAG = {1,1,-1,-1}, QPSK.cpp:15
        }
    }
    out_re[i] = ar;
    if (kind == SDRX_QPSK) out_im[i] = ai;
}

int launch_psk_modulate(int kind, const uint8_t *sym, size_t len, double *out_re, double *out_im, size_t zero_prefix,
                        hipStream_t s)
{
    if (!sym || !out_re || (kind == SDRX_QPSK && !out_im) || (kind != SDRX_BPSK && kind != SDRX_QPSK)) {
        set_error("psk_modulate: bad argument");
        return SDRX_EINVAL;
    }
    if (len == 0) return SDRX_OK;
    size_t blocks = (len * 8 + 255) / 256;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(psk_modulate_kernel, dim3((unsigned)blocks), dim3(256), 0, s, kind, sym, len, out_re, out_im,
                       zero_prefix);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
