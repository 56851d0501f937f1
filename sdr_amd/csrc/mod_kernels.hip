// mod_kernels.hip -- the transmit-side copies and the stand-alone FSK convolution for gfx950.
//
//   am_modulate_kernel  : AM<T>::modulate (streams/AM.h:41-54) = raw_upsample (Resample.h:15-27): every bit repeated
//                         floor(len_mod / len) times, the rest of the output left as it was
//   fm_modulate_kernel  : FM<T>::modulate (streams/FM.h:44-74): per bit, the mark or the space table of spb samples
//   fm_convolve_kernel  : FM<T>::convolve (streams/FM.h:83-108, with the 7th parameter `ds` of the repair SURVEY 8c
//                         describes): out[k] += sum_j x[(k+1)ds-1-j] * w[len2-1-j], j ascending, zero outside x
// All three are HBM-bound copies (8 bytes written per sample, 1/rep or 1/spb bytes read); nchan independent bit
// streams per launch.  The cosine tables are host arithmetic (libm), built exactly as the reference builds them.
#include "sdrx_internal.h"

namespace sdrx {

// Each lane writes two consecutive doubles (16 bytes): a wave stores 1 KB per instruction.
__global__ void __launch_bounds__(256) am_modulate_kernel(const uint8_t *__restrict__ bits, size_t in_pitch,
                                                          double *__restrict__ out, size_t out_pitch, size_t n_write,
                                                          unsigned rep, unsigned magic)
{
    const size_t chan = blockIdx.y;
    const uint8_t *b = bits + chan * in_pitch;
    double *o = out + chan * out_pitch;
    const size_t o0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (o0 >= n_write) return;
    // o / rep: a 32-bit magic multiply where it is exact (o < 2^20, rep < 2^11), else the division
    auto bit_of = [&](size_t i) -> double {
        const size_t q = (magic && i < (1u << 20)) ? (size_t)__umulhi((unsigned)i, magic) : i / rep;
        return (double)b[q];
    };
    const double v0 = bit_of(o0);
    if (o0 + 1 < n_write && (((size_t)(o + o0)) & 15) == 0) {
        typedef double v2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(v2{v0, bit_of(o0 + 1)}, (v2 *)(o + o0));
    } else {
        o[o0] = v0;
        if (o0 + 1 < n_write) o[o0 + 1] = bit_of(o0 + 1);
    }
}

int launch_am_modulate(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, double *out, size_t len_mod,
                       size_t out_pitch, hipStream_t s)
{
    if (!bits || !out) {
        set_error("am_modulate: null pointer");
        return SDRX_EINVAL;
    }
    if (len == 0 || nchan == 0) return SDRX_OK;
    const size_t rep = len_mod / len; // floor(len_mod / len): the inner loop's bound, Resample.h:20
    if (rep == 0) return SDRX_OK;     // nothing is written (and the outer loop never advances out_index)
    if (rep > 0xffffffffULL || nchan > 65535) return SDRX_EUNSUPPORTED;
    const size_t n_write = len * rep < len_mod ? len * rep : len_mod;
    const size_t blocks = (n_write + 511) / 512;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    const unsigned magic = rep < 2048 ? magic_u32((unsigned)rep) : 0;
    hipLaunchKernelGGL(am_modulate_kernel, dim3((unsigned)blocks, (unsigned)nchan), dim3(256), 0, s, bits, in_pitch, out,
                       out_pitch, n_write, (unsigned)rep, magic);
    SDRX_HIP(hipGetLastError());
    note_kernel("am_modulate_kernel");
    return SDRX_OK;
}

// One workgroup per run of `bpw` bits; both tables (up to FM_MAX_SPB_LDS doubles each) and the run's bits are staged in
// LDS once, then the run's bpw * spb outputs are written as one flat stream: a lane writes two consecutive doubles per
// step whatever spb is (a wave stores 1 KB per instruction; with one lane per table sample, spb = 16 used a quarter of
// every wave and reached 0.38 of the HBM peak).
constexpr unsigned FM_MAX_SPB_LDS = 8192; // 2 x 64 KB of LDS; longer bits read the tables from global memory (L2)
constexpr unsigned FM_MAX_BPW = 1024;

__global__ void __launch_bounds__(256) fm_modulate_kernel(const uint8_t *__restrict__ bits, size_t len, size_t in_pitch,
                                                          const double *__restrict__ w_mark,
                                                          const double *__restrict__ w_space, unsigned spb, unsigned bpw,
                                                          double *__restrict__ out, size_t out_pitch, int in_lds)
{
    extern __shared__ __align__(16) double tab[]; // [0, spb) mark, [spb, 2 spb) space (when in_lds)
    __shared__ uint8_t sel[FM_MAX_BPW];           // 1: space (FM.h:62: `if (data_in[i] == 0)` selects the space wave)
    const size_t chan = blockIdx.y;
    const size_t i0 = (size_t)blockIdx.x * bpw;
    const unsigned nb = (unsigned)((len - i0) < (size_t)bpw ? (len - i0) : (size_t)bpw);
    const uint8_t *b = bits + chan * in_pitch + i0;
    double *o = out + chan * out_pitch + i0 * (size_t)spb;
    for (unsigned k = threadIdx.x; k < nb; k += blockDim.x) sel[k] = b[k] == 0 ? 1 : 0;
    if (in_lds) {
        for (unsigned k = threadIdx.x; k < spb; k += blockDim.x) {
            tab[k] = w_mark[k];
            tab[spb + k] = w_space[k];
        }
    }
    __syncthreads();
    const unsigned long long total = (unsigned long long)nb * spb;
    auto value = [&](unsigned long long p) -> double {
        const unsigned i = (unsigned)(p / spb), k = (unsigned)(p - (unsigned long long)i * spb);
        if (in_lds) return tab[(sel[i] ? spb : 0u) + k];
        return (sel[i] ? w_space : w_mark)[k];
    };
    typedef double v2 __attribute__((ext_vector_type(2)));
    const unsigned head = ((((size_t)o) & 15) == 0) ? 0u : 1u; // a run that starts 8 bytes off a 16-byte boundary:
    if (head && threadIdx.x == 0) o[0] = value(0);             // one double first, then pairs
    if (total < (1ull << 32)) {
        // (bit, sample) of a lane's pair advance by the same step every turn: one division per lane, none per pair
        const unsigned tot = (unsigned)total, step = 2u * blockDim.x, qstep = step / spb, rstep = step - qstep * spb;
        unsigned p = head + 2u * threadIdx.x;
        unsigned i = p / spb, k = p - i * spb;
        auto at = [&](unsigned bi, unsigned bk) -> double {
            if (in_lds) return tab[(sel[bi] ? spb : 0u) + bk];
            return (sel[bi] ? w_space : w_mark)[bk];
        };
#pragma unroll 4
        for (; p < tot; p += step) {
            const double v0 = at(i, k);
            if (p + 1 < tot) {
                const bool wrap = k + 1 == spb;
                __builtin_nontemporal_store(v2{v0, at(wrap ? i + 1 : i, wrap ? 0u : k + 1)}, (v2 *)(o + p));
            } else {
                o[p] = v0;
            }
            i += qstep;
            k += rstep;
            if (k >= spb) { k -= spb; i++; }
        }
    } else {
        for (unsigned long long p = head + 2ull * threadIdx.x; p < total; p += 2ull * blockDim.x) {
            const double v0 = value(p);
            if (p + 1 < total) __builtin_nontemporal_store(v2{v0, value(p + 1)}, (v2 *)(o + p));
            else o[p] = v0;
        }
    }
}

int launch_fm_modulate(const uint8_t *bits, size_t len, size_t nchan, size_t in_pitch, const double *w_mark,
                       const double *w_space, size_t spb, double *out, size_t out_pitch, hipStream_t s)
{
    if (!bits || !out || !w_mark || !w_space) {
        set_error("fm_modulate: null pointer");
        return SDRX_EINVAL;
    }
    if (len == 0 || nchan == 0 || spb == 0) return SDRX_OK;
    if (spb > 0x7fffffffULL || nchan > 65535) return SDRX_EUNSUPPORTED;
    // bits per workgroup: at least 64, and enough that a workgroup writes >= 4096 samples (32 KB)
    size_t bpw = (4096 + spb - 1) / spb;
    bpw = bpw < 64 ? 64 : (bpw > FM_MAX_BPW ? FM_MAX_BPW : bpw);
    const size_t blocks = (len + bpw - 1) / bpw;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    const int in_lds = spb <= FM_MAX_SPB_LDS;
    const size_t lds = in_lds ? 2 * spb * sizeof(double) : 0;
    if (lds > 48 * 1024)
        SDRX_HIP(hipFuncSetAttribute((const void *)fm_modulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fm_modulate_kernel, dim3((unsigned)blocks, (unsigned)nchan), dim3(256), lds, s, bits, len, in_pitch,
                       w_mark, w_space, (unsigned)spb, (unsigned)bpw, out, out_pitch, in_lds);
    SDRX_HIP(hipGetLastError());
    note_kernel("fm_modulate_kernel");
    return SDRX_OK;
}

// One lane per output: the sum is a serial chain in the reference's order (it starts from the output's prior value),
// so parallelism is across outputs only.  Lanes of a wave read windows ds apart; the public helper is not the
// demodulator's path (that is fsk_corr_wave_kernel, which stages windows through LDS).
__global__ void __launch_bounds__(256) fm_convolve_kernel(const double *__restrict__ x, size_t len1,
                                                          const double *__restrict__ w, size_t len2,
                                                          double *__restrict__ out, size_t n_out, size_t ds)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const size_t i = ds - 1 + k * ds; // FM.h:94: for (i = ds-1; i < length1; i += ds)
    double acc = out[k];
    for (size_t j = 0; j < len2; j++) {
        // real_data[i + length2 - j]: input1[i - j] for 0 <= i - j < length1, the zero padding otherwise
        const double xv = (j <= i && i - j < len1) ? x[i - j] : 0.0;
        acc += xv * w[len2 - j - 1];
    }
    out[k] = acc;
}

int launch_fm_convolve(const double *x, size_t len1, const double *w, size_t len2, double *out, size_t ds, hipStream_t s)
{
    if (!x || !w || !out) {
        set_error("fm_convolve: null pointer");
        return SDRX_EINVAL;
    }
    if (ds == 0) return SDRX_EINVAL;
    if (len1 < ds) return SDRX_OK;
    const size_t n_out = (len1 - (ds - 1) + ds - 1) / ds; // outputs i = ds-1, 2ds-1, ... < len1
    const size_t blocks = (n_out + 255) / 256;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(fm_convolve_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, len1, w, len2, out, n_out, ds);
    SDRX_HIP(hipGetLastError());
    note_kernel("fm_convolve_kernel");
    return SDRX_OK;
}

} // namespace sdrx
