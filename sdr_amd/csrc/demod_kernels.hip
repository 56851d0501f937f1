// demod_kernels.hip -- stand-alone AM / ones-zeros / FSK demodulators for gfx950.
//
//   am_demod_kernel    : AM<T>::demodulate (AM.h:63-77) == kernel_am_demod (am.cu:72-81)
//   ones_zeros_kernel  : the receiver's ones/zeros count (rtlsdr_receiver.cpp:2063-2074), which
//                        ones_zeros.cu:28-52 was meant to compute
//   fsk_corr_kernel    : FM<T>::demodulate (FM.h:83-148 with the SURVEY section 8c repair)
#include "sdrx_internal.h"

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
    size_t blocks = (nbits + FSK_NT - 1) / FSK_NT;
    if (blocks > 0x7fffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(fsk_corr_kernel, dim3((unsigned)blocks), dim3(FSK_NT), 0, s, re, w1, w2, out, nbits,
                       (unsigned)ds);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
