// launch-shape probe for the two-rail streaming kernels (am_demod ds=1, ones/zeros): 2 x 2 GiB in, 1 B/sample out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, int WG, int MAP, int NT>
__global__ void __launch_bounds__(WG) am(const double *__restrict__ re, const double *__restrict__ im, uint8_t *__restrict__ out,
                                         size_t n, double thr2, unsigned wgs_per_xcd)
{
    unsigned b = blockIdx.x;
    if (MAP) b = (b & 7) * wgs_per_xcd + (b >> 3);
    const size_t pair0 = (size_t)b * (WG * K) + threadIdx.x;
    const v2d *r2 = (const v2d *)re, *i2 = (const v2d *)im;
    const size_t npair = n / 2;
    v2d a[K], c[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t p = pair0 + (size_t)k * WG;
        if (p < npair) {
            a[k] = NT ? __builtin_nontemporal_load(r2 + p) : r2[p];
            c[k] = NT ? __builtin_nontemporal_load(i2 + p) : i2[p];
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t p = pair0 + (size_t)k * WG;
        if (p < npair) {
            const double e0 = a[k].x * a[k].x + c[k].x * c[k].x, e1 = a[k].y * a[k].y + c[k].y * c[k].y;
            const unsigned short v = (unsigned short)((e0 >= thr2 ? 1u : 0u) | (e1 >= thr2 ? 0x100u : 0u));
            if (NT == 2) __builtin_nontemporal_store(v, (unsigned short *)(out + 2 * p));
            else *(unsigned short *)(out + 2 * p) = v;
        }
    }
}

template <int K, int WG, int MAP, int NT> void run(const char *name, const double *re, const double *im, uint8_t *out, size_t n)
{
    unsigned blocks = (unsigned)((n / 2 + (size_t)WG * K - 1) / ((size_t)WG * K));
    blocks = (blocks + 7) & ~7u;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; i++) am<K, WG, MAP, NT><<<blocks, WG>>>(re, im, out, n, 6400.0, blocks / 8);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 60; i++) am<K, WG, MAP, NT><<<blocks, WG>>>(re, im, out, n, 6400.0, blocks / 8);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 60;
    printf("%-52s %.3f ms  %.0f GB/s (17 B/sample)  %.3f of 8 TB/s\n", name, ms, 17.0 * n / ms / 1e6, 17.0 * n / ms / 1e6 / 8000.0);
    fflush(stdout);
}

__global__ void fill(double *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)((i * 2654435761u) % 255) - 127.0;
}

int main()
{
    const size_t n = 1ull << 28;
    double *re, *im; uint8_t *out;
    CK(hipMalloc(&re, n * 8)); CK(hipMalloc(&im, n * 8)); CK(hipMalloc(&out, n));
    fill<<<65536, 256>>>(re, n); fill<<<65536, 256>>>(im, n);
    CK(hipDeviceSynchronize());
    run<4, 256, 0, 1>("K=4 wg256 linear nt-load (shipped)", re, im, out, n);
    run<4, 256, 1, 1>("K=4 wg256 xcd-contiguous nt-load", re, im, out, n);
    run<4, 256, 0, 2>("K=4 wg256 linear nt-load nt-store", re, im, out, n);
    run<8, 256, 0, 1>("K=8 wg256 linear nt-load", re, im, out, n);
    run<8, 128, 0, 1>("K=8 wg128 linear nt-load", re, im, out, n);
    run<8, 128, 1, 1>("K=8 wg128 xcd-contiguous nt-load", re, im, out, n);
    run<8, 128, 1, 2>("K=8 wg128 xcd-contiguous nt-load nt-store", re, im, out, n);
    run<5, 128, 1, 1>("K=5 wg128 xcd-contiguous nt-load", re, im, out, n);
    run<2, 256, 0, 1>("K=2 wg256 linear nt-load", re, im, out, n);
    run<16, 64, 1, 1>("K=16 wg64 xcd-contiguous nt-load", re, im, out, n);
    run<4, 256, 0, 0>("K=4 wg256 linear plain load", re, im, out, n);
    return 0;
}
