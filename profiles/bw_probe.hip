// hipcc -O3 --offload-arch=gfx950 -o bw_probe profiles/bw_probe.hip && ./bw_probe   (results: DESIGN.md section 4)
// read-bandwidth probe: what the HBM gives a streaming read kernel on this box, by access shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each wave reads TILE consecutive 1 KB rows (64 lanes x 16 B); MAP: 0 = block b -> tile b, 1 = XCD-contiguous
__global__ void fill(unsigned long long *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        p[i] = (z & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    }
}

template <int LOADS, int MAP, int NT>
__global__ void rd(const v2d *__restrict__ x, size_t n_v, double *sink, unsigned wgs_per_xcd)
{
    const unsigned waves = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned b = blockIdx.x;
    if (MAP) b = (b & 7) * wgs_per_xcd + (b >> 3);
    const size_t tile = (size_t)b * waves + w;
    const v2d *p = x + tile * (size_t)(64 * LOADS) + lane;
    v2d acc = {0, 0};
#pragma unroll
    for (int i = 0; i < LOADS; i++) {
        v2d v = NT ? __builtin_nontemporal_load(p + 64 * i) : p[64 * i];
        acc += v;
    }
    if (acc.x == 1.2345e300) sink[0] = acc.y;
}

template <int LOADS, int MAP, int NT> void run(const char *name, const v2d *x, size_t bytes, double *sink, int wg)
{
    const size_t per_wg = (size_t)(wg / 64) * 64 * LOADS * 16;
    unsigned blocks = (unsigned)(bytes / per_wg);
    blocks &= ~7u;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 60; i++) rd<LOADS, MAP, NT><<<blocks, wg>>>(x, 0, sink, blocks / 8);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 100; i++) rd<LOADS, MAP, NT><<<blocks, wg>>>(x, 0, sink, blocks / 8);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 100;
    printf("%-44s wg %4d  %.3f ms  %.0f GB/s\n", name, wg, ms, (double)blocks * per_wg / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const size_t bytes = 4ull << 30;
    v2d *x;
    double *sink;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(x, 0, bytes));
    fill<<<65536, 256>>>((unsigned long long *)x, bytes / 8);
    CK(hipDeviceSynchronize());
    for (int wg : {128}) {
        run<10, 0, 0>("10 loads/lane, linear", x, bytes, sink, wg);
        run<10, 1, 0>("10 loads/lane, XCD-contiguous", x, bytes, sink, wg);
        run<10, 1, 1>("10 loads/lane, XCD-contiguous, nontemporal", x, bytes, sink, wg);
    }
    run<4, 1, 0>("4 loads/lane, XCD-contiguous", x, bytes, sink, 256);
    run<16, 1, 0>("16 loads/lane, XCD-contiguous", x, bytes, sink, 128);
    run<16, 1, 1>("16 loads/lane, XCD-contiguous, nontemporal", x, bytes, sink, 128);
    run<32, 1, 0>("32 loads/lane, XCD-contiguous", x, bytes, sink, 128);
    run<32, 0, 0>("32 loads/lane, linear", x, bytes, sink, 128);
    return 0;
}
