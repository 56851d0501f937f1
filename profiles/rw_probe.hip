// hipcc -O3 --offload-arch=gfx950 -o rw_probe profiles/rw_probe.hip && ./rw_probe   (results: profiles/r02_rw_probe.json)
// read + write probe: what the HBM gives a kernel that reads every 16-byte element of a buffer once and writes it back
// in place (the traffic shape of the recurrences, normalize's scale pass and the fft: 1 byte written per byte read),
// by access kind (ordinary / nontemporal), elements in flight per lane and grid shape (one pass or persistent).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PER, int NT>
__global__ __launch_bounds__(256) void rw(v2d *x, size_t n_v)
{
    const size_t tiles = n_v / (256 * PER);
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        v2d *p = x + t * (256 * PER) + threadIdx.x;
        v2d v[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = NT ? __builtin_nontemporal_load(p + 256 * i) : p[256 * i];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            v[i] = v[i] * 1.0000001;
            if (NT) __builtin_nontemporal_store(v[i], p + 256 * i);
            else p[256 * i] = v[i];
        }
    }
}

template <int PER, int NT> void run(v2d *x, size_t n_v, unsigned grid, const char *shape)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t tiles = n_v / (256 * PER);
    const unsigned g = grid ? grid : (unsigned)tiles;
    for (int i = 0; i < 5; i++) rw<PER, NT><<<g, 256>>>(x, n_v);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; i++) rw<PER, NT><<<g, 256>>>(x, n_v);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 20;
    printf("{\"per_lane\": %d, \"nontemporal\": %d, \"grid\": \"%s\", \"ms\": %.4f, \"TBps_read_plus_write\": %.3f, \"frac_of_8TBps\": %.3f},\n",
           PER, NT, shape, ms, 32.0 * n_v / ms / 1e9, 32.0 * n_v / ms / 1e9 / 8.0);
}

int main()
{
    const size_t n_v = (size_t)1 << 26; // 1 GiB of 16-byte elements, as the kernel-rate table's shapes
    v2d *x;
    CK(hipMalloc(&x, n_v * sizeof(v2d)));
    CK(hipMemset(x, 0, n_v * sizeof(v2d)));
    printf("[\n");
    run<1, 0>(x, n_v, 0, "one tile per workgroup");
    run<1, 1>(x, n_v, 0, "one tile per workgroup");
    run<4, 0>(x, n_v, 0, "one tile per workgroup");
    run<4, 1>(x, n_v, 0, "one tile per workgroup");
    run<8, 0>(x, n_v, 0, "one tile per workgroup");
    run<8, 1>(x, n_v, 0, "one tile per workgroup");
    run<4, 0>(x, n_v, 2048, "persistent, 2048 workgroups");
    run<4, 1>(x, n_v, 2048, "persistent, 2048 workgroups");
    run<8, 0>(x, n_v, 768, "persistent, 768 workgroups");
    run<8, 1>(x, n_v, 768, "persistent, 768 workgroups");
    run<8, 1>(x, n_v, 2048, "persistent, 2048 workgroups");
    printf("{}]\n");
    return 0;
}
