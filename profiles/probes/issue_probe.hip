// Probe: what bounds a LONE wave of fp64 arithmetic on a gfx950 SIMD -- the dependent-result latency or a per-wave issue
// interval -- and what a second / third / fourth wave on the SAME SIMD costs.  The PSK loops (psk_kernels.hip) are one or
// two such waves per 64 channels; their step time is (instructions per step) x (what this probe measures).
//   ./issue_probe          prints cycles per instruction (s_memtime, 100 MHz constant clock converted with the measured ratio
//                          against wall time) for: a dependent chain, 2 / 4 / 8 independent chains interleaved, for
//                          v_add_f64, v_mul_f64, v_fma_f64, v_cndmask_b32, and the same with 1, 2, 4, 8 waves per workgroup
//                          of ONE workgroup (waves of a workgroup are dealt round-robin to the 4 SIMDs: 4 waves = 1 per SIMD,
//                          8 waves = 2 per SIMD, 16 waves = 4 per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__device__ int g_lanes = 64; // lanes of each wave that stay active (does the SIMD skip the passes of idle lanes?)
template <int OP, int CHAINS> __global__ void __launch_bounds__(1024) k(double *out, int iters, double seed)
{
    if ((int)(threadIdx.x & 63) >= g_lanes) return;
    double a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x * 1e-3;
    const double m = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 64 / CHAINS; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
                if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int *)&a[c]) : "v"(7) : );
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c];
    if (s == 12345.678) out[threadIdx.x] = s;
}

template <int OP, int CHAINS> double run(int waves, double *d)
{
    const int iters = 20000;
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(1), dim3(64 * waves), 0, 0, d, 100, 1.0);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(1), dim3(64 * waves), 0, 0, d, iters, 1.0);
    hipDeviceSynchronize();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return sec / ((double)iters * 64); // seconds per instruction of ONE wave
}

int main()
{
    double *d;
    hipMalloc(&d, 8192 * 8);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double ghz = clk_khz / 1e6;
    printf("clock rate attribute %.3f GHz (cycles below = seconds x this)\n", ghz);
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cndmask_b32"};
    for (int waves : {1, 4, 8, 16}) {
        printf("waves in the workgroup: %d (%d per SIMD)\n", waves, (waves + 3) / 4);
#define ROW(OP)                                                                                                         \
    printf("  %-14s chains 1: %6.2f   2: %6.2f   4: %6.2f   8: %6.2f  cycles per instruction per wave\n", names[OP],       \
           run<OP, 1>(waves, d) * ghz * 1e9, run<OP, 2>(waves, d) * ghz * 1e9, run<OP, 4>(waves, d) * ghz * 1e9,            \
           run<OP, 8>(waves, d) * ghz * 1e9);
        ROW(0) ROW(1) ROW(2) ROW(3)
    }
    for (int lanes : {32, 16, 1}) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_lanes), &lanes, sizeof(int));
        printf("one wave, %d active lanes: v_add_f64 chains 1: %6.2f  4: %6.2f   v_fma_f64 chains 1: %6.2f  4: %6.2f\n", lanes,
               run<0, 1>(1, d) * ghz * 1e9, run<0, 4>(1, d) * ghz * 1e9, run<2, 1>(1, d) * ghz * 1e9, run<2, 4>(1, d) * ghz * 1e9);
    }
    return 0;
}
