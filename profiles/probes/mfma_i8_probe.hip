// Probe: v_mfma_i32_16x16x64_i8 on gfx950 -- operand / result lane layout, and whether it runs beside vector work.
//   layout: A[i][k] (16 x 64 int8), B[k][j] (64 x 16 int8), D[i][j] (16 x 16 int32); candidate: A lane = 16 (k / 16) + i holding
//           bytes k % 16 of its four registers, B lane = 16 (k / 16) + j likewise, D lane = 16 (i / 4) + j, register i % 4.
//   rate:   one wave per SIMD issuing only the matrix instruction; the same with a second wave per SIMD issuing only fp32
//           vector work; that vector wave alone.  If matrix and vector instructions share a pipe the combined time is the sum.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_i8_probe.hip -o mfma_i8_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout(const signed char *A, const signed char *B, int *D)
{
    const int l = threadIdx.x, g = l / 16, r = l % 16;
    v4i a, b, c = {0, 0, 0, 0};
    signed char ab[16], bb[16];
    for (int t = 0; t < 16; t++) { ab[t] = A[r * 64 + 16 * g + t]; bb[t] = B[(16 * g + t) * 16 + r]; }
    memcpy(&a, ab, 16);
    memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[l * 4 + v] = c[v];
}

// mode bit 0: waves with even index in their SIMD pair run the matrix loop; bit 1: the others run the vector loop
__global__ void __launch_bounds__(512) rate(int *out, float *fout, int iters, int mode)
{
    const int wave = threadIdx.x >> 6; // 8 waves: 0..3 -> one per SIMD, 4..7 -> a second per SIMD
    if (wave < 4) {
        if (!(mode & 1)) return;
        v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {1, 1, 2, 1}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; it++) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 12345) out[threadIdx.x] = 1;
    } else {
        if (!(mode & 2)) return;
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(1.0000001f), "v"(1e-9f));
            }
        }
        if (x0 + x1 + x2 + x3 == 12345.f) fout[threadIdx.x] = 1.f;
    }
}

static double run(int mode, int iters, int *d, float *f)
{
    hipLaunchKernelGGL(rate, dim3(1), dim3(512), 0, 0, d, f, 100, mode);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(rate, dim3(1), dim3(512), 0, 0, d, f, iters, mode);
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    signed char A[16 * 64], B[64 * 16];
    int D[256];
    srand(7);
    for (int i = 0; i < 1024; i++) { A[i] = (signed char)(rand() % 255 - 127); B[i] = (signed char)(rand() % 256 - 128); }
    signed char *dA, *dB;
    int *dD;
    float *dF;
    (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dD, 4096); (void)hipMalloc(&dF, 4096);
    (void)hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
        const int i = 4 * (l / 16) + v, j = l % 16;
        int s = 0;
        for (int k = 0; k < 64; k++) s += (int)A[i * 64 + k] * (int)B[k * 16 + j];
        ok += s == D[l * 4 + v];
    }
    printf("layout A lane=16(k/16)+i bytes k%%16, B lane=16(k/16)+j, D lane=16(i/4)+j reg i%%4: %d/256 exact\n", ok);
    const int iters = 200000;
    const double tm = run(1, iters, dD, dF), tv = run(2, iters, dD, dF), tb = run(3, iters, dD, dF);
    printf("matrix only %.3f ms (%.1f ns per instruction per wave), vector only %.3f ms (%.2f ns per v_fma_f32), both %.3f ms\n", tm * 1e3,
           tm / (iters * 4.0) * 1e9, tv * 1e3, tv / (iters * 16.0) * 1e9, tb * 1e3);
    return 0;
}
