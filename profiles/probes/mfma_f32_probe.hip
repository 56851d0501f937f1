// Probe: operand / result lane layout of v_mfma_f32_16x16x4_f32 on gfx950, and whether it is bit for bit the
// fmaf chain over its four k-steps.  Build: hipcc --offload-arch=gfx950 -O2 mfma_f32_probe.hip -o mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D)
{
    const int l = threadIdx.x;
    // A[i][kk] row-major 16x4, B[kk][j] row-major 4x16
    const float a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[l * 4 + v] = c[v];
}
int main()
{
    float A[64], B[64], D[256], *dA, *dB, *dD;
    srand(3);
    for (int i = 0; i < 64; i++) { A[i] = (float)rand() / RAND_MAX - 0.5f; B[i] = (float)rand() / RAND_MAX * 3.f - 1.5f; }
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A, 256, hipMemcpyHostToDevice); hipMemcpy(dB, B, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
    // candidate layouts
    int okA = 0, okB = 0, exactA = 0;
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
        const int j = l % 16;
        const int iA = 4 * (l / 16) + v; // f32 layout (expected)
        const int iB = 4 * v + l / 16;   // the f64 instruction's layout
        float rA = 0.f, rB = 0.f;
        for (int kk = 0; kk < 4; kk++) { rA = fmaf(A[iA * 4 + kk], B[kk * 16 + j], rA); rB = fmaf(A[iB * 4 + kk], B[kk * 16 + j], rB); }
        okA += fabsf(rA - D[l * 4 + v]) < 1e-5f; okB += fabsf(rB - D[l * 4 + v]) < 1e-5f;
        exactA += memcmp(&rA, &D[l * 4 + v], 4) == 0;
    }
    printf("layout i=4*(l/16)+v: %d/256 close, %d/256 bit-exact with the ascending fmaf chain; layout i=4*v+l/16: %d/256 close\n", okA, exactA, okB);
    return 0;
}
