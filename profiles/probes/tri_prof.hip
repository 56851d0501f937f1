// Probe: where the three waves of psk_loops_tri_kernel spend their time (work of each wave's block against waiting at the
// block barrier), from s_memtime stamps compiled in by SDRX_TRI_PROFILE.  The kernel is the library's own source, launched
// directly on noise; its printf reports ticks of the s_memtime counter, and main() times the launch so that ticks convert.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-comment -DSDRX_TRI_PROFILE profiles/probes/tri_prof.hip -o build/tri_prof
// usage: tri_prof [log2 channels = 10] [log2 samples per channel after the matched filter = 14] [kind: 0 BPSK, 1 QPSK]
#include "../../sdr_amd/csrc/psk_kernels.hip"
#include <chrono>
#include <vector>
namespace sdrx {
void set_error(const char *, ...) {}
Tuning &tuning() { static Tuning t; return t; }
thread_local LastKernel t_last_kernel;
}
int main(int argc, char **argv)
{
    const int lc = argc > 1 ? atoi(argv[1]) : 10, lm = argc > 2 ? atoi(argv[2]) : 14, kind = argc > 3 ? atoi(argv[3]) : 1;
    const size_t nchan = (size_t)1 << lc, m = (size_t)1 << lm;
    std::vector<double> h(nchan * m * 2);
    unsigned long long z = 88172645463325252ULL;
    for (auto &v : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = ((double)(z >> 11) / 9007199254740992.0 - 0.5) * 1.2; }
    double *mf; sdrx_psk_state *st; uint8_t *dec;
    if (hipMalloc(&mf, h.size() * 8) != hipSuccess || hipMalloc(&st, nchan * sizeof(sdrx_psk_state)) != hipSuccess ||
        hipMalloc(&dec, nchan * (m / 2 + 1)) != hipSuccess) return 1;
    (void)hipMemcpy(mf, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemset(st, 0, nchan * sizeof(sdrx_psk_state));
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        if (sdrx::launch_psk_loops(kind, mf, m, nchan, m, st, dec, m / 2 + 1, 1, 1, nullptr, nullptr, 0)) return 2;
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        printf("launch %d: %.3f ms (%zu channels x %zu samples, %s)\n", rep, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3,
               nchan, m, kind ? "QPSK" : "BPSK");
    }
    return 0;
}
