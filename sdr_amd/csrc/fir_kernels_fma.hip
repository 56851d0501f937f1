// fir_kernels_fma.hip -- the fused multiply-add (fast mode 1) instances of the fp64 FIR kernels; see
// fir_kernels.hip -- time-tiled, register-blocked FIR for gfx950 with fused demod epilogues.
//
// Replaces Filter::do_sample / Filter::execute (RadioNode/third-party/FIR-filter-class/filt.cpp:248-273)
// and, fused behind it, decimate<T> (Resample.h:33-52), AM<T>::demodulate (AM.h:63-77) and the
// phase-difference discriminator (rtl_fm.c:422-435,518-546).  The same kernel is the PSK matched
// filter (BPSK.cpp:279-290: T = 129, D = 4, zero history).
//
// Parallelisation: the reference is one serial loop per stream; here every OUTPUT keeps the
// reference's summation (acc = 0; acc += x[i-k]*h[k], k ascending, one rounding per mul and per
// add) and outputs are spread over lanes, tiles and channels.  A 256-thread workgroup stages the
// tile's input span (its T-1 sample overlap-save halo comes from the previous tile or from the
// carried history) into LDS with one coalesced pass, then each lane walks its window newest-first,
// feeding R register accumulators per rail from each LDS read; the tap for a step is the same for
// every lane, so taps travel through scalar loads, not LDS.  LDS layout: lane blocks of Sp samples
// at an odd pitch Spad, which makes the per-step ds_read (b128 for I/Q, b64 for one rail)
// bank-conflict free for any decimation factor.
#include "sdrx_internal.h"

#include "../../include/sdrx_math.h"
#include "fir_common.h"

#include <array>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#ifndef SDRX_STAGE_UNR
#define SDRX_STAGE_UNR 12
#endif
#ifndef SDRX_STEADY_UB
#define SDRX_STEADY_UB 8 // measured on config 5 (T = 1024, real rail), exact / fma chain: UB = 4 16.4 / 10.9 ms, 8 15.8 / 9.5, 16 20.9 / 10.8
#endif

namespace sdrx {

#define FT double
#define SDRX_FNS f64
#define SDRX_FIR_PART 2
#include "fir_impl.inc"
#undef SDRX_FIR_PART
#undef FT
#undef SDRX_FNS

} // namespace sdrx
