// fir_kernels_f32.hip -- the fp32 twins of the FIR kernels (sdrx_fir_job.dtype = SDRX_F32): float samples,
// taps, state and outputs, fused multiply-add always.  Same structure as the fp64 kernels
// (fir_impl.inc); checked against the fp64 CPU path to 1e-5 relative, not bit for bit.
#include "sdrx_internal.h"

#include "../../include/sdrx_math.h"
#include "fir_common.h"

#include <array>
#include <cmath>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#ifndef SDRX_STAGE_UNR
#define SDRX_STAGE_UNR 12
#endif
#ifndef SDRX_STEADY_UB
#define SDRX_STEADY_UB 4
#endif

namespace sdrx {

#define FT float
#define SDRX_FNS f32
#define SDRX_FIR_AMV 1 /* + the verified fp32 form of the fp64 FIR -> AM chain (cfg1) */
#include "fir_impl.inc"
#undef FT
#undef SDRX_FNS

} // namespace sdrx
