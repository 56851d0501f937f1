// fir_common.h -- helpers shared by the fp64 and fp32 instantiations of fir_impl.inc
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/sdrx_math.h"

namespace sdrx {

// f(integral_constant<int, I>) for I in [LO, HI): loop bounds that depend on I stay compile-time
template <int LO, int HI, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        static_for<LO + 1, HI>(f);
    }
}

__device__ __forceinline__ double fma_t(double x, double h, double acc) { return __builtin_fma(x, h, acc); }
__device__ __forceinline__ float fma_t(float x, float h, float acc) { return __builtin_fmaf(x, h, acc); }
// fp64: include/sdrx_math.h, the definition shared with the CPU checker; fp32 twin: the device library
__device__ __forceinline__ double atan2_t(double y, double x) { return sdrx_atan2(y, x); }
__device__ __forceinline__ float atan2_t(float y, float x) { return atan2f(y, x); }

// whole-wave shift by one lane through DPP (pure VALU, no LDS); lanes shifted in from outside read 0
template <int CTRL> __device__ __forceinline__ double dpp_move(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

} // namespace sdrx
