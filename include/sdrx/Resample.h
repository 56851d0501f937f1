// sdrx/Resample.h -- drop-in for RadioNode/src/libraries/DigitalSignalProcessing/Resample.h:12-54 and
// Resample.cpp:10-70: the six index-shuffling helpers every stream class (and the reference's TEST_FM / TEST_Resample)
// uses.  They move samples between caller-owned host arrays without arithmetic on the path (upsample's ramp aside), so
// they stay host code; inside the GPU path the same steps are folded into the kernels (the FIR kernels compute only
// the kept outputs, the shapers skip the stuffed zeros, am_modulate_kernel is raw_upsample).  The reference's quirks are
// kept where a caller can see them: decimate<T> and decimateComplexArray clear only element [dst] after the kept
// samples (Resample.h:46-49, Resample.cpp:63-67), decimate(BBP_Block*) drops the first sample and leaves
// n_points = dst + 1 (Resample.cpp:38-48).
#ifndef RADIONODE_RESAMPLE_H
#define RADIONODE_RESAMPLE_H

#include "common.h"

// Resample.cpp:10-22: a ramp of `ratio` per repeated sample, ceil(ratio) outputs per input; in_length is not consulted
inline size_t upsample(RADIO_DATA_TYPE *in_block, size_t /*in_length*/, RADIO_DATA_TYPE *out_block, size_t out_length,
                       RADIO_DATA_TYPE ratio)
{
    const RADIO_DATA_TYPE per_in = ceil(ratio);
    size_t out = 0;
    for (size_t in = 0; out < out_length; in++) {
        out_block[out++] = in_block[in];
        for (int j = 1; j < per_in && out < out_length; j++) out_block[out++] = in_block[in] + (ratio * j);
    }
    return out_length;
}

// Resample.h:15-27: every input repeated floor(ratio) times
template <class data_type>
size_t raw_upsample(data_type *in_block, size_t in_length, data_type *out_block, size_t out_length, data_type ratio)
{
    const auto reps = floor(ratio);
    size_t out = 0;
    for (size_t in = 0; out < out_length && in < in_length; in++)
        for (int j = 0; (j < reps) && (out < out_length); j++) out_block[out++] = in_block[in];
    return out_length;
}

// Resample.cpp:24-36: every input followed by ceil(ratio) - 1 zeros
inline size_t upsample_fill_w_zeros(RADIO_DATA_TYPE *in_block, size_t in_length, RADIO_DATA_TYPE *out_block,
                                    size_t out_length, RADIO_DATA_TYPE ratio)
{
    const RADIO_DATA_TYPE zeros = ceil(ratio) - 1;
    size_t out = 0;
    for (size_t in = 0; out < out_length && in < in_length; in++) {
        out_block[out++] = in_block[in];
        for (int j = 0; j < zeros && out < out_length; j++) out_block[out++] = 0;
    }
    return out_length;
}

// Resample.cpp:38-48: keeps points decimate_by, 2*decimate_by, ... (the first point is dropped), n_points = kept + 1
inline void decimate(BBP_Block *block, size_t decimate_by = 2)
{
    size_t kept = 0;
    for (size_t src = decimate_by; src < block->n_points; src += decimate_by) block->points[kept++] = block->points[src];
    block->n_points = kept + 1;
}

// Resample.h:33-52: keeps 0, by, 2*by, ...; returns their count (ceil(length / by)); clears only block[count]
template <class data_type> size_t decimate(data_type *block, size_t length, size_t decimate_by = 2)
{
    size_t kept = 0;
    for (size_t src = 0; src < length; src += decimate_by) block[kept++] = block[src];
    if (kept < length) block[kept] = 0; // the reference's tail loop writes block[dst] on every pass
    return kept;
}

// Resample.cpp:50-70: the same on a valarray of complex samples
inline size_t decimateComplexArray(Complex_Array &block, size_t length, size_t decimate_by = 2)
{
    size_t kept = 0;
    for (size_t src = 0; src < length; src += decimate_by) {
        block[kept].real(block[src].real());
        block[kept].imag(block[src].imag());
        kept++;
    }
    if (kept < length) {
        block[kept].real(0);
        block[kept].imag(0);
    }
    return kept;
}

#endif // RADIONODE_RESAMPLE_H
