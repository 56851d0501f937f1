// sdrx/AM.h -- drop-in for RadioNode/src/libraries/streams/AM.h:19-81.
#ifndef RADIONODE_AM_H
#define RADIONODE_AM_H

#include "Resample.h"            // AM.h:8
#include "SafeBufferPoolQueue.h" // AM.h:7
#include "common.h"

template <class data_type> class AM {
    static_assert(sizeof(data_type) == sizeof(double), "the GPU path is fp64 (RADIO_DATA_TYPE)");

  public:
    static constexpr double THRESHOLD = 80; // AM.h:23
    AM() {}
    ~AM() {}
    void modulate(uint8_t *data_in, size_t length, data_type *modulated_real, data_type * /*modulated_imag*/,
                  size_t length_modulated) // AM.h:41-54
    {
        sdrx_am_modulate(data_in, length, modulated_real, length_modulated);
    }
    size_t demodulate(data_type *real, data_type *imag, uint8_t *output, size_t number_of_points, size_t downsample) // AM.h:63-77
    {
        return sdrx_am_demodulate(real, imag, output, number_of_points, downsample);
    }
};

#endif
