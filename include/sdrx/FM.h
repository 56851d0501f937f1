// sdrx/FM.h -- drop-in for RadioNode/src/libraries/streams/FM.h:16-157 (with the two-line repair
// that makes the reference header compile: convolve's missing `ds` parameter and output[j]).
#ifndef RADIONODE_FM_H
#define RADIONODE_FM_H

#include "Resample.h" // FM.h:7
#include "common.h"
#include "filt.h"     // FM.h:9

template <class data_type> class FM {
    static_assert(sizeof(data_type) == sizeof(double), "the GPU path is fp64 (RADIO_DATA_TYPE)");

  public:
    FM(double tx_sample_rate, double rx_sample_rate, double MarkFreq = 19000, double SpaceFreq = 20000)
        : m_tx_sample_rate(tx_sample_rate), m_rx_sample_rate(rx_sample_rate), MARK_FREQ(MarkFreq), SPACE_FREQ(SpaceFreq) {}
    ~FM() {}
    void modulate(uint8_t *data_in, size_t length, data_type *modulated_real, data_type * /*modulated_imag*/,
                  size_t /*length_modulated*/, size_t samples_per_bit) // FM.h:44-74
    {
        sdrx_fm_modulate(m_tx_sample_rate, MARK_FREQ, SPACE_FREQ, data_in, length, modulated_real, samples_per_bit);
    }
    // FM.h:83-108.  The reference's declaration has six parameters and uses an undeclared `ds` (it does not compile);
    // this is the repaired signature SURVEY 8c describes, the one its demodulate() calls with seven arguments.
    void convolve(const data_type *input1, size_t length1, const data_type *input2, size_t length2, data_type *output,
                  size_t lengthOutput, size_t ds)
    {
        sdrx_fm_convolve(input1, length1, input2, length2, output, lengthOutput, ds);
    }
    size_t demodulate(data_type *real, data_type * /*imag*/, uint8_t *output, size_t number_of_points, size_t ds) // FM.h:112-148
    {
        return sdrx_fm_demodulate(m_rx_sample_rate, MARK_FREQ, SPACE_FREQ, real, output, number_of_points, ds);
    }

  private:
    double m_tx_sample_rate, m_rx_sample_rate, MARK_FREQ, SPACE_FREQ;
};

#endif
