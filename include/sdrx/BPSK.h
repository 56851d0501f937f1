// sdrx/BPSK.h -- drop-in for RadioNode/src/libraries/streams/BPSK.h:20-309.
// modulate / demodulate / decode_pulseShape / symbol_timing_synchronization /
// phase_shift_correction / phase_ambiguity_correction run on the MI355X through libsdrx.
#ifndef RADIONODE_BPSK_H
#define RADIONODE_BPSK_H

#include "Resample.h"
#include "psk_common.h"

class BPSK : public sdrx_psk_dropin {
  public:
    static const RADIO_DATA_TYPE LUT_REAL[2];      // BPSK.h:23, BPSK.cpp:14
    static const RADIO_DATA_TYPE LUT_180[2];       // BPSK.h:24, BPSK.cpp:15
    static const RADIO_DATA_TYPE (&RRC_COEFF)[129]; // BPSK.h:25, BPSK.cpp:18-50: the library's one host table

    enum PhaseAmbiguity { Ambiguity_0 = 0, Ambiguity_180 = 1 };

    BPSK(uint64_t UniqueWord = 0x4E, size_t UniqueWordLength = 8) // BPSK.cpp:52-90
        : sdrx_psk_dropin(SDRX_BPSK, UniqueWord, nullptr, UniqueWordLength), m_RealData(nullptr), m_length(0),
          m_phase_ambiguity_cc(10, 5, 5)
    {
        for (int i = 0; i < 64; i++) { uw_0[i] = m_st.uw[0][i]; uw_180[i] = m_st.uw[1][i]; }
    }
    ~BPSK() {}

    void modulate(uint8_t *data_in, size_t length, RADIO_DATA_TYPE *modulated_real, size_t length_modulated) // BPSK.cpp:96-108
    {
        (void)sdrx_psk_modulate(SDRX_BPSK, data_in, length, modulated_real, nullptr, length_modulated);
    }
    void demodulate(BBP_Block *input, BBP_Block *output) // BPSK.cpp:148-176
    {
        std::vector<double> iq, out(input->number_of_points() / 8 + 2);
        std::vector<uint64_t> hits(4096);
        block_to_iq(input, iq);
        sync_in();
        const size_t m = sdrx_psk_demodulate(m_h, iq.data(), input->number_of_points(), out.data(), hits.data(), hits.size());
        sync_out();
        print_hits(hits);
        output->reset();
        for (size_t i = 0; i < m; i++) {
            output->points[i].real(out[i]);
            output->points[i].imag(0);
            output->n_points++;
        }
    }
    // BPSK.cpp:178-211: the overload the receiver is written to call (rtlsdr_receiver.cpp:2096)
    size_t demodulate(Complex_Array &input, uint8_t *output, size_t number_of_points, size_t downsample,
                      RADIO_DATA_TYPE *real_plot_data, RADIO_DATA_TYPE *imag_plot_data)
    {
        const size_t points = decimateComplexArray(input, number_of_points, downsample);
        std::vector<uint64_t> hits(4096);
        sync_in();
        const size_t m = sdrx_psk_demodulate_plot(m_h, reinterpret_cast<const double *>(&input[0]), points, output,
                                                  real_plot_data, imag_plot_data, hits.data(), hits.size());
        sync_out();
        print_hits(hits);
        return m;
    }
    size_t decode_pulseShape(BBP_Block *input, RADIO_DATA_TYPE *real, RADIO_DATA_TYPE *imag) // BPSK.cpp:257-300
    {
        std::vector<double> iq;
        block_to_iq(input, iq);
        return sdrx_psk_decode_pulse_shape(m_h, iq.data(), input->number_of_points(), real, imag);
    }
    size_t decode_pulseShape(Complex_Array &input, size_t number_of_points, RADIO_DATA_TYPE *real, RADIO_DATA_TYPE *imag) // BPSK.cpp:213-253
    {
        return sdrx_psk_decode_pulse_shape(m_h, reinterpret_cast<const double *>(&input[0]), number_of_points, real, imag);
    }
    void symbol_timing_synchronization(RADIO_DATA_TYPE *real, RADIO_DATA_TYPE *imag, size_t num_of_points) // BPSK.cpp:302-317
    {
        sync_in();
        sdrx_psk_timing(m_h, real, imag, num_of_points);
        sync_out();
    }
    size_t phase_shift_correction(RADIO_DATA_TYPE *real, RADIO_DATA_TYPE *imag, size_t num_of_points) // BPSK.cpp:393-414
    {
        sync_in();
        const size_t m = sdrx_psk_phase_correction(m_h, real, imag, num_of_points);
        sync_out();
        return m;
    }
    void phase_ambiguity_correction(RADIO_DATA_TYPE *real, RADIO_DATA_TYPE * /*imag*/, size_t num_of_points) // BPSK.cpp:446-487
    {
        std::vector<uint64_t> hits(4096);
        sync_in();
        (void)sdrx_psk_ambiguity(m_h, real, num_of_points, hits.data(), hits.size());
        sync_out();
        print_hits(hits);
    }
    RADIO_DATA_TYPE PhaseShiftCorrection_OutputLookup(RADIO_DATA_TYPE real, RADIO_DATA_TYPE /*imag*/) { return real < 0 ? 1 : 0; } // BPSK.cpp:434-444
    void encode_LUT(uint8_t *data_in, size_t length) // BPSK.cpp:112-120 (table lookup into the caller-owned m_RealData)
    {
        m_length = length;
        for (size_t i = 0; i < m_length; i++) m_RealData[i] = (data_in[i] & 1) ? -1.0 : 1.0;
    }
    void encode_pulseShape(RADIO_DATA_TYPE *output_data, size_t output_length) // BPSK.cpp:123-146
    {
        std::vector<uint8_t> sym(m_length ? m_length : 1);
        for (size_t i = 0; i < m_length; i++) sym[i] = m_RealData[i] < 0 ? 1 : 0;
        (void)sdrx_psk_modulate(SDRX_BPSK, sym.data(), m_length, output_data, nullptr, output_length);
    }

    RADIO_DATA_TYPE *m_RealData;
    size_t m_length;
    confidence_counter m_phase_ambiguity_cc;
    RADIO_DATA_TYPE uw_0[64], uw_180[64];

  private:
    void sync_in()
    {
        const int32_t cc[4] = {m_phase_ambiguity_cc.value(), 5, 5, 5};
        push(cc);
    }
    void sync_out()
    {
        int32_t cc[4];
        pull(cc);
        m_phase_ambiguity_cc.set(cc[0]);
    }
    void print_hits(const std::vector<uint64_t> &hits) // the "Phase 0" / "Phase 180" prints of BPSK.cpp:469,474
    {
        for (int i = 0; i < m_st.n_hits && i < (int)hits.size(); i++)
            std::cout << ((hits[i] & 7) == 0 ? "Phase 0" : "Phase 180") << std::endl;
    }
};

inline const RADIO_DATA_TYPE BPSK::LUT_REAL[2] = {1, -1};
inline const RADIO_DATA_TYPE BPSK::LUT_180[2] = {1, 0};
inline const RADIO_DATA_TYPE (&BPSK::RRC_COEFF)[129] = *reinterpret_cast<const RADIO_DATA_TYPE (*)[129]>(sdrx_rrc_coeff());

#endif
