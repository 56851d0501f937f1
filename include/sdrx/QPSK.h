// sdrx/QPSK.h -- drop-in for RadioNode/src/libraries/streams/QPSK.h:20-289.
#ifndef RADIONODE_QPSK_H
#define RADIONODE_QPSK_H

#include "Resample.h"
#include "psk_common.h"

class QPSK : public sdrx_psk_dropin {
  public:
    static const RADIO_DATA_TYPE LUT_REAL[4];      // QPSK.h:23-28, QPSK.cpp:14-19
    static const RADIO_DATA_TYPE LUT_IMAG[4];
    static const RADIO_DATA_TYPE LUT_REAL___0[4];
    static const RADIO_DATA_TYPE LUT_REAL__90[4];
    static const RADIO_DATA_TYPE LUT_REAL_180[4];
    static const RADIO_DATA_TYPE LUT_REAL_270[4];
    static const RADIO_DATA_TYPE (&RRC_COEFF)[129];         // QPSK.h:29, QPSK.cpp:32-64: the library's one host table
    static const RADIO_DATA_TYPE (&DEFAULT_UNIQUE_WORD)[64]; // QPSK.h:30, QPSK.cpp:20-28

    enum PhaseAmbiguity { Ambiguity_0 = 0, Ambiguity_180 = 1 };

    QPSK(const RADIO_DATA_TYPE *UniqueWord = DEFAULT_UNIQUE_WORD, size_t UniqueWordLength = 16) // QPSK.h:43
        : sdrx_psk_dropin(SDRX_QPSK, 0, UniqueWord, UniqueWordLength), m_RealData(nullptr), m_ImagData(nullptr), m_length(0),
          m_phase_ambiguity_cc_000(10, 5, 5), m_phase_ambiguity_cc_090(10, 5, 5), m_phase_ambiguity_cc_180(10, 5, 5),
          m_phase_ambiguity_cc_270(10, 5, 5)
    {
        for (int i = 0; i < 64; i++) {
            uw_000[i] = m_st.uw[0][i]; uw_090[i] = m_st.uw[1][i]; uw_180[i] = m_st.uw[2][i]; uw_270[i] = m_st.uw[3][i];
        }
    }
    ~QPSK() {}

    void modulate(uint8_t *data_in, size_t length, RADIO_DATA_TYPE *modulated_real, RADIO_DATA_TYPE *modulated_imag,
                  size_t length_modulated) // QPSK.cpp:127-142
    {
        (void)sdrx_psk_modulate(SDRX_QPSK, data_in, length, modulated_real, modulated_imag, length_modulated);
    }
    void demodulate(BBP_Block *input, BBP_Block *output) // QPSK.cpp:190-219
    {
        std::vector<double> iq, out(input->number_of_points() / 8 + 2);
        std::vector<uint64_t> hits(4096);
        block_to_iq(input, iq);
        sync_in();
        m_length = sdrx_psk_demodulate(m_h, iq.data(), input->number_of_points(), out.data(), hits.data(), hits.size());
        sync_out();
        print_hits(hits);
        output->reset();
        for (size_t i = 0; i < m_length; i++) {
            output->points[i].real(out[i]);
            output->points[i].imag(0);
            output->n_points++;
        }
    }
    // The stage methods work on the caller-owned members m_RealData / m_ImagData / m_length, as in the reference.
    size_t decode_pulseShape(BBP_Block *input) // QPSK.cpp:222-262
    {
        std::vector<double> iq;
        block_to_iq(input, iq);
        std::vector<double> re(input->number_of_points() / 4 + 1), im(input->number_of_points() / 4 + 1);
        const size_t m = sdrx_psk_decode_pulse_shape(m_h, iq.data(), input->number_of_points(), re.data(), im.data());
        for (size_t i = 0; i < m_length && i < m; i++) { m_RealData[i] = re[i]; m_ImagData[i] = im[i]; } // memcpy of m_length: QPSK.cpp:254-255
        return m_length;
    }
    void symbol_timing_synchronization() // QPSK.cpp:264-279
    {
        sync_in();
        sdrx_psk_timing(m_h, m_RealData, m_ImagData, m_length);
        sync_out();
    }
    void phase_shift_correction() // QPSK.cpp:355-376
    {
        sync_in();
        m_length = sdrx_psk_phase_correction(m_h, m_RealData, m_ImagData, m_length);
        sync_out();
    }
    void phase_ambiguity_correction() // QPSK.cpp:416-509
    {
        std::vector<uint64_t> hits(4096);
        sync_in();
        (void)sdrx_psk_ambiguity(m_h, m_RealData, m_length, hits.data(), hits.size());
        sync_out();
        print_hits(hits);
    }
    RADIO_DATA_TYPE PhaseShiftCorrection_OutputLookup(RADIO_DATA_TYPE real, RADIO_DATA_TYPE imag) // QPSK.cpp:396-414
    {
        if (real < 0 && imag < 0) return 3;
        if (real > 0 && imag < 0) return 2;
        if (real < 0 && imag > 0) return 1;
        return 0;
    }
    void encode_LUT(uint8_t *data_in, size_t length) // QPSK.cpp:144-153
    {
        m_length = length;
        for (size_t i = 0; i < m_length; i++) {
            m_RealData[i] = (data_in[i] & 1) ? -1.0 : 1.0;
            m_ImagData[i] = (data_in[i] & 2) ? -1.0 : 1.0;
        }
    }
    void encode_pulseShape(RADIO_DATA_TYPE *real_data, RADIO_DATA_TYPE *imag_data, size_t output_length) // QPSK.cpp:156-188
    {
        std::vector<uint8_t> sym(m_length ? m_length : 1);
        for (size_t i = 0; i < m_length; i++) sym[i] = (uint8_t)((m_RealData[i] < 0 ? 1 : 0) | (m_ImagData[i] < 0 ? 2 : 0));
        (void)sdrx_psk_modulate(SDRX_QPSK, sym.data(), m_length, real_data, imag_data, output_length);
    }

    RADIO_DATA_TYPE *m_RealData, *m_ImagData;
    size_t m_length;
    confidence_counter m_phase_ambiguity_cc_000, m_phase_ambiguity_cc_090, m_phase_ambiguity_cc_180, m_phase_ambiguity_cc_270;
    RADIO_DATA_TYPE uw_000[64], uw_090[64], uw_180[64], uw_270[64];

  private:
    void sync_in()
    {
        const int32_t cc[4] = {m_phase_ambiguity_cc_000.value(), m_phase_ambiguity_cc_090.value(),
                               m_phase_ambiguity_cc_180.value(), m_phase_ambiguity_cc_270.value()};
        push(cc);
    }
    void sync_out()
    {
        int32_t cc[4];
        pull(cc);
        m_phase_ambiguity_cc_000.set(cc[0]); m_phase_ambiguity_cc_090.set(cc[1]);
        m_phase_ambiguity_cc_180.set(cc[2]); m_phase_ambiguity_cc_270.set(cc[3]);
    }
    void print_hits(const std::vector<uint64_t> &hits) // QPSK.cpp:448,456,464,472
    {
        static const char *name[4] = {"No Phase ", "90 Phase ", "180 Phase ", "270 Phase "};
        for (int i = 0; i < m_st.n_hits && i < (int)hits.size(); i++)
            std::cout << name[hits[i] & 3] << (hits[i] >> 3) << std::endl;
        if (m_phase_ambiguity_cc_090.high_confidence()) std::cout << "90 Degree Phase Detected" << std::endl; // QPSK.cpp:483
        else if (m_phase_ambiguity_cc_180.high_confidence()) std::cout << "180 Degree Phase Detected" << std::endl;
        else if (m_phase_ambiguity_cc_270.high_confidence()) std::cout << "270 Degree Phase Detected" << std::endl;
    }
};

inline const RADIO_DATA_TYPE QPSK::LUT_REAL[4] = {1, -1, 1, -1};
inline const RADIO_DATA_TYPE QPSK::LUT_IMAG[4] = {1, 1, -1, -1};
inline const RADIO_DATA_TYPE QPSK::LUT_REAL___0[4] = {0, 1, 2, 3};
inline const RADIO_DATA_TYPE QPSK::LUT_REAL__90[4] = {2, 0, 3, 1};
inline const RADIO_DATA_TYPE QPSK::LUT_REAL_180[4] = {3, 2, 1, 0};
inline const RADIO_DATA_TYPE QPSK::LUT_REAL_270[4] = {1, 3, 0, 2};
inline const RADIO_DATA_TYPE (&QPSK::RRC_COEFF)[129] = *reinterpret_cast<const RADIO_DATA_TYPE (*)[129]>(sdrx_rrc_coeff());
inline const RADIO_DATA_TYPE (&QPSK::DEFAULT_UNIQUE_WORD)[64] =
    *reinterpret_cast<const RADIO_DATA_TYPE (*)[64]>(sdrx_qpsk_default_unique_word());

#endif
