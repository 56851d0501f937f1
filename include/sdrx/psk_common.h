// sdrx/psk_common.h -- what BPSK.h and QPSK.h share in the reference: the public loop structs
// (BPSK.h:130-276 == QPSK.h:104-250), the scalar per-sample helpers, and the mirroring of that
// public state to and from the device-side sdrx_psk_state around every block-level GPU call.
#ifndef SDRX_DROPIN_PSK_COMMON_H
#define SDRX_DROPIN_PSK_COMMON_H

#include <iostream>
#include <vector>

#include "common.h"

class sdrx_psk_dropin {
  public:
    struct SymbolTimingSynchronizationVars {
        RADIO_DATA_TYPE mu, I, Q, muDelay;
        SymbolTimingSynchronizationVars() : mu(0), I(0), Q(0), muDelay(0) {}
    } SymbolTimingSynchronizationVariables;
    struct InterpData {
        RADIO_DATA_TYPE delay1, delay2, delay3, delay4, delay5;
        InterpData() : delay1(0), delay2(0), delay3(0), delay4(0), delay5(0) {}
    };
    InterpData Interp_I_Channel, Interp_Q_Channel;
    struct GardnerDetectorData {
        RADIO_DATA_TYPE delay1, delay2, delay3, delay4;
        GardnerDetectorData() : delay1(1), delay2(1), delay3(1), delay4(1) {}
    } GardnerDetectorVariables;
    struct LoopFilterVars {
        RADIO_DATA_TYPE delay_x, delay_y;
        LoopFilterVars() : delay_x(0), delay_y(0) {}
    } LoopFilterVariables;
    struct NcoControlVars {
        RADIO_DATA_TYPE delay;
        NcoControlVars() : delay(0) {}
    } NcoControlVariables;
    RADIO_DATA_TYPE Register_mu, Register_I, Register_Q;
    struct PhaseCorrectionVars {
        RADIO_DATA_TYPE dds_real, dds_imag, dds_delay;
        PhaseCorrectionVars() : dds_real(1), dds_imag(0), dds_delay(0) {}
    } PhaseCorrectionVariables;
    struct PhaseLoopFilterVars {
        RADIO_DATA_TYPE delay_x, delay_y;
        PhaseLoopFilterVars() : delay_x(0), delay_y(0) {}
    } PhaseLoopFilterVariables;

    // ---- scalar per-sample helpers of the reference's public surface (BPSK.cpp:319-389, 417-444).
    // They act on one sample and on the public structs above; the block-level methods, which are
    // the hot path, never call them: those run in psk_loops_kernel on the GPU.
    RADIO_DATA_TYPE InterpolationFilter(InterpData &data, RADIO_DATA_TYPE in, RADIO_DATA_TYPE mu)
    {
        const RADIO_DATA_TYPE K = -0.5;
        RADIO_DATA_TYPE V1 = data.delay1 - K * in, V2 = V1 + data.delay2, V3 = V2 - data.delay3;
        RADIO_DATA_TYPE V4 = K * in - data.delay1 + data.delay4, V5 = V4 + data.delay2, V6 = data.delay3 + V5;
        RADIO_DATA_TYPE V7 = V3 * mu + V6, V8 = data.delay5 + mu * V7;
        data.delay3 = data.delay2; data.delay2 = data.delay1; data.delay1 = K * in;
        data.delay5 = data.delay4; data.delay4 = in;
        return V8;
    }
    RADIO_DATA_TYPE GardnerDetector(RADIO_DATA_TYPE i, RADIO_DATA_TYPE q, RADIO_DATA_TYPE strobe)
    {
        GardnerDetectorData &g = GardnerDetectorVariables;
        RADIO_DATA_TYPE r = 0;
        if (strobe >= 0.5)
            r = (g.delay3 * (-SIGN_FUNC(q) + SIGN_FUNC(g.delay4)) + (g.delay1 * (-SIGN_FUNC(i) + SIGN_FUNC(g.delay2))));
        g.delay2 = g.delay1; g.delay1 = i; g.delay4 = g.delay3; g.delay3 = q;
        return r;
    }
    RADIO_DATA_TYPE LoopFilter(RADIO_DATA_TYPE val)
    {
        const RADIO_DATA_TYPE K1 = -0.00587880145347487, K2 = -2.35152058138995e-05;
        RADIO_DATA_TYPE r = LoopFilterVariables.delay_y + K2 * val + K1 * (val - LoopFilterVariables.delay_x);
        LoopFilterVariables.delay_x = val;
        LoopFilterVariables.delay_y = r;
        return r;
    }
    RADIO_DATA_TYPE NcoControl(RADIO_DATA_TYPE input)
    {
        RADIO_DATA_TYPE r = NcoControlVariables.delay - floor(NcoControlVariables.delay);
        NcoControlVariables.delay = r - 0.5 - input;
        return r;
    }
    RADIO_DATA_TYPE StrobeEnabledRegister(RADIO_DATA_TYPE input, RADIO_DATA_TYPE strobe, RADIO_DATA_TYPE &reg)
    {
        if (strobe >= 1) reg = input;
        return reg;
    }
    RADIO_DATA_TYPE PhaseLoopFilter(RADIO_DATA_TYPE val)
    {
        const RADIO_DATA_TYPE K1 = 0.0620001240002480, K2 = 0.000992001984003968;
        RADIO_DATA_TYPE r = PhaseLoopFilterVariables.delay_y - K1 * PhaseLoopFilterVariables.delay_x + (K1 + K2) * val;
        PhaseLoopFilterVariables.delay_x = val;
        PhaseLoopFilterVariables.delay_y = r;
        return r;
    }
    void complex_multiply(RADIO_DATA_TYPE ra, RADIO_DATA_TYPE ia, RADIO_DATA_TYPE rb, RADIO_DATA_TYPE ib,
                          RADIO_DATA_TYPE &rc, RADIO_DATA_TYPE &ic)
    {
        rc = ra * rb - ia * ib;
        ic = ia * rb + ra * ib;
    }
    inline RADIO_DATA_TYPE constrainAngle(RADIO_DATA_TYPE x)
    {
        x = fmod(x, 2.0 * PI);
        if (x < 0.0) x += 2.0 * PI;
        return x;
    }

    RADIO_DATA_TYPE normalization_max;
    RADIO_DATA_TYPE m_differential_decode_delay;
    size_t UW_SIZE;

  protected:
    sdrx_psk_dropin(int kind, uint64_t bpsk_uw, const RADIO_DATA_TYPE *qpsk_uw, size_t uw_len)
        : Register_mu(0), Register_I(0), Register_Q(0), normalization_max(0), m_differential_decode_delay(1),
          UW_SIZE(uw_len), m_kind(kind), m_h(sdrx_psk_create(kind, bpsk_uw, qpsk_uw, uw_len))
    {
        sdrx_psk_get_state(m_h, &m_st);
    }
    ~sdrx_psk_dropin() { sdrx_psk_destroy(m_h); }
    sdrx_psk_dropin(const sdrx_psk_dropin &) = delete;
    sdrx_psk_dropin &operator=(const sdrx_psk_dropin &) = delete;

    // public members -> device state (tests poke the members between stage calls)
    void push(const int32_t cc[4])
    {
        sdrx_psk_state &s = m_st;
        s.nco_delay = NcoControlVariables.delay; s.reg_mu = Register_mu; s.reg_Q = Register_Q; s.reg_I = Register_I;
        s.mu_delay = SymbolTimingSynchronizationVariables.muDelay;
        const InterpData *ip[2] = {&Interp_Q_Channel, &Interp_I_Channel};
        double *dst[2] = {s.iq_d, s.ii_d};
        for (int c = 0; c < 2; c++) {
            dst[c][0] = ip[c]->delay1; dst[c][1] = ip[c]->delay2; dst[c][2] = ip[c]->delay3;
            dst[c][3] = ip[c]->delay4; dst[c][4] = ip[c]->delay5;
        }
        s.g1 = GardnerDetectorVariables.delay1; s.g2 = GardnerDetectorVariables.delay2;
        s.g3 = GardnerDetectorVariables.delay3; s.g4 = GardnerDetectorVariables.delay4;
        s.lf_x = LoopFilterVariables.delay_x; s.lf_y = LoopFilterVariables.delay_y;
        s.dds_delay = PhaseCorrectionVariables.dds_delay;
        s.plf_x = PhaseLoopFilterVariables.delay_x; s.plf_y = PhaseLoopFilterVariables.delay_y;
        for (int k = 0; k < 4; k++) s.cc[k] = cc[k];
        s.uw_size = (int32_t)UW_SIZE;
        sdrx_psk_set_state(m_h, &s);
    }
    // device state -> public members
    void pull(int32_t cc[4])
    {
        sdrx_psk_get_state(m_h, &m_st);
        const sdrx_psk_state &s = m_st;
        NcoControlVariables.delay = s.nco_delay; Register_mu = s.reg_mu; Register_Q = s.reg_Q; Register_I = s.reg_I;
        SymbolTimingSynchronizationVariables.muDelay = s.mu_delay;
        SymbolTimingSynchronizationVariables.mu = s.reg_mu;
        InterpData *ip[2] = {&Interp_Q_Channel, &Interp_I_Channel};
        const double *src[2] = {s.iq_d, s.ii_d};
        for (int c = 0; c < 2; c++) {
            ip[c]->delay1 = src[c][0]; ip[c]->delay2 = src[c][1]; ip[c]->delay3 = src[c][2];
            ip[c]->delay4 = src[c][3]; ip[c]->delay5 = src[c][4];
        }
        GardnerDetectorVariables.delay1 = s.g1; GardnerDetectorVariables.delay2 = s.g2;
        GardnerDetectorVariables.delay3 = s.g3; GardnerDetectorVariables.delay4 = s.g4;
        LoopFilterVariables.delay_x = s.lf_x; LoopFilterVariables.delay_y = s.lf_y;
        PhaseCorrectionVariables.dds_delay = s.dds_delay;
        PhaseLoopFilterVariables.delay_x = s.plf_x; PhaseLoopFilterVariables.delay_y = s.plf_y;
        for (int k = 0; k < 4; k++) cc[k] = s.cc[k];
    }
    static void block_to_iq(const BBP_Block *in, std::vector<double> &iq)
    {
        const size_t n = in->number_of_points();
        iq.resize(2 * (n ? n : 1));
        for (size_t i = 0; i < n; i++) { iq[2 * i] = in->points[i].real(); iq[2 * i + 1] = in->points[i].imag(); }
    }

    int m_kind;
    sdrx_psk_t *m_h;
    sdrx_psk_state m_st;
};

#endif
