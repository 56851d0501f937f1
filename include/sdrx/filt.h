// sdrx/filt.h -- drop-in for RadioNode/third-party/FIR-filter-class/filt.h:131-162.
// Same class name, constructors, methods and error codes; execute()/do_sample() run on the
// MI355X through libsdrx (sdrx_fir_*), the shift register lives on the device between calls.
#ifndef _FILTER_H
#define _FILTER_H

#define MAX_NUM_FILTER_TAPS 1000 // filt.h:121

#include <math.h>
#include <stdio.h>

#include "common.h"

enum filterType { LPF, HPF, BPF }; // filt.h:131

class Filter {
  public:
    Filter(filterType filt_t, int num_taps, double Fs, double Fx) // filt.cpp:45-70
        : m_num_taps(num_taps), m_Fs(Fs), m_h(sdrx_fir_create((int)filt_t, 0, num_taps, Fs, Fx, 0.0, MAX_NUM_FILTER_TAPS)) {}
    Filter(filterType filt_t, int num_taps, double Fs, double Fl, double Fu) // filt.cpp:73-102
        : m_num_taps(num_taps), m_Fs(Fs), m_h(sdrx_fir_create((int)filt_t, 1, num_taps, Fs, Fl, Fu, MAX_NUM_FILTER_TAPS)) {}
    ~Filter() { sdrx_fir_destroy(m_h); }
    Filter(const Filter &) = delete;
    Filter &operator=(const Filter &) = delete;

    void init() { sdrx_fir_reset(m_h); }                                             // filt.cpp:236-246
    double do_sample(double data_sample) { return sdrx_fir_do_sample(m_h, data_sample); } // filt.cpp:248-265
    int get_error_flag() { return sdrx_fir_error_flag(m_h); }                        // filt.h:156
    void get_taps(double *taps) { sdrx_fir_get_taps(m_h, taps); }                    // filt.cpp:156-166
    void execute(RADIO_DATA_TYPE *filter, size_t length) { (void)sdrx_fir_execute(m_h, filter, length); } // filt.cpp:267-273

    int write_taps_to_file(char *filename) // filt.cpp:168-186 (text dump of the taps, host I/O)
    {
        if (get_error_flag() != 0) return -1;
        std::valarray<double> t(0.0, (size_t)m_num_taps);
        get_taps(&t[0]);
        FILE *fd = fopen(filename, "w");
        if (fd == NULL) return -1;
        fprintf(fd, "%d\n", m_num_taps);
        for (int i = 0; i < m_num_taps; i++) fprintf(fd, "%15.6f\n", t[i]);
        fclose(fd);
        return 0;
    }
    int write_freqres_to_file(char *filename) // filt.cpp:190-234 (diagnostic dump, host I/O)
    {
        const int NP = 1000;
        if (get_error_flag() != 0) return -1;
        std::valarray<double> t(0.0, (size_t)m_num_taps), mag(0.0, (size_t)NP);
        get_taps(&t[0]);
        const double dw = M_PI / (NP - 1.0);
        double mag_max = -1;
        for (int i = 0; i < NP; i++) {
            const double w = i * dw;
            double yr = 0, yi = 0;
            for (int k = 0; k < m_num_taps; k++) {
                yr += t[k] * cos(k * w);
                yi -= t[k] * sin(k * w);
            }
            mag[i] = sqrt(yr * yr + yi * yi);
            if (mag[i] > mag_max) mag_max = mag[i];
        }
        if (mag_max <= 0.0) return -2;
        FILE *fd = fopen(filename, "w");
        if (fd == NULL) return -3;
        for (int i = 0; i < NP; i++) {
            double d = -100;
            if (mag[i] != 0) {
                d = 20 * log10(mag[i] / mag_max);
                if (d < -100) d = -100;
            }
            fprintf(fd, "%10.6e %10.6e\n", i * dw * (m_Fs / 2) / M_PI, d);
        }
        fclose(fd);
        return 0;
    }

  private:
    int m_num_taps;
    double m_Fs;
    sdrx_fir_t *m_h;
};

#endif
