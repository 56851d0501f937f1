// sdrx/common.h -- the reference's boundary types (RadioNode/src/libraries/common/
// Common_Deffinitions.h:11-35, BBP_Block.h:12-55, confidence_counter.h) for code that builds
// against the drop-in headers instead of the reference tree.  Types only; no DSP.
#ifndef SDRX_DROPIN_COMMON_H
#define SDRX_DROPIN_COMMON_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <complex>
#include <mutex>
#include <valarray>

#include "../sdrx.h"

typedef double RADIO_DATA_TYPE;                // Common_Deffinitions.h:11
typedef std::complex<RADIO_DATA_TYPE> Complex; // :12
typedef std::valarray<Complex> Complex_Array;  // :13
const RADIO_DATA_TYPE PI = 3.141592653589793238460; // :14
#define MHZ_TO_HZ(freq) (freq * 1000000)
#define BLOCK_READ_SIZE 240000 / 2 // :20 (unparenthesised, as in the reference)
enum HardwareType { CPU, GPU };    // :23-27
enum GainMode { Auto, Manual };    // :31-35

#ifndef SIGN_FUNC
#define SIGN_FUNC(val) ((val == 0.0) ? 0.0 : ((val < 0.0) ? -1.0 : 1.0)) // BPSK.h:16
#endif

/// BBP_Block.h:12-55
struct BBP_Block {
    Complex_Array points;
    size_t n_points;
    BBP_Block() : points(Complex(0, 0), BLOCK_READ_SIZE), n_points(0) { reset(); }
    explicit BBP_Block(size_t size) : points(Complex(0, 0), size), n_points(0) { reset(); }
    size_t number_of_points() const { return n_points; }
    void reset()
    {
        n_points = 0;
        points = Complex(0, 0);
    }
};

/// confidence_counter.h:12-88 -- the PSK classes expose their counters through this type; its value
/// is mirrored from / to the device-side counter around every block call.
class confidence_counter {
  public:
    confidence_counter(int32_t max, int32_t low_threshold, int32_t high_threshold)
        : m_counter(max / 2), m_low_threshold(low_threshold), m_high_threshold(high_threshold), m_max(max) {}
    int32_t value() const { return m_counter; }
    bool high_confidence() const { return m_counter >= m_high_threshold; }
    bool low_confidence() const { return m_counter < m_low_threshold; }
    void operator++() { increment(1); }
    void operator--() { decrement(1); }
    void operator++(int) { increment(1); }
    void operator--(int) { decrement(1); }
    void decrement(int32_t dec)
    {
        std::lock_guard<std::mutex> lk(m_lock);
        m_counter -= dec;
        if (m_counter < 0) m_counter = 0;
    }
    void increment(int32_t inc)
    {
        std::lock_guard<std::mutex> lk(m_lock);
        m_counter += inc;
        if (m_counter > m_max) m_counter = m_max;
    }
    void reset()
    {
        std::lock_guard<std::mutex> lk(m_lock);
        m_counter = m_max / 2;
    }
    void set(int32_t v) { m_counter = v; } // drop-in only: mirror of the device counter

  private:
    int32_t m_counter, m_low_threshold, m_high_threshold, m_max;
    std::mutex m_lock;
};

#endif
