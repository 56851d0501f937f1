// dropin_driver.cpp -- exercises the C++ drop-in headers (include/sdrx/*.h) the way the reference's
// own unit tests drive the reference classes (RadioNode/src/test/TEST_*.cpp), on the GPU.
// Usage: dropin_driver <indir> <outdir>.  Inputs and outputs are raw little-endian arrays;
// tests/test_gpu_dropin_cpp.py writes the inputs and checks every output against the CPU oracle.
#include <cstdio>
#include <string>
#include <vector>

#include <sdrx/AM.h>
#include <sdrx/BPSK.h>
#include <sdrx/FM.h>
#include <sdrx/QPSK.h>
#include <sdrx/Resample.h>
#include <sdrx/filt.h>

template <class T> static std::vector<T> rd(const std::string &p)
{
    std::vector<T> v;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}
template <class T> static void wr(const std::string &p, const T *d, size_t n)
{
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", p.c_str()); exit(2); }
    fwrite(d, sizeof(T), n, f);
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string in = std::string(argv[1]) + "/", out = std::string(argv[2]) + "/";
    if (sdrx_init(0) != SDRX_OK) { fprintf(stderr, "sdrx_init: %s\n", sdrx_last_error()); return 3; }

    { // Filter: LPF 64 in two ragged chunks + do_sample, HPF 51, BPF 129, error object
        std::vector<double> x = rd<double>(in + "x.f64");
        Filter lpf(LPF, 64, 2.0e6, 1.0e5);
        std::vector<double> y = x;
        lpf.execute(y.data(), 1000);
        lpf.execute(y.data() + 1000, y.size() - 1001);
        y.back() = lpf.do_sample(y.back());
        wr(out + "fir_lpf.f64", y.data(), y.size());
        std::vector<double> taps(64);
        lpf.get_taps(taps.data());
        wr(out + "fir_lpf_taps.f64", taps.data(), taps.size());
        Filter hpf(HPF, 51, 2.0e6, 3.0e5);
        y = x;
        hpf.execute(y.data(), y.size());
        wr(out + "fir_hpf.f64", y.data(), y.size());
        Filter bpf(BPF, 129, 2.0e6, 1.0e5, 4.0e5);
        y = x;
        bpf.execute(y.data(), y.size());
        wr(out + "fir_bpf.f64", y.data(), y.size());
        Filter bad(LPF, 1001, 2.0e6, 1.0e5);
        double e[3] = {(double)bad.get_error_flag(), bad.do_sample(1.0), (double)lpf.get_error_flag()};
        wr(out + "fir_err.f64", e, 3);
    }
    { // AM: TEST(AM, ModDemod) shape
        std::vector<uint8_t> bits = rd<uint8_t>(in + "bits.u8");
        std::vector<double> mod(bits.size() * 8, 0.0);
        AM<RADIO_DATA_TYPE> am;
        am.modulate(bits.data(), bits.size(), mod.data(), 0, mod.size());
        wr(out + "am_mod.f64", mod.data(), mod.size());
        for (double &v : mod) v *= 127.0;
        std::vector<uint8_t> o(bits.size());
        size_t n = am.demodulate(mod.data(), mod.data(), o.data(), mod.size(), 8);
        wr(out + "am_bits.u8", o.data(), n);
    }
    { // FM: TEST(FM, modulate/demodulate) shape
        std::vector<uint8_t> bits = rd<uint8_t>(in + "bits.u8");
        bits.resize(200);
        std::vector<double> mod(bits.size() * 100, 0.0), im(bits.size() * 100, 0.0);
        FM<RADIO_DATA_TYPE> tx(100, 10, 20, 10), rx(100, 100, 20, 10);
        tx.modulate(bits.data(), bits.size(), mod.data(), im.data(), mod.size(), 100);
        wr(out + "fm_mod.f64", mod.data(), mod.size());
        std::vector<uint8_t> o(bits.size());
        size_t n = rx.demodulate(mod.data(), im.data(), o.data(), mod.size(), 100);
        wr(out + "fm_bits.u8", o.data(), n);
    }
    { // BPSK: TEST(BPSK, Full_ModDemod) shape, two consecutive blocks, then public state
        std::vector<uint8_t> bits = rd<uint8_t>(in + "bits.u8");
        const size_t L = bits.size(), N = L * 8;
        BPSK bpsk(0x14EB, 16);
        std::vector<double> mod(N, 0.0);
        bpsk.modulate(bits.data(), L, mod.data(), N);
        wr(out + "bpsk_mod.f64", mod.data(), N);
        BBP_Block inb(N), outb(N);
        std::vector<double> dec;
        for (int blk = 0; blk < 2; blk++) {
            inb.reset();
            for (size_t i = 0; i < N; i++) {
                inb.points[i].real(mod[i]);
                inb.points[i].imag(mod[i]);
                inb.n_points++;
            }
            bpsk.demodulate(&inb, &outb);
            for (size_t i = 0; i < outb.number_of_points(); i++) dec.push_back(outb.points[i].real());
        }
        wr(out + "bpsk_dec.f64", dec.data(), dec.size());
        double st[6] = {bpsk.NcoControlVariables.delay, bpsk.Register_mu, bpsk.PhaseCorrectionVariables.dds_delay,
                        bpsk.LoopFilterVariables.delay_y, bpsk.PhaseLoopFilterVariables.delay_y,
                        (double)bpsk.m_phase_ambiguity_cc.value()};
        wr(out + "bpsk_state.f64", st, 6);
    }
    { // QPSK: TEST(QPSK, Deconstructed_ModDemod) shape: every stage through the public members
        std::vector<uint8_t> sym = rd<uint8_t>(in + "sym.u8");
        const size_t L = sym.size(), N = L * 8;
        QPSK qpsk;
        std::vector<double> mr(N, 0.0), mi(N, 0.0);
        qpsk.modulate(sym.data(), L, mr.data(), mi.data(), N);
        wr(out + "qpsk_mod_re.f64", mr.data(), N);
        wr(out + "qpsk_mod_im.f64", mi.data(), N);
        BBP_Block blk(N), outb(N);
        for (size_t i = 0; i < N; i++) {
            blk.points[i].real(mr[i]);
            blk.points[i].imag(mi[i]);
            blk.n_points++;
        }
        qpsk.m_length = blk.number_of_points() / 4;
        std::vector<double> re(qpsk.m_length, 0.0), im(qpsk.m_length, 0.0);
        qpsk.m_RealData = re.data();
        qpsk.m_ImagData = im.data();
        qpsk.decode_pulseShape(&blk);
        wr(out + "qpsk_corr_re.f64", re.data(), qpsk.m_length);
        wr(out + "qpsk_corr_im.f64", im.data(), qpsk.m_length);
        qpsk.symbol_timing_synchronization();
        wr(out + "qpsk_time_re.f64", re.data(), qpsk.m_length);
        wr(out + "qpsk_time_im.f64", im.data(), qpsk.m_length);
        qpsk.phase_shift_correction();
        wr(out + "qpsk_phase.f64", re.data(), qpsk.m_length);
        qpsk.phase_ambiguity_correction();
        wr(out + "qpsk_amb.f64", re.data(), qpsk.m_length);
        QPSK q2;
        q2.demodulate(&blk, &outb);
        std::vector<double> dec;
        for (size_t i = 0; i < outb.number_of_points(); i++) dec.push_back(outb.points[i].real());
        wr(out + "qpsk_dec.f64", dec.data(), dec.size());
    }
    { // round 2 members: public statics, Resample.h helpers, FM::convolve, BPSK::demodulate's second overload
        std::vector<double> st;
        for (int i = 0; i < 2; i++) st.push_back(BPSK::LUT_REAL[i]);
        for (int i = 0; i < 2; i++) st.push_back(BPSK::LUT_180[i]);
        for (int i = 0; i < 129; i++) st.push_back(BPSK::RRC_COEFF[i]);
        for (int i = 0; i < 129; i++) st.push_back(QPSK::RRC_COEFF[i]);
        const RADIO_DATA_TYPE *luts[6] = {QPSK::LUT_REAL, QPSK::LUT_IMAG, QPSK::LUT_REAL___0, QPSK::LUT_REAL__90,
                                          QPSK::LUT_REAL_180, QPSK::LUT_REAL_270};
        for (const RADIO_DATA_TYPE *l : luts)
            for (int i = 0; i < 4; i++) st.push_back(l[i]);
        for (int i = 0; i < 64; i++) st.push_back(QPSK::DEFAULT_UNIQUE_WORD[i]);
        st.push_back((double)sizeof(BPSK::RRC_COEFF) / sizeof(double));
        wr(out + "statics.f64", st.data(), st.size());

        std::vector<double> x = rd<double>(in + "x.f64");
        std::vector<double> up(80 + 4, 7.25), zs(80 + 4, 7.25), rp(75 + 4, 7.25);
        raw_upsample<RADIO_DATA_TYPE>(x.data(), 10, up.data(), 80, 8.0);
        upsample_fill_w_zeros(x.data(), 10, zs.data(), 80, 8.0);
        upsample(x.data(), 10, rp.data(), 75, 7.5);
        wr(out + "rs_raw.f64", up.data(), up.size());
        wr(out + "rs_zeros.f64", zs.data(), zs.size());
        wr(out + "rs_ramp.f64", rp.data(), rp.size());
        std::vector<double> d(x.begin(), x.begin() + 1001);
        double r = (double)decimate<RADIO_DATA_TYPE>(d.data(), 1001, 4);
        d.push_back(r);
        wr(out + "rs_decimate.f64", d.data(), d.size());
        BBP_Block blk(64);
        for (size_t i = 0; i < 50; i++) { blk.points[i] = Complex(x[2 * i], x[2 * i + 1]); blk.n_points++; }
        decimate(&blk, 3);
        std::vector<double> bo;
        for (size_t i = 0; i < 50; i++) { bo.push_back(blk.points[i].real()); bo.push_back(blk.points[i].imag()); }
        bo.push_back((double)blk.n_points);
        wr(out + "rs_decimate_block.f64", bo.data(), bo.size());

        FM<RADIO_DATA_TYPE> fm(100, 100, 20, 10);
        std::vector<double> w(x.begin() + 2000, x.begin() + 2150), acc(10, 0.5);
        fm.convolve(x.data(), 1003, w.data(), 150, acc.data(), 10, 100);
        wr(out + "fm_convolve.f64", acc.data(), acc.size());

        BPSK b2(0x14EB, 16);
        for (int call = 0; call < 2; call++) {
            std::vector<double> iq = rd<double>(in + "bpsk2_iq" + std::to_string(call) + ".f64");
            const size_t n = iq.size() / 2, ds = call == 0 ? 2 : 3;
            Complex_Array a(Complex(0, 0), n);
            for (size_t i = 0; i < n; i++) a[i] = Complex(iq[2 * i], iq[2 * i + 1]);
            std::vector<uint8_t> bits(n, 0);
            std::vector<double> pr(n, 0.0), pi(n, 0.0);
            const size_t m = b2.demodulate(a, bits.data(), n, ds, pr.data(), pi.data());
            wr(out + "bpsk2_bits" + std::to_string(call) + ".u8", bits.data(), m);
            wr(out + "bpsk2_plot_re" + std::to_string(call) + ".f64", pr.data(), n);
            wr(out + "bpsk2_plot_im" + std::to_string(call) + ".f64", pi.data(), n);
            for (size_t i = 0; i < n; i++) { iq[2 * i] = a[i].real(); iq[2 * i + 1] = a[i].imag(); }
            wr(out + "bpsk2_after" + std::to_string(call) + ".f64", iq.data(), iq.size());
        }
    }
    printf("dropin_driver ok\n");
    return 0;
}
