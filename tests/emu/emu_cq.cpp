// emu_cq.cpp -- host-side SIMT emulation of the chirp-z band kernel body (tests only).
// Compiles hpfw_amd/csrc/fft_lds.h with -DHPFW_SIMT_EMU (threads become a loop, LDS becomes a
// bounds-checked array) and compares every band of one clip with the CPU oracle.  This checks the
// index arithmetic of the fused / padded / pruned passes without a GPU; the GPU parity tests
// (tests/test_gpu_parity.py) remain the proof for the real kernels.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../hpfw_amd/csrc/fft_lds.h"
#include "../../hpfw_amd/csrc/plan.h"
#include "../../oracle/hpfw_oracle.h"

using hpfw::cf;

template <class T>
struct Checked {
    T *p;
    size_t n;
    T &operator[](long i) const
    {
        if (i < 0 || (size_t)i >= n) {
            std::fprintf(stderr, "LDS index %ld out of [0,%zu)\n", i, n);
            std::abort();
        }
        return p[i];
    }
};

template <int NP>
static void run_band(const hpfw::HostPlan &hp, const hpfw::BluesteinClass &bc, int j, const cf *x, float *mag)
{
    using P = hpfw::Size<NP>;
    int nt = NP % 3 == 0 ? NP / 12 : NP / 16;
    if (nt < 64) nt = 64;
    if (nt > 1024) nt = 1024;
    std::vector<cf> lds_mem(P::DATA);
    // poison so that a read of never-written LDS shows up as NaN
    for (auto &v : lds_mem) v = {__builtin_nanf(""), __builtin_nanf("")};
    Checked<cf> lds{lds_mem.data(), lds_mem.size()};
    std::vector<float> red_mem(nt);
    Checked<float> red{red_mem.data(), red_mem.size()};
    const hpfw::XsPtr xs{x + (hp.start[j] - hp.kmin)};
    const cf *g = reinterpret_cast<const cf *>(hp.g.data()) + hp.g_off[j];
    hpfw::CqTwiddles tw;
    tw.tab = reinterpret_cast<const cf *>(bc.gtw.data());
    for (int k = 0; k < 4; ++k) tw.off[k] = bc.goff[k];
    tw.mid_off = bc.mid_off;
    hpfw::cq_band_body<NP>(lds, red, nt, xs, g, hp.lg[j], tw, reinterpret_cast<const cf *>(bc.vrev.data()), hp.c,
                             mag + (size_t)j * hp.c, [](float m) { return m; });
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 44100 * 3;
    hpfw::HostPlan hp;
    std::string why;
    if (!hpfw::build_plan(n, hp, why)) {
        std::fprintf(stderr, "plan: %s\n", why.c_str());
        return 2;
    }
    hpfw_oracle_plan *op = hpfw_oracle_plan_create(n);
    std::vector<int16_t> pcm(n);
    unsigned s = 12345;
    for (long i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        double t = (double)i / 44100.0;
        double v = 0.3 * __builtin_sin(6.2831853 * 440.0 * t) + 0.2 * __builtin_sin(6.2831853 * 1567.9 * t * (1 + 0.02 * t)) +
                   0.05 * ((double)(s >> 8) / 8388608.0 - 1.0);
        pcm[i] = (int16_t)(v * 20000.0);
    }
    const long nk = hp.kmax - hp.kmin;
    std::vector<float> x(2 * nk), ref((size_t)121 * hp.c), got((size_t)121 * hp.c, -1.0f);
    hpfw_oracle_spectrum(op, pcm.data(), x.data());
    hpfw_oracle_cqmag(op, x.data(), ref.data());
    const cf *xc = reinterpret_cast<const cf *>(x.data());
    for (const hpfw::BluesteinClass &bc : hp.classes) {
        for (int j : bc.bands) {
            switch (bc.p) {
            case 64: run_band<64>(hp, bc, j, xc, got.data()); break;
            case 96: run_band<96>(hp, bc, j, xc, got.data()); break;
            case 128: run_band<128>(hp, bc, j, xc, got.data()); break;
            case 192: run_band<192>(hp, bc, j, xc, got.data()); break;
            case 256: run_band<256>(hp, bc, j, xc, got.data()); break;
            case 384: run_band<384>(hp, bc, j, xc, got.data()); break;
            case 512: run_band<512>(hp, bc, j, xc, got.data()); break;
            case 768: run_band<768>(hp, bc, j, xc, got.data()); break;
            case 1024: run_band<1024>(hp, bc, j, xc, got.data()); break;
            case 1536: run_band<1536>(hp, bc, j, xc, got.data()); break;
            case 2048: run_band<2048>(hp, bc, j, xc, got.data()); break;
            case 3072: run_band<3072>(hp, bc, j, xc, got.data()); break;
            case 4096: run_band<4096>(hp, bc, j, xc, got.data()); break;
            case 6144: run_band<6144>(hp, bc, j, xc, got.data()); break;
            case 8192: run_band<8192>(hp, bc, j, xc, got.data()); break;
            case 12288: run_band<12288>(hp, bc, j, xc, got.data()); break;
            case 16384: run_band<16384>(hp, bc, j, xc, got.data()); break;
            default: std::fprintf(stderr, "unexpected size %d\n", bc.p); return 2;
            }
        }
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i)
        if (!(ref[i] == got[i])) {
            if (bad < 5) std::fprintf(stderr, "band %zu col %zu: emu %.9g oracle %.9g\n", i / hp.c, i % hp.c, got[i], ref[i]);
            ++bad;
        }
    std::printf("n=%ld classes=%zu values=%zu mismatches=%ld\n", n, hp.classes.size(), ref.size(), bad);
    hpfw_oracle_plan_destroy(op);
    return bad ? 1 : 0;
}
