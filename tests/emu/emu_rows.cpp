// emu_rows.cpp -- host-side SIMT emulation of the forward row-transform kernel body (tests only):
// hpfw_amd/csrc/fft_rows.h compiled with -DHPFW_SIMT_EMU, every residue pair of one clip, then
// the length-n1 DFT of the specification on the host, compared with the oracle's forward bins.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>

#include "../../hpfw_amd/csrc/fft_rows.h"
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

int main(int argc, char **argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 44100 * 3;
    const int nthreads = argc > 2 ? std::atoi(argv[2]) : 384;
    hpfw::HostPlan hp;
    std::string why;
    if (!hpfw::build_plan(n, hp, why)) {
        std::fprintf(stderr, "plan: %s\n", why.c_str());
        return 2;
    }
    std::vector<int16_t> pcm(n);
    unsigned s = 777;
    for (long i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        double t = (double)i / 44100.0;
        pcm[i] = (int16_t)(9000.0 * std::sin(6.2831853 * 523.25 * t) + 6000.0 * std::sin(6.2831853 * 2093.0 * t + 1.0) +
                           2000.0 * ((double)(s >> 8) / 8388608.0 - 1.0));
    }
    hpfw::RowsArgs a;
    a.n1 = hp.n1;
    a.n2 = hp.n2;
    a.h = hp.h;
    for (size_t g = 0; g < hp.groups.size(); ++g) a.groups.tw_off[g] = hp.rows_gtw_off[g];
    a.hpad = (hp.h + 31) / 32 * 32;
    a.pair_stride = 1;
    a.groups.n = (int)hp.groups.size();
    for (size_t g = 0; g < hp.groups.size(); ++g) {
        a.groups.r1[g] = hp.groups[g].first;
        a.groups.r2[g] = hp.groups[g].second;
    }
    a.gtw = reinterpret_cast<const cf *>(hp.rows_gtw.data());
    a.tw_big = reinterpret_cast<const cf *>(hp.tw_big.data());
    a.pos_n2 = hp.pos_n2.data();
    a.kb_last = hp.kb_last.data();
    std::vector<float> yp((size_t)2 * hp.n1 * a.hpad, NAN);
    const size_t lds_n = (size_t)hp.n2;
    for (int a0 = 0; a0 < hp.n1; a0 += 2) {
        std::vector<hpfw::i16x2> pairs(hp.n2);
        for (int t = 0; t < hp.n2; ++t) {
            pairs[t].x = pcm[a0 + (long)hp.n1 * t];
            pairs[t].y = (a0 + 1 < hp.n1) ? pcm[a0 + 1 + (long)hp.n1 * t] : 0;
        }
        std::vector<cf> lds_mem(lds_n, cf{NAN, NAN});
        Checked<cf> lds{lds_mem.data(), lds_mem.size()};
        float *ya = yp.data() + (size_t)2 * a0 * a.hpad;
        float *yb = (a0 + 1 < hp.n1) ? yp.data() + (size_t)2 * (a0 + 1) * a.hpad : nullptr;
        // alternate between the compile-time group sequence (when it applies) and the run-time one
        if (hpfw::Groups6300::matches(a.groups, 0, a.groups.n) && (a0 & 2) == 0)
            hpfw::rows_body<hpfw::Groups6300>(lds, a, nthreads, pairs.data(), a0, ya, yb);
        else
            hpfw::rows_body<hpfw::RuntimeGroups>(lds, a, nthreads, pairs.data(), a0, ya, yb);
    }
    // S6: X[n2 k1 + k2] = sum_a T_n1[a k1] Y'[a][k2]
    const long nk = hp.kmax - hp.kmin;
    std::vector<float> got(2 * nk), ref(2 * nk);
    for (long k = hp.kmin; k < hp.kmax; ++k) {
        long k1 = k / hp.n2, k2 = k % hp.n2;
        bool conj = false;
        if (k2 >= hp.h) {
            k1 = hp.n1 - 1 - k1;
            k2 = hp.n2 - k2;
            conj = true;
        }
        float ar = 0.f, ai = 0.f;
        for (long aa = 0; aa < hp.n1; ++aa) {
            const hpfw::HostCf d = hp.tw_n1[(size_t)((aa * k1) % hp.n1)];
            const cf y = {yp[(size_t)(2 * aa) * a.hpad + k2], yp[(size_t)(2 * aa + 1) * a.hpad + k2]};
            ar = __builtin_fmaf(d.r, y.r, ar);
            ar = __builtin_fmaf(-d.i, y.i, ar);
            ai = __builtin_fmaf(d.i, y.r, ai);
            ai = __builtin_fmaf(d.r, y.i, ai);
        }
        got[2 * (k - hp.kmin)] = ar;
        got[2 * (k - hp.kmin) + 1] = conj ? -ai : ai;
    }
    hpfw_oracle_plan *op = hpfw_oracle_plan_create(n);
    hpfw_oracle_spectrum(op, pcm.data(), ref.data());
    long bad = 0;
    for (long i = 0; i < 2 * nk; ++i)
        if (!(got[i] == ref[i])) {
            if (bad < 5) std::fprintf(stderr, "bin %ld: emu %.9g oracle %.9g\n", i / 2 + hp.kmin, got[i], ref[i]);
            ++bad;
        }
    std::printf("n=%ld n1=%d n2=%d groups=%d gtw=%d values=%ld mismatches=%ld\n", n, hp.n1, hp.n2, a.groups.n, (int)hp.rows_gtw.size(),
                2 * nk, bad);
    hpfw_oracle_plan_destroy(op);
    return bad ? 1 : 0;
}
