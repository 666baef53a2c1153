// emu_rows.cpp -- host-side SIMT emulation of the forward transform's row stage (tests only):
// hpfw_amd/csrc/fft_rows.h compiled with -DHPFW_SIMT_EMU.  The column stage of S6 (exact integer sums, one rounding,
// the plan's fixed-point twiddles wq) is evaluated here; every row then goes through
// rows2_body -- load times the twiddles between the stages, FFT_n2 in (bounds-checked) LDS, the pruned stores of the row and of its mirror -- and the bins
// gathered out of the rows layout are compared with the oracle's forward bins.
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
    pcm[0] = 32767; // the ends of the sample range go through the digit split of the column stage (GPU) and through this sum alike
    pcm[1] = -32768;
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
    a.tw_big = nullptr;
    a.pos_n2 = hp.pos_n2.data();
    a.kb_last = hp.kb_last.data();
    const hpfw::Rows2Out o{hp.n1, hp.hq, hp.q2lo, hp.q2w, reinterpret_cast<const cf *>(hp.ts_seed.data()),
                           reinterpret_cast<const cf *>(hp.ts_step.data()), (hp.n2 + 3) / 4,
                           28 /* one contiguous row: every piece in block 0 */, 0, hp.n2 /* Im row n2 floats behind Re */, 0};
    std::vector<cf> x((size_t)hp.n1 * hp.q2w, cf{NAN, NAN});
    const size_t lds_n = (size_t)hp.n2;
    std::vector<float> z((size_t)2 * hp.n2); // Re row, Im row
    for (int q1 = 0; q1 < hp.hq; ++q1) {
        for (int k2 = 0; k2 < hp.n2; ++k2) { // S6 column stage, as the plan's tables state it
            long long gr = 0, gi = 0;
            for (int k1 = 0; k1 < hp.n1; ++k1) {
                const size_t m = (size_t)(((long long)q1 * k1) % hp.n1);
                gr += (long long)hp.wq[2 * m] * pcm[(size_t)hp.n2 * k1 + k2];
                gi += (long long)hp.wq[2 * m + 1] * pcm[(size_t)hp.n2 * k1 + k2];
            }
            z[(size_t)k2] = (float)gr;               // one rounding of the exact integer
            z[(size_t)hp.n2 + k2] = (float)gi;
        }
        std::vector<cf> lds_mem(lds_n, cf{NAN, NAN});
        Checked<cf> lds{lds_mem.data(), lds_mem.size()};
        // alternate between the compile-time group sequence (when it applies) and the run-time one
        if (hpfw::Groups6300::matches(a.groups, 0, a.groups.n) && (q1 & 1) == 0)
            hpfw::rows2_body<hpfw::Groups6300>(lds, a, nthreads, z.data(), q1, o, x.data());
        else
            hpfw::rows2_body<hpfw::RuntimeGroups>(lds, a, nthreads, z.data(), q1, o, x.data());
    }
    const long nk = hp.kmax - hp.kmin;
    std::vector<float> got(2 * nk), ref(2 * nk);
    for (long k = hp.kmin; k < hp.kmax; ++k) { // the bins out of the rows layout: x[k mod n1][k / n1 - q2lo]
        const cf v = x[(size_t)(k % hp.n1) * hp.q2w + (size_t)(k / hp.n1 - hp.q2lo)];
        got[2 * (k - hp.kmin)] = v.r;
        got[2 * (k - hp.kmin) + 1] = v.i;
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
