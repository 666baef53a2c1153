// plan.h -- host-side geometry and tables of the extraction path (DESIGN.md "Arithmetic
// specification" S2, S5-S7).  Everything here runs once per distinct clip length.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace hpfw {

struct HostCf {
    float r, i;
};

struct BluesteinClass {
    int p = 0;
    std::vector<int> radix;
    std::vector<HostCf> tw;    // T_p
    std::vector<HostCf> gtw;   // per-butterfly twiddles of the fused groups, [entry][butterfly] per group
    int goff[4] = {0, 0, 0, 0}; // start of each outer group's table in gtw
    int mid_off = 0;           // start of the innermost group's entries
    std::vector<HostCf> vrev;  // DFT_p(chirp) at digit-reversed positions
    std::vector<int> bands;    // bands using this size
    // p > 16384 does not fit the LDS: the first `outer` radix-4 passes (and the last `outer` inverse ones)
    // run through global memory, the rest on blocks of len0 = p / 4^outer points in LDS; gtw then holds
    // the groups of a length-len0 transform with the twiddles of length p
    int len0 = 0, outer = 0;
};

struct HostPlan {
    int64_t n = 0;
    int n1 = 0, n2 = 0, h = 0;
    int kmin = 0, kmax = 0, k1lo = 0, k1hi = 0;
    int m = 0, c = 0, n_frames = 0, n_hp = 0;
    std::vector<int> radix;                 // passes of the length-n2 FFT
    std::vector<std::pair<int, int>> groups; // the same passes, fused in pairs (second = 1: single pass)
    std::vector<HostCf> rows_gtw;           // per-butterfly twiddles of every group, [entry][butterfly]
    std::vector<int> rows_gtw_off;          // start of each group's table in rows_gtw
    std::vector<HostCf> tw_n2, tw_n1;
    std::vector<HostCf> tw_big;             // the STFT of the Mel front-end only (build_frame_transform): rows of ones
    // S6, column stage first (7-smooth lengths): the clip as it lies is an [n1][n2] sample matrix
    int hq = 0;                             // rows q1 = 0 .. n1 / 2 of the column stage (the rest follow by conjugation)
    int q2lo = 0, q2w = 0;                  // outputs q2lo .. q2lo + q2w - 1 of every row transform hold consumed bins
    int cols_mt = 0, cols_ks = 0;           // 32-row tiles of the (Re, Im) interleaved rows; 32-sample steps of k1
    std::vector<int32_t> wq;                // [n1][2]: rint(2^22 T_n1[m])
    std::vector<int8_t> cols_image;         // digits of wq as the matrix instruction's A operand (k_forward.hip)
    std::vector<double> cols_corr;          // [2 hq]: 128 sum_k1 wq[(q1 k1) mod n1] (Re, Im): the samples' +128 digit offset
    // the twiddles between the stages, T_N[q1 k2] 2^-37, are formed by the row stage from every fourth one:
    // ts[q1][4 m + e] = ts_seed[q1][m] (e = 0), ts_seed[q1][m] * ts_step[q1][e] (S1; e = 1, 2, 3)
    std::vector<HostCf> ts_seed;            // [hq][ceil(n2 / 4)]: T_N[4 q1 m] 2^-37
    std::vector<HostCf> ts_step;            // [hq][4]: T_N[q1 e] (entry 0 unused)
    std::vector<int> pos_n2;
    std::vector<int> kb_last;               // last group's block b holds outputs kb_last[b] + (n2 / len) f
    int start[121], lg[121], psize[121];
    std::vector<int64_t> g_off;             // [121]
    int64_t g_total = 0, big_m = 0;         // sum of lg; M (the longest band: every band's output length)
    std::vector<HostCf> g;                  // window * chirp / (M P), concatenated (empty when the device generates it)
    std::vector<BluesteinClass> classes;
    // clip lengths with a prime factor above 7: the forward DFT as a chirp-z (Bluestein) convolution of
    // length bz_l = n1 * n2 >= N + (kmax - kmin) - 1, n2 = 6300; n1 need not be smooth
    bool bluestein = false;
    int64_t bz_l = 0;
    // its tables (chirp, T_L, Bhat, w[k] / L) are generated on the device: kernels.h BzArgs
};

// Returns false (and a reason) when the clip length is unsupported.
// geometry_only: stop after the sizes (n1, n2, M, C, bins consumed) are known, build no table
// Conventions of essentia's NSGConstantQ that cannot be checked offline (essentia is not vendored; DESIGN.md
// appendix A): a maintainer holding one real essentia output flips these until the geometry and the spectrogram
// match, without touching a kernel.  0 = the restatement's defaults.
enum : unsigned {
    kConvHannPeriodic = 1u,   // window 0.5 - 0.5 cos(2 pi i / L) instead of the symmetric 2 pi i / (L - 1)
    kConvLgHalfEven = 2u,     // Lg = round-half-to-even(bw / fftres) instead of round-half-away-from-zero
    kConvFloatGeometry = 4u,  // fftres, band frequencies, posit and Lg evaluated in float (essentia's Real) instead of double
    kConvNoIfftScale = 8u,    // band transforms without the 1/M of the inverse FFT (only the 1e-10 power floor sees it)
    kConvAll = 15u
};
// While one of these lives, build_plan on this thread does not spawn its own team of host threads (the caller is
// already one of many: the collectors' reader threads prepare the tables of their files' lengths)
struct PlanSerial {
    PlanSerial();
    ~PlanSerial();
    bool before;
};
// force_bluestein: take the chirp-z forward transform even when the length is 7-smooth (tests)
// host_windows: build the constant-Q window table g on the host too (the product generates it on the device)
bool build_plan(int64_t n_samples, HostPlan &out, std::string &why, bool geometry_only = false,
                bool force_bluestein = false, unsigned conventions = 0, bool host_windows = true);
// the two per-band constants of the window table (S5): shared by plan.cpp and the device generator's launch
inline double cq_window_scale(unsigned conventions, int64_t big_m, int psize)
{
    return 1.0 / (((conventions & kConvNoIfftScale) ? 1.0 : (double)big_m) * (double)psize);
}
inline int64_t cq_hann_den(unsigned conventions, int64_t lg) { return (conventions & kConvHannPeriodic) ? lg : lg - 1; }

// the row transform alone for frames of n2 samples (STFT of the Mel front-end): radix, groups, rows_gtw,
// pos_n2 of `out`, and tw_big = two rows of ones
bool build_frame_transform(int n2, HostPlan &out, std::string &why);
// window [4410] and filterbank operand image [2208][64] of the Mel front-end
void mel_tables(std::vector<float> &window, std::vector<float> &cpack);

// e^{-2 pi i m / n} with exact octant symmetry, evaluated in double (S2)
void twiddle_d(int64_t m, int64_t n, double &re, double &im);
// mixed-radix digit reversal of the DIF pass list
int64_t digit_pos(int64_t k, int64_t n, const std::vector<int> &radix);
bool make_radix_list(int64_t n, std::vector<int> &radix);
// the table of one fused group: entry e of the butterflies j0 = 0 .. m2 - 1 of a block (device_math.h tw_entry)
void append_group_twiddles(const std::vector<HostCf> &tw, int64_t n, int64_t len, int r1, int r2,
                           std::vector<HostCf> &out);

} // namespace hpfw
