// plan.cpp -- geometry of essentia's NSGConstantQ as hpfw configures it, and the constant tables
// of the forward / chirp-z transforms.
//
// Geometry follows the reference call site include/hpfw/spectrum/cqt.h:54-61
// (NSGConstantQ: inputSize = clip length, gamma 0, binsPerOctave 24, minimumWindow 96, window
// "hann", minFrequency 130.81, maxFrequency 4186.01; essentia defaults otherwise: rasterize "full")
// and essentia's designWindow(): f_j = fmin 2^(j/24), bw_j = (2^(1/24) - 2^(-1/24)) f_j,
// posit_j = floor(f_j / fftres), Lg_j = max(round(bw_j / fftres), 96), M = max_j Lg_j.
// hpfw keeps ceil(M / 3) columns (cqt.h:73-81).
#include "plan.h"
#include "trig_d.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace hpfw {

namespace {
constexpr double kSampleRate = 44100.0;
constexpr double kMinFreq = 130.81;
constexpr double kMaxFreq = 4186.01;
constexpr int kBpo = 24;
constexpr int kMinWindow = 96;
constexpr int kDown = 3;
constexpr int kBins = 121;
constexpr int kCtx = 20;
constexpr int kLag = 80;
constexpr int64_t kN2Max = 6826; // one in-LDS transform: with half of its twiddle table, two fit a CU's LDS

HostCf twiddle_f(int64_t m, int64_t n)
{
    double c, s;
    twiddle_d(m, n, c, s);
    return {(float)c, (float)s};
}

// twiddle_d(m, n) for m = 0 .. n - 1 as (re, im) pairs, computed once per n and process (the values are a function of
// (m, n) alone).  The transforms of a corpus share a handful of lengths -- the row length 6300, the chirp-z sizes 2^a and
// 3 * 2^a and their thirds -- while every file brings a new clip length: without this a length's host tables cost some
// 160 000 sincos, three quarters of them for twiddles computed for the previous file already.  Sizes above 2^16 (the
// clip-length-sized tables of the 7-smooth path) are not kept.
const std::vector<double> *twiddle_d_table(int64_t n)
{
    if (n > (1 << 16)) return nullptr;
    static std::mutex mtx;
    static std::map<int64_t, std::unique_ptr<std::vector<double>>> tables;
    {
        std::scoped_lock lock(mtx);
        auto it = tables.find(n);
        if (it != tables.end()) return it->second.get();
    }
    auto t = std::make_unique<std::vector<double>>((size_t)(2 * n));
    for (int64_t m = 0; m < n; ++m) twiddle_d(m, n, (*t)[(size_t)(2 * m)], (*t)[(size_t)(2 * m + 1)]);
    std::scoped_lock lock(mtx);
    auto &slot = tables[n];
    if (!slot) slot = std::move(t); // (another thread may have been faster: its table holds the same values)
    return slot.get();
}

std::vector<HostCf> twiddle_table(int64_t n)
{
    std::vector<HostCf> t((size_t)n);
    if (const std::vector<double> *d = twiddle_d_table(n)) {
        for (int64_t m = 0; m < n; ++m) t[(size_t)m] = {(float)(*d)[(size_t)(2 * m)], (float)(*d)[(size_t)(2 * m + 1)]};
        return t;
    }
    for (int64_t m = 0; m < n; ++m) t[(size_t)m] = twiddle_f(m, n);
    return t;
}

// e^{+i pi 3 m^2 / M} = conj(e^{-2 pi i r / 2M}), r = 3 m^2 mod 2M reduced exactly in integers, by S2b (trig_d.h): the same
// bits as the device's generation of the window table (k_cq_tables.hip) and as the oracle
void chirp_d(int64_t m, int64_t big_m, double &c, double &s)
{
    const int64_t mm = m < 0 ? -m : m;
    const int64_t r = (int64_t)(((__int128)3 * mm * mm) % (2 * big_m));
    double re, im;
    unit_d(r, 2 * big_m, re, im);
    c = re;
    s = -im;
}

// S5: iterative radix-2 decimation-in-time FFT in double, twiddles from twiddle_d, butterfly
// t = w b (4 mul, 1 sub, 1 add), a' = a + t, b' = a - t; nothing fused.
void fft_r2_double(std::vector<double> &re, std::vector<double> &im)
{
    const int64_t n = (int64_t)re.size();
    for (int64_t i = 1, j = 0; i < n; ++i) {
        int64_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    const std::vector<double> *tab = twiddle_d_table(n);
    for (int64_t len = 2; len <= n; len <<= 1) {
        const int64_t half = len >> 1, ts = n / len;
        for (int64_t j = 0; j < half; ++j) {
            double wr, wi;
            if (tab) {
                wr = (*tab)[(size_t)(2 * ts * j)];
                wi = (*tab)[(size_t)(2 * ts * j + 1)];
            } else {
                twiddle_d(ts * j, n, wr, wi);
            }
            for (int64_t base = 0; base < n; base += len) {
                const int64_t ia = base + j, ib = ia + half;
                const double tr = wr * re[ib] - wi * im[ib];
                const double ti = wr * im[ib] + wi * re[ib];
                re[ib] = re[ia] - tr;
                im[ib] = im[ia] - ti;
                re[ia] = re[ia] + tr;
                im[ia] = im[ia] + ti;
            }
        }
    }
}

// DFT of length 2^a or 3 * 2^a in double (S5): for 3 m points, X[k] = (F0[k mod m] + W^k F1[k mod m]) +
// W^2k F2[k mod m] with F_r the radix-2 transform of x[3 t + r] and W = e^{-2 pi i / (3 m)}
void dft_double(std::vector<double> &re, std::vector<double> &im)
{
    const int64_t n = (int64_t)re.size();
    if (n % 3 != 0) {
        fft_r2_double(re, im);
        return;
    }
    const int64_t m = n / 3;
    std::vector<double> fr[3], fi[3];
    for (int r = 0; r < 3; ++r) {
        fr[r].resize((size_t)m);
        fi[r].resize((size_t)m);
        for (int64_t t = 0; t < m; ++t) {
            fr[r][(size_t)t] = re[(size_t)(3 * t + r)];
            fi[r][(size_t)t] = im[(size_t)(3 * t + r)];
        }
        fft_r2_double(fr[r], fi[r]);
    }
    const std::vector<double> *tab = twiddle_d_table(n);
    for (int64_t k = 0; k < n; ++k) {
        const size_t km = (size_t)(k % m);
        double w1r, w1i, w2r, w2i;
        if (tab) {
            const size_t k2 = (size_t)((2 * k) % n);
            w1r = (*tab)[(size_t)(2 * k)];
            w1i = (*tab)[(size_t)(2 * k + 1)];
            w2r = (*tab)[2 * k2];
            w2i = (*tab)[2 * k2 + 1];
        } else {
            twiddle_d(k, n, w1r, w1i);
            twiddle_d((2 * k) % n, n, w2r, w2i);
        }
        const double ar = fr[0][km] + (w1r * fr[1][km] - w1i * fi[1][km]);
        const double ai = fi[0][km] + (w1r * fi[1][km] + w1i * fr[1][km]);
        re[(size_t)k] = ar + (w2r * fr[2][km] - w2i * fi[2][km]);
        im[(size_t)k] = ai + (w2r * fi[2][km] + w2i * fr[2][km]);
    }
}

// chirp-z length of a band (S7): the smallest of {2^a, 3 * 2^a}, at least 64, that holds `need` points
int64_t chirpz_length(int64_t need)
{
    int64_t p2 = 64, p3 = 96;
    while (p2 < need) p2 <<= 1;
    while (p3 < need) p3 <<= 1;
    return p3 < p2 ? p3 : p2;
}
} // namespace

// Twiddles of one fused (r1, r2) DIF group at sub-length len of a length-n transform, laid out
// [entry pair][butterfly][2] (nb = n / (r1 r2) butterflies): entry q2 (r1-1) + (s-1) = T_n[ts1 (j0 + q2 m2) s],
// entry (r1-1) r2 + (s2-1) = T_n[ts2 j0 s2], with j0 = b mod m2.  The inverse DIT group uses the
// same values (conjugated by the kernel).
void append_group_twiddles(const std::vector<HostCf> &tw, int64_t n, int64_t len, int r1, int r2,
                           std::vector<HostCf> &out)
{
    // the entries of butterfly b depend on j0 = b mod m2 only: the table holds the m2 butterflies of one block (the inner
    // groups' tables are a few KB and stay in the L1 of every CU)
    const int64_t m1 = len / r1, m2 = m1 / r2, ts1 = n / len, ts2 = n / m1, nb = m2;
    const size_t off = out.size();
    const int entries = (r1 - 1) * r2 + (r2 - 1);
    // entry e of butterfly b at [e >> 1][b][e & 1] (device_math.h tw_entry); an odd count leaves the last half empty
    out.resize(off + (size_t)((entries + 1) / 2) * 2 * nb, HostCf{0.0f, 0.0f});
    auto at = [&](int e, int64_t b) -> HostCf & { return out[off + ((size_t)(e >> 1) * nb + b) * 2 + (e & 1)]; };
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t j0 = b;
        for (int q2 = 0; q2 < r2; ++q2)
            for (int s = 1; s < r1; ++s) at(q2 * (r1 - 1) + (s - 1), b) = tw[(size_t)(ts1 * (j0 + q2 * m2) * s)];
        for (int s2 = 1; s2 < r2; ++s2) at((r1 - 1) * r2 + (s2 - 1), b) = tw[(size_t)(ts2 * j0 * s2)];
    }
}

void twiddle_d(int64_t m, int64_t n, double &re, double &im)
{
    m %= n;
    if (m < 0) m += n;
    const int64_t a = 8 * m;
    const int oct = (int)(a / n);
    const int64_t r = a - (int64_t)oct * n;
    const int64_t t = (oct & 1) ? (n - r) : r;
    const double alpha = M_PI * (double)t / (double)(4 * n);
    double ca, sa;
    sincos(alpha, &sa, &ca); // glibc's sincos in the library and in the oracle: cos() and sin() differ from it in the last bit for 0.1 % of arguments
    double c, s;
    switch (oct) {
    case 0: c = ca; s = sa; break;
    case 1: c = sa; s = ca; break;
    case 2: c = -sa; s = ca; break;
    case 3: c = -ca; s = sa; break;
    case 4: c = -ca; s = -sa; break;
    case 5: c = -sa; s = -ca; break;
    case 6: c = sa; s = -ca; break;
    default: c = ca; s = -sa; break;
    }
    re = c;
    im = -s;
}

int64_t digit_pos(int64_t k, int64_t n, const std::vector<int> &radix)
{
    int64_t pos = 0, len = n;
    for (int r : radix) {
        len /= r;
        pos += (k % r) * len;
        k /= r;
    }
    return pos;
}

// primes descending, pairs of 2 merged into 4: [7.. 5.. 4.. 3.. 2]
bool make_radix_list(int64_t n, std::vector<int> &radix)
{
    int c2 = 0, c3 = 0, c5 = 0, c7 = 0;
    while (n % 7 == 0) { n /= 7; ++c7; }
    while (n % 5 == 0) { n /= 5; ++c5; }
    while (n % 3 == 0) { n /= 3; ++c3; }
    while (n % 2 == 0) { n /= 2; ++c2; }
    if (n != 1) return false;
    radix.clear();
    radix.insert(radix.end(), c7, 7);
    radix.insert(radix.end(), c5, 5);
    radix.insert(radix.end(), c2 / 2, 4);
    radix.insert(radix.end(), c3, 3);
    if (c2 & 1) radix.push_back(2);
    return radix.size() <= 24;
}

// fn(i) for i in [0, n) on a small team of host threads: the table entries are pure functions of their
// index, so the result does not depend on the split (a new clip length costs one table build, and a
// corpus of full-length tracks brings a new length with almost every file)
thread_local bool g_plan_serial = false; // PlanSerial: the caller is itself one of many host threads

template <class F>
static void parallel_rows(int64_t n, F fn)
{
    unsigned team = g_plan_serial ? 1u : std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if ((int64_t)team > n) team = (unsigned)n;
    if (team <= 1) {
        for (int64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int64_t> next{0};
    auto work = [&] {
        for (int64_t i; (i = next.fetch_add(1)) < n;) fn(i);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < team; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

PlanSerial::PlanSerial() : before(g_plan_serial) { g_plan_serial = true; }
PlanSerial::~PlanSerial() { g_plan_serial = before; }

bool build_plan(int64_t n, HostPlan &p, std::string &why, bool geometry_only, bool force_bluestein, unsigned conv, bool host_windows)
{
    if (n < 2) {
        why = "clip too short";
        return false;
    }
    p = HostPlan();
    p.n = n;
    // ---- bands ----
    const double fftres = kSampleRate / (double)n;
    const double q = std::pow(2.0, 1.0 / kBpo) - std::pow(2.0, -1.0 / kBpo);
    const int nb = (int)std::floor(kBpo * std::log2(kMaxFreq / kMinFreq)) + 1;
    if (nb != kBins) {
        why = "band count mismatch";
        return false;
    }
    int64_t kmin = n, kmax = 0, big_m = 0;
    for (int j = 0; j < kBins; ++j) {
        const double f = kMinFreq * std::pow(2.0, (double)j / kBpo);
        int64_t posit;
        double bw; // window length before rounding
        if (conv & kConvFloatGeometry) { // essentia's Real is float
            const float fres = (float)kSampleRate / (float)n;
            const float qf = std::pow(2.0f, 1.0f / (float)kBpo) - std::pow(2.0f, -1.0f / (float)kBpo);
            const float ff = (float)kMinFreq * std::pow(2.0f, (float)j / (float)kBpo);
            posit = (int64_t)std::floor(ff / fres);
            bw = (double)(qf * ff / fres);
        } else {
            posit = (int64_t)std::floor(f / fftres);
            bw = q * f / fftres;
        }
        int64_t lg = (conv & kConvLgHalfEven) ? (int64_t)std::nearbyint(bw) : (int64_t)std::round(bw);
        if (lg < kMinWindow) lg = kMinWindow;
        const int64_t st = posit - lg / 2;
        p.start[j] = (int)st;
        p.lg[j] = (int)lg;
        if (st < kmin) kmin = st;
        if (st + lg > kmax) kmax = st + lg;
        if (lg > big_m) big_m = lg;
    }
    if (kmin < 0 || kmax > n / 2) {
        why = "clip too short: constant-Q bands leave the positive half spectrum";
        return false;
    }
    p.kmin = (int)kmin;
    p.kmax = (int)kmax;
    p.m = (int)big_m;
    p.c = (int)((big_m + kDown - 1) / kDown);
    p.n_frames = p.c - (kCtx - 1);
    p.n_hp = p.n_frames - kLag;
    if (p.n_frames < 0) p.n_frames = 0;
    if (p.n_hp < 0) p.n_hp = 0;
    if (chirpz_length(big_m + p.c - 1) > (1 << 19)) { // the longest band's transform; checked before any table is built
        why = "clip too long: chirp-z length above 2^19 (about 25 minutes)";
        return false;
    }
    // ---- forward split N = n1 * n2 ----
    // d0 = the smallest divisor with N / d0 <= kN2Max; among the divisors in [d0, 5 d0 / 4] prefer an
    // even n2 (its twiddle table halves exactly), then the fewest pairs of radix passes, then the
    // smallest n1
    int64_t n1 = 0, d0 = 0;
    int best_odd = 2, best_groups = 1 << 30;
    {
        int64_t rest = n;
        for (int f : {2, 3, 5, 7})
            while (rest % f == 0) rest /= f;
        p.bluestein = force_bluestein || rest != 1;
    }
    if (p.bluestein) {
        // S15: a chirp-z convolution of length L = n1 * 6300 >= N + (kmax - kmin) - 1 with n1 = 16 a, the smallest
        // that fits: the first transform's column stage splits into transforms of length a and 16 (k_bluestein.hip)
        const int64_t need = n + (kmax - kmin) - 1;
        n1 = (need + 16 * 6300 - 1) / (16 * 6300) * 16;
        p.bz_l = n1 * 6300;
    }
    for (int64_t d = 1; !p.bluestein && d <= n; ++d) {
        if (n % d || n / d > kN2Max) continue;
        if (d0 == 0) d0 = d;
        if (4 * d > 5 * d0) break;
        std::vector<int> r;
        if (!make_radix_list(n / d, r)) break; // not 7-smooth: reported below
        const int odd = (int)((n / d) & 1), groups = ((int)r.size() + 1) / 2;
        if (odd < best_odd || (odd == best_odd && groups < best_groups)) {
            best_odd = odd;
            best_groups = groups;
            n1 = d;
        }
    }
    if (n1 == 0) {
        why = "clip length has a prime factor other than 2, 3, 5, 7";
        return false;
    }
    const int64_t n2 = p.bluestein ? 6300 : n / n1;
    std::vector<int> tmp;
    std::vector<int> desc;
    if (!make_radix_list(n2, desc) || (!p.bluestein && !make_radix_list(n1, tmp))) {
        why = "clip length has a prime factor other than 2, 3, 5, 7";
        return false;
    }
    // pass order of the row transform: the descending list taken alternately from its front and its
    // back ([7,7,5,5,4,3] -> [7,3,7,4,5,5]) so that neighbouring passes have small products
    for (size_t lo = 0, hi = desc.size(); lo < hi;) {
        p.radix.push_back(desc[lo++]);
        if (lo < hi) p.radix.push_back(desc[--hi]);
    }
    if (n1 > 8192) {
        why = "clip too long";
        return false;
    }
    p.n1 = (int)n1;
    p.n2 = (int)n2;
    p.h = (int)(n2 / 2 + 1);
    p.k1lo = (int)(kmin / n2);
    p.k1hi = (int)((kmax - 1) / n2);
    if (geometry_only) return true;
    p.tw_n2 = twiddle_table(n2);
    p.tw_n1 = twiddle_table(n1);
    // fuse consecutive passes in pairs of at most 36 points (kernel-side scheduling, same arithmetic;
    // a 49-point pair spills registers)
    for (size_t i = 0; i < p.radix.size();) {
        if (i + 1 < p.radix.size() && p.radix[i] * p.radix[i + 1] <= 36) {
            p.groups.push_back({p.radix[i], p.radix[i + 1]});
            i += 2;
        } else {
            p.groups.push_back({p.radix[i], 1});
            i += 1;
        }
    }
    if (p.groups.size() > 12) {
        why = "too many radix passes";
        return false;
    }
    // per-butterfly twiddle tables of the fused groups (values copied from T_n2)
    {
        int64_t len = n2;
        for (const auto &g : p.groups) {
            p.rows_gtw_off.push_back((int)p.rows_gtw.size());
            append_group_twiddles(p.tw_n2, n2, len, g.first, g.second, p.rows_gtw);
            len /= g.first * g.second;
        }
    }
    if (!p.bluestein) { // the chirp-z path's own tables (chirp, T_L, Bhat, w[k] / L) are generated on the device
        // S6: fixed-point twiddles of the column stage, their digits as the int8 matrix instruction's A operand, the
        // correction for the samples' digit offset, and the twiddles between the stages
        const int64_t hq = n1 / 2 + 1;
        p.hq = (int)hq;
        p.q2lo = (int)(kmin / n1);
        p.q2w = (int)((kmax - 1) / n1) - p.q2lo + 1;
        p.cols_mt = (int)((2 * hq + 31) / 32);
        p.cols_ks = (int)((n1 + 31) / 32);
        p.wq.resize((size_t)2 * n1);
        for (int64_t m = 0; m < n1; ++m) {
            double c, s;
            twiddle_d(m, n1, c, s);
            p.wq[(size_t)(2 * m)] = (int32_t)std::rint(c * 4194304.0);
            p.wq[(size_t)(2 * m + 1)] = (int32_t)std::rint(s * 4194304.0);
        }
        p.cols_corr.assign((size_t)2 * hq, 0.0);
        for (int64_t q1 = 0; q1 < hq; ++q1) {
            int64_t sr = 0, si = 0;
            for (int64_t k1 = 0; k1 < n1; ++k1) {
                sr += p.wq[(size_t)(2 * ((q1 * k1) % n1))];
                si += p.wq[(size_t)(2 * ((q1 * k1) % n1) + 1)];
            }
            p.cols_corr[(size_t)(2 * q1)] = (double)(128 * sr); // |.| < 2^43: exact
            p.cols_corr[(size_t)(2 * q1 + 1)] = (double)(128 * si);
        }
        // image [tile][step][digit][lane][16]: byte e of lane l = digit of W[row = 32 tile + (l & 31)][k1 = 32 step + 16 (l >> 5) + e],
        // W[2 q1] = Re wq[(q1 k1) mod n1], W[2 q1 + 1] = Im; zero beyond the hq rows and the n1 samples
        p.cols_image.assign((size_t)p.cols_mt * p.cols_ks * 3 * 1024, 0);
        parallel_rows(p.cols_mt, [&](int64_t mt) {
            for (int64_t ks = 0; ks < p.cols_ks; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 16; ++e) {
                        const int64_t row = 32 * mt + (l & 31), k1 = 32 * ks + 16 * (l >> 5) + e, q1 = row >> 1;
                        if (q1 >= hq || k1 >= n1) continue;
                        const int w = p.wq[(size_t)(2 * ((q1 * k1) % n1) + (row & 1))];
                        const int d0 = ((w + 128) & 255) - 128, w1 = (w - d0) >> 8, d1 = ((w1 + 128) & 255) - 128, d2 = (w1 - d1) >> 8;
                        const int d[3] = {d0, d1, d2};
                        for (int i = 0; i < 3; ++i)
                            p.cols_image[((((size_t)mt * p.cols_ks + ks) * 3 + i) * 64 + l) * 16 + e] = (int8_t)d[i];
                    }
        });
        const int64_t nq = (n2 + 3) / 4;
        p.ts_seed.resize((size_t)(hq * nq));
        p.ts_step.resize((size_t)(hq * 4));
        parallel_rows(hq, [&](int64_t q1) {
            for (int64_t m = 0; m < nq; ++m) {
                const HostCf t = twiddle_f((q1 * 4 * m) % n, n);
                p.ts_seed[(size_t)(q1 * nq + m)] = {std::ldexp(t.r, -37), std::ldexp(t.i, -37)};
            }
            p.ts_step[(size_t)(q1 * 4)] = {1.0f, 0.0f};
            for (int e = 1; e < 4; ++e) p.ts_step[(size_t)(q1 * 4 + e)] = twiddle_f((q1 * e) % n, n);
        });
    }
    p.pos_n2.resize((size_t)n2);
    for (int64_t k = 0; k < n2; ++k) p.pos_n2[(size_t)k] = (int)digit_pos(k, n2, p.radix);
    // the last fused group (R1, R2) works on blocks of len = R1 R2 consecutive positions; block b holds
    // the outputs k2 = kb + nb f, f = s + R1 s2 at position len b + s R2 + s2 (nb = n2 / len)
    {
        const int r1 = p.groups.back().first, r2 = p.groups.back().second, len = r1 * r2;
        const int nb = (int)(n2 / len);
        p.kb_last.assign((size_t)nb, -1);
        for (int k = 0; k < nb; ++k) p.kb_last[(size_t)(p.pos_n2[(size_t)k] / len)] = k;
        for (int b = 0; b < nb; ++b)
            for (int s = 0; s < r1; ++s)
                for (int s2 = 0; s2 < r2; ++s2)
                    if (p.kb_last[(size_t)b] < 0 ||
                        p.pos_n2[(size_t)(p.kb_last[(size_t)b] + nb * (s + r1 * s2))] != len * b + s * r2 + s2) {
                        why = "internal: last-group output map";
                        return false;
                    }
    }
    // ---- chirp-z classes ----
    p.g_off.assign(kBins, 0);
    int64_t goff = 0;
    for (int j = 0; j < kBins; ++j) {
        const int64_t need = p.lg[j] + p.c - 1;
        const int64_t ps = chirpz_length(need);
        if (ps > (1 << 19)) {
            why = "clip too long: chirp-z length above 2^19";
            return false;
        }
        p.psize[j] = (int)ps;
        size_t k = 0;
        for (; k < p.classes.size(); ++k)
            if (p.classes[k].p == ps) break;
        if (k == p.classes.size()) {
            BluesteinClass bc;
            bc.p = (int)ps;
            p.classes.push_back(std::move(bc));
        }
        p.classes[k].bands.push_back(j);
        p.g_off[j] = goff;
        goff += p.lg[j];
    }
    // the tables of every class (butterfly twiddles, DFT of the chirp), classes side by side
    std::atomic<bool> bad_class{false};
    parallel_rows((int64_t)p.classes.size(), [&](int64_t ci) {
        BluesteinClass &bc = p.classes[(size_t)ci];
        const int64_t ps = bc.p;
        make_radix_list(ps, bc.radix);
        bc.tw = twiddle_table(ps);
        // per-butterfly twiddle tables of the fused groups, in the order the kernel walks them
        // (fft_lds.h GroupOf): pairs of passes of the list [4.., 3, 2]; the innermost group's
        // entries do not depend on the butterfly and are stored once
        {
            auto next_radix = [](int64_t len) { return len % 4 == 0 ? 4 : (len % 3 == 0 ? 3 : 2); };
            // above 16384 points: peel radix-4 passes off the front until a block fits the LDS
            int64_t len = ps;
            while (ps > 16384 && len > 8192) {
                len /= 4;
                ++bc.outer;
            }
            bc.len0 = (int)len;
            int g = 0;
            for (;;) {
                const int r1 = next_radix(len);
                const int r2 = len / r1 > 1 ? next_radix(len / r1) : 1;
                const int64_t rest = len / (r1 * r2);
                if (rest == 1) {
                    std::vector<HostCf> tmp;
                    append_group_twiddles(bc.tw, ps, len, r1, r2, tmp);
                    bc.mid_off = (int)bc.gtw.size();
                    for (int e = 0; e < (r1 - 1) * r2; ++e) bc.gtw.push_back(tmp[(size_t)e]); // m2 = 1: one butterfly, entries in order
                    break;
                }
                if (g >= 4) {
                    bad_class = true;
                    return;
                }
                bc.goff[g++] = (int)bc.gtw.size();
                append_group_twiddles(bc.tw, ps, len, r1, r2, bc.gtw);
                len /= r1 * r2;
            }
        }
        std::vector<double> re((size_t)ps, 0.0), im((size_t)ps, 0.0);
        for (int64_t mm = -(ps - p.c); mm <= p.c - 1; ++mm) { // v[m mod P] = e^{-i pi 3 m^2 / M}
            double cc, ss;
            chirp_d(mm, big_m, cc, ss);
            const int64_t idx = mm < 0 ? mm + ps : mm;
            re[(size_t)idx] = cc;
            im[(size_t)idx] = -ss;
        }
        dft_double(re, im);
        bc.vrev.resize((size_t)ps);
        for (int64_t kk = 0; kk < ps; ++kk) {
            const int64_t pos = digit_pos(kk, ps, bc.radix);
            bc.vrev[(size_t)pos] = {(float)re[(size_t)kk], (float)im[(size_t)kk]};
        }
    });
    if (bad_class) {
        why = "internal: too many fused groups";
        return false;
    }
    // G_j[i] = hann_Lg[i] e^{+i pi 3 i^2 / M} / (M P): S5, trig_d.h cq_window_d -- the product generates this table on the
    // device (k_cq_tables.hip: a corpus of tracks brings a new length with every file, and the table was most of a length's
    // host cost and of its upload); the host builds it for the emulation of tests/emu and for the table checksums only
    p.g_total = goff;
    p.big_m = big_m;
    if (!host_windows) return true;
    p.g.resize((size_t)goff);
    parallel_rows(kBins, [&](int64_t j) {
        const int64_t lg = p.lg[j];
        const double scale = cq_window_scale(conv, big_m, p.psize[j]);
        const int64_t hann_den = cq_hann_den(conv, lg);
        HostCf *gj = p.g.data() + p.g_off[(size_t)j];
        for (int64_t i = 0; i < lg; ++i) cq_window_d(i, hann_den, big_m, scale, gj[i].r, gj[i].i);
    });
    return true;
}

} // namespace hpfw

// ---- host-only diagnostic: FNV-1a checksums of the plan tables (no device involved) ----------
namespace {
uint64_t fnv1a(const void *data, size_t bytes, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < bytes; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}
} // namespace

namespace hpfw {

// The row transform alone, for frames of n2 real samples taken in pairs (the STFT of the Mel front-end):
// the same pass order, fused groups, butterfly tables and output positions as the forward transform's
// rows; the table that multiplies the split spectra holds ones (two rows: the pair's two frames).
bool build_frame_transform(int n2, HostPlan &p, std::string &why)
{
    p = HostPlan();
    std::vector<int> desc;
    if (n2 < 2 || n2 > kN2Max || !make_radix_list(n2, desc)) {
        why = "frame length not supported";
        return false;
    }
    for (size_t lo = 0, hi = desc.size(); lo < hi;) {
        p.radix.push_back(desc[lo++]);
        if (lo < hi) p.radix.push_back(desc[--hi]);
    }
    p.n = n2;
    p.n1 = 2;
    p.n2 = n2;
    p.h = n2 / 2 + 1;
    p.tw_n2 = twiddle_table(n2);
    for (size_t i = 0; i < p.radix.size();) {
        if (i + 1 < p.radix.size() && p.radix[i] * p.radix[i + 1] <= 36) {
            p.groups.push_back({p.radix[i], p.radix[i + 1]});
            i += 2;
        } else {
            p.groups.push_back({p.radix[i], 1});
            i += 1;
        }
    }
    int64_t len = n2;
    for (const auto &g : p.groups) {
        p.rows_gtw_off.push_back((int)p.rows_gtw.size());
        append_group_twiddles(p.tw_n2, n2, len, g.first, g.second, p.rows_gtw);
        len /= g.first * g.second;
    }
    p.tw_big.assign((size_t)2 * p.h, HostCf{1.0f, 0.0f});
    p.pos_n2.resize((size_t)n2);
    for (int64_t k = 0; k < n2; ++k) p.pos_n2[(size_t)k] = (int)digit_pos(k, n2, p.radix);
    { // the last group's output map, as in build_plan (the compile-time group sequences use it)
        const int r1 = p.groups.back().first, r2 = p.groups.back().second, glen = r1 * r2;
        const int nb = n2 / glen;
        p.kb_last.assign((size_t)nb, -1);
        for (int k = 0; k < nb; ++k) p.kb_last[(size_t)(p.pos_n2[(size_t)k] / glen)] = k;
        for (int b = 0; b < nb; ++b)
            for (int s = 0; s < r1; ++s)
                for (int s2 = 0; s2 < r2; ++s2)
                    if (p.kb_last[(size_t)b] < 0 ||
                        p.pos_n2[(size_t)(p.kb_last[(size_t)b] + nb * (s + r1 * s2))] != glen * b + s * r2 + s2) {
                        why = "internal: last-group output map";
                        return false;
                    }
    }
    return true;
}

// Tables of the Mel front-end: essentia Windowing "hann" of 4410 points normalised to
// area 2, and MelBands(2206 -> 33; 0..22050 Hz, htkMel, weighting "warping", normalize "unit_sum") as the
// MFMA operand image cpack [2208 bins][64 rows] (rows >= 33 and bins >= 2206 are zero).
void mel_tables(std::vector<float> &window, std::vector<float> &cpack)
{
    const int n = 4410, nbins = 2206, nbands = 33;
    window.resize((size_t)n);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)i / (double)(n - 1));
    for (int i = 0; i < n; ++i)
        window[(size_t)i] = (float)((0.5 - 0.5 * std::cos(2.0 * M_PI * (double)i / (double)(n - 1))) * (2.0 / sum));
    auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> fb((size_t)nbands + 2);
    const double lo = hz2mel(0.0), hi = hz2mel(22050.0), inc = (hi - lo) / (nbands + 1);
    for (int i = 0; i < nbands + 2; ++i) fb[(size_t)i] = mel2hz(lo + inc * i);
    const double fscale = (44100.0 / 2.0) / (double)(nbins - 1);
    cpack.assign((size_t)2208 * 64, 0.0f);
    std::vector<double> c((size_t)nbins);
    for (int i = 0; i < nbands; ++i) {
        const double w0 = hz2mel(fb[(size_t)i]), w1 = hz2mel(fb[(size_t)i + 1]), w2 = hz2mel(fb[(size_t)i + 2]);
        const int jb = (int)(fb[(size_t)i] / fscale + 0.5), je = (int)(fb[(size_t)i + 2] / fscale + 0.5);
        double weight = 0.0;
        std::fill(c.begin(), c.end(), 0.0);
        for (int j = jb; j <= je && j < nbins; ++j) {
            const double bf = j * fscale;
            if (bf >= fb[(size_t)i] && bf < fb[(size_t)i + 1]) c[(size_t)j] = (hz2mel(bf) - w0) / (w1 - w0);
            else if (bf >= fb[(size_t)i + 1] && bf < fb[(size_t)i + 2]) c[(size_t)j] = (w2 - hz2mel(bf)) / (w2 - w1);
            weight += c[(size_t)j];
        }
        for (int j = 0; j < nbins; ++j) cpack[(size_t)j * 64 + i] = (float)(weight > 0.0 ? c[(size_t)j] / weight : 0.0);
    }
}

} // namespace hpfw

// the smallest supported clip length >= n_samples (include/hpfw_gpu.h), or -1: every length from the
// shortest clip that yields a hashprint up to the longest the tables allow is supported (lengths with prime
// factors above 7 take the chirp-z forward transform), so this is n_samples itself inside that range
extern "C" int64_t hpfw_gpu_supported_length(int64_t n_samples)
{
    for (int64_t n = n_samples > 2 ? n_samples : 2; n <= (int64_t)44100 * 1600; ++n) {
        hpfw::HostPlan hp;
        std::string why;
        if (hpfw::build_plan(n, hp, why, true)) {
            if (hp.n_hp > 0) return n;
        } else if (why.find("too long") != std::string::npos) {
            return -1;
        }
    }
    return -1;
}

extern "C" int hpfw_gpu_plan_checksum_ex(int64_t n_samples, int force_bluestein, unsigned conventions, uint64_t *out8);

extern "C" int hpfw_gpu_plan_checksum(int64_t n_samples, uint64_t *out8)
{
    // negative length: the chirp-z tables of |n_samples| even when it is 7-smooth
    return hpfw_gpu_plan_checksum_ex(n_samples < 0 ? -n_samples : n_samples, n_samples < 0, 0, out8);
}

extern "C" int hpfw_gpu_plan_checksum_ex(int64_t n_samples, int force_bluestein, unsigned conventions, uint64_t *out8)
{
    hpfw::HostPlan p;
    std::string why;
    if (!out8 || conventions > hpfw::kConvAll || !hpfw::build_plan(n_samples, p, why, false, force_bluestein != 0, conventions)) return -2;
    out8[0] = fnv1a(p.tw_n2.data(), p.tw_n2.size() * 8);
    out8[1] = fnv1a(p.tw_n1.data(), p.tw_n1.size() * 8);
    {   // the full table as the row stage forms it (S1 product of seed and step), row by row: the oracle's table
        uint64_t ht = 1469598103934665603ull;
        const int64_t nq = (p.n2 + 3) / 4;
        for (int64_t q1 = 0; q1 < p.hq; ++q1)
            for (int64_t k2 = 0; k2 < p.n2; ++k2) {
                const hpfw::HostCf a = p.ts_seed[(size_t)(q1 * nq + k2 / 4)], w = p.ts_step[(size_t)(q1 * 4 + (k2 & 3))];
                float pair[2] = {a.r, a.i};
                if (k2 & 3) {
                    const float pp = a.i * w.i, qq = a.i * w.r;
                    pair[0] = __builtin_fmaf(a.r, w.r, -pp);
                    pair[1] = __builtin_fmaf(a.r, w.i, qq);
                }
                ht = fnv1a(pair, 8, ht);
            }
        out8[2] = fnv1a(p.wq.data(), p.wq.size() * 4, ht);
    }
    // (the twiddles between the two stages, then the column stage's fixed-point twiddles.)  chirp-z lengths: neither
    // (slot 2 = the hash of nothing); the tables that stand in their place are generated on the device and checked there
    // (hpfw_gpu_chirpz_table_checksums)
    out8[3] = fnv1a(p.pos_n2.data(), p.pos_n2.size() * 4);
    uint64_t h = fnv1a(p.start, sizeof(p.start));
    h = fnv1a(p.lg, sizeof(p.lg), h);
    out8[4] = fnv1a(p.psize, sizeof(p.psize), h);
    out8[5] = fnv1a(p.g.data(), p.g.size() * 8);
    uint64_t ht = 1469598103934665603ull, hv = ht;
    for (const hpfw::BluesteinClass &bc : p.classes) {
        ht = fnv1a(bc.tw.data(), bc.tw.size() * 8, ht);
        hv = fnv1a(bc.vrev.data(), bc.vrev.size() * 8, hv);
    }
    out8[6] = ht;
    out8[7] = hv;
    return 0;
}
