// trig_d.h -- S2b: e^{-2 pi i m / n} in double for tables that are generated on the device AND restated on the host and in
// the oracle.  The octant reduction of S2 (plan.cpp twiddle_d) in integers, then cosine and sine of alpha in [0, pi / 4] as
// Taylor polynomials in alpha^2 evaluated by explicit fma chains (truncation below 1e-19; coefficients = the correctly
// rounded 1 / k!): every operation is an IEEE double multiply or fma, so the device, plan.cpp and the oracle's C restatement
// give the same bits, which a libm call would not promise.
// Used by: the chirp-z forward transform's tables (k_bluestein.hip, S15) and, from round 4 on, the constant-Q stage's
// windows and chirps (S5: k_cq_tables.hip on the device, plan.cpp for the host-side emulation and the table checksums).
#pragma once
#include <cstdint>

#if defined(__HIP__)
#define HPFW_HD __host__ __device__ __forceinline__
#else
#define HPFW_HD inline
#endif

namespace hpfw {

HPFW_HD void cos_sin_d(double x, double &c, double &s)
{
    const double z = x * x;
    double ps = 0x1.952c77030ad4ap-49;                 // 1 / 17!
    ps = __builtin_fma(ps, z, -0x1.ae7f3e733b81fp-41); // -1 / 15!
    ps = __builtin_fma(ps, z, 0x1.6124613a86d09p-33);
    ps = __builtin_fma(ps, z, -0x1.ae64567f544e4p-26);
    ps = __builtin_fma(ps, z, 0x1.71de3a556c734p-19);
    ps = __builtin_fma(ps, z, -0x1.a01a01a01a01ap-13);
    ps = __builtin_fma(ps, z, 0x1.1111111111111p-7);
    ps = __builtin_fma(ps, z, -0x1.5555555555555p-3); // -1 / 3!
    s = __builtin_fma(x * z, ps, x);
    double pc = -0x1.6827863b97d97p-53;               // -1 / 18!
    pc = __builtin_fma(pc, z, 0x1.ae7f3e733b81fp-45); // 1 / 16!
    pc = __builtin_fma(pc, z, -0x1.93974a8c07c9dp-37);
    pc = __builtin_fma(pc, z, 0x1.1eed8eff8d898p-29);
    pc = __builtin_fma(pc, z, -0x1.27e4fb7789f5cp-22);
    pc = __builtin_fma(pc, z, 0x1.a01a01a01a01ap-16);
    pc = __builtin_fma(pc, z, -0x1.6c16c16c16c17p-10);
    pc = __builtin_fma(pc, z, 0x1.5555555555555p-5);
    pc = __builtin_fma(pc, z, -0.5);
    c = __builtin_fma(z, pc, 1.0);
}

// e^{-2 pi i m / n}, 0 <= m < n
HPFW_HD void unit_d(int64_t m, int64_t n, double &re, double &im)
{
    const int64_t a = 8 * m;
    const int oct = (int)(a / n);
    const int64_t r = a - (int64_t)oct * n;
    const int64_t t = (oct & 1) ? (n - r) : r;
    const double alpha = 3.14159265358979323846 * (double)t / (double)(4 * n);
    double ca, sa, c, s;
    cos_sin_d(alpha, ca, sa);
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

// S5, the constant-Q stage's window table: G_j[i] = hann_Lg[i] e^{+i pi 3 i^2 / M} * scale, i < Lg, in double, rounded once.
//   hann_Lg[i] = 0.5 - 0.5 cos(2 pi i / den), den = Lg - 1 (essentia Windowing "hann"; Lg for the periodic convention)
//   e^{+i pi 3 i^2 / M} = conj(e^{-2 pi i r / 2M}), r = 3 i^2 mod 2M (exact in 64 bits: i < 2^20)
HPFW_HD void cq_window_d(int64_t i, int64_t hann_den, int64_t big_m, double scale, float &out_re, float &out_im)
{
    double hc, hs, cc, cs;
    unit_d(i % hann_den, hann_den, hc, hs);
    const double w = 0.5 - 0.5 * hc;
    unit_d((3 * i * i) % (2 * big_m), 2 * big_m, cc, cs);
    out_re = (float)(w * cc * scale);
    out_im = (float)(w * -cs * scale);
    (void)hs;
}

} // namespace hpfw
