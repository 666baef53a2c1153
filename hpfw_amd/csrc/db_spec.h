// db_spec.h -- the dB term of DESIGN.md S8, t(p) = (float)(10 log10_d(max(p, 1e-10))), shared by the
// kernels that produce spectrogram values (convert.h:7-16 of the reference).
#pragma once
#include <hip/hip_runtime.h>

namespace hpfw {

// log10 in double by a fixed sequence of IEEE operations
__device__ __forceinline__ double log10_spec(double x)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double r = 1.0 / 23.0;
    r = __builtin_fma(r, z, 1.0 / 21.0);
    r = __builtin_fma(r, z, 1.0 / 19.0);
    r = __builtin_fma(r, z, 1.0 / 17.0);
    r = __builtin_fma(r, z, 1.0 / 15.0);
    r = __builtin_fma(r, z, 1.0 / 13.0);
    r = __builtin_fma(r, z, 1.0 / 11.0);
    r = __builtin_fma(r, z, 1.0 / 9.0);
    r = __builtin_fma(r, z, 1.0 / 7.0);
    r = __builtin_fma(r, z, 1.0 / 5.0);
    r = __builtin_fma(r, z, 1.0 / 3.0);
    r = __builtin_fma(r, z, 1.0);
    const double lm = 2.0 * s * r;
    return __builtin_fma((double)e, 0.30102999566398119521, lm * 0.43429448190325182765);
}

__device__ __forceinline__ float db_term(float pw)
{
    const float xx = pw < 1e-10f ? 1e-10f : pw;
    return (float)(10.0 * log10_spec((double)xx));
}

} // namespace hpfw
