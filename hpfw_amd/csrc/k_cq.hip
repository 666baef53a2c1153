// k_cq.hip -- the inverse half of a2, a3 and a4: forward DFT bins -> |c_j[3c]| -> dB spectrogram.
//
// Replaces, per constant-Q band j, essentia NSGConstantQ::compute's window multiply + length-M
// inverse FFT, and hpfw's magnitude / every-third-sample loop (reference
// include/hpfw/spectrum/cqt.h:66-81), then amplitude_to_db (include/hpfw/spectrum/convert.h:7-25).
// M = 7255 = 5 * 1451 for a 30 s clip and only samples 3c are kept, so each band is a chirp-z
// (Bluestein) transform of power-of-two length P_j >= Lg_j + C - 1 held entirely in LDS
// (DESIGN.md S7): a = X[s_j + i] G_j[i]; A = FFT_P(a); B = A . V_P; b = IFFT_P(B); |b[c]|.
#include "kernels.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

constexpr int kCqThreads = 256;

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
    return v;
}

__global__ __launch_bounds__(kCqThreads) void cq_kernel(CqPlanDev cp, CqClassDev cc, const cf *__restrict__ x,
                                                        float *__restrict__ mag, unsigned *__restrict__ magmax)
{
    cf *a = reinterpret_cast<cf *>(smem_raw);
    const int tid = threadIdx.x;
    const int j = cc.band[blockIdx.x];
    const int clip = blockIdx.y;
    const int lg = cp.lg[j];
    const cf *xs = x + (int64_t)clip * cp.nk + (cp.start[j] - cp.kmin);
    const cf *g = cp.g + cp.g_off[j];
    for (int i = tid; i < cc.p; i += kCqThreads) {
        cf v = {0.0f, 0.0f};
        if (i < lg) v = c_mul(xs[i], g[i]);
        a[i] = v;
    }
    __syncthreads();
    lds_fft_dif(a, cc.p, cc.radix, cc.tw, tid, kCqThreads);
    for (int i = tid; i < cc.p; i += kCqThreads) a[i] = c_mul(a[i], cc.vrev[i]);
    __syncthreads();
    lds_fft_idit(a, cc.p, cc.radix, cc.tw, tid, kCqThreads);
    float *out = mag + ((int64_t)clip * kBins + j) * cp.c;
    float mx = 0.0f;
    for (int i = tid; i < cp.c; i += kCqThreads) {
        const cf v = a[i];
        const float m = sqrtf(__builtin_fmaf(v.r, v.r, v.i * v.i));
        out[i] = m;
        mx = fmaxf(mx, m);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) atomicMax(&magmax[clip], __float_as_uint(mx)); // mag >= 0: bit order = value order
}

// per-clip maximum for the stage entry point that starts from given magnitudes
__global__ __launch_bounds__(256) void magmax_kernel(const float *__restrict__ mag, int64_t per_clip,
                                                     unsigned *__restrict__ magmax)
{
    const int clip = blockIdx.y;
    const float *m = mag + (int64_t)clip * per_clip;
    float mx = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (int64_t)gridDim.x * 256)
        mx = fmaxf(mx, m[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(&magmax[clip], __float_as_uint(mx));
}

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

// mag and db may be the same buffer (each element is read, then written, by one thread)
__global__ __launch_bounds__(256) void db_kernel(const float *mag, const unsigned *__restrict__ magmax,
                                                 int64_t per_clip, float *db)
{
    __shared__ float ref_s;
    const int clip = blockIdx.y;
    if (threadIdx.x == 0) {
        const float mm = __uint_as_float(magmax[clip]);
        ref_s = db_term(mm * mm);
    }
    __syncthreads();
    const float ref = ref_s;
    const float *m = mag + (int64_t)clip * per_clip;
    float *o = db + (int64_t)clip * per_clip;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (int64_t)gridDim.x * 256) {
        const float v = m[i];
        const float l = db_term(v * v) - ref;
        o[i] = l < -80.0f ? -80.0f : l;
    }
}

static int g_cq_lds_set = 0;

void launch_cq_class(const CqPlanDev &cp, const CqClassDev &cc, const cf *d_x, int n_clips, float *d_mag,
                     unsigned *d_magmax, hipStream_t s)
{
    if (!g_cq_lds_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cq_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        g_cq_lds_set = 1;
    }
    dim3 grid(cc.n_bands, n_clips);
    hipLaunchKernelGGL(cq_kernel, grid, dim3(kCqThreads), (size_t)cc.p * sizeof(cf), s, cp, cc, d_x, d_mag,
                       d_magmax);
}

void launch_magmax(const float *d_mag, int n_clips, int64_t per_clip, unsigned *d_magmax, hipStream_t s)
{
    int bx = (int)((per_clip + 256 * 8 - 1) / (256 * 8));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(magmax_kernel, dim3(bx, n_clips), dim3(256), 0, s, d_mag, per_clip, d_magmax);
}

void launch_db(const float *d_mag, const unsigned *d_magmax, int n_clips, int64_t per_clip, float *d_db,
               hipStream_t s)
{
    int bx = (int)((per_clip + 256 * 4 - 1) / (256 * 4));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(db_kernel, dim3(bx, n_clips), dim3(256), 0, s, d_mag, d_magmax, per_clip, d_db);
}

} // namespace hpfw
