// k_cq.hip -- the inverse half of a2, a3 and a4: forward DFT bins -> |c_j[3c]| -> dB spectrogram.
//
// Replaces, per constant-Q band j, essentia NSGConstantQ::compute's window multiply + length-M
// inverse FFT, and hpfw's magnitude / every-third-sample loop (reference
// include/hpfw/spectrum/cqt.h:66-81), then amplitude_to_db (include/hpfw/spectrum/convert.h:7-25).
// M = 7255 = 5 * 1451 for a 30 s clip and only samples 3c are kept, so each band is a chirp-z
// (Bluestein) transform of power-of-two length P_j >= Lg_j + C - 1 held entirely in LDS
// (DESIGN.md S7): a = X[s_j + i] G_j[i]; A = FFT_P(a); B = A . V_P; b = IFFT_P(B); |b[c]|.
#include "kernels.h"
#include "db_spec.h"

namespace hpfw {

extern __shared__ __align__(16) unsigned char smem_raw[];

// the maximum over the wave, in every lane: inside the rows of sixteen lanes on the data-parallel-primitive path (two quad
// permutes, two row rotations -- a permute through the LDS hardware per step cost the workgroup 1.1-1.6 k cycles,
// tools/cq_stamps.py), then the four rows' values read as scalars.  (max is exact and does not depend on the order.)
template <int CTRL>
__device__ __forceinline__ float wave_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, wave_dpp<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmaxf(v, wave_dpp<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmaxf(v, wave_dpp<0x124>(v)); // row_ror:4
    v = fmaxf(v, wave_dpp<0x128>(v)); // row_ror:8
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

constexpr int cq_threads(int n)
{
    const int t = n % 3 == 0 ? n / 12 : n / 16; // one fused 16-point (12-point) butterfly per thread and pass
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

// One workgroup = one band of one clip.  The body (fft_lds.h) is shared with the host-side SIMT
// emulation of tests/emu; here HPFW_FOR_THREADS is the thread itself and HPFW_BARRIER a barrier.
// DBT: store the dB term t(m^2) = (float)(10 log10(max(m^2, 1e-10))) of each magnitude instead of the
// magnitude (extraction: the dB conversion then is S = max(t - t_max, -80) wherever S is read, and
// no separate pass over the spectrogram is needed; the chirp-z has VALU slots to spare for it).
// Registers (round 4, scalar complex arithmetic): 70-91 VGPRs as compiled, no scratch (8192: 73, 6144: 70, 12288: 76,
// 4096: 91, 3072: 76) -- what holds the large classes to 3-4 waves per SIMD is their LDS (70-104 KB per workgroup), not
// registers.  The classes of at most 128 threads (at most two waves per workgroup) are held to five waves = 96 VGPRs, which
// costs no scratch and lets a fifth workgroup onto the CU where the LDS has room.  Held to six -- 80 VGPRs, 44-84 bytes of
// scratch in round 3's build -- the classes whose LDS footprint admits six measured slower, 4.0 against 3.85 ms per 1000 clips.
constexpr int cq_waves(int n) { return cq_threads(n) <= 128 ? 5 : 1; }

#if defined(HPFW_CQ_STAMPS)
static __device__ long long *g_cq_stamps = nullptr; // [clip][121][8]
extern "C" void hpfw_gpu_debug_set_cq_stamps(void *d)
{
    long long *p = static_cast<long long *>(d);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cq_stamps), &p, sizeof(p));
}
#define HPFW_CQ_STP (g_cq_stamps ? g_cq_stamps + ((int64_t)clip * kBins + j) * 8 : nullptr)
#else
#define HPFW_CQ_STP nullptr
#endif

template <int NP, bool DBT>
__global__ __launch_bounds__(cq_threads(NP), cq_waves(NP)) void cq_kernel(CqPlanDev cp, CqClassDev cc,
                                                              const cf *__restrict__ x, float *__restrict__ mag,
                                                              float *__restrict__ wavemax)
{
    using P = Size<NP>;
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    float *red = reinterpret_cast<float *>(lds + P::DATA); // one float per thread behind the data
    const int j = cc.band[blockIdx.x];
    const int clip = blockIdx.y;
    float *out = mag + ((int64_t)clip * kBins + j) * cp.c;
    // natural order (chirp-z forward transform, stage entry point) -- or a band so narrow that a row of the rows layout
    // holds only a few of its bins: walking the slice element by element then costs less than walking n1 short rows
    if (cp.xn1 == 1 || cp.nq2[j] < cp.rows_min) {
        const XsBand xs{cp.view(x, clip), cp.start[j]};
        cq_band_body<NP>(lds, red, cq_threads(NP), xs, cp.g + cp.g_off[j], cp.lg[j], cc.gtw, cc.vrev, cp.c, out,
                           [](float m) { return DBT ? db_term(m * m) : m; }, HPFW_CQ_STP);
    } else {           // rows k mod n1, as the row stage of S6 leaves them
        const XsBandRows xs{x + (int64_t)clip * cp.xclip, cp.xn1, cp.xw, cp.xq0, cp.start[j], cp.q2a[j], cp.nq2[j], cp.nq2_magic[j]};
        cq_band_body<NP>(lds, red, cq_threads(NP), xs, cp.g2 + cp.g2_off[j], cp.lg[j], cc.gtw, cc.vrev, cp.c, out,
                           [](float m) { return DBT ? db_term(m * m) : m; }, HPFW_CQ_STP);
    }
    // wave maximum -> wavemax[clip][band][wave]: plain stores (clipmax_kernel reduces them); one
    // atomicMax per wave on a per-clip word cost 0.4 ms per 1000 clips in contention
    const float mx = wave_max(red[threadIdx.x]);
    float *slot = wavemax + ((int64_t)clip * kBins + j) * kCqMaxWaves;
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = mx;
    if (threadIdx.x < kCqMaxWaves && threadIdx.x >= (blockDim.x >> 6)) slot[threadIdx.x] = -INFINITY; // slots of absent waves
#if defined(HPFW_CQ_STAMPS)
    if (g_cq_stamps && threadIdx.x == 0) {
        long long *stp = HPFW_CQ_STP;
        stp[5] = __builtin_amdgcn_s_memtime();
        stp[6] = NP;
        stp[7] = __builtin_amdgcn_s_getreg((15 << 11) | 4);
    }
#endif
}

// band maxima for the stage entry point that starts from given magnitudes: one workgroup per (band, clip)
__global__ __launch_bounds__(256) void magmax_kernel(const float *__restrict__ mag, int c, float *__restrict__ wavemax)
{
    const int band = blockIdx.x, clip = blockIdx.y;
    const float *m = mag + ((int64_t)clip * kBins + band) * c;
    float mx = 0.0f;
    for (int i = threadIdx.x; i < c; i += 256) mx = fmaxf(mx, m[i]);
    mx = wave_max(mx);
    float *slot = wavemax + ((int64_t)clip * kBins + band) * kCqMaxWaves;
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = mx;
    if (threadIdx.x < kCqMaxWaves && threadIdx.x >= 4) slot[threadIdx.x] = -INFINITY;
}

// clipmax[clip] = max over the clip's wave maxima
__global__ __launch_bounds__(256) void clipmax_kernel(const float *__restrict__ wavemax, float *__restrict__ clipmax)
{
    __shared__ float part[4];
    const int clip = blockIdx.x;
    const float *w = wavemax + (int64_t)clip * kBins * kCqMaxWaves;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < kBins * kCqMaxWaves; i += 256) mx = fmaxf(mx, w[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) clipmax[clip] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// mag and db may be the same buffer (each element is read, then written, by one thread)
__global__ __launch_bounds__(256) void db_kernel(const float *mag, const float *__restrict__ clipmax,
                                                 int64_t per_clip, float *db)
{
    __shared__ float ref_s;
    const int clip = blockIdx.y;
    if (threadIdx.x == 0) {
        const float mm = clipmax[clip];
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

// S = max(t - t_max, -80) in place, for consumers that want the dB spectrogram itself
// (the covariance of filter learning, the stage entry point)
__global__ __launch_bounds__(256) void db_finish_kernel(float *t, const float *__restrict__ clipmax, int64_t per_clip)
{
    const int clip = blockIdx.y;
    const float ref = clipmax[clip];
    float *o = t + (int64_t)clip * per_clip;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (int64_t)gridDim.x * 256) {
        const float l = o[i] - ref;
        o[i] = l < -80.0f ? -80.0f : l;
    }
}

template <int NP>
static void launch_cq_t(const CqPlanDev &cp, const CqClassDev &cc, const cf *d_x, int n_clips, float *d_mag,
                        float *d_wavemax, bool db_term_out, hipStream_t s)
{
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cq_kernel<NP, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cq_kernel<NP, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    dim3 grid(cc.n_bands, n_clips);
    const size_t lds = (size_t)Size<NP>::DATA * sizeof(cf) + cq_threads(NP) * sizeof(float);
    if (db_term_out)
        hipLaunchKernelGGL((cq_kernel<NP, true>), grid, dim3(cq_threads(NP)), lds, s, cp, cc, d_x, d_mag, d_wavemax);
    else
        hipLaunchKernelGGL((cq_kernel<NP, false>), grid, dim3(cq_threads(NP)), lds, s, cp, cc, d_x, d_mag, d_wavemax);
}

void launch_cq_class(const CqPlanDev &cp, const CqClassDev &cc, const cf *d_x, int n_clips, float *d_mag,
                     float *d_wavemax, bool db_term_out, hipStream_t s)
{
    switch (cc.p) {
    case 64: launch_cq_t<64>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 96: launch_cq_t<96>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 128: launch_cq_t<128>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 192: launch_cq_t<192>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 256: launch_cq_t<256>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 384: launch_cq_t<384>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 512: launch_cq_t<512>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 768: launch_cq_t<768>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 1024: launch_cq_t<1024>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 1536: launch_cq_t<1536>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 2048: launch_cq_t<2048>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 3072: launch_cq_t<3072>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 4096: launch_cq_t<4096>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 6144: launch_cq_t<6144>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 8192: launch_cq_t<8192>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    case 12288: launch_cq_t<12288>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break;
    default: launch_cq_t<16384>(cp, cc, d_x, n_clips, d_mag, d_wavemax, db_term_out, s); break; // the plan admits nothing larger
    }
}

void launch_magmax(const float *d_mag, int n_clips, int c, float *d_wavemax, hipStream_t s)
{
    hipLaunchKernelGGL(magmax_kernel, dim3(kBins, n_clips), dim3(256), 0, s, d_mag, c, d_wavemax);
}

void launch_clipmax(const float *d_wavemax, float *d_clipmax, int n_clips, hipStream_t s)
{
    hipLaunchKernelGGL(clipmax_kernel, dim3(n_clips), dim3(256), 0, s, d_wavemax, d_clipmax);
}

void launch_db_finish(float *d_t, const float *d_clipmax, int n_clips, int64_t per_clip, hipStream_t s)
{
    int bx = (int)((per_clip + 256 * 4 - 1) / (256 * 4));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(db_finish_kernel, dim3(bx, n_clips), dim3(256), 0, s, d_t, d_clipmax, per_clip);
}

void launch_db(const float *d_mag, const float *d_wavemax, float *d_clipmax, int n_clips, int64_t per_clip,
               float *d_db, hipStream_t s)
{
    hipLaunchKernelGGL(clipmax_kernel, dim3(n_clips), dim3(256), 0, s, d_wavemax, d_clipmax);
    int bx = (int)((per_clip + 256 * 4 - 1) / (256 * 4));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(db_kernel, dim3(bx, n_clips), dim3(256), 0, s, d_mag, d_clipmax, per_clip, d_db);
}

} // namespace hpfw
