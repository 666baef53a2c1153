// pk_mfma_repro.hip -- packed FP32 arithmetic beside an int8 matrix kernel: a self-checking reproducer.
// (not part of the library; the finding behind DESIGN.md "packed FP32 is not used")
//
// Round 3's row transform and chirp-z kernels did their complex arithmetic with v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32.  Their results were bit-exact on a GPU of their own and went wrong -- sixteen consecutive elements at a
// time, always lanes 48..63 of a wave -- whenever hashprint_q_kernel (or the LDS-staged column kernel: both issue int8
// matrix instructions with their operands read from LDS) ran on another stream or in another process at the same time
// (tools/interfere.py, tools/rows_snapshots.py).  Built without packed instructions the same kernels are exact beside any
// neighbour.  This program shows the effect without the library's transform code: every thread computes the first fused
// group of the row transform (radix 7, twiddles, radix 3, twiddles; 21 points from LDS) TWICE, once with packed
// instructions and once with the same IEEE operations as scalar instructions, and compares the two bit for bit.  The two
// can only differ if the hardware returns a wrong result.  The neighbour is the library's hashprint_q_kernel
// (hpfw_gpu_hashprints_from_db) on a second stream.
//
//   pk_mfma_repro [rounds] [neighbour: 1 hashprint_q (default), 0 none, 2..6 synthetic kernels (below)]
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (tools/build_probes.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/hpfw_gpu.h"

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            std::exit(2);                                                                                              \
        }                                                                                                              \
    } while (0)

struct cf {
    float r, i;
};
typedef float v2f __attribute__((ext_vector_type(2)));

// the same operations twice: Sc as scalar IEEE instructions, Pk as the packed instructions round 3 used
struct Sc {
    static __device__ __forceinline__ cf add(cf a, cf b) { return {a.r + b.r, a.i + b.i}; }
    static __device__ __forceinline__ cf sub(cf a, cf b) { return {a.r - b.r, a.i - b.i}; }
    static __device__ __forceinline__ cf add_mi(cf a, cf d) { return {a.r + d.i, a.i - d.r}; }
    static __device__ __forceinline__ cf sub_mi(cf a, cf d) { return {a.r - d.i, a.i + d.r}; }
    static __device__ __forceinline__ cf mul(cf a, cf w)
    {
        const float p = a.i * w.i, q = a.i * w.r;
        return {__builtin_fmaf(a.r, w.r, -p), __builtin_fmaf(a.r, w.i, q)};
    }
    static __device__ __forceinline__ cf fma_s(float s, cf a, cf b) { return {__builtin_fmaf(s, a.r, b.r), __builtin_fmaf(s, a.i, b.i)}; }
    static __device__ __forceinline__ cf scale(float s, cf a) { return {s * a.r, s * a.i}; }
};
struct Pk {
    static __device__ __forceinline__ v2f p(cf a) { return v2f{a.r, a.i}; }
    static __device__ __forceinline__ cf u(v2f v) { return cf{v.x, v.y}; }
    static __device__ __forceinline__ cf add(cf a, cf b)
    {
        v2f o;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(p(a)), "v"(p(b)));
        return u(o);
    }
    static __device__ __forceinline__ cf sub(cf a, cf b)
    {
        v2f o;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(p(a)), "v"(p(b)));
        return u(o);
    }
    static __device__ __forceinline__ cf add_mi(cf a, cf d)
    {
        v2f o;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(o) : "v"(p(a)), "v"(p(d)));
        return u(o);
    }
    static __device__ __forceinline__ cf sub_mi(cf a, cf d)
    {
        v2f o;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o) : "v"(p(a)), "v"(p(d)));
        return u(o);
    }
    static __device__ __forceinline__ cf mul(cf a, cf w)
    {
        v2f t, o;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(p(a)), "v"(p(w)));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(o) : "v"(p(a)), "v"(p(w)), "v"(t));
        return u(o);
    }
    static __device__ __forceinline__ cf fma_s(float s, cf a, cf b)
    {
        v2f o;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(o) : "s"(v2f{s, s}), "v"(p(a)), "v"(p(b)));
        return u(o);
    }
    static __device__ __forceinline__ cf scale(float s, cf a)
    {
        v2f o;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(o) : "s"(v2f{s, s}), "v"(p(a)));
        return u(o);
    }
};

template <class M>
__device__ __forceinline__ void dft3(cf *u)
{
    const float s = 0.86602540378443864676f;
    cf t1 = M::add(u[1], u[2]);
    cf d = M::sub(u[1], u[2]);
    cf m1 = M::fma_s(-0.5f, t1, u[0]);
    cf sd = M::scale(s, d);
    u[0] = M::add(u[0], t1);
    u[1] = M::add_mi(m1, sd);
    u[2] = M::sub_mi(m1, sd);
}

template <class M>
__device__ __forceinline__ void dft7(cf *u)
{
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
    cf a1 = M::add(u[1], u[6]), a2 = M::add(u[2], u[5]), a3 = M::add(u[3], u[4]);
    cf b1 = M::sub(u[1], u[6]), b2 = M::sub(u[2], u[5]), b3 = M::sub(u[3], u[4]);
    cf p1 = M::fma_s(c3, a3, M::fma_s(c2, a2, M::fma_s(c1, a1, u[0])));
    cf p2 = M::fma_s(c1, a3, M::fma_s(c3, a2, M::fma_s(c2, a1, u[0])));
    cf p3 = M::fma_s(c2, a3, M::fma_s(c1, a2, M::fma_s(c3, a1, u[0])));
    cf q1 = M::fma_s(s3, b3, M::fma_s(s2, b2, M::scale(s1, b1)));
    cf q2 = M::fma_s(-s1, b3, M::fma_s(-s3, b2, M::scale(s2, b1)));
    cf q3 = M::fma_s(s2, b3, M::fma_s(-s1, b2, M::scale(s3, b1)));
    u[0] = M::add(M::add(M::add(u[0], a1), a2), a3);
    u[1] = M::add_mi(p1, q1);
    u[6] = M::sub_mi(p1, q1);
    u[2] = M::add_mi(p2, q2);
    u[5] = M::sub_mi(p2, q2);
    u[3] = M::add_mi(p3, q3);
    u[4] = M::sub_mi(p3, q3);
}

// one fused (7, 3) group on 21 points in registers: in[q2][q] -> out[s][s2]; tw: 20 per-butterfly twiddles
template <class M>
__device__ __forceinline__ void group73(const cf (&in)[3][7], const cf (&tw)[20], cf (&out)[7][3])
{
    cf e[7][3];
#pragma unroll
    for (int q2 = 0; q2 < 3; ++q2) {
        cf u[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) u[q] = in[q2][q];
        dft7<M>(u);
        e[0][q2] = u[0];
#pragma unroll
        for (int s = 1; s < 7; ++s) e[s][q2] = M::mul(u[s], tw[q2 * 6 + (s - 1)]);
    }
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        cf v[3];
#pragma unroll
        for (int q2 = 0; q2 < 3; ++q2) v[q2] = e[s][q2];
        dft3<M>(v);
        out[s][0] = v[0];
        out[s][1] = M::mul(v[1], tw[18]);
        out[s][2] = M::mul(v[2], tw[19]);
    }
}

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b)
{
    unsigned x = a * 0x9e3779b1u ^ (b + 0x7f4a7c15u) * 0x85ebca6bu;
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    return x;
}
__device__ __forceinline__ float unit(unsigned h) { return (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }

constexpr int kRec = 48;
struct Report {
    unsigned long long bad, compared;
    unsigned quarter[4]; // mismatching values by lane quarter (lanes 0-15, 16-31, 32-47, 48-63)
    unsigned n;
    unsigned rec[kRec][6]; // wg, tid, pass << 8 | output index, which half, packed bits, scalar bits
};

constexpr int kN2 = 6300, kButterflies = 300;

__global__ __launch_bounds__(512, 6) void victim_kernel(Report *rep, const cf *__restrict__ twiddles, int passes, unsigned salt)
{
    extern __shared__ cf lds[];
    const int tid = threadIdx.x;
    const unsigned wg = blockIdx.x;
    unsigned bad = 0, compared = 0;
    for (int p = 0; p < passes; ++p) {
        for (int i = tid; i < kN2; i += 512) lds[i] = cf{unit(mix(wg * 64 + p + salt, 2 * i)), unit(mix(wg * 64 + p + salt, 2 * i + 1))};
        __syncthreads();
        if (tid < kButterflies) {
            cf in[3][7], tw[20], op[7][3], os[7][3];
#pragma unroll
            for (int q2 = 0; q2 < 3; ++q2)
#pragma unroll
                for (int q = 0; q < 7; ++q) in[q2][q] = lds[tid + q2 * 300 + q * 900];
#pragma unroll
            for (int e = 0; e < 20; ++e) tw[e] = twiddles[e * kButterflies + tid];
            group73<Pk>(in, tw, op);
            group73<Sc>(in, tw, os);
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const unsigned pr = __float_as_uint(op[s][s2].r), pi = __float_as_uint(op[s][s2].i);
                    const unsigned sr = __float_as_uint(os[s][s2].r), si = __float_as_uint(os[s][s2].i);
                    compared += 2;
                    if (pr != sr || pi != si) {
                        bad += (pr != sr) + (pi != si);
                        atomicAdd(&rep->quarter[(tid & 63) >> 4], (unsigned)((pr != sr) + (pi != si)));
                        const unsigned slot = atomicAdd(&rep->n, 1u);
                        if (slot < kRec) {
                            rep->rec[slot][0] = wg;
                            rep->rec[slot][1] = tid;
                            rep->rec[slot][2] = (p << 8) | (s * 3 + s2);
                            rep->rec[slot][3] = (pr != sr ? 1 : 0) | (pi != si ? 2 : 0);
                            rep->rec[slot][4] = pr != sr ? pr : pi;
                            rep->rec[slot][5] = pr != sr ? sr : si;
                        }
                    }
                    lds[tid + s * 900 + s2 * 300] = os[s][s2];
                }
        }
        __syncthreads();
    }
    if (bad) atomicAdd(&rep->bad, (unsigned long long)bad);
    if ((tid & 63) == 0) atomicAdd(&rep->compared, (unsigned long long)compared * 64);
}

// ---- synthetic neighbours (2..6): which ingredient of hashprint_q_kernel does it? ----
// 2: int8 matrix instructions on register operands; 3: the same with the B operand re-read from LDS every step
// (ds_read_b128, as hashprint_q_kernel does); 4: the LDS reads and integer adds alone; 5: f32 matrix instructions on
// register operands; 6: int8 matrix instructions, B operand from LDS, only 64 registers (more waves per SIMD)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND, int ACCS>
__global__ __launch_bounds__(256, KIND == 6 ? 4 : 2) void neighbour_kernel(int *sink, int iters)
{
    extern __shared__ v4i nl[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3904; i += 256) nl[i] = v4i{(int)mix(i, 1), (int)mix(i, 2), (int)mix(i, 3), (int)mix(i, 4)};
    __syncthreads();
    v4i a = v4i{(int)mix(tid, 5), (int)mix(tid, 6), (int)mix(tid, 7), (int)mix(tid, 8)};
    v4i b = v4i{(int)mix(tid, 9), (int)mix(tid, 10), (int)mix(tid, 11), (int)mix(tid, 12)};
    int s = 0;
    if (KIND == 5) {
        v16f acc[2] = {v16f{0}, v16f{0}};
        float fa = (float)(tid & 7), fb = (float)(tid & 3);
        for (int it = 0; it < iters * 4; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc[1], 0, 0, 0);
        }
        s = (int)(acc[0][0] + acc[1][3]);
    } else {
        v4i acc[ACCS];
#pragma unroll
        for (int f = 0; f < ACCS; ++f) acc[f] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < ACCS; ++f) {
                v4i bb = b;
                if (KIND == 3 || KIND == 4 || KIND == 6) bb = nl[(lane + 61 * f + 7 * it) % 3840];
                if (KIND == 4)
                    acc[f] += bb;
                else
                    acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bb, acc[f], 0, 0, 0);
            }
            a.x ^= it;
        }
#pragma unroll
        for (int f = 0; f < ACCS; ++f) s += acc[f].x + acc[f].y + acc[f].z + acc[f].w;
    }
    if (s == 0x12345678) sink[0] = s;
}

template <int KIND, int ACCS>
static void launch_neighbour(int *sink, hipStream_t st)
{
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(neighbour_kernel<KIND, ACCS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once = true;
    }
    hipLaunchKernelGGL((neighbour_kernel<KIND, ACCS>), dim3(256 * 8), dim3(256), 62464, st, sink, 600);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 30;
    const int neighbour = argc > 2 ? std::atoi(argv[2]) : 1;
    hpfw_gpu *h = nullptr;
    if (hpfw_gpu_create(0, &h)) {
        std::fprintf(stderr, "create: %s\n", hpfw_gpu_last_error());
        return 2;
    }
    {
        std::vector<float> f((size_t)64 * 2420);
        unsigned s = 12345;
        for (auto &v : f) {
            s = s * 1664525u + 1013904223u;
            v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f;
        }
        if (hpfw_gpu_set_filters(h, f.data())) return 2;
    }
    const int n_clips = 256, c = 2419;
    float *d_db;
    uint64_t *d_hp;
    CK(hipMalloc(&d_db, (size_t)n_clips * 121 * c * 4));
    CK(hipMalloc(&d_hp, (size_t)n_clips * (c - 99) * 8));
    {
        std::vector<float> db((size_t)n_clips * 121 * c);
        unsigned s = 777;
        for (auto &v : db) {
            s = s * 1664525u + 1013904223u;
            v = -(float)(s >> 8) * (80.0f / 16777216.0f);
        }
        CK(hipMemcpy(d_db, db.data(), db.size() * 4, hipMemcpyHostToDevice));
    }
    cf *d_tw;
    {
        std::vector<cf> tw((size_t)20 * kButterflies);
        unsigned s = 4242;
        for (auto &v : tw) {
            s = s * 1664525u + 1013904223u;
            const float a = (float)(s >> 8) * (6.2831853f / 16777216.0f);
            v = cf{__builtin_cosf(a), -__builtin_sinf(a)};
        }
        CK(hipMalloc(&d_tw, tw.size() * sizeof(cf)));
        CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    }
    Report *rep;
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMemset(rep, 0, sizeof(Report)));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int *sink;
    CK(hipMalloc(&sink, 64));
    auto run_neighbour = [&]() {
        switch (neighbour) {
        case 0: break;
        case 1:
            if (hpfw_gpu_hashprints_from_db(h, d_db, n_clips, c, d_hp, sb)) {
                std::fprintf(stderr, "neighbour: %s\n", hpfw_gpu_last_error());
                std::exit(2);
            }
            break;
        case 2: launch_neighbour<2, 40>(sink, sb); break;
        case 3: launch_neighbour<3, 40>(sink, sb); break;
        case 4: launch_neighbour<4, 40>(sink, sb); break;
        case 5: launch_neighbour<5, 2>(sink, sb); break;
        default: launch_neighbour<6, 8>(sink, sb); break;
        }
    };
    for (int r = 0; r < rounds; ++r) {
        run_neighbour();
        run_neighbour();
        hipLaunchKernelGGL(victim_kernel, dim3(3392), dim3(512), kN2 * sizeof(cf), sa, rep, d_tw, 6, (unsigned)r * 977u);
        run_neighbour();
        run_neighbour();
    }
    CK(hipDeviceSynchronize());
    Report hr;
    CK(hipMemcpy(&hr, rep, sizeof(hr), hipMemcpyDeviceToHost));
    std::printf("{\"neighbour\": \"%s\", \"rounds\": %d, \"values_compared\": %llu, \"values_differing\": %llu, \"by_lane_quarter\": [%u, %u, %u, %u],\n",
                neighbour == 0 ? "none" : neighbour == 1 ? "hashprint_q_kernel" : neighbour == 2 ? "int8 mfma, registers" : neighbour == 3 ? "int8 mfma, B from LDS"
                : neighbour == 4 ? "LDS reads, integer adds" : neighbour == 5 ? "f32 mfma, registers" : "int8 mfma, B from LDS, 64 registers", rounds, hr.compared, hr.bad, hr.quarter[0], hr.quarter[1], hr.quarter[2], hr.quarter[3]);
    std::printf(" \"first\": [");
    for (unsigned i = 0; i < hr.n && i < 16; ++i)
        std::printf("%s{\"wg\": %u, \"tid\": %u, \"pass\": %u, \"output\": %u, \"halves\": %u, \"packed\": \"%08x\", \"scalar\": \"%08x\"}", i ? ", " : "",
                    hr.rec[i][0], hr.rec[i][1], hr.rec[i][2] >> 8, hr.rec[i][2] & 255, hr.rec[i][3], hr.rec[i][4], hr.rec[i][5]);
    std::printf("]}\n");
    hpfw_gpu_destroy(h);
    return 0;
}
