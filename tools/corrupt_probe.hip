// corrupt_probe.hip -- WHAT of a resident workgroup changes when hashprint_q_kernel runs beside it?
// (not part of the library; diagnosis of tests/test_gpu_multi.py::test_two_processes_share_the_gpu)
//
// tools/interfere.py showed: the row transform and the chirp-z kernels go wrong when hashprint_q_kernel (or the
// LDS-staged column kernel) runs at the same time on another stream, and only then.  Here the victim is a kernel that
// computes nothing: it writes a pattern to its LDS, vector registers and matrix accumulators, spins, and checks them.
// The aggressor is the library's hpfw_gpu_hashprints_from_db on a second stream.
//
//   corrupt_probe <victim_lds_bytes> <victim_threads> [rounds] [aggressor: 1 hashprint_q, 0 none]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/hpfw_gpu.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            std::exit(2);                                                                                              \
        }                                                                                                              \
    } while (0)

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b)
{
    unsigned x = a * 0x9e3779b1u ^ (b + 0x7f4a7c15u) * 0x85ebca6bu;
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    return x;
}

constexpr int kMaxRec = 64;
struct Report {
    unsigned long long lds_bad, vgpr_bad, launches_bad;
    unsigned n_lds, n_vgpr;
    unsigned lds_rec[kMaxRec][5];  // wg, word index, got, want, check pass
    unsigned vgpr_rec[kMaxRec][5]; // wg, tid, k, got, want
};

// the pattern is checked `passes` times with a spin before each: a late overwrite shows as a later pass failing
template <int THREADS, int NREG>
__global__ __launch_bounds__(THREADS) void victim_kernel(Report *rep, int lds_words, long long spin_ticks, int passes, unsigned salt)
{
    extern __shared__ unsigned lds[];
    const unsigned tid = threadIdx.x, wg = blockIdx.x;
    for (int i = tid; i < lds_words; i += THREADS) lds[i] = mix(wg ^ salt, i);
    unsigned r[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
        r[k] = mix(wg * THREADS + tid, k ^ salt);
        asm volatile("" : "+v"(r[k]));
    }
    __syncthreads();
    unsigned bad_l = 0, bad_v = 0;
    for (int p = 0; p < passes; ++p) {
        const long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
        for (int i = tid; i < lds_words; i += THREADS) {
            const unsigned got = lds[i], want = mix(wg ^ salt, i);
            if (got != want) {
                ++bad_l;
                const unsigned slot = atomicAdd(&rep->n_lds, 1u);
                if (slot < kMaxRec) {
                    rep->lds_rec[slot][0] = wg;
                    rep->lds_rec[slot][1] = i;
                    rep->lds_rec[slot][2] = got;
                    rep->lds_rec[slot][3] = want;
                    rep->lds_rec[slot][4] = p;
                }
                lds[i] = want; // (count a damaged word once)
            }
        }
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
            asm volatile("" : "+v"(r[k]));
            const unsigned want = mix(wg * THREADS + tid, k ^ salt);
            if (r[k] != want) {
                ++bad_v;
                const unsigned slot = atomicAdd(&rep->n_vgpr, 1u);
                if (slot < kMaxRec) {
                    rep->vgpr_rec[slot][0] = wg;
                    rep->vgpr_rec[slot][1] = tid;
                    rep->vgpr_rec[slot][2] = k | (p << 16);
                    rep->vgpr_rec[slot][3] = r[k];
                    rep->vgpr_rec[slot][4] = want;
                }
                r[k] = want;
            }
        }
        __syncthreads();
    }
    if (bad_l) atomicAdd(&rep->lds_bad, (unsigned long long)bad_l);
    if (bad_v) atomicAdd(&rep->vgpr_bad, (unsigned long long)bad_v);
    if ((bad_l || bad_v) && tid == 0) atomicAdd(&rep->launches_bad, 1ull);
}

// ---- dynamic victims: `mode` 1 LDS exchange through barriers, 2 table loads, 3 packed arithmetic ----
// mode 1: every pass each thread writes f(pass, tid) to slot perm(tid), barrier, reads slot tid and checks it against
//         f(pass, inverse perm), barrier.  A read that overtakes the write, or a lost write, shows.
// mode 2: 16-byte loads from a read-only table (mix(0, word index)), index walking the table; checked every load
// mode 3: a chain of v_pk_fma_f32 on values whose result is known (integers small enough to be exact)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void dynamic_kernel(Report *rep, int mode, int passes, const v4i *__restrict__ table, int table_vecs,
                                                          unsigned salt)
{
    extern __shared__ unsigned lds[];
    const unsigned tid = threadIdx.x, wg = blockIdx.x;
    unsigned bad = 0;
    if (mode == 1) {
        // perm: tid -> (tid * 37 + 11) mod THREADS (37 odd and coprime to the power-of-two THREADS); 12 slots per thread at a
        // stride of THREADS words plus a skew, as an in-LDS transform pass does
        uint2 *lds2 = reinterpret_cast<uint2 *>(lds);   // 8-byte elements, as the transforms' complex values
        for (int p = 0; p < passes; ++p) {
            const unsigned dst = (tid * 37u + 11u + p) & (THREADS - 1);
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const unsigned w = mix(wg * 4096 + p, tid * 16 + e) ^ salt;
                lds2[e * (THREADS + 3) + dst] = uint2{w, ~w};
            }
            __syncthreads();
            // who wrote slot tid: src with (src * 37 + 11 + p) = tid mod THREADS; 37^-1 mod 2^k by Newton
            unsigned inv = 37u;
            inv *= 2u - 37u * inv;
            inv *= 2u - 37u * inv;
            inv *= 2u - 37u * inv;
            inv *= 2u - 37u * inv;
            const unsigned src = ((tid - 11u - p) * inv) & (THREADS - 1);
            uint2 got[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) got[e] = lds2[e * (THREADS + 3) + tid];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const unsigned want = mix(wg * 4096 + p, src * 16 + e) ^ salt;
                if (got[e].x != want || got[e].y != ~want) {
                    ++bad;
                    const unsigned slot = atomicAdd(&rep->n_lds, 1u);
                    if (slot < kMaxRec) {
                        rep->lds_rec[slot][0] = wg;
                        rep->lds_rec[slot][1] = tid * 16 + e;
                        rep->lds_rec[slot][2] = got[e].x;
                        rep->lds_rec[slot][3] = want;
                        rep->lds_rec[slot][4] = p;
                    }
                }
            }
            __syncthreads();
        }
    } else if (mode == 2) {
        unsigned idx = (wg * 977u + tid) % table_vecs;
        for (int p = 0; p < passes; ++p) {
            v4i v[4];
            unsigned at[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                at[e] = idx;
                v[e] = table[idx];
                idx = (idx + THREADS * 3 + e) % table_vecs;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned want = mix(0, at[e] * 4 + k);
                    if ((unsigned)v[e][k] != want) {
                        ++bad;
                        const unsigned slot = atomicAdd(&rep->n_vgpr, 1u);
                        if (slot < kMaxRec) {
                            rep->vgpr_rec[slot][0] = wg;
                            rep->vgpr_rec[slot][1] = tid;
                            rep->vgpr_rec[slot][2] = at[e] * 4 + k;
                            rep->vgpr_rec[slot][3] = (unsigned)v[e][k];
                            rep->vgpr_rec[slot][4] = want;
                        }
                    }
                }
        }
    } else {
        typedef float v2f __attribute__((ext_vector_type(2)));
        for (int p = 0; p < passes; ++p) {
            v2f acc = v2f{(float)(tid & 7), (float)(wg & 7)};
            const v2f m = v2f{1.0f, -1.0f}, a = v2f{3.0f, 5.0f};
#pragma unroll 16
            for (int e = 0; e < 256; ++e) {
                v2f o;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(acc), "v"(m), "v"(a));
                acc = o;
            }
            // x -> x + 3 (256 times); y -> -y + 5 (an even number of times: back to y)
            const float wx = (float)(tid & 7) + 768.0f, wy = (float)(wg & 7);
            if (acc.x != wx || acc.y != wy) {
                ++bad;
                const unsigned slot = atomicAdd(&rep->n_vgpr, 1u);
                if (slot < kMaxRec) {
                    rep->vgpr_rec[slot][0] = wg;
                    rep->vgpr_rec[slot][1] = tid;
                    rep->vgpr_rec[slot][2] = p;
                    rep->vgpr_rec[slot][3] = __float_as_uint(acc.x);
                    rep->vgpr_rec[slot][4] = __float_as_uint(acc.y);
                }
            }
        }
    }
    if (bad) {
        atomicAdd(mode == 1 ? &rep->lds_bad : &rep->vgpr_bad, (unsigned long long)bad);
        atomicAdd(&rep->launches_bad, 1ull);
    }
}

__global__ void fill_table(unsigned *t, int words)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) t[i] = mix(0, i);
}

int main(int argc, char **argv)
{
    const int lds_bytes = argc > 1 ? std::atoi(argv[1]) : 50400;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 512;
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 40;
    const int aggressor = argc > 4 ? std::atoi(argv[4]) : 1;
    const int mode = argc > 5 ? std::atoi(argv[5]) : 0;
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
    Report *rep;
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMemset(rep, 0, sizeof(Report)));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_kernel<512, 40>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_kernel<256, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = 2048;
    const int table_vecs = 65536 / 16 * 4; // 256 KB
    unsigned *d_table;
    CK(hipMalloc(&d_table, (size_t)table_vecs * 16));
    hipLaunchKernelGGL(fill_table, dim3(64), dim3(256), 0, 0, d_table, table_vecs * 4);
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(dynamic_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(dynamic_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int r = 0; r < rounds && mode; ++r) {
        if (aggressor && hpfw_gpu_hashprints_from_db(h, d_db, n_clips, c, d_hp, sb)) return 2;
        const int passes = mode == 1 ? 60 : (mode == 2 ? 120 : 40);
        if (threads == 512)
            hipLaunchKernelGGL((dynamic_kernel<512>), dim3(grid), dim3(512), lds_bytes, sa, rep, mode, passes, reinterpret_cast<const v4i *>(d_table),
                               table_vecs, (unsigned)r);
        else
            hipLaunchKernelGGL((dynamic_kernel<256>), dim3(grid), dim3(256), lds_bytes, sa, rep, mode, passes, reinterpret_cast<const v4i *>(d_table),
                               table_vecs, (unsigned)r);
        if (aggressor && hpfw_gpu_hashprints_from_db(h, d_db, n_clips, c, d_hp, sb)) return 2;
    }
    for (int r = 0; r < rounds && !mode; ++r) {
        if (aggressor && hpfw_gpu_hashprints_from_db(h, d_db, n_clips, c, d_hp, sb)) {
            std::fprintf(stderr, "aggressor: %s\n", hpfw_gpu_last_error());
            return 2;
        }
        if (threads == 512)
            hipLaunchKernelGGL((victim_kernel<512, 40>), dim3(grid), dim3(512), lds_bytes, sa, rep, lds_bytes / 4, 2000ll, 4, (unsigned)r);
        else
            hipLaunchKernelGGL((victim_kernel<256, 64>), dim3(grid), dim3(256), lds_bytes, sa, rep, lds_bytes / 4, 2000ll, 4, (unsigned)r);
        if (aggressor && hpfw_gpu_hashprints_from_db(h, d_db, n_clips, c, d_hp, sb)) return 2;
    }
    CK(hipDeviceSynchronize());
    Report hr;
    CK(hipMemcpy(&hr, rep, sizeof(hr), hipMemcpyDeviceToHost));
    std::printf("{\"mode\": %d, \"victim_lds\": %d, \"victim_threads\": %d, \"aggressor\": %d, \"rounds\": %d, \"lds_bad\": %llu, \"vgpr_bad\": %llu, \"wgs_bad\": %llu,\n",
                mode, lds_bytes, threads, aggressor, rounds, hr.lds_bad, hr.vgpr_bad, hr.launches_bad);
    std::printf(" \"lds_rec\": [");
    for (unsigned i = 0; i < hr.n_lds && i < 24; ++i)
        std::printf("%s[%u, %u, \"%08x\", \"%08x\", %u]", i ? ", " : "", hr.lds_rec[i][0], hr.lds_rec[i][1], hr.lds_rec[i][2], hr.lds_rec[i][3], hr.lds_rec[i][4]);
    std::printf("],\n \"vgpr_rec\": [");
    for (unsigned i = 0; i < hr.n_vgpr && i < 24; ++i)
        std::printf("%s[%u, %u, %u, \"%08x\", \"%08x\"]", i ? ", " : "", hr.vgpr_rec[i][0], hr.vgpr_rec[i][1], hr.vgpr_rec[i][2], hr.vgpr_rec[i][3], hr.vgpr_rec[i][4]);
    std::printf("]}\n");
    hpfw_gpu_destroy(h);
    return 0;
}
