// simd_share_probe.hip -- does a wave doing ordinary work stay correct while ANOTHER wave on the same SIMD runs int8
// matrix instructions back to back?  (not part of the library; diagnosis of test_two_processes_share_the_gpu)
//
// One kernel, workgroups of 8 waves: waves 0..3 (one per SIMD) issue v_mfma_i32_16x16x64_i8 in a loop like
// hashprint_q_kernel's main loop (or, with `mfma` = 0, idle), waves 4..7 (the same four SIMDs) do work whose result
// is known and check it:
//   mode 1  chains of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with operand modifiers on exact values
//   mode 2  8-byte LDS stores and loads inside the wave's own 4 KB (permuted lanes), checked
//   mode 3  16-byte loads from a read-only table, checked
//   mode 4  scalar (wave-uniform) loads from the table, checked
//   simd_share_probe <mode> <mfma 0/1> [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            std::exit(2);                                                                                              \
        }                                                                                                              \
    } while (0)

__host__ __device__ inline unsigned mix(unsigned a, unsigned b)
{
    unsigned x = a * 0x9e3779b1u ^ (b + 0x7f4a7c15u) * 0x85ebca6bu;
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    return x;
}

struct Report {
    unsigned long long bad, checks;
    unsigned n;
    unsigned rec[32][4];
};

__device__ __forceinline__ void note(Report *rep, unsigned a, unsigned b, unsigned c, unsigned d)
{
    const unsigned slot = atomicAdd(&rep->n, 1u);
    if (slot < 32) {
        rep->rec[slot][0] = a;
        rep->rec[slot][1] = b;
        rep->rec[slot][2] = c;
        rep->rec[slot][3] = d;
    }
}

__global__ __launch_bounds__(512, 1) void share_kernel(Report *rep, int mode, int mfma, int iters, const v4i *__restrict__ table, int table_vecs,
                                                     int *sink)
{
    extern __shared__ unsigned lds[];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    if (wave < 4) {
        if (!mfma) return;
        v4i a = v4i{(int)mix(tid, 1), (int)mix(tid, 2), (int)mix(tid, 3), (int)mix(tid, 4)};
        v4i b = v4i{(int)mix(tid, 5), (int)mix(tid, 6), (int)mix(tid, 7), (int)mix(tid, 8)};
        v4i acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[f], 0, 0, 0);
            a.x ^= it;
        }
        int s = 0;
#pragma unroll
        for (int f = 0; f < 8; ++f) s += acc[f].x + acc[f].y + acc[f].z + acc[f].w;
        if (s == 0x12345678) sink[0] = s;
        return;
    }
    unsigned bad = 0, checks = 0;
    if (mode == 1) {
        for (int it = 0; it < iters; ++it) {
            // complex products of small Gaussian integers, exact in f32: (a + i b)(c + i d) as the library's c_mul does it
            const float ar = (float)((mix(tid, it) & 255) - 128), ai = (float)((mix(tid, it + 7777) & 255) - 128);
            const float wr = (float)((mix(wg, it) & 255) - 128), wi = (float)((mix(lane, it) & 255) - 128);
            v2f pa = v2f{ar, ai}, pw = v2f{wr, wi}, t, o, s1, s2;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(pa), "v"(pw));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(o) : "v"(pa), "v"(pw), "v"(t));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(s1) : "v"(o), "v"(pa));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(s2) : "v"(o), "v"(pa));
            const float er = ar * wr - ai * wi, ei = ar * wi + ai * wr;
            // s1 = o + (-i) pa = (o.r + pa.i, o.i - pa.r); s2 = o - (-i) pa = (o.r - pa.i, o.i + pa.r)
            const bool ok = o.x == er && o.y == ei && s1.x == er + ai && s1.y == ei - ar && s2.x == er - ai && s2.y == ei + ar;
            ++checks;
            if (!ok) {
                ++bad;
                note(rep, wg, tid, __float_as_uint(o.x), __float_as_uint(er));
            }
        }
    } else if (mode == 2) {
        uint2 *mine = reinterpret_cast<uint2 *>(lds) + (wave - 4) * 512; // 4 KB per victim wave
        for (int it = 0; it < iters; ++it) {
            const unsigned dst = (lane * 37u + 11u + it) & 63u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned w = mix(wg * 8192 + it, lane * 8 + e);
                mine[e * 64 + dst] = uint2{w, ~w};
            }
            unsigned inv = 37u;
            inv *= 2u - 37u * inv;
            inv *= 2u - 37u * inv;
            inv *= 2u - 37u * inv;
            inv *= 2u - 37u * inv;
            const unsigned src = ((lane - 11u - it) * inv) & 63u;
            uint2 got[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) got[e] = mine[e * 64 + lane];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned want = mix(wg * 8192 + it, src * 8 + e);
                ++checks;
                if (got[e].x != want || got[e].y != ~want) {
                    ++bad;
                    note(rep, wg, tid, got[e].x, want);
                }
            }
        }
    } else if (mode == 3) {
        unsigned idx = (wg * 977u + tid) % table_vecs;
        for (int it = 0; it < iters; ++it) {
            v4i v[4];
            unsigned at[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                at[e] = idx;
                v[e] = table[idx];
                idx = (idx + 1543u + e) % table_vecs;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned want = mix(0, at[e] * 4 + k);
                    ++checks;
                    if ((unsigned)v[e][k] != want) {
                        ++bad;
                        note(rep, wg, tid, (unsigned)v[e][k], want);
                    }
                }
        }
    } else {
        unsigned idx = (wg * 977u + wave * 31u) % table_vecs;
        for (int it = 0; it < iters; ++it) {
            const unsigned at = __builtin_amdgcn_readfirstlane(idx);
            const v4i v = table[at]; // uniform address: a scalar load
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned want = mix(0, at * 4 + k);
                ++checks;
                if ((unsigned)v[k] != want) {
                    ++bad;
                    note(rep, wg, tid, (unsigned)v[k], want);
                }
            }
            idx = (idx + 1543u) % table_vecs;
        }
    }
    if (bad) atomicAdd(&rep->bad, (unsigned long long)bad);
    if (lane == 0) atomicAdd(&rep->checks, (unsigned long long)checks);
}

__global__ void fill_table(unsigned *t, int words)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) t[i] = mix(0, i);
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? std::atoi(argv[1]) : 1;
    const int mfma = argc > 2 ? std::atoi(argv[2]) : 1;
    const int launches = argc > 3 ? std::atoi(argv[3]) : 20;
    const int table_vecs = 16384;
    unsigned *d_table;
    int *sink;
    Report *rep;
    CK(hipMalloc(&d_table, (size_t)table_vecs * 16));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMemset(rep, 0, sizeof(Report)));
    hipLaunchKernelGGL(fill_table, dim3(64), dim3(256), 0, 0, d_table, table_vecs * 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int l = 0; l < launches; ++l)
        hipLaunchKernelGGL(share_kernel, dim3(256 * 4), dim3(512), 16384, 0, rep, mode, mfma, 2000, reinterpret_cast<const v4i *>(d_table), table_vecs,
                           sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Report h;
    CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("{\"mode\": %d, \"mfma\": %d, \"launches\": %d, \"ms\": %.2f, \"checks_per_lane0\": %llu, \"bad\": %llu, \"rec\": [", mode, mfma, launches, ms,
                h.checks, h.bad);
    for (unsigned i = 0; i < h.n && i < 8; ++i)
        std::printf("%s[%u, %u, \"%08x\", \"%08x\"]", i ? ", " : "", h.rec[i][0], h.rec[i][1], h.rec[i][2], h.rec[i][3]);
    std::printf("]}\n");
    return 0;
}
