// mfma_i8_probe.hip -- what v_mfma_i32_32x32x32_i8 costs on gfx950 (not part of the library).
//
// One wave per SIMD (256-thread workgroups, one per CU by LDS), back-to-back instructions on 1, 2 or 4 independent
// accumulators and on a single dependent chain; cycles by s_memtime around the loop (shader-clock ticks), the clock the
// chip holds by s_memrealtime (100 MHz), and the chip-wide rate by HIP events.  Also the same loop with the operands
// re-read from LDS (ds_read_b128) as project_q_kernel does, at one and at two waves per SIMD.
//   usage: mfma_i8_probe [random]     (any argument: random operands; zeros clock higher)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kIters = 4096;

// ACC independent accumulators, LDSOPS: 0 = operands in registers, 1 = both operands of every instruction re-read from LDS
// (ds_read_b128), the reads of the next group of ACC instructions issued before this group's instructions
template <int ACC, int LDSOPS>
__global__ __launch_bounds__(512) void probe(const v4i *__restrict__ src, int *out, long long *stamps)
{
    extern __shared__ v4i lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    v16i acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a) acc[a] = v16i{0};
    v4i a0 = src[lane], b0 = src[64 + lane];
    v4i an[ACC], bn[ACC], ac[ACC], bc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a) {
        an[a] = lds[(a * 128 + lane) & 4095];
        bn[a] = lds[(a * 128 + 64 + lane) & 4095];
    }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters / ACC; ++it) {
        if (LDSOPS) {
#pragma unroll
            for (int a = 0; a < ACC; ++a) {
                ac[a] = an[a];
                bc[a] = bn[a];
            }
#pragma unroll
            for (int a = 0; a < ACC; ++a) {
                an[a] = lds[(((it + 1) * ACC + a) * 128 + lane) & 4095];
                bn[a] = lds[(((it + 1) * ACC + a) * 128 + 64 + lane) & 4095];
            }
        }
#pragma unroll
        for (int a = 0; a < ACC; ++a)
            acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(LDSOPS ? ac[a] : a0, LDSOPS ? bc[a] : b0, acc[a], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) {
        stamps[(blockIdx.x * (blockDim.x / 64) + (tid >> 6)) * 2] = t1 - t0;
        stamps[(blockIdx.x * (blockDim.x / 64) + (tid >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int ACC, int LDSOPS>
static void run(const v4i *d_src, int *d_out, long long *d_st, int threads, const char *name)
{
    const int wgs = 256 * 8;
    const size_t lds = 100 * 1024; // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<ACC, LDSOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) probe<ACC, LDSOPS><<<wgs, threads, lds>>>(d_src, d_out, d_st);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) probe<ACC, LDSOPS><<<wgs, threads, lds>>>(d_src, d_out, d_st);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const int waves = wgs * threads / 64;
    std::vector<long long> st((size_t)waves * 2);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), ghz(waves);
    for (int w = 0; w < waves; ++w) {
        cyc[w] = (double)st[2 * w] / kIters;
        ghz[w] = st[2 * w + 1] > 0 ? (double)st[2 * w] / (double)st[2 * w + 1] * 0.1 : 0.0;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double ops = (double)reps * waves * kIters * 2.0 * 32 * 32 * 32;
    std::printf("{\"case\": \"%s\", \"waves_per_simd\": %d, \"accumulators\": %d, \"operands\": \"%s\", \"cycles_per_mfma_per_wave_median\": %.2f, "
                "\"cycles_per_mfma_per_simd\": %.2f, \"clock_ghz_median\": %.3f, \"ms\": %.4f, \"tops\": %.1f}\n",
                name, threads / 256, ACC, LDSOPS ? "lds" : "registers", cyc[waves / 2], cyc[waves / 2] / (threads / 256), ghz[waves / 2],
                ms / reps, ops / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const bool rnd = argc > 1;
    std::vector<int> h(4096 * 4, 0);
    unsigned s = 12345;
    if (rnd) for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int)s; }
    v4i *d_src;
    int *d_out;
    long long *d_st;
    hipMalloc(&d_src, h.size() * 4);
    hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_out, (size_t)256 * 8 * 512 * 4);
    hipMalloc(&d_st, (size_t)256 * 8 * 8 * 2 * 8);
    std::printf("{\"operands_data\": \"%s\", \"instruction\": \"v_mfma_i32_32x32x32_i8\", \"ops_per_instruction\": 65536}\n", rnd ? "random" : "zeros");
    run<1, 0>(d_src, d_out, d_st, 256, "dependent chain");
    run<2, 0>(d_src, d_out, d_st, 256, "2 accumulators");
    run<4, 0>(d_src, d_out, d_st, 256, "4 accumulators");
    run<4, 0>(d_src, d_out, d_st, 512, "4 accumulators, 2 waves per SIMD");
    run<4, 1>(d_src, d_out, d_st, 256, "4 accumulators, operands from LDS");
    run<4, 1>(d_src, d_out, d_st, 512, "4 accumulators, operands from LDS, 2 waves per SIMD");
    return 0;
}
