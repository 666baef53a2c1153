// mfma_probe.hip -- micro-benchmark behind the projection kernel's tuning (not part of the library).
// Measures v_mfma_f32_32x32x2_f32 throughput for the kernel's inner-loop shapes:
//   mode 0: MFMA only (operands in registers)      mode 1: + B operand from LDS (ds_read2_b32)
//   mode 2: + A operand streamed from global memory, one bin ahead (as project_kernel does)
// usage: mfma_probe <waves_per_simd 1|2|4>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kIters = 200; // bins per wave: 200 * 40 MFMAs

// mode 3: mode 2 + the S slab re-staged from global memory every 11 bins (two barriers)
// mode 4: mode 3 with the filter stream double-buffered by unrolling two bins (no register copies)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ fpack, float *out, const float *__restrict__ sglob)
{
    __shared__ float s_tile[11 * 276];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5;
    for (int i = tid; i < 11 * 276; i += 256) s_tile[i] = sglob[i];
    __syncthreads();
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const float4 *ap = reinterpret_cast<const float4 *>(fpack) + lane * 5;
    float4 a_cur[5], a_nxt[5];
    for (int i = 0; i < 5; ++i) a_cur[i] = ap[i];
    const int nl = wave * 64 + (lane & 31);
    for (int it = 0; it < kIters; ++it) {
        const int b = it % 11;
        if (MODE >= 3 && b == 0) {
            __syncthreads();
            for (int i = tid; i < 11 * 275; i += 256) s_tile[(i / 275) * 276 + i % 275] = sglob[(size_t)(it % 121) * 2419 + (blockIdx.x % 9) * 256 + i % 275 + (i / 275) * 2419];
            __syncthreads();
        }
        if (MODE >= 2)
            for (int i = 0; i < 5; ++i) a_nxt[i] = ap[(size_t)((it + 1) % 121) * 320 + i];
        const float *srow = s_tile + b * 276 + nl + kh;
#pragma unroll
        for (int tp = 0; tp < 10; ++tp) {
            const float4 a4 = a_cur[tp >> 1];
            const float a0 = (tp & 1) ? a4.z : a4.x, a1 = (tp & 1) ? a4.w : a4.y;
            float b0 = a4.x, b1 = a4.y;
            if (MODE >= 1) { b0 = srow[2 * tp]; b1 = srow[2 * tp + 32]; }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
        if (MODE >= 2)
            for (int i = 0; i < 5; ++i) a_cur[i] = a_nxt[i];
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc00[r] + acc01[r] + acc10[r] + acc11[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const float *d_f, float *d_o, const float *d_s, int wgs, const char *name)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<MODE><<<wgs, 256>>>(d_f, d_o, d_s);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) probe<MODE><<<wgs, 256>>>(d_f, d_o, d_s);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = 5.0 * wgs * 4 * kIters * 40.0 * 4096.0;
    std::printf("%-28s wgs=%d  %.3f ms  %.1f TFLOP/s\n", name, wgs, ms / 5, flop / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? std::atoi(argv[1]) : 4;
    const bool rnd = argc > 2;   // any second argument: random operands (DVFS: zeros clock higher)
    std::vector<float> f(64 * 2420, 0.001f);
    unsigned st = 12345;
    if (rnd) for (auto &v : f) { st = st * 1664525u + 1013904223u; v = ((st >> 8) / 8388608.0f - 1.0f) * 0.05f; }
    float *d_f, *d_o;
    hipMalloc(&d_f, f.size() * 4);
    hipMemcpy(d_f, f.data(), f.size() * 4, hipMemcpyHostToDevice);
    const int wgs = 256 * wps * 8;
    hipMalloc(&d_o, (size_t)wgs * 256 * 4);
    float *d_s;
    hipMalloc(&d_s, (size_t)140 * 2419 * 4);
    {
        std::vector<float> sv((size_t)140 * 2419, 0.0f);
        if (rnd) for (auto &v : sv) { st = st * 1664525u + 1013904223u; v = -80.0f * ((st >> 8) / 16777216.0f); }
        hipMemcpy(d_s, sv.data(), sv.size() * 4, hipMemcpyHostToDevice);
    }
    run<0>(d_f, d_o, d_s, wgs, "mfma only");
    run<1>(d_f, d_o, d_s, wgs, "mfma + LDS B");
    run<2>(d_f, d_o, d_s, wgs, "mfma + LDS B + global A");
    run<3>(d_f, d_o, d_s, wgs, "... + S staging / barriers");
    return 0;
}
