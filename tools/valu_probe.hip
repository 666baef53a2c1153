// valu_probe.hip -- issue rate of scalar and packed f32 VALU instructions on gfx950 at 1, 2 and 4
// waves per SIMD (measurement tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe.bin && tools/valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void probe(float *out, int iters, float seed)
{
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + i + threadIdx.x;
        p[i] = f2{seed + i, seed - i + threadIdx.x};
    }
    const float m = seed * 0.5f;
    const f2 pm = {m, m + 1.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pm));
                if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int waves_per_simd, float *d_out, double ghz, int cus)
{
    const int iters = 4000;
    dim3 grid(cus), block(256 * waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<grid, block>>>(d_out, 10, 1.0f);
    hipEventRecord(e0);
    probe<MODE><<<grid, block>>>(d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves_per_simd; // wave-instructions issued on one SIMD
    const double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction on a SIMD (at %.2f GHz)\n", name,
           waves_per_simd, ms, cycles / instr_per_simd, ghz);
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const double ghz = pr.clockRate * 1e-6;
    printf("%s, %d CUs, clock %.2f GHz\n", pr.name, pr.multiProcessorCount, ghz);
    float *d;
    hipMalloc(&d, (size_t)pr.multiProcessorCount * 1024 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, d, ghz, pr.multiProcessorCount);
        run<1>("v_pk_fma_f32", w, d, ghz, pr.multiProcessorCount);
        run<2>("v_add_f32", w, d, ghz, pr.multiProcessorCount);
        run<3>("v_pk_add_f32", w, d, ghz, pr.multiProcessorCount);
        run<4>("v_pk_mul_f32", w, d, ghz, pr.multiProcessorCount);
    }
    return 0;
}
