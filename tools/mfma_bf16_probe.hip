// Probe: numerical behaviour of v_mfma_f32_32x32x16_bf16 (gfx950).  One wave per matrix triple.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    // A [32][16], B [16][32], C/D [32][32], one set per block
    const int lane = threadIdx.x, blk = blockIdx.x;
    const uint16_t* a = A + (size_t)blk * 512; const uint16_t* b = B + (size_t)blk * 512;
    const float* c = C + (size_t)blk * 1024; float* d = D + (size_t)blk * 1024;
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) {
        const int kk = 8 * (lane >> 5) + i;
        uint16_t ua = a[(lane & 31) * 16 + kk], ub = b[kk * 32 + (lane & 31)];
        av[i] = __builtin_bit_cast(__bf16, ua); bv[i] = __builtin_bit_cast(__bf16, ub);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); acc[r] = c[row * 32 + (lane & 31)]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); d[row * 32 + (lane & 31)] = acc[r]; }
}
int main(int argc, char** argv) {
    const char* in = argv[1]; const char* out = argv[2]; int nb = atoi(argv[3]);
    std::vector<uint16_t> A((size_t)nb * 512), B((size_t)nb * 512); std::vector<float> C((size_t)nb * 1024), D((size_t)nb * 1024);
    FILE* f = fopen(in, "rb"); fread(A.data(), 2, A.size(), f); fread(B.data(), 2, B.size(), f); fread(C.data(), 4, C.size(), f); fclose(f);
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(out, "wb"); fwrite(D.data(), 4, D.size(), f); fclose(f);
    printf("done %d\n", nb); return 0;
}
