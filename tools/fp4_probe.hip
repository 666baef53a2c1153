// fp4_probe.hip -- does v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands +-1 give the
// exact 64-bit Hamming dot product?  acc(m, n) must equal 64 - 2 popcount(x_m ^ y_n).
//   hipcc --offload-arch=gfx950 -O3 tools/fp4_probe.hip -o tools/fp4_probe.bin && tools/fp4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 8 bits -> 8 nibbles: bit = 0 -> +1.0 (0x2), bit = 1 -> -1.0 (0xA)
__host__ __device__ inline uint32_t expand8(uint32_t x)
{
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return 0x22222222u | (t << 3);
}

__global__ void probe(const uint64_t *x, const uint64_t *y, float *out, int *cycles)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const uint32_t xa = (uint32_t)(x[r] >> (32 * h)), yb = (uint32_t)(y[r] >> (32 * h));
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        a[i] = (int)expand8((xa >> (8 * i)) & 0xff);
        b[i] = (int)expand8((yb >> (8 * i)) & 0xff);
    }
    f32x16 acc = {0};
    const int one = 0x7f7f7f7f; // E8M0 scale 2^0 in every byte
    long t0 = clock64();
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, one, 0, one);
#pragma unroll
    for (int i = 0; i < 63; ++i) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, one, 0, one);
    long t1 = clock64();
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[row * 32 + r] = acc[reg];
    }
    if (lane == 0) *cycles = (int)(t1 - t0);
}

int main()
{
    std::vector<uint64_t> x(32), y(32);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 32; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = s;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; y[i] = s;
    }
    y[3] = x[5];
    uint64_t *dx, *dy; float *dout; int *dc;
    hipMalloc(&dx, 256); hipMalloc(&dy, 256); hipMalloc(&dout, 4096); hipMalloc(&dc, 4);
    hipMemcpy(dx, x.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, dy, dout, dc);
    std::vector<float> out(1024);
    int cyc = 0;
    hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(&cyc, dc, 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            const float want = 64.0f * (64 - 2 * __builtin_popcountll(x[m] ^ y[n])); // 64 accumulations
            if (out[m * 32 + n] != want) {
                if (bad < 5) printf("(%d,%d): got %g want %g\n", m, n, out[m * 32 + n], want);
                ++bad;
            }
        }
    printf("mismatches=%d of 1024; 64 dependent MFMAs took %d cycles (%.1f each)\n", bad, cyc, cyc / 64.0);
    return bad != 0;
}
