// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access widths this
// library uses?  MI355X_MICROARCH.md calibrates them for 16 B per lane only (FETCH_SIZE = half the bytes of a wide
// coalesced read); most kernels here load 4 or 8 bytes per lane.  Streams a 1 GiB buffer (past the 256 MiB
// Infinity Cache) with coalesced loads / stores of 2, 4, 8 and 16 bytes per lane; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/fetch_calib.bin     (and again with WRITE_SIZE)
// and divide the counter (KiB) by the known byte count: profiles/summarize.py does that.
#include <hip/hip_runtime.h>

#include <cstdio>

template <class T>
__global__ __launch_bounds__(256) void calib_read(const T *__restrict__ p, size_t n, float *__restrict__ out)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T v = p[i];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        acc += (float)b[0];
    }
    if (acc == -1.0f) out[0] = acc; // never true: keeps the loads alive
}

template <class T>
__global__ __launch_bounds__(256) void calib_write(T *__restrict__ p, size_t n, unsigned fill)
{
    T v;
    unsigned *w = reinterpret_cast<unsigned *>(&v);
    for (unsigned k = 0; k < (sizeof(T) + 3) / 4; ++k) w[k] = fill;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

struct b2 { unsigned short x; };
struct b4 { unsigned x; };
struct __attribute__((aligned(8))) b8 { unsigned x, y; };
struct __attribute__((aligned(16))) b16 { unsigned x, y, z, w; };

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *buf = nullptr;
    float *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&out, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_read<b2>, dim3(grid), dim3(256), 0, 0, (const b2 *)buf, bytes / 2, out);
        hipLaunchKernelGGL(calib_read<b4>, dim3(grid), dim3(256), 0, 0, (const b4 *)buf, bytes / 4, out);
        hipLaunchKernelGGL(calib_read<b8>, dim3(grid), dim3(256), 0, 0, (const b8 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(calib_read<b16>, dim3(grid), dim3(256), 0, 0, (const b16 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(calib_write<b2>, dim3(grid), dim3(256), 0, 0, (b2 *)buf, bytes / 2, 1u);
        hipLaunchKernelGGL(calib_write<b4>, dim3(grid), dim3(256), 0, 0, (b4 *)buf, bytes / 4, 2u);
        hipLaunchKernelGGL(calib_write<b8>, dim3(grid), dim3(256), 0, 0, (b8 *)buf, bytes / 8, 3u);
        hipLaunchKernelGGL(calib_write<b16>, dim3(grid), dim3(256), 0, 0, (b16 *)buf, bytes / 16, 4u);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::printf("fetch_calib: 8 kernels x 2 over %zu bytes\n", bytes);
    return 0;
}
