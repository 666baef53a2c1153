// k_cq_tables.hip -- the constant-Q stage's window table of a clip length, generated on the device.
//
// essentia's NSGConstantQ is constructed per file with inputSize = the file's sample count (reference
// include/hpfw/spectrum/cqt.h:54-55), so every file of a real corpus brings windows of its own: 121 bands, sum Lg = a
// quarter of a million complex values for a 30 s clip.  On the host that table was three quarters of a new length's cost
// (one libm cosine and one product per value: 3 of 4 ms on a core) and two thirds of its upload (2 of 2.9 MB); here it is
// one pass of double arithmetic per value -- trig_d.h cq_window_d, the same text plan.cpp and the oracle evaluate -- on the
// stream that carries the length's other tables.
#include "kernels.h"
#include "trig_d.h"

namespace hpfw {

// G_j[i] = hann_Lg[i] e^{+i pi 3 i^2 / M} * scale_j;  blockIdx.y = band
__global__ __launch_bounds__(256) void cq_window_kernel(const int *__restrict__ lg, const int64_t *__restrict__ g_off, CqWindowBands b,
                                                        int64_t big_m, cf *__restrict__ g)
{
    const int j = blockIdx.y;
    const int n = lg[j];
    cf *gj = g + g_off[j];
    const double scale = b.scale[j];
    const int64_t den = b.hann_den[j];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        cf v;
        cq_window_d(i, den, big_m, scale, v.r, v.i);
        gj[i] = v;
    }
}

// the same windows in the order the rows layout of the forward bins is read (kernels.h XsBandRows): band j holds
// n1 * nq2[j] entries, entry q1 nq2 + tq = the window at bin q1 + n1 (q2a + tq) - start, zero outside the band
__global__ __launch_bounds__(256) void cq_window_rows_kernel(CqPlanDev c, int n1, cf *__restrict__ g2)
{
    const int j = blockIdx.y;
    const int nq2 = c.nq2[j], q2a = c.q2a[j], start = c.start[j], n = c.lg[j];
    const cf *gj = c.g + c.g_off[j];
    cf *out = g2 + c.g2_off[j];
    const int total = n1 * nq2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int q1 = e / nq2, tq = e - q1 * nq2;
        const int64_t i = q1 + (int64_t)n1 * (q2a + tq) - start;
        out[e] = (i >= 0 && i < n) ? gj[i] : cf{0.0f, 0.0f};
    }
}

void launch_cq_windows(const CqPlanDev &c, const CqWindowBands &b, int64_t big_m, int lg_max, cf *d_g, hipStream_t s)
{
    const int bx = (lg_max + 255) / 256;
    hipLaunchKernelGGL(cq_window_kernel, dim3(bx < 1 ? 1 : (bx > 16 ? 16 : bx), kBins), dim3(256), 0, s, c.lg, c.g_off, b, big_m, d_g);
}

void launch_cq_windows_rows(const CqPlanDev &c, int n1, int max_entries, cf *d_g2, hipStream_t s)
{
    const int bx = (max_entries + 255) / 256;
    hipLaunchKernelGGL(cq_window_rows_kernel, dim3(bx < 1 ? 1 : (bx > 32 ? 32 : bx), kBins), dim3(256), 0, s, c, n1, d_g2);
}

} // namespace hpfw
