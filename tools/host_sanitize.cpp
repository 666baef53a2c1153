// Host-only code of the library (table building, Mel tables, eigen-solver, supported lengths) under the
// sanitizers -- GPU sanitizers are not available on this pool, the device side is covered by parity tests:
//   bash tools/host_sanitize.sh        (AddressSanitizer + UBSan, then ThreadSanitizer)
#include "plan.h"
#include <cstdio>
#include <string>
#include <vector>
#include <random>
extern "C" int hpfw_gpu_host_top_eigenvectors(const float *a, int n, int m, float *vec, double *val);
extern "C" long long hpfw_gpu_supported_length(long long n);
int main()
{
    const long long lens[] = {1323000, 88200, 220500, 2646000, 7938000, 54432, 1327104};
    for (long long n : lens) {
        hpfw::HostPlan p;
        std::string why;
        bool ok = hpfw::build_plan(n, p, why, false);
        std::printf("%lld %d %s classes=%zu\n", n, (int)ok, why.c_str(), p.classes.size());
    }
    hpfw::HostPlan f; std::string why;
    std::printf("frame %d\n", (int)hpfw::build_frame_transform(4410, f, why));
    std::vector<float> w, c; hpfw::mel_tables(w, c);
    std::printf("mel %zu %zu\n", w.size(), c.size());
    const int n = 200, m = 16;
    std::mt19937 g(1); std::normal_distribution<float> d;
    std::vector<float> x(n * 300), a(n * n, 0.f);
    for (auto &v : x) v = d(g);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < 300; ++k) s += x[i*300+k]*x[j*300+k]; a[i*n+j] = (float)(s/300); }
    std::vector<float> vec(m * n); std::vector<double> val(m);
    const int rc = hpfw_gpu_host_top_eigenvectors(a.data(), n, m, vec.data(), val.data());
    std::printf("eig rc %d %f\n", rc, val[0]);
    for (long long q : {10LL, 1323001LL, 5000000LL, 158760000LL}) std::printf("supp %lld -> %lld\n", q, hpfw_gpu_supported_length(q));
    return 0;
}
