// eigen_host.cpp -- HashprintHandle::calc_filters on the host (reference
// include/hpfw/core/hashprint_handle.h:105-112: SelfAdjointEigenSolver of the accumulated covariance,
// eigenvectors by descending eigenvalue, the first 64 as filter rows).  Runs once per index();
// the reference does it serially with Eigen (18 s for 2420 x 2420); here 0.4 s on 16 host threads.
//
// Only the 64 leading eigenvectors are needed, so: Householder reduction to tridiagonal form
// (double), the 64 largest eigenvalues by Sturm-sequence bisection, their eigenvectors by inverse
// iteration on the tridiagonal matrix, back-transformation with the stored reflectors.
// The sign of an eigenvector is arbitrary in the reference (whatever Eigen returns); here the
// component of largest magnitude is made positive so that results are reproducible.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

namespace hpfw {

namespace {

// A team of host threads for the row loops of the reduction (4840 short parallel phases per solve, so
// the workers spin on a generation counter instead of sleeping).  Every row is computed by one
// thread in the same order as the serial code: results do not depend on the number of threads.
class Team {
public:
    explicit Team(int n_threads) : n_(n_threads < 1 ? 1 : n_threads)
    {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~Team()
    {
        stop_.store(true, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_acq_rel);
        for (auto &w : workers_) w.join();
    }
    int size() const { return n_; }
    // fn(begin, end) over a static partition of [0, count)
    void run(int count, const std::function<void(int, int)> &fn)
    {
        if (n_ == 1 || count < 64) {
            fn(0, count);
            return;
        }
        fn_ = &fn;
        count_ = count;
        done_.store(0, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_acq_rel);
        part(0);
        while (done_.load(std::memory_order_acquire) != n_ - 1) std::this_thread::yield();
    }

private:
    void part(int t) const
    {
        const int per = (count_ + n_ - 1) / n_;
        const int b = std::min(count_, t * per), e = std::min(count_, b + per);
        if (b < e) (*fn_)(b, e);
    }
    void loop(int t)
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g;
            int spins = 0;
            while ((g = gen_.load(std::memory_order_acquire)) == seen)
                if (++spins > 2000) std::this_thread::yield();
            seen = g;
            if (stop_.load(std::memory_order_acquire)) return;
            part(t);
            done_.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    const int n_;
    std::vector<std::thread> workers_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> done_{0};
    std::atomic<bool> stop_{false};
    const std::function<void(int, int)> *fn_ = nullptr;
    int count_ = 0;
};

int team_size()
{
    if (const char *e = std::getenv("HPFW_EIGEN_THREADS")) return std::max(1, std::atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hc ? hc : 1, 16); // measured on the MI355X host: 0.38 s at 16 threads, 4.9 s at 1, slower again beyond 32
}

// A (n x n, symmetric, full storage, row-major) -> tridiagonal (d, e); reflector i is
// v = [1; A[i+2..n-1][i]] acting on rows/cols i+1..n-1, with factor tau[i].
void tridiagonalize(std::vector<double> &a, int n, std::vector<double> &d, std::vector<double> &e,
                    std::vector<double> &tau, Team &team)
{
    d.assign(n, 0.0);
    e.assign(n > 1 ? n - 1 : 0, 0.0);
    tau.assign(n > 1 ? n - 1 : 0, 0.0);
    std::vector<double> v(n), p(n), w(n);
    for (int i = 0; i + 1 < n; ++i) {
        const int m = n - i - 1; // order of the trailing block
        double alpha = a[(size_t)(i + 1) * n + i];
        double xnorm2 = 0.0;
        for (int r = i + 2; r < n; ++r) xnorm2 += a[(size_t)r * n + i] * a[(size_t)r * n + i];
        double t = 0.0, beta = alpha;
        if (xnorm2 > 0.0) {
            beta = -std::copysign(std::sqrt(alpha * alpha + xnorm2), alpha);
            t = (beta - alpha) / beta;
            const double scale = 1.0 / (alpha - beta);
            v[0] = 1.0;
            for (int r = i + 2; r < n; ++r) v[r - i - 1] = a[(size_t)r * n + i] * scale;
        }
        e[i] = beta;
        tau[i] = t;
        d[i] = a[(size_t)i * n + i];
        if (t != 0.0) {
            // p = t * B v, B = trailing block
            team.run(m, [&](int r0, int r1) {
                for (int r = r0; r < r1; ++r) {
                    const double *row = &a[(size_t)(i + 1 + r) * n + (i + 1)];
                    double s = 0.0;
                    for (int c = 0; c < m; ++c) s += row[c] * v[c];
                    p[r] = t * s;
                }
            });
            double pv = 0.0;
            for (int r = 0; r < m; ++r) pv += p[r] * v[r];
            const double k = 0.5 * t * pv;
            for (int r = 0; r < m; ++r) w[r] = p[r] - k * v[r];
            // B -= v w^T + w v^T
            team.run(m, [&](int r0, int r1) {
                for (int r = r0; r < r1; ++r) {
                    double *row = &a[(size_t)(i + 1 + r) * n + (i + 1)];
                    const double vr = v[r], wr = w[r];
                    for (int c = 0; c < m; ++c) row[c] -= vr * w[c] + wr * v[c];
                }
            });
            // keep the reflector in column i below the subdiagonal
            for (int r = i + 2; r < n; ++r) a[(size_t)r * n + i] = v[r - i - 1];
        } else {
            for (int r = i + 2; r < n; ++r) a[(size_t)r * n + i] = 0.0;
        }
    }
    d[n - 1] = a[(size_t)(n - 1) * n + (n - 1)];
}

// number of eigenvalues of the tridiagonal (d, e) that are < x
int sturm_count(const std::vector<double> &d, const std::vector<double> &e, double x, double tiny)
{
    int cnt = 0;
    double q = d[0] - x;
    if (q < 0) ++cnt;
    for (size_t i = 1; i < d.size(); ++i) {
        if (std::fabs(q) < tiny) q = q < 0 ? -tiny : tiny;
        q = d[i] - x - e[i - 1] * e[i - 1] / q;
        if (q < 0) ++cnt;
    }
    return cnt;
}

// solve (T - lambda I) y = b for tridiagonal T by Gaussian elimination with partial pivoting
void solve_shifted(const std::vector<double> &d, const std::vector<double> &e, double lambda, double tiny,
                   std::vector<double> &y)
{
    const int n = (int)d.size();
    std::vector<double> diag(n), up1(n, 0.0), up2(n, 0.0), rhs(y);
    // row i: sub e[i-1], diag d[i]-lambda, super e[i]
    std::vector<double> sub(n, 0.0);
    for (int i = 0; i < n; ++i) {
        diag[i] = d[i] - lambda;
        if (i + 1 < n) up1[i] = e[i];
        if (i > 0) sub[i] = e[i - 1];
    }
    for (int i = 0; i + 1 < n; ++i) {
        if (std::fabs(sub[i + 1]) > std::fabs(diag[i])) { // swap rows i and i+1
            std::swap(diag[i], sub[i + 1]);
            std::swap(up1[i], diag[i + 1]);
            std::swap(up2[i], up1[i + 1]);
            std::swap(rhs[i], rhs[i + 1]);
        }
        if (std::fabs(diag[i]) < tiny) diag[i] = tiny;
        const double f = sub[i + 1] / diag[i];
        diag[i + 1] -= f * up1[i];
        up1[i + 1] -= f * up2[i];
        rhs[i + 1] -= f * rhs[i];
    }
    if (std::fabs(diag[n - 1]) < tiny) diag[n - 1] = tiny;
    y[n - 1] = rhs[n - 1] / diag[n - 1];
    if (n > 1) y[n - 2] = (rhs[n - 2] - up1[n - 2] * y[n - 1]) / diag[n - 2];
    for (int i = n - 3; i >= 0; --i) y[i] = (rhs[i] - up1[i] * y[i + 1] - up2[i] * y[i + 2]) / diag[i];
}

} // namespace

// cov: n x n symmetric, row-major float.  out: m rows of length n (row r = unit eigenvector of the
// r-th largest eigenvalue).  evals (optional): the m eigenvalues.  Returns 0, or -1 on bad input.
int top_eigenvectors(const float *cov, int n, int m, float *out, double *evals)
{
    if (!cov || !out || n < 1 || m < 1 || m > n) return -1;
    std::vector<double> a((size_t)n * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) a[(size_t)r * n + c] = 0.5 * ((double)cov[(size_t)r * n + c] + (double)cov[(size_t)c * n + r]);
    std::vector<double> d, e, tau;
    Team team(team_size());
    tridiagonalize(a, n, d, e, tau, team);
    // Gershgorin bounds and scale
    double lo = d[0], hi = d[0], tnorm = 0.0;
    for (int i = 0; i < n; ++i) {
        const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0);
        lo = std::min(lo, d[i] - r);
        hi = std::max(hi, d[i] + r);
        tnorm = std::max(tnorm, std::fabs(d[i]) + r);
    }
    const double tiny = std::max(tnorm, 1e-300) * 1e-18;
    std::vector<double> lam(m);
    for (int r = 0; r < m; ++r) { // r-th largest: count(x) <= n - 1 - r < count(x + )
        const int target = n - 1 - r; // eigenvalue index in ascending order
        double a0 = lo, b0 = hi;
        for (int it = 0; it < 200 && b0 - a0 > 1e-15 * std::max(std::fabs(a0), std::fabs(b0)) + 1e-300; ++it) {
            const double mid = 0.5 * (a0 + b0);
            if (sturm_count(d, e, mid, tiny) <= target) a0 = mid; else b0 = mid;
        }
        lam[r] = 0.5 * (a0 + b0);
    }
    std::vector<std::vector<double>> vecs(m, std::vector<double>(n));
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    for (int r = 0; r < m; ++r) {
        std::vector<double> &y = vecs[r];
        for (int i = 0; i < n; ++i) { // deterministic pseudo-random start
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            y[i] = (double)((seed >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
        }
        // separate coincident eigenvalues slightly so that inverse iteration does not return the same vector
        double shift = lam[r];
        for (int k = 0; k < r; ++k)
            if (std::fabs(lam[k] - shift) < 1e-12 * tnorm) shift -= 1e-11 * tnorm;
        for (int it = 0; it < 5; ++it) {
            solve_shifted(d, e, shift, tiny, y);
            for (int k = 0; k < r; ++k) { // modified Gram-Schmidt against the vectors already found
                double dot = 0.0;
                for (int i = 0; i < n; ++i) dot += vecs[k][i] * y[i];
                for (int i = 0; i < n; ++i) y[i] -= dot * vecs[k][i];
            }
            double nrm = 0.0;
            for (int i = 0; i < n; ++i) nrm += y[i] * y[i];
            nrm = std::sqrt(nrm);
            if (nrm == 0.0) break;
            for (int i = 0; i < n; ++i) y[i] /= nrm;
        }
    }
    // back-transform: z = H(0) H(1) ... H(n-2) y, the vectors shared out over the team
    std::vector<std::thread> pool;
    const int nt = std::min(team.size(), m);
    auto back = [&](int r_begin, int r_end) {
    for (int r = r_begin; r < r_end; ++r) {
        std::vector<double> z = vecs[r];
        for (int i = n - 2; i >= 0; --i) {
            if (tau[i] == 0.0) continue;
            double s = z[i + 1];
            for (int q = i + 2; q < n; ++q) s += a[(size_t)q * n + i] * z[q];
            s *= tau[i];
            z[i + 1] -= s;
            for (int q = i + 2; q < n; ++q) z[q] -= s * a[(size_t)q * n + i];
        }
        int imax = 0;
        for (int i = 1; i < n; ++i)
            if (std::fabs(z[i]) > std::fabs(z[imax])) imax = i;
        const double sgn = z[imax] < 0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) out[(size_t)r * n + i] = (float)(sgn * z[i]);
        if (evals) evals[r] = lam[r];
    }
    };
    const int per = (m + nt - 1) / nt;
    for (int t = 1; t < nt; ++t) pool.emplace_back(back, std::min(m, t * per), std::min(m, (t + 1) * per));
    back(0, std::min(m, per));
    for (auto &th : pool) th.join();
    return 0;
}

} // namespace hpfw

// C entry point for tests of the host solver (no device involved)
extern "C" int hpfw_gpu_host_top_eigenvectors(const float *cov, int n, int m, float *out, double *evals)
{
    return hpfw::top_eigenvectors(cov, n, m, out, evals);
}
