// fft_lds.h -- FFTs of length 2^a or 3 * 2^a held in LDS for the chirp-z (Bluestein) bands.
//
// The arithmetic is the specification's sequence of radix-4 passes, then one radix-3 and / or one
// radix-2 pass (the descending radix list [4.., 3, 2] of the length); here two
// consecutive passes are fused in registers (16 points per thread, one LDS round trip instead of
// two), the twiddles of a fused group come from a per-butterfly table in global memory (built by
// the host from T_N, entry e of butterfly b at [e][b]: coalesced, L2 resident, no index
// arithmetic), and the data sit in a padded image (one spare slot per 16) so that the stride-16
// accesses of the innermost pass do not collide on LDS banks.  None of this changes a rounding.
#pragma once
#include "device_math.h"

namespace hpfw {

HPFW_DEVICE int pad16(int i) { return i + (i >> 4); }

// Padded positions of the points base + q2 M2 + q M1 of a butterfly (base = blk LEN + j0, j0 < M2) without a shift and an
// add per access: where the spare slots fall is known at compile time in most groups, and the position is then one
// per-thread value plus a constant the LDS instruction carries as its offset.
//   FOLD:  LEN a multiple of 16 and M2 a multiple or a divisor of 16 -- (j0 + c) >> 4 = (j0 >> 4) + (c >> 4) for every
//          offset c (a multiple of M2): position = pad16(base) + c + (c >> 4);
//   ROWS:  only M1 a multiple of 16 -- one padded position per q2, plus q (M1 + M1 / 16);
//   else the general form.
template <int LEN, int M1, int M2, int R2>
struct PadAt {
    static constexpr bool FOLD = LEN % 16 == 0 && (M2 % 16 == 0 || 16 % M2 == 0);
    static constexpr bool ROWS = !FOLD && M1 % 16 == 0;
    int base;
    int p[FOLD ? 1 : (ROWS ? R2 : 1)];
    HPFW_DEVICE_MEMBER explicit PadAt(int b) : base(b)
    {
        if constexpr (FOLD) {
            p[0] = pad16(b);
        } else if constexpr (ROWS) {
#pragma unroll
            for (int q2 = 0; q2 < R2; ++q2) p[q2] = pad16(b + q2 * M2);
        } else {
            p[0] = 0;
        }
    }
    HPFW_DEVICE_MEMBER int operator()(int q2, int q) const // the position of base + q2 M2 + q M1
    {
        if constexpr (FOLD) {
            const int c = q2 * M2 + q * M1;
            return p[0] + c + (c >> 4);
        } else if constexpr (ROWS) {
            return p[q2] + q * (M1 + M1 / 16);
        } else {
            return pad16(base + q2 * M2 + q * M1);
        }
    }
};

template <int N_>
struct Size {
    static constexpr int N = N_;
    static constexpr int DATA = N + N / 16; // padded data slots (complex)
};

constexpr bool cq_size_ok(int n) // 2^a or 3 * 2^a, 64 <= n <= 16384
{
    if (n < 64 || n > 16384) return false;
    if (n % 3 == 0) n /= 3;
    return (n & (n - 1)) == 0;
}

constexpr int kCqMaxGroups = 4;

// where each fused group's twiddles start in the class table (complex elements); entry layout as
// in plan.cpp append_group_twiddles: q2 (R1-1) + (s-1) for stage 1, (R1-1) R2 + (s2-1) for stage 2
struct CqTwiddles {
    const cf *tab;
    int off[kCqMaxGroups]; // outer groups, outermost first
    int mid_off;           // innermost group: one butterfly's entries (they do not depend on b)
};

// ---- one fused group of the forward DIF: radix R1 at sub-length LEN, then radix R2 (or 1) ----
// PRUNE: only inputs with index < nz inside the first quarter can be non-zero (nz <= LEN / R1 and
// LEN == N): a radix-R1 butterfly whose inputs 1..R1-1 are zero returns its input 0 on every
// output, so those loads and adds are skipped and the zero padding is never read.
// tw(e): entry e of THIS butterfly's twiddles -- TwAtUse (a load where it is used) or TwPairs (fetched before the barrier in
// front of the group: device_math.h)
template <int N, int LEN, int R1, int R2, bool PRUNE, class Lds, class Tw>
HPFW_DEVICE void dif_butterfly(Lds &lds, int b, int nz, const Tw &tw)
{
    constexpr int M1 = LEN / R1, M2 = M1 / R2;
    {
        const int blk = b / M2, j0 = b % M2;
        const int base = blk * LEN + j0;
        const PadAt<LEN, M1, M2, R2> at(base);
        cf e[R1][R2];
#pragma unroll
        for (int q2 = 0; q2 < R2; ++q2) {
            const int j = j0 + q2 * M2;
            cf u[R1];
            if (PRUNE) {
                cf u0 = {0.0f, 0.0f};
                if (j < nz) u0 = lds[at(q2, 0)]; // LEN = N: blk = 0 and base + q2 M2 = j
#pragma unroll
                for (int s = 0; s < R1; ++s) u[s] = u0;
            } else {
#pragma unroll
                for (int q = 0; q < R1; ++q) u[q] = lds[at(q2, q)];
                Dft<R1>::run(u);
            }
            e[0][q2] = u[0];
#pragma unroll
            for (int s = 1; s < R1; ++s) e[s][q2] = c_mul(u[s], tw(q2 * (R1 - 1) + (s - 1)));
        }
#pragma unroll
        for (int s = 0; s < R1; ++s) {
            if constexpr (R2 > 1) {
                cf v[R2];
#pragma unroll
                for (int q2 = 0; q2 < R2; ++q2) v[q2] = e[s][q2];
                Dft<R2>::run(v);
                lds[at(0, s)] = v[0];
#pragma unroll
                for (int s2 = 1; s2 < R2; ++s2) lds[at(s2, s)] = c_mul(v[s2], tw((R1 - 1) * R2 + (s2 - 1)));
            } else {
                lds[at(0, s)] = e[s][0];
            }
        }
    }
}

template <int N, int LEN, int R1, int R2, bool PRUNE, class Lds>
HPFW_DEVICE void dif_group(Lds &lds, const cf *__restrict__ gt, int tid, int nthreads, int nz)
{
    constexpr int M2 = LEN / (R1 * R2), NB = Size<N>::N / (R1 * R2);
    for (int b = tid; b < NB; b += nthreads) dif_butterfly<N, LEN, R1, R2, PRUNE>(lds, b, nz, TwAtUse{gt, M2, b % M2});
}

// inverse butterfly of the specification: swap, forward codelet, swap.  For radix 2, 3 and 4 the same operations on the
// same values, written without the swaps (which cost register moves): multiplying by +i
// instead of -i, x - (-y) for x + y and x + (-y) for x - y, which round identically.
template <int R>
struct Idft {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = {u[q].i, u[q].r};
        Dft<R>::run(u);
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = {u[q].i, u[q].r};
    }
};
template <>
struct Idft<2> {
    HPFW_DEVICE_STATIC void run(cf *u) { Dft<2>::run(u); }
};
template <>
struct Idft<3> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        const float s = 0.86602540378443864676f;
        cf t1 = c_add(u[1], u[2]);
        cf d = c_sub(u[1], u[2]);
        cf m1 = c_fma_s(-0.5f, t1, u[0]);
        cf sd = c_scale(s, d); // m1 +- i s d
        u[0] = c_add(u[0], t1);
        u[1] = c_sub_mi(m1, sd);
        u[2] = c_add_mi(m1, sd);
    }
};
template <>
struct Idft<4> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        cf t0 = c_add(u[0], u[2]);
        cf t1 = c_sub(u[0], u[2]);
        cf t2 = c_add(u[1], u[3]);
        cf d = c_sub(u[1], u[3]);
        u[0] = c_add(t0, t2);
        u[2] = c_sub(t0, t2);
        u[1] = c_sub_mi(t1, d); // t1 +- i d
        u[3] = c_add_mi(t1, d);
    }
};
template <int R>
HPFW_DEVICE void idft(cf *u)
{
    Idft<R>::run(u);
}

// ---- one fused group of the inverse DIT: radix R2 (or 1) at sub-length LEN / R1, then R1 ----
// Same table as the forward group (the kernel conjugates).  Outputs with index >= keep are not
// stored (only the first C samples of the last group are used).
template <int N, int LEN, int R1, int R2, class Lds, class Tw>
HPFW_DEVICE void idit_butterfly(Lds &lds, int b, int keep, const Tw &tw)
{
    constexpr int M1 = LEN / R1, M2 = M1 / R2;
    {
        const int blk = b / M2, j0 = b % M2;
        const int base = blk * LEN + j0;
        const PadAt<LEN, M1, M2, R2> at(base);
        cf o[R1][R2];
#pragma unroll
        for (int q = 0; q < R1; ++q) {
            cf v[R2];
            v[0] = lds[at(0, q)];
#pragma unroll
            for (int q2 = 1; q2 < R2; ++q2)
                v[q2] = c_mulc(lds[at(q2, q)], tw((R1 - 1) * R2 + (q2 - 1)));
            if constexpr (R2 > 1) idft<R2>(v);
#pragma unroll
            for (int s2 = 0; s2 < R2; ++s2) o[q][s2] = v[s2];
        }
#pragma unroll
        for (int s2 = 0; s2 < R2; ++s2) {
            const int j = j0 + s2 * M2;
            cf u[R1];
            u[0] = o[0][s2];
#pragma unroll
            for (int q = 1; q < R1; ++q) u[q] = c_mulc(o[q][s2], tw(s2 * (R1 - 1) + (q - 1)));
            idft<R1>(u);
#pragma unroll
            for (int s = 0; s < R1; ++s) {
                const int idx = blk * LEN + j + s * M1;
                if (idx < keep) lds[at(s2, s)] = u[s];
            }
        }
    }
}

template <int N, int LEN, int R1, int R2, class Lds>
HPFW_DEVICE void idit_group(Lds &lds, const cf *__restrict__ gt, int tid, int nthreads, int keep)
{
    constexpr int M2 = LEN / (R1 * R2), NB = Size<N>::N / (R1 * R2);
    for (int b = tid; b < NB; b += nthreads) idit_butterfly<N, LEN, R1, R2>(lds, b, keep, TwAtUse{gt, M2, b % M2});
}

// ---- the innermost group of both transforms in one register pass:
// last DIF group (R1, R2 at LEN), pointwise product with V (digit-reversed positions = the LDS
// positions), first inverse DIT group (R2, then R1) -- the same LEN points in both directions.
// mt: the group's stage-1 entries (q2 (R1-1) + (s-1)), the same for every butterfly.
template <int N, int LEN, int R1, int R2, class Lds>
HPFW_DEVICE void mid_group(Lds &lds, const cf *__restrict__ mt, int tid, int nthreads, const cf *__restrict__ vrev)
{
    using P = Size<N>;
    static_assert(LEN == R1 * R2, "the innermost group spans whole sub-transforms");
    constexpr int M1 = LEN / R1, M2 = 1;
    constexpr int NB = P::N / LEN;
    for (int b = tid; b < NB; b += nthreads) {
        const int base = b * LEN;
        // the block's LEN points start at a multiple of LEN: LEN a multiple of 16, or a divisor (no spare slot inside the block)
        const PadAt<(LEN % 16 == 0 || 16 % LEN == 0) ? 16 : LEN, M1, M2, R2> at(base);
        cf e[R1][R2];
        // forward: radix R1 over q at j = q2 (j0 = 0), then radix R2
#pragma unroll
        for (int q2 = 0; q2 < R2; ++q2) {
            cf u[R1];
#pragma unroll
            for (int q = 0; q < R1; ++q) u[q] = lds[at(q2, q)];
            Dft<R1>::run(u);
            e[0][q2] = u[0];
#pragma unroll
            for (int s = 1; s < R1; ++s) e[s][q2] = c_mul(u[s], mt[q2 * (R1 - 1) + (s - 1)]);
        }
#pragma unroll
        for (int s = 0; s < R1; ++s) {
            cf v[R2];
#pragma unroll
            for (int q2 = 0; q2 < R2; ++q2) v[q2] = e[s][q2];
            if constexpr (R2 > 1) {
                Dft<R2>::run(v);
                // the stage-2 twiddles are T_N[TS2 * j0 * s2] with j0 = 0, i.e. T_N[0] = (1, -0):
                // that product returns its operand (only the sign of a zero can differ), so it is skipped
            }
            // pointwise product at positions base + s M1 + s2: the R2 factors lie side by side, fetched two at a time
            // where that keeps 16-byte alignment (LEN and M1 even)
            if constexpr (R2 % 2 == 0 && M1 % 2 == 0 && LEN % 2 == 0) {
                const cf2 *__restrict__ vp = reinterpret_cast<const cf2 *>(vrev + base + s * M1);
#pragma unroll
                for (int s2 = 0; s2 < R2; s2 += 2) {
                    const cf2 w = vp[s2 >> 1];
                    v[s2] = c_mul(v[s2], w.a);
                    v[s2 + 1] = c_mul(v[s2 + 1], w.b);
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < R2; ++s2) v[s2] = c_mul(v[s2], vrev[base + s * M1 + s2]);
            }
            // inverse inner radix R2 at j0 = 0: the twiddles conj(T_N[0]) are skipped likewise
            if constexpr (R2 > 1) idft<R2>(v);
#pragma unroll
            for (int s2 = 0; s2 < R2; ++s2) e[s][s2] = v[s2];
        }
        // inverse outer radix R1 at j = s2
#pragma unroll
        for (int s2 = 0; s2 < R2; ++s2) {
            cf u[R1];
            u[0] = e[0][s2];
#pragma unroll
            for (int q = 1; q < R1; ++q) u[q] = c_mulc(e[q][s2], mt[s2 * (R1 - 1) + (q - 1)]);
            idft<R1>(u);
#pragma unroll
            for (int s = 0; s < R1; ++s) lds[at(s2, s)] = u[s];
        }
    }
}

// ---- drivers: fused groups of the pass list [4 .., 3, 2] ----
// LEN = the sub-transform length still to be done: 2^a or 3 * 2^a.  The next pass is radix 4 while
// 4 divides LEN, then 3, then 2; a group fuses it with the pass after it (16, 12, 8 or 6 points per
// thread) when there is one.  The innermost group is handled by mid_group; the outer ones by
// dif_group on the way in and idit_group on the way out.  (plan.cpp walks the same list to lay out
// the twiddle tables.)
constexpr int cq_next_radix(int len) { return len % 4 == 0 ? 4 : (len % 3 == 0 ? 3 : 2); }

template <int LEN>
struct GroupOf {
    static constexpr int R1 = cq_next_radix(LEN);
    static constexpr int R2 = LEN / R1 > 1 ? cq_next_radix(LEN / R1) : 1;
    static constexpr int REST = LEN / (R1 * R2);
};

template <int N, int LEN, int G, class Lds>
HPFW_DEVICE void cq_transform(Lds &lds, const CqTwiddles &tw, int nthreads, int nz, int keep,
                              const cf *__restrict__ vrev)
{
    using GO = GroupOf<LEN>;
    constexpr bool FIRST = (G == 0);
    if constexpr (GO::REST == 1) {
        const cf *mt = tw.tab + tw.mid_off;
        HPFW_FOR_THREADS(t, nthreads) { mid_group<N, LEN, GO::R1, GO::R2>(lds, mt, t, nthreads, vrev); }
        HPFW_BARRIER();
    } else {
        static_assert(G < kCqMaxGroups, "too many fused groups");
        const cf *gt = tw.tab + tw.off[G];
        if (FIRST && nz <= LEN / GO::R1) {
            HPFW_FOR_THREADS(t, nthreads) { dif_group<N, LEN, GO::R1, GO::R2, true>(lds, gt, t, nthreads, nz); }
        } else {
            HPFW_FOR_THREADS(t, nthreads) { dif_group<N, LEN, GO::R1, GO::R2, false>(lds, gt, t, nthreads, nz); }
        }
        HPFW_BARRIER();
        cq_transform<N, GO::REST, G + 1>(lds, tw, nthreads, nz, keep, vrev);
        HPFW_FOR_THREADS(t, nthreads)
        {
            idit_group<N, LEN, GO::R1, GO::R2>(lds, gt, t, nthreads, FIRST ? keep : N);
        }
        HPFW_BARRIER();
    }
}

#if !defined(HPFW_SIMT_EMU)
// (-DHPFW_CQ_PREFETCH; off: measured 2.5 % SLOWER than the loads where they are used -- eight runs on one box, the stage's
// time over the hashprint kernel's of the same run 1.472 against 1.436 -- unlike the row stage, where the same change took a
// quarter out of the first group: here the compiler already issues a group's loads together, other workgroups and
// classes cover the wait, and the 16 more registers cost more than the L2 round trip.)
// The same sequence with the twiddles of every outer group fetched BEFORE the barrier in front of it (one butterfly per
// thread in those groups: N / 16 or N / 12 of them for cq_threads(N) threads): they arrive while the workgroup waits at
// the barrier, where otherwise every group began with an L2 round trip.  `mine`: this level's pairs, fetched by the
// caller for the way in; `outer`: the level above, whose pairs are fetched again in front of this level's last barrier,
// for its way out.
struct CqNoPairs {
    HPFW_DEVICE_MEMBER void refetch() const {}
};
template <int R1, int R2>
struct CqPairs : TwPairs<R1, R2> {
    const cf *gt;
    int nb, b; // b = tid mod nb: a valid butterfly of the table for every thread, so the fetch needs no guard (a guarded
               // one would keep the old values alive through the inner levels)
    HPFW_DEVICE_MEMBER void refetch() { this->fetch(gt, nb, b); }
};

template <int N, int LEN, int G, class Lds, class Mine, class Outer>
HPFW_DEVICE void cq_transform_fetched(Lds &lds, const CqTwiddles &tw, int nthreads, int nz, int keep, const cf *__restrict__ vrev,
                                      Mine &mine, Outer &outer)
{
    using GO = GroupOf<LEN>;
    const int tid = threadIdx.x;
    if constexpr (GO::REST == 1) {
        mid_group<N, LEN, GO::R1, GO::R2>(lds, tw.tab + tw.mid_off, tid, nthreads, vrev);
        outer.refetch();
        HPFW_BARRIER();
    } else {
        constexpr int NB = Size<N>::N / (GO::R1 * GO::R2), M2 = LEN / (GO::R1 * GO::R2);
        if (tid < NB) {
            if (G == 0 && nz <= LEN / GO::R1)
                dif_butterfly<N, LEN, GO::R1, GO::R2, true>(lds, tid, nz, mine);
            else
                dif_butterfly<N, LEN, GO::R1, GO::R2, false>(lds, tid, nz, mine);
        }
        using GN = GroupOf<GO::REST>;
        if constexpr (GN::REST == 1) {       // the innermost group comes next: its twiddles are scalar loads
            CqNoPairs none;
            HPFW_BARRIER();
            cq_transform_fetched<N, GO::REST, G + 1>(lds, tw, nthreads, nz, keep, vrev, none, mine);
        } else {
            constexpr int M2N = GO::REST / (GN::R1 * GN::R2);
            CqPairs<GN::R1, GN::R2> next;
            next.gt = tw.tab + tw.off[G + 1];
            next.nb = M2N;
            next.b = tid % M2N;
            next.refetch();
            HPFW_BARRIER();
            cq_transform_fetched<N, GO::REST, G + 1>(lds, tw, nthreads, nz, keep, vrev, next, mine);
        }
        if (tid < NB) idit_butterfly<N, LEN, GO::R1, GO::R2>(lds, tid, G == 0 ? keep : N, mine);
        outer.refetch();
        HPFW_BARRIER();
        (void)M2;
    }
}
#endif

// ---- the whole band: window*chirp, forward FFT, times V, inverse FFT, magnitudes ----
// lds: Size<NP>::DATA complex slots; red: one float per thread (the largest value it stored).
// xs.for_each(tid, nthreads, lg, g, f): f(i, X[start_j + i] g[i]) for every i < lg, in whatever order the layout of the
// forward bins makes cheap (a plain pointer in the emulation; kernels.h XsBand / XsBandRows in the kernels: the forward
// transform leaves the bins in rows k mod n1, and g is then the window permuted likewise).
// fin(m): what is stored for magnitude m -- m itself, or its dB term (monotone in m, so the
// largest stored value belongs to the largest magnitude either way).
struct XsPtr {
    const cf *p;
    HPFW_DEVICE_MEMBER cf operator()(int i) const { return p[i]; }
    template <class F>
    HPFW_DEVICE_MEMBER void for_each(int tid, int nthreads, int lg, const cf *__restrict__ g, F f) const
    {
        for (int i = tid; i < lg; i += nthreads) f(i, c_mul(p[i], g[i]));
    }
};

// stp: diagnosis builds (-DHPFW_CQ_STAMPS, tools/cq_stamps.py) only -- s_memtime of thread 0 at the phase boundaries
#if defined(HPFW_CQ_STAMPS) && !defined(HPFW_SIMT_EMU)
#define HPFW_CQ_STAMP(k)                                                                  \
    do {                                                                                  \
        if (stp && threadIdx.x == 0) stp[k] = __builtin_amdgcn_s_memtime();               \
    } while (0)
#else
#define HPFW_CQ_STAMP(k) ((void)0)
#endif
template <int NP, class Lds, class Red, class Xs, class Fin>
HPFW_DEVICE void cq_band_body(Lds &lds, Red &red, int nthreads, const Xs &xs, const cf *__restrict__ g,
                              int lg, const CqTwiddles &tw, const cf *__restrict__ vrev, int c,
                              float *__restrict__ out_mag, Fin fin, long long *stp = nullptr)
{
    HPFW_CQ_STAMP(0);
    using P = Size<NP>;
    static_assert(cq_size_ok(NP), "chirp-z lengths are 2^a or 3 * 2^a");
    const bool prune = lg <= P::N / 4; // the first pass is radix 4 for every admitted length
    HPFW_FOR_THREADS(tid, nthreads)
    {
        xs.for_each(tid, nthreads, lg, g, [&](int i, cf v) { lds[pad16(i)] = v; });
        if (!prune)
            for (int i = lg + tid; i < P::N; i += nthreads) lds[pad16(i)] = {0.0f, 0.0f};
    }
#if !defined(HPFW_SIMT_EMU) && defined(HPFW_CQ_PREFETCH)
    if constexpr (GroupOf<NP>::REST != 1) {
        using G0 = GroupOf<NP>;
        constexpr int NB0 = NP / (G0::R1 * G0::R2), M20 = NP / (G0::R1 * G0::R2);
        static_assert(NB0 <= 1024, "one butterfly per thread in the outer groups");
        CqPairs<G0::R1, G0::R2> first;
        first.gt = tw.tab + tw.off[0];
        first.nb = M20;
        first.b = (int)threadIdx.x % M20;
        first.refetch();
        CqNoPairs none;
        HPFW_BARRIER();
        cq_transform_fetched<NP, NP, 0>(lds, tw, nthreads, lg, c, vrev, first, none);
    } else
#endif
    {
        HPFW_CQ_STAMP(1);
        HPFW_BARRIER();
        HPFW_CQ_STAMP(2);
        cq_transform<NP, NP, 0>(lds, tw, nthreads, lg, c, vrev);
        HPFW_CQ_STAMP(3);
    }
    HPFW_FOR_THREADS(tid, nthreads)
    {
        float mx = fin(0.0f);
        for (int i = tid; i < c; i += nthreads) {
            const cf v = lds[pad16(i)];
            const float m = fin(__builtin_sqrtf(HPFW_FMAF(v.r, v.r, v.i * v.i)));
            out_mag[i] = m;
            mx = m > mx ? m : mx;
        }
        red[tid] = mx;
    }
    HPFW_CQ_STAMP(4);
    (void)stp;
}

} // namespace hpfw
