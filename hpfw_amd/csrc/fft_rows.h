// fft_rows.h -- the length-n2 mixed-radix FFT of one pair of residue sequences, held in LDS
// (DESIGN.md S4, S6).  Same arithmetic as the specification's radix-7/5/4/3/2 passes; two
// consecutive passes are fused in registers (up to 49 points per thread and LDS round trip) and,
// when 4 | n2, the twiddles T_n2[m] come from the first quarter of the table kept in LDS
// (T[m + n2/4] = -i T[m] exactly, by the table's construction).
#pragma once
#include "device_math.h"

namespace hpfw {

constexpr int kRowsMaxGroups = 12;
constexpr int kRowsMaxPoints = 36; // largest fused pair (plan.cpp uses the same bound)

struct i16x2 { // two consecutive residues of one time step
    short x, y;
};

struct RowGroups {
    int n;                    // number of fused groups
    int r1[kRowsMaxGroups];   // first radix of the group
    int r2[kRowsMaxGroups];   // second radix, or 1
};

struct RowsArgs {
    int n1, n2, h;            // N = n1 * n2, h = n2 / 2 + 1
    int quad;                 // 1: quadrant twiddle table in LDS (4 | n2), 0: global table
    RowGroups groups;
    const cf *tw_n2;          // T_{n2} (global)
    const cf *tw_big;         // T_N[a * k2]  [n1][h]
    const int *pos_n2;        // digit-reversed position of output k2
};

// T_n2[m], 0 <= m < n2
template <class Lds>
HPFW_DEVICE cf rows_tw(const Lds &lds, const RowsArgs &a, int m)
{
    if (a.quad) {
        const int nq = a.n2 >> 2;
        const int q = (m >= nq) + (m >= 2 * nq) + (m >= 3 * nq);
        const cf e = lds[a.n2 + (m - q * nq)];
        // (-i)^q * e
        const float re = (q & 1) ? e.i : e.r;
        const float im = (q & 1) ? e.r : e.i;
        const bool neg_re = (q == 2) || (q == 3);
        const bool neg_im = (q == 1) || (q == 2);
        return {neg_re ? -re : re, neg_im ? -im : im};
    }
    return a.tw_n2[m];
}

// one fused group of the forward DIF at sub-length len: radix R1, then radix R2 (or 1)
template <int R1, int R2, class Lds>
HPFW_DEVICE void rows_group(Lds &lds, const RowsArgs &a, int len, int tid, int nthreads)
{
    const int n = a.n2;
    const int m1 = len / R1, m2 = m1 / R2;
    const int ts1 = n / len, ts2 = n / m1;
    const int nb = n / (R1 * R2);
    const float inv_m2 = 1.0f / (float)m2;
    for (int b = tid; b < nb; b += nthreads) {
        int blk = (int)((float)b * inv_m2);
        if (blk * m2 > b) --blk;
        if ((blk + 1) * m2 <= b) ++blk;
        const int j0 = b - blk * m2;
        const int base = blk * len + j0;
        cf e[R1][R2];
#pragma unroll
        for (int q2 = 0; q2 < R2; ++q2) {
            const int j = j0 + q2 * m2;
            cf u[R1];
#pragma unroll
            for (int q = 0; q < R1; ++q) u[q] = lds[base + q2 * m2 + q * m1];
            Dft<R1>::run(u);
            e[0][q2] = u[0];
#pragma unroll
            for (int s = 1; s < R1; ++s) e[s][q2] = c_mul(u[s], rows_tw(lds, a, ts1 * j * s));
        }
#pragma unroll
        for (int s = 0; s < R1; ++s) {
            if constexpr (R2 > 1) {
                cf v[R2];
#pragma unroll
                for (int q2 = 0; q2 < R2; ++q2) v[q2] = e[s][q2];
                Dft<R2>::run(v);
                lds[base + s * m1] = v[0];
#pragma unroll
                for (int s2 = 1; s2 < R2; ++s2) lds[base + s * m1 + s2 * m2] = c_mul(v[s2], rows_tw(lds, a, ts2 * j0 * s2));
            } else {
                lds[base + s * m1] = e[s][0];
            }
        }
    }
}

template <int R1, class Lds>
HPFW_DEVICE void rows_group_r2(Lds &lds, const RowsArgs &a, int len, int r2, int tid, int nthreads)
{
    // the plan fuses two passes only when their product is <= kRowsMaxPoints (register budget)
    switch (r2) {
    case 1: rows_group<R1, 1>(lds, a, len, tid, nthreads); break;
    case 2: rows_group<R1, 2>(lds, a, len, tid, nthreads); break;
    case 3: rows_group<R1, 3>(lds, a, len, tid, nthreads); break;
    case 4: rows_group<R1, 4>(lds, a, len, tid, nthreads); break;
    case 5:
        if constexpr (R1 * 5 <= kRowsMaxPoints) rows_group<R1, 5>(lds, a, len, tid, nthreads);
        break;
    default:
        if constexpr (R1 * 7 <= kRowsMaxPoints) rows_group<R1, 7>(lds, a, len, tid, nthreads);
        break;
    }
}

// The whole row transform of residues (a0, a0 + 1): pairs[t] = (x[a0 + n1 t], x[a0 + 1 + n1 t]) as
// two int16 (second is 0 when a0 + 1 == n1).  lds: n2 (+ n2/4 when quad) complex slots.
// ya / yb: rows a0 and a0 + 1 of Y' (yb may be null).
template <class Lds>
HPFW_DEVICE void rows_body(Lds &lds, const RowsArgs &a, int nthreads, const i16x2 *__restrict__ pairs, int a0,
                           cf *__restrict__ ya, cf *__restrict__ yb)
{
    const int n2 = a.n2;
    HPFW_FOR_THREADS(tid, nthreads)
    {
        if (a.quad)
            for (int i = tid; i < (n2 >> 2); i += nthreads) lds[n2 + i] = a.tw_n2[i];
        for (int t = tid; t < n2; t += nthreads) {
            const i16x2 p = pairs[t];
            lds[t] = {(float)p.x / 32768.0f, (float)p.y / 32768.0f};
        }
    }
    HPFW_BARRIER();
    int len = n2;
    for (int g = 0; g < a.groups.n; ++g) {
        const int r1 = a.groups.r1[g], r2 = a.groups.r2[g];
        HPFW_FOR_THREADS(tid, nthreads)
        {
            switch (r1) {
            case 2: rows_group_r2<2>(lds, a, len, r2, tid, nthreads); break;
            case 3: rows_group_r2<3>(lds, a, len, r2, tid, nthreads); break;
            case 4: rows_group_r2<4>(lds, a, len, r2, tid, nthreads); break;
            case 5: rows_group_r2<5>(lds, a, len, r2, tid, nthreads); break;
            default: rows_group_r2<7>(lds, a, len, r2, tid, nthreads); break;
            }
        }
        HPFW_BARRIER();
        len /= r1 * r2;
    }
    const cf *twa = a.tw_big + (int64_t)a0 * a.h;
    const cf *twb = twa + a.h;
    HPFW_FOR_THREADS(tid, nthreads)
    {
        for (int k2 = tid; k2 < a.h; k2 += nthreads) {
            const cf zk = lds[a.pos_n2[k2]];
            const cf zm = lds[a.pos_n2[k2 == 0 ? 0 : n2 - k2]];
            const cf va = {0.5f * (zk.r + zm.r), 0.5f * (zk.i - zm.i)};
            const cf vb = {0.5f * (zk.i + zm.i), 0.5f * (zm.r - zk.r)};
            ya[k2] = c_mul(va, twa[k2]);
            if (yb) yb[k2] = c_mul(vb, twb[k2]);
        }
    }
}

} // namespace hpfw
