// fft_rows.h -- the length-n2 mixed-radix FFT of one pair of residue sequences, held in LDS
// (DESIGN.md S4, S6).  Same arithmetic as the specification's radix-7/5/4/3/2 passes; two
// consecutive passes are fused in registers (up to 36 points per thread and LDS round trip).
// The twiddles of a fused group are read from a per-butterfly table in global memory (L2 resident,
// built by the host from T_n2: entries e, e + 1 of butterfly b side by side at [e >> 1][b], so a wave reads one
// contiguous KB per pair of entries, device_math.h tw_entry): no index arithmetic and no LDS traffic for twiddles.  n2 <= 6826 keeps the data
// under 55 KB of LDS, so several workgroups share a CU and one's loads hide behind another's passes.
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
    int tw_off[kRowsMaxGroups]; // offset of the group's twiddle table in gtw (complex elements)
};

// diagnosis builds (-DHPFW_ROWS_SNAP, tools/rows_snapshots.py): the LDS image of every workgroup after the load and after
// each fused group goes to RowsArgs::snap [workgroup][4][n2]
#if defined(HPFW_ROWS_SNAP) && !defined(HPFW_SIMT_EMU)
#define HPFW_SNAP(lds, a, slot, n2v)                                                                                   \
    do {                                                                                                               \
        if ((a).snap) {                                                                                                \
            cf *dst_ = (a).snap + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)) * (int64_t)(n2v);       \
            for (int i_ = threadIdx.x; i_ < (n2v); i_ += blockDim.x) dst_[i_] = (lds)[i_];                              \
            __syncthreads(); /* the next group writes in place */                                                      \
        }                                                                                                              \
    } while (0)
#else
#define HPFW_SNAP(lds, a, slot, n2v) ((void)0)
#endif

// diagnosis builds (-DHPFW_ROWS_STAMPS, tools/rows_stamps.py): s_memtime ticks wave 0 of every workgroup spends up to
// each barrier of the row transform, RowsArgs::stamps [workgroup][8]
#if defined(HPFW_ROWS_STAMPS) && !defined(HPFW_SIMT_EMU)
#define HPFW_STAMP(a, k)                                                                                               \
    do {                                                                                                               \
        if ((a).stamps && threadIdx.x == 0)                                                                            \
            (a).stamps[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
#else
#define HPFW_STAMP(a, k) ((void)0)
#endif

struct RowsArgs {
#if defined(HPFW_ROWS_SNAP)
    cf *snap;
#endif
#if defined(HPFW_ROWS_STAMPS)
    long long *stamps;
#endif
    int n1, n2, h;            // N = n1 * n2, h = n2 / 2 + 1
    int hpad;                 // row stride (floats) of the planar output, a multiple of 32
    int pair_stride;          // (the Mel front-end's frame pairs: 1)
    RowGroups groups;
    const cf *gtw;            // per-butterfly twiddles of every group (see group_twiddle_count)
    const cf *tw_big;         // the table the split spectra are multiplied by (Mel front-end: rows of ones)
    const int *pos_n2;        // digit-reversed position of output k2
    const int *kb_last;       // output k2 = kb_last[b] + (n2 / len) f of the last group's block b (see rows_last_*)
};

// one fused group of the forward DIF at sub-length len: radix R1, then radix R2 (or 1), in place
// N2C, LENC: the transform length and the sub-length when the group sequence is fixed at compile time (0 = taken
// from the arguments at run time): every LDS address is then the thread's base plus an immediate offset, no
// address or 1 / m2 is computed or kept in registers, and the loop over butterflies disappears when the workgroup
// holds at least one thread per butterfly.
// the butterfly of one fused group on registers: in place when `out` is null, else to out[s R2 + s2] (the two-phase groups)
template <int R1, int R2, class Lds, class Tw>
HPFW_DEVICE void rows_butterfly(Lds &lds, int base, int m1, int m2, const Tw &tw, cf *out)
{
    cf e[R1][R2];
#pragma unroll
    for (int q2 = 0; q2 < R2; ++q2) {
        cf u[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) u[q] = lds[base + q2 * m2 + q * m1];
        Dft<R1>::run(u);
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
            if (out) {
                out[s * R2] = v[0];
#pragma unroll
                for (int s2 = 1; s2 < R2; ++s2) out[s * R2 + s2] = c_mul(v[s2], tw((R1 - 1) * R2 + (s2 - 1)));
            } else {
                lds[base + s * m1] = v[0];
#pragma unroll
                for (int s2 = 1; s2 < R2; ++s2) lds[base + s * m1 + s2 * m2] = c_mul(v[s2], tw((R1 - 1) * R2 + (s2 - 1)));
            }
        } else {
            if (out)
                out[s] = e[s][0];
            else
                lds[base + s * m1] = e[s][0];
        }
    }
}

template <int R1, int R2, int N2C = 0, int LENC = 0, class Lds>
HPFW_DEVICE void rows_group(Lds &lds, const RowsArgs &a, const cf *__restrict__ gt, int len_rt, int tid, int nthreads)
{
    const int n = N2C ? N2C : a.n2;
    const int len = LENC ? LENC : len_rt;
    const int m1 = len / R1, m2 = m1 / R2;
    const int nb = n / (R1 * R2);
    const float inv_m2 = 1.0f / (float)m2;
    for (int b = tid; b < nb; b += nthreads) {
        int blk;
        if constexpr (N2C != 0) {
            blk = b / m2; // a constant divisor: multiply and shift
        } else {
            blk = (int)((float)b * inv_m2);
            if (blk * m2 > b) --blk;
            if ((blk + 1) * m2 <= b) ++blk;
        }
        const int j0 = b - blk * m2;
        // the entries depend on j0 only: the table holds the m2 butterflies of one block
        rows_butterfly<R1, R2>(lds, blk * len + j0, m1, m2, TwAtUse{gt, m2, j0}, static_cast<cf *>(nullptr));
    }
}

// butterfly `tid` of a compile-time group (one per thread), its twiddles fetched earlier; in place, or into out
template <int R1, int R2, int N2, int LEN, class Lds>
HPFW_DEVICE void rows_group_pre(Lds &lds, int tid, const TwPairs<R1, R2> &tw, cf *out)
{
    constexpr int m1 = LEN / R1, m2 = m1 / R2, nb = N2 / (R1 * R2);
    if (tid >= nb) return;
    const int blk = tid / m2, j0 = tid - blk * m2;
    rows_butterfly<R1, R2>(lds, blk * LEN + j0, m1, m2, tw, out);
}
template <int R1, int R2, int N2, int LEN>
HPFW_DEVICE void rows_group_fetch(const cf *__restrict__ gt, int tid, TwPairs<R1, R2> &tw)
{
    constexpr int m2 = LEN / (R1 * R2), nb = N2 / (R1 * R2);
    if (tid < nb) tw.fetch(gt, m2, tid % m2);
}

// The group before the last, one butterfly per thread, in two halves around a barrier (its stores
// are not in place): the same butterfly as rows_group into registers, out[s R2 + s2]; then element
// j0 of the next group's block b' = blk R1 R2 + s R2 + s2 (in place: position b' m2 + j0) goes to
// j0 (n / m2) + b'.  The last group, whose thread b owns block b, then reads with its lanes on
// consecutive addresses instead of m2 complex apart (8-way bank conflicts for m2 = 20).
template <int R1, int R2, int N2C = 0, int LENC = 0, class Lds>
HPFW_DEVICE void rows_pre_compute(Lds &lds, const RowsArgs &a, const cf *__restrict__ gt, int len_rt, int tid, cf *out)
{
    const int n = N2C ? N2C : a.n2;
    const int len = LENC ? LENC : len_rt;
    const int m1 = len / R1, m2 = m1 / R2;
    const int nb = n / (R1 * R2);
    if (tid >= nb) return;
    const int blk = tid / m2, j0 = tid - blk * m2;
    rows_butterfly<R1, R2>(lds, blk * len + j0, m1, m2, TwAtUse{gt, m2, j0}, out);
}

template <int R1, int R2, int N2C = 0, int LENC = 0, class Lds>
HPFW_DEVICE void rows_pre_store(Lds &lds, const RowsArgs &a, int len_rt, int tid, const cf *out)
{
    const int n = N2C ? N2C : a.n2;
    const int len = LENC ? LENC : len_rt;
    const int m2 = len / (R1 * R2);
    const int nb = n / (R1 * R2);
    if (tid >= nb) return;
    const int blk = tid / m2, j0 = tid - blk * m2;
    const int o0 = j0 * (n / m2) + blk * (R1 * R2);
#pragma unroll
    for (int t = 0; t < R1 * R2; ++t) lds[o0 + t] = out[t];
}

// The last group (len = R1 R2: thread b owns block b) when the data are transposed, in two halves
// around a barrier: compute from element e of block b at e nb + b into registers; then store output
// f = s + R1 s2 of the block -- frequency k2 = kb_last[b] + nb f -- at lds[k2]: natural order, so the
// Hermitian split reads lds[k2] and lds[n2 - k2] on consecutive lanes and needs no position table.
template <int R1, int R2, int N2C = 0, class Lds>
HPFW_DEVICE void rows_last_compute(Lds &lds, const RowsArgs &a, const cf *__restrict__ gt, int tid, cf *out)
{
    const int nb = (N2C ? N2C : a.n2) / (R1 * R2);
    if (tid >= nb) return;
    // the last group's sub-length is R1 R2: j0 = 0 for every butterfly, so its twiddles are the same for all of them --
    // butterfly 0's entries, read through a wave-uniform address (scalar loads) instead of one vector load per entry
    const int TWB = 0, TWN = 1;
    cf e[R1][R2];
#pragma unroll
    for (int q2 = 0; q2 < R2; ++q2) {
        cf u[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) u[q] = lds[(q2 + q * R2) * nb + tid];
        Dft<R1>::run(u);
        e[0][q2] = u[0];
#pragma unroll
        for (int s = 1; s < R1; ++s) e[s][q2] = c_mul(u[s], tw_entry(gt, q2 * (R1 - 1) + (s - 1), TWN, TWB));
    }
#pragma unroll
    for (int s = 0; s < R1; ++s) {
        if constexpr (R2 > 1) {
            cf v[R2];
#pragma unroll
            for (int q2 = 0; q2 < R2; ++q2) v[q2] = e[s][q2];
            Dft<R2>::run(v);
            out[s] = v[0];
#pragma unroll
            for (int s2 = 1; s2 < R2; ++s2) out[s + R1 * s2] = c_mul(v[s2], tw_entry(gt, (R1 - 1) * R2 + (s2 - 1), TWN, TWB));
        } else {
            out[s] = e[s][0];
        }
    }
}

template <int R1, int R2, int N2C = 0, class Lds>
HPFW_DEVICE void rows_last_store(Lds &lds, const RowsArgs &a, int tid, const cf *out)
{
    const int nb = (N2C ? N2C : a.n2) / (R1 * R2);
    if (tid >= nb) return;
    const int k0 = a.kb_last[tid];
#pragma unroll
    for (int f = 0; f < R1 * R2; ++f) lds[k0 + f * nb] = out[f];
}

template <int R1, class Lds>
HPFW_DEVICE void rows_group_r2(Lds &lds, const RowsArgs &a, const cf *gt, int len, int r2, int tid, int nthreads)
{
    // the plan fuses two passes only when their product is <= kRowsMaxPoints (register budget)
    switch (r2) {
    case 1: rows_group<R1, 1>(lds, a, gt, len, tid, nthreads); break;
    case 2: rows_group<R1, 2>(lds, a, gt, len, tid, nthreads); break;
    case 3: rows_group<R1, 3>(lds, a, gt, len, tid, nthreads); break;
    case 4: rows_group<R1, 4>(lds, a, gt, len, tid, nthreads); break;
    case 5:
        if constexpr (R1 * 5 <= kRowsMaxPoints) rows_group<R1, 5>(lds, a, gt, len, tid, nthreads);
        break;
    default:
        if constexpr (R1 * 7 <= kRowsMaxPoints) rows_group<R1, 7>(lds, a, gt, len, tid, nthreads);
        break;
    }
}

// The group sequence either comes from the plan at run time (any 7-smooth n2) ...
struct RuntimeGroups {
    static constexpr bool kNatural = false; // outputs stay at their digit-reversed positions
    static constexpr int kProduct = 0;      // the transform length is a run-time value
    template <class Lds>
    HPFW_DEVICE_STATIC void run(Lds &lds, const RowsArgs &a, int nthreads)
    {
        int len = a.n2;
        for (int g = 0; g < a.groups.n; ++g) {
            const int r1 = a.groups.r1[g], r2 = a.groups.r2[g];
            const cf *gt = a.gtw + a.groups.tw_off[g];
            HPFW_FOR_THREADS(tid, nthreads)
            {
                switch (r1) {
                case 2: rows_group_r2<2>(lds, a, gt, len, r2, tid, nthreads); break;
                case 3: rows_group_r2<3>(lds, a, gt, len, r2, tid, nthreads); break;
                case 4: rows_group_r2<4>(lds, a, gt, len, r2, tid, nthreads); break;
                case 5: rows_group_r2<5>(lds, a, gt, len, r2, tid, nthreads); break;
                default: rows_group_r2<7>(lds, a, gt, len, r2, tid, nthreads); break;
                }
            }
            HPFW_BARRIER();
            len /= r1 * r2;
        }
    }
};

// ... or is fixed at compile time: the kernel then contains exactly these butterflies and its
// register allocation is theirs (the run-time dispatcher is sized by its largest case).
template <int... RS>
struct StaticGroups;

template <>
struct StaticGroups<> {
    static constexpr int kProduct = 1;
    static bool matches(const RowGroups &, int g, int n) { return g == n; }
};

// Needs at least two groups and one thread per butterfly of the last two (kMinThreads); those two
// use the transposed layout and leave the outputs in natural order.
template <int R1, int R2, int... Rest>
struct StaticGroups<R1, R2, Rest...> {
    static constexpr bool kNatural = true;
    static constexpr int kProduct = R1 * R2 * StaticGroups<Rest...>::kProduct;
    // butterflies of the two-phase groups: n2 / (R1 R2) with n2 the product of ALL radices, so only
    // the complete list knows it; StaticGroups<...>::min_threads(n2) is used by the launcher
    static int min_threads(int n2)
    {
        if constexpr (sizeof...(Rest) == 0 || sizeof...(Rest) == 2) {
            const int here = n2 / (R1 * R2);
            if constexpr (sizeof...(Rest) == 2) {
                const int rest = StaticGroups<Rest...>::min_threads(n2);
                return here > rest ? here : rest;
            } else {
                return here;
            }
        } else {
            return StaticGroups<Rest...>::min_threads(n2);
        }
    }
    // N2 = the product of the whole list, LEN = the sub-length this group starts from: both known at compile time
    template <int N2, int LEN, class Lds>
    HPFW_DEVICE_STATIC void run_from(Lds &lds, const RowsArgs &a, int nthreads, int g)
    {
        const cf *gt = a.gtw + a.groups.tw_off[g];
        if constexpr (sizeof...(Rest) == 0) {
            HPFW_CARRY(cf, outv, R1 * R2, nthreads);
            HPFW_FOR_THREADS(tid, nthreads) { rows_last_compute<R1, R2, N2>(lds, a, gt, tid, HPFW_CARRY_AT(outv, R1 * R2, tid)); }
            HPFW_BARRIER();
            HPFW_STAMP(a, 5);
            HPFW_FOR_THREADS(tid, nthreads) { rows_last_store<R1, R2, N2>(lds, a, tid, HPFW_CARRY_AT(outv, R1 * R2, tid)); }
            HPFW_BARRIER();
            HPFW_STAMP(a, 6);
            HPFW_SNAP(lds, a, g + 1, N2);
        } else if constexpr (sizeof...(Rest) == 2) {
            HPFW_CARRY(cf, outv, R1 * R2, nthreads);
            HPFW_FOR_THREADS(tid, nthreads) { rows_pre_compute<R1, R2, N2, LEN>(lds, a, gt, LEN, tid, HPFW_CARRY_AT(outv, R1 * R2, tid)); }
            HPFW_BARRIER();
            HPFW_STAMP(a, 3);
            HPFW_FOR_THREADS(tid, nthreads) { rows_pre_store<R1, R2, N2, LEN>(lds, a, LEN, tid, HPFW_CARRY_AT(outv, R1 * R2, tid)); }
            HPFW_BARRIER();
            HPFW_STAMP(a, 4);
            HPFW_SNAP(lds, a, g + 1, N2);
            StaticGroups<Rest...>::template run_from<N2, LEN / (R1 * R2)>(lds, a, nthreads, g + 1);
        } else {
            HPFW_FOR_THREADS(tid, nthreads) { rows_group<R1, R2, N2, LEN>(lds, a, gt, LEN, tid, nthreads); }
            HPFW_BARRIER();
            HPFW_STAMP(a, 2);
            HPFW_SNAP(lds, a, g + 1, N2);
            StaticGroups<Rest...>::template run_from<N2, LEN / (R1 * R2)>(lds, a, nthreads, g + 1);
        }
    }
#if !defined(HPFW_SIMT_EMU)
    // The same sequence with every group's twiddles fetched before the barrier in front of it (one butterfly per thread
    // throughout: the launcher checked min_threads).  `mine`: this group's, fetched by the caller.
    using Pairs = TwPairs<R1, R2>;
    template <int N2, int LEN>
    HPFW_DEVICE_STATIC void fetch(const RowsArgs &a, int g, Pairs &mine)
    {
        if constexpr (sizeof...(Rest) != 0) rows_group_fetch<R1, R2, N2, LEN>(a.gtw + a.groups.tw_off[g], threadIdx.x, mine);
    }
    template <int N2, int LEN, class Lds>
    HPFW_DEVICE_STATIC void run_fetched(Lds &lds, const RowsArgs &a, int g, const Pairs &mine)
    {
        const int tid = threadIdx.x;
        if constexpr (sizeof...(Rest) == 0) {
            run_from<N2, LEN>(lds, a, 0, g);          // (the last group's twiddles are the same for every butterfly: scalar loads)
        } else if constexpr (sizeof...(Rest) == 2) {
            cf outv[R1 * R2];
            rows_group_pre<R1, R2, N2, LEN>(lds, tid, mine, outv);
            HPFW_BARRIER();
            HPFW_STAMP(a, 3);
            rows_pre_store<R1, R2, N2, LEN>(lds, a, LEN, tid, outv);
            HPFW_BARRIER();
            HPFW_STAMP(a, 4);
            HPFW_SNAP(lds, a, g + 1, N2);
            StaticGroups<Rest...>::template run_from<N2, LEN / (R1 * R2)>(lds, a, 0, g + 1);
        } else {
            using Next = StaticGroups<Rest...>;
            rows_group_pre<R1, R2, N2, LEN>(lds, tid, mine, static_cast<cf *>(nullptr));
            typename Next::Pairs next;
            Next::template fetch<N2, LEN / (R1 * R2)>(a, g + 1, next);
            HPFW_BARRIER();
            HPFW_STAMP(a, 2);
            HPFW_SNAP(lds, a, g + 1, N2);
            Next::template run_fetched<N2, LEN / (R1 * R2)>(lds, a, g + 1, next);
        }
    }
#endif
    template <class Lds>
    HPFW_DEVICE_STATIC void run(Lds &lds, const RowsArgs &a, int nthreads)
    {
        static_assert(sizeof...(Rest) >= 2, "the transposed layout needs a group before the last");
        run_from<kProduct, kProduct>(lds, a, nthreads, 0); // the launcher checked a.n2 == kProduct (matches())
    }
    static bool matches(const RowGroups &g, int i, int n)
    {
        return i < n && g.r1[i] == R1 && g.r2[i] == R2 && StaticGroups<Rest...>::matches(g, i + 1, n);
    }
    // the whole list against a plan: same groups, hence the same product
    static bool matches_plan(const RowsArgs &a) { return matches(a.groups, 0, a.groups.n) && a.n2 == kProduct; }
};

// n2 = 6300 = 7 3 5 3 5 4: every clip length that is a multiple of 1/7 s at 44.1 kHz up to 50 s
using Groups6300 = StaticGroups<7, 3, 5, 3, 5, 4>;

// The whole row transform of residues (a0, a0 + 1): pairs[t] = (x[a0 + n1 t], x[a0 + 1 + n1 t]) as
// two int16 (second is 0 when a0 + 1 == n1).  lds: n2 complex slots.
// Output: rows 2 a0 .. 2 a0 + 3 of the planar matrix Y' [2 n1][hpad] (row 2a = Re, 2a + 1 = Im of
// residue a) -- the B operand of the column-DFT MFMA kernel; ya / yb point at the Re rows of a0 and
// a0 + 1 (yb may be null).
// load(t) -> the two real samples of time step t as a cf (real part: first sequence): PairLoad for the
// forward transform's residue pairs; the STFT of the Mel front-end passes windowed frames (k_mel.hip).
struct PairLoad {
    const i16x2 *pairs;
    int stride;
    HPFW_DEVICE_MEMBER i16x2 raw(int t) const { return pairs[(int64_t)t * stride]; }
    HPFW_DEVICE_STATIC cf conv(i16x2 p, int) { return {(float)p.x / 32768.0f, (float)p.y / 32768.0f}; }
};

template <class Groups, class Lds, class Load>
HPFW_DEVICE void rows_body_from(Lds &lds, const RowsArgs &a, int nthreads, const Load &load, int a0,
                                float *__restrict__ ya, float *__restrict__ yb)
{
    // a compile-time group sequence knows its length: loop bounds and the half-spectrum size become constants
    const int n2 = Groups::kProduct ? Groups::kProduct : a.n2;
    const int h = Groups::kProduct ? Groups::kProduct / 2 + 1 : a.h;
    const int hpad = Groups::kProduct ? (Groups::kProduct / 2 + 1 + 31) / 32 * 32 : a.hpad;
    HPFW_FOR_THREADS(tid, nthreads)
    {
        // loads in batches of kLd so that their latencies overlap
        constexpr int kLd = 13;
        for (int t0 = tid; t0 < n2; t0 += kLd * nthreads) {
            decltype(load.raw(0)) p[kLd];
#pragma unroll
            for (int e = 0; e < kLd; ++e) {
                const int t = t0 + e * nthreads;
                p[e] = load.raw(t < n2 ? t : 0);
            }
#pragma unroll
            for (int e = 0; e < kLd; ++e) {
                const int t = t0 + e * nthreads;
                if (t < n2) lds[t] = load.conv(p[e], t);
            }
        }
    }
    HPFW_BARRIER();
    Groups::run(lds, a, nthreads);
    // Hermitian split + twiddle.  The table reads (digit-reversal positions, T_N rows) of kEpi outputs
    // are issued together before any store, so their latencies overlap instead of adding up.
    const cf *__restrict__ twa = a.tw_big + (int64_t)a0 * h;
    const cf *__restrict__ twb = yb ? twa + h : twa;
    const int *__restrict__ pos = a.pos_n2;
    constexpr bool kNat = Groups::kNatural;
    constexpr int kEpi = 9;
    HPFW_FOR_THREADS(tid, nthreads)
    {
        for (int k0 = tid; k0 < h; k0 += kEpi * nthreads) {
            int pk[kEpi], pm[kEpi];
            cf wa[kEpi], wb[kEpi];
#pragma unroll
            for (int e = 0; e < kEpi; ++e) {
                const int k2 = k0 + e * nthreads;
                const int kk = k2 < h ? k2 : 0;
                pk[e] = kNat ? kk : pos[kk];
                pm[e] = kNat ? (kk == 0 ? 0 : n2 - kk) : pos[kk == 0 ? 0 : n2 - kk];
                wa[e] = twa[kk];
                wb[e] = twb[kk];
            }
#pragma unroll
            for (int e = 0; e < kEpi; ++e) {
                const int k2 = k0 + e * nthreads;
                if (k2 < h) {
                    const cf zk = lds[pk[e]];
                    const cf zm = lds[pm[e]];
                    const cf va = {0.5f * (zk.r + zm.r), 0.5f * (zk.i - zm.i)};
                    const cf vb = {0.5f * (zk.i + zm.i), 0.5f * (zm.r - zk.r)};
                    const cf oa = c_mul(va, wa[e]);
                    ya[k2] = oa.r;
                    ya[hpad + k2] = oa.i;
                    if (yb) {
                        const cf ob = c_mul(vb, wb[e]);
                        yb[k2] = ob.r;
                        yb[hpad + k2] = ob.i;
                    }
                }
            }
        }
    }
}

template <class Groups, class Lds>
HPFW_DEVICE void rows_body(Lds &lds, const RowsArgs &a, int nthreads, const i16x2 *__restrict__ pairs, int a0,
                           float *__restrict__ ya, float *__restrict__ yb)
{
    rows_body_from<Groups>(lds, a, nthreads, PairLoad{pairs, a.pair_stride}, a0, ya, yb);
}

// ---- S6, the row stage of the forward transform (7-smooth lengths) ------------------------------------------------
// The column stage (k_forward.hip, fwd_cols_q_kernel) leaves z[q1][k2] = G[q1][k2] T_N[q1 k2], q1 <= n1 / 2.  Row q1:
// Z = FFT_n2(z[q1]) (S4), X[q1 + n1 q2] = Z[q2], and the bins of row n1 - q1 by X[k] = conj(X[N - k]):
// X[(n1 - q1) + n1 q2] = conj(Z[n2 - 1 - q2]).  Only q2lo <= q2 < q2lo + q2w holds consumed bins; they are stored as
// x[row][q2 - q2lo], rows of q2w elements (the constant-Q stage gathers its slices from that layout: kernels.h XsView).
struct Rows2Out {
    int n1, hq;      // rows of the column stage: hq = n1 / 2 + 1 computed, n1 in the output
    int q2lo, q2w;
    // the twiddles between the stages, T_N[q1 k2] 2^-37, from every fourth one: [hq][nq] seeds T_N[4 q1 m] 2^-37 and
    // [hq][4] steps T_N[q1 e]; element 4 m + e is seed (e = 0) or seed * step[e] (S1)
    const cf *seed, *step;
    int nq;
    // where z lies (kernels.h ColsQArgs::zclip): blocks of 2^zshift4 four-column pieces, zpitch floats from block to block,
    // zrow floats from a row's Re to its Im inside a block; the emulation passes one contiguous row (zshift4 = 30)
    int zshift4;
    long long zpitch;
    int zrow;
    long long zclip;
};

struct alignas(16) f4 {
    float x, y, z, w;
};

// zre: the row's Re values in the blocked layout of Rows2Out (its Im values o.zrow floats on), the column stage's integers rounded to f32
template <class Groups, class Lds>
HPFW_DEVICE void rows2_body(Lds &lds, const RowsArgs &a, int nthreads, const float *__restrict__ zre, int q1, const Rows2Out &o,
                            cf *__restrict__ xclip)
{
    const int n2 = Groups::kProduct ? Groups::kProduct : a.n2;
    HPFW_STAMP(a, 0);
    HPFW_FOR_THREADS(tid, nthreads)
    {
        constexpr int kLd = 4; // loads in batches so that their latencies overlap
        if ((n2 & 3) == 0) {   // both planes start 16-byte aligned: four elements per pair of loads
            const float *__restrict__ zim = zre + o.zrow;
            const int bmask = (1 << o.zshift4) - 1;
            // piece t (columns 4 t .. 4 t + 3) of the row: block t >> zshift4, piece t & bmask inside it
            auto piece = [&](const float *row, int t) { return *reinterpret_cast<const f4 *>(row + (int64_t)(t >> o.zshift4) * o.zpitch + 4 * (t & bmask)); };
            const cf *__restrict__ seeds = o.seed + (int64_t)q1 * o.nq;
            const cf s1 = o.step[q1 * 4 + 1], s2 = o.step[q1 * 4 + 2], s3 = o.step[q1 * 4 + 3];
            const int nq = n2 / 4;
            for (int t0 = tid; t0 < nq; t0 += kLd * nthreads) {
                f4 vr[kLd], vi[kLd];
                cf w0[kLd];
#pragma unroll
                for (int e = 0; e < kLd; ++e) {
                    const int t = t0 + e * nthreads;
                    vr[e] = piece(zre, t < nq ? t : 0);
                    vi[e] = piece(zim, t < nq ? t : 0);
                    w0[e] = seeds[t < nq ? t : 0];
                }
#pragma unroll
                for (int e = 0; e < kLd; ++e) {
                    const int t = t0 + e * nthreads;
                    if (t < nq) {
                        const cf o0 = c_mul(cf{vr[e].x, vi[e].x}, w0[e]), o1 = c_mul(cf{vr[e].y, vi[e].y}, c_mul(w0[e], s1));
                        const cf o2 = c_mul(cf{vr[e].z, vi[e].z}, c_mul(w0[e], s2)), o3 = c_mul(cf{vr[e].w, vi[e].w}, c_mul(w0[e], s3));
#if defined(HPFW_SIMT_EMU)
                        lds[4 * t] = o0;
                        lds[4 * t + 1] = o1;
                        lds[4 * t + 2] = o2;
                        lds[4 * t + 3] = o3;
#else
                        // two 16-byte stores per lane, 32 bytes from lane to lane: a store instruction is served in groups of
                        // eight lanes over 32 banks, and lanes l and l + 4 of a group would meet on the same banks -- so the
                        // second four of every eight write their halves in the other order
                        const bool sw = (tid >> 2) & 1;
                        cf *at = &lds[4 * t];
                        *reinterpret_cast<cf2 *>(at + (sw ? 2 : 0)) = sw ? cf2{o2, o3} : cf2{o0, o1};
                        *reinterpret_cast<cf2 *>(at + (sw ? 0 : 2)) = sw ? cf2{o0, o1} : cf2{o2, o3};
#endif
                    }
                }
            }
        } else {
            for (int t = tid; t < n2; t += nthreads) {
                const cf w0 = o.seed[(int64_t)q1 * o.nq + (t >> 2)];
                const int64_t at = (int64_t)(t >> (o.zshift4 + 2)) * o.zpitch + (t & ((4 << o.zshift4) - 1));
                lds[t] = c_mul(cf{zre[at], zre[o.zrow + at]}, (t & 3) ? c_mul(w0, o.step[q1 * 4 + (t & 3)]) : w0);
            }
        }
    }
#if !defined(HPFW_SIMT_EMU)
    if constexpr (Groups::kProduct != 0) {
        // the compile-time sequence: the first group's twiddles are on their way while the workgroup meets at the barrier
        typename Groups::Pairs tw0;
        Groups::template fetch<Groups::kProduct, Groups::kProduct>(a, 0, tw0);
        HPFW_BARRIER();
        HPFW_STAMP(a, 1);
        HPFW_SNAP(lds, a, 0, n2);
        Groups::template run_fetched<Groups::kProduct, Groups::kProduct>(lds, a, 0, tw0);
    } else
#endif
    {
        HPFW_BARRIER();
        HPFW_STAMP(a, 1);
        HPFW_SNAP(lds, a, 0, n2);
        Groups::run(lds, a, nthreads);
    }
    const int *__restrict__ pos = a.pos_n2;
    constexpr bool kNat = Groups::kNatural;
    const bool mirror_row = q1 >= 1 && o.n1 - q1 >= o.hq; // row n1 - q1 is not computed itself
    HPFW_FOR_THREADS(tid, nthreads)
    {
        for (int i = tid; i < 2 * o.q2w; i += nthreads) {
            const bool mir = i >= o.q2w;
            if (mir && !mirror_row) break;
            const int j = mir ? i - o.q2w : i;
            const int q2 = o.q2lo + j;
            const int src = mir ? n2 - 1 - q2 : q2;
            cf v = lds[kNat ? src : pos[src]];
            if (mir) v.i = -v.i;
            xclip[(int64_t)(mir ? o.n1 - q1 : q1) * o.q2w + j] = v;
        }
    }
    HPFW_STAMP(a, 7);
}

} // namespace hpfw
