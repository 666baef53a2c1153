// device_math.h -- complex f32 helpers, radix-2/3/4/5/7 butterflies and the in-LDS FFT passes.
//
// Every rounding is explicit (the library is built with -ffp-contract=off): the sequence of IEEE
// operations is the one fixed in DESIGN.md "Arithmetic specification" (S1, S3, S4), so results
// can be compared bit for bit with a host evaluation of the same specification.
#pragma once
#include "simt.h"

namespace hpfw {

struct cf {
    float r, i;
};

// Per-butterfly twiddle tables of the fused FFT groups (plan.cpp append_group_twiddles): entry e of butterfly b of a table
// with nb butterflies lies at [e >> 1][b][e & 1] -- two consecutive entries of a butterfly side by side, so that a lane
// fetches them with ONE 16-byte load and a wave with one contiguous KB (8-byte accesses run at 0.5-0.7 of that rate,
// and the transforms in LDS are bound by exactly these table loads).
struct alignas(16) cf2 {
    cf a, b;
};
HPFW_DEVICE cf tw_entry(const cf *__restrict__ gt, int e, int nb, int b)
{
    // (an unsigned 32-bit index: the load then takes the table's address from scalar registers and the index as its
    // 32-bit offset, instead of a 64-bit address computed with two adds per entry)
    const cf2 p = reinterpret_cast<const cf2 *>(gt)[(unsigned)((e >> 1) * nb + b)];
    return (e & 1) ? p.b : p.a;
}

// entries per butterfly of a fused (R1, R2) group: stage 1 has (R1 - 1) R2, stage 2 has R2 - 1;
// entry q2 (R1 - 1) + (s - 1) = T_n[ts1 (j0 + q2 m2) s], entry (R1 - 1) R2 + (s2 - 1) = T_n[ts2 j0 s2]
constexpr int group_twiddle_count(int r1, int r2) { return (r1 - 1) * r2 + (r2 - 1); }


// Where a fused group's per-butterfly twiddles come from.  TwAtUse: one load per entry where it is multiplied in -- what
// the compiler makes of it under the register bound of three workgroups per CU is a load and a full wait per entry, ten
// L2 round trips in a row for the (7, 3) group (tools/rows_stamps.py: that group took 39 % of a workgroup's time,
// 3.4 times the (5, 3) group).  TwPairs<R1, R2>: all entry pairs of one butterfly fetched together (16 bytes each),
// BEFORE the barrier in front of the group, so that they arrive while the workgroup waits there anyway.
struct TwAtUse {
    const cf *gt;
    int nb, b;
    HPFW_DEVICE_MEMBER cf operator()(int e) const { return tw_entry(gt, e, nb, b); }
};
template <int R1, int R2>
struct TwPairs {
    static constexpr int kPairs = (group_twiddle_count(R1, R2) + 1) / 2;
    cf2 p[kPairs];
    HPFW_DEVICE_MEMBER void fetch(const cf *__restrict__ gt, int nb, int b)
    {
#pragma unroll
        for (int k = 0; k < kPairs; ++k) p[k] = reinterpret_cast<const cf2 *>(gt)[(unsigned)(k * nb + b)];
    }
    HPFW_DEVICE_MEMBER cf operator()(int e) const { return (e & 1) ? p[e >> 1].b : p[e >> 1].a; }
};

// The complex helpers.  Each is a fixed sequence of IEEE operations on scalar f32 instructions (the host-side emulation of
// tests/emu compiles the same text).
//
// NO PACKED FP32.  Round 3 issued these operations as v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on the (Re, Im) register
// pair.  On MI355X those instructions return wrong results in lanes 48..63 of a wave -- one instruction, sixteen lanes at
// a time -- while a kernel that feeds int8 matrix instructions from LDS (hashprint_q_kernel, the LDS-staged column
// kernel) runs on the same compute units, from another stream or another process; alone on the GPU they are exact,
// which is why every single-process test passed.  tools/pk_mfma_repro.hip shows it without this library's transform
// code (packed against scalar evaluation of the same butterflies in one thread: 0 of 8.2e9 values differ alone, 8.5e5
// beside hashprint_q_kernel, all of them in lanes 48..63); DESIGN.md section 9 has the measurements.  The library is
// therefore built with the target feature packed-fp32-ops switched off (csrc/Makefile), so that the compiler cannot form
// such an instruction from this text or any other either, and the build checks the code object for them.  The scalar
// form costs nothing: 9.94 against 10.03 ms per 1000 clips (the inline-asm packed form kept the compiler from
// scheduling around it).
HPFW_DEVICE cf c_add(cf a, cf b) { return {a.r + b.r, a.i + b.i}; }
HPFW_DEVICE cf c_sub(cf a, cf b) { return {a.r - b.r, a.i - b.i}; }
// a + (-i) d and a - (-i) d
HPFW_DEVICE cf c_add_mi(cf a, cf d) { return {a.r + d.i, a.i - d.r}; }
HPFW_DEVICE cf c_sub_mi(cf a, cf d) { return {a.r - d.i, a.i + d.r}; }
// a * w
HPFW_DEVICE cf c_mul(cf a, cf w)
{
    float p = a.i * w.i;
    float q = a.i * w.r;
    return {HPFW_FMAF(a.r, w.r, -p), HPFW_FMAF(a.r, w.i, q)};
}
// a * conj(w)
HPFW_DEVICE cf c_mulc(cf a, cf w)
{
    float p = a.i * w.i;
    float q = a.r * w.i;
    return {HPFW_FMAF(a.r, w.r, p), HPFW_FMAF(a.i, w.r, -q)};
}
HPFW_DEVICE cf c_fma_s(float s, cf a, cf b)
{
    return {HPFW_FMAF(s, a.r, b.r), HPFW_FMAF(s, a.i, b.i)};
}
HPFW_DEVICE cf c_scale(float s, cf a) { return {s * a.r, s * a.i}; }

// ---- forward DFT butterflies (sign -), in place on u[0..R) -------------------------------
template <int R>
struct Dft;

template <>
struct Dft<2> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        cf a = u[0], b = u[1];
        u[0] = c_add(a, b);
        u[1] = c_sub(a, b);
    }
};

template <>
struct Dft<3> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        const float s = 0.86602540378443864676f;
        cf t1 = c_add(u[1], u[2]);
        cf d = c_sub(u[1], u[2]);
        cf m1 = c_fma_s(-0.5f, t1, u[0]);
        cf sd = c_scale(s, d); // m1 +- (-i) s d
        u[0] = c_add(u[0], t1);
        u[1] = c_add_mi(m1, sd);
        u[2] = c_sub_mi(m1, sd);
    }
};

template <>
struct Dft<4> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        cf t0 = c_add(u[0], u[2]);
        cf t1 = c_sub(u[0], u[2]);
        cf t2 = c_add(u[1], u[3]);
        cf d = c_sub(u[1], u[3]);
        u[0] = c_add(t0, t2);
        u[2] = c_sub(t0, t2);
        u[1] = c_add_mi(t1, d); // t1 +- (-i) d
        u[3] = c_sub_mi(t1, d);
    }
};

template <>
struct Dft<5> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        cf a1 = c_add(u[1], u[4]), a2 = c_add(u[2], u[3]);
        cf b1 = c_sub(u[1], u[4]), b2 = c_sub(u[2], u[3]);
        cf p1 = c_fma_s(c2, a2, c_fma_s(c1, a1, u[0]));
        cf p2 = c_fma_s(c1, a2, c_fma_s(c2, a1, u[0]));
        cf q1 = c_fma_s(s2, b2, c_scale(s1, b1));
        cf q2 = c_fma_s(-s1, b2, c_scale(s2, b1));
        u[0] = c_add(c_add(u[0], a1), a2);
        u[1] = c_add_mi(p1, q1); // p +- (-i) q
        u[4] = c_sub_mi(p1, q1);
        u[2] = c_add_mi(p2, q2);
        u[3] = c_sub_mi(p2, q2);
    }
};

template <>
struct Dft<7> {
    HPFW_DEVICE_STATIC void run(cf *u)
    {
        const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f,
                    c3 = -0.90096886790241912624f;
        const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f,
                    s3 = 0.43388373911755812048f;
        cf a1 = c_add(u[1], u[6]), a2 = c_add(u[2], u[5]), a3 = c_add(u[3], u[4]);
        cf b1 = c_sub(u[1], u[6]), b2 = c_sub(u[2], u[5]), b3 = c_sub(u[3], u[4]);
        cf p1 = c_fma_s(c3, a3, c_fma_s(c2, a2, c_fma_s(c1, a1, u[0])));
        cf p2 = c_fma_s(c1, a3, c_fma_s(c3, a2, c_fma_s(c2, a1, u[0])));
        cf p3 = c_fma_s(c2, a3, c_fma_s(c1, a2, c_fma_s(c3, a1, u[0])));
        cf q1 = c_fma_s(s3, b3, c_fma_s(s2, b2, c_scale(s1, b1)));
        cf q2 = c_fma_s(-s1, b3, c_fma_s(-s3, b2, c_scale(s2, b1)));
        cf q3 = c_fma_s(s2, b3, c_fma_s(-s1, b2, c_scale(s3, b1)));
        u[0] = c_add(c_add(c_add(u[0], a1), a2), a3);
        u[1] = c_add_mi(p1, q1); // p +- (-i) q
        u[6] = c_sub_mi(p1, q1);
        u[2] = c_add_mi(p2, q2);
        u[5] = c_sub_mi(p2, q2);
        u[3] = c_add_mi(p3, q3);
        u[4] = c_sub_mi(p3, q3);
    }
};

struct RadixList {
    int n;        // number of passes
    int r[24];
};

#if !defined(HPFW_SIMT_EMU)
// ---- one in-place pass over an array of n complex values held in LDS ---------------------
// Forward decimation in frequency: sub-transform length `len`, radix R, m = len / R.
// Butterfly (base, j): gather a[base + j + q m], DFT_R, multiply output s >= 1 by T_n[ts j s],
// ts = n / len; scatter to the same places.  `tw` is the table T_n (global memory).
template <int R>
HPFW_DEVICE void dif_pass(cf *a, int n, int len, const cf *__restrict__ tw, int tid,
                                         int nthreads)
{
    const int m = len / R;
    const int ts = n / len;
    const int nb = n / R;
    for (int b = tid; b < nb; b += nthreads) {
        const int blk = b / m;
        const int j = b - blk * m;
        cf *p = a + blk * len + j;
        cf u[R];
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = p[q * m];
        Dft<R>::run(u);
        p[0] = u[0];
        const int tj = ts * j;
#pragma unroll
        for (int s = 1; s < R; ++s) p[s * m] = c_mul(u[s], tw[tj * s]);
    }
}

// Inverse decimation in time (sign +, unnormalised): sub-transform length len = m R.
// Butterfly (base, j): gather a[base + j + q m] * conj(T_n[ts j q]) (q >= 1), inverse DFT_R
// evaluated as swap(DFT_R(swap(.))), scatter.
template <int R>
HPFW_DEVICE void idit_pass(cf *a, int n, int m, const cf *__restrict__ tw, int tid,
                                          int nthreads)
{
    const int len = m * R;
    const int ts = n / len;
    const int nb = n / R;
    for (int b = tid; b < nb; b += nthreads) {
        const int blk = b / m;
        const int j = b - blk * m;
        cf *p = a + blk * len + j;
        cf u[R];
        const int tj = ts * j;
        {
            cf v = p[0];
            u[0] = {v.i, v.r};
        }
#pragma unroll
        for (int q = 1; q < R; ++q) {
            cf v = c_mulc(p[q * m], tw[tj * q]);
            u[q] = {v.i, v.r};
        }
        Dft<R>::run(u);
#pragma unroll
        for (int s = 0; s < R; ++s) p[s * m] = {u[s].i, u[s].r};
    }
}

// Full in-LDS transforms; every thread of the block must call them (they contain barriers).
HPFW_DEVICE void lds_fft_dif(cf *a, int n, const RadixList &rl,
                                            const cf *__restrict__ tw, int tid, int nthreads)
{
    int len = n;
    for (int p = 0; p < rl.n; ++p) {
        const int r = rl.r[p];
        switch (r) {
        case 2: dif_pass<2>(a, n, len, tw, tid, nthreads); break;
        case 3: dif_pass<3>(a, n, len, tw, tid, nthreads); break;
        case 4: dif_pass<4>(a, n, len, tw, tid, nthreads); break;
        case 5: dif_pass<5>(a, n, len, tw, tid, nthreads); break;
        default: dif_pass<7>(a, n, len, tw, tid, nthreads); break;
        }
        len /= r;
        __syncthreads();
    }
}

HPFW_DEVICE void lds_fft_idit(cf *a, int n, const RadixList &rl,
                                             const cf *__restrict__ tw, int tid, int nthreads)
{
    int m = 1;
    for (int p = rl.n - 1; p >= 0; --p) {
        const int r = rl.r[p];
        switch (r) {
        case 2: idit_pass<2>(a, n, m, tw, tid, nthreads); break;
        case 3: idit_pass<3>(a, n, m, tw, tid, nthreads); break;
        case 4: idit_pass<4>(a, n, m, tw, tid, nthreads); break;
        case 5: idit_pass<5>(a, n, m, tw, tid, nthreads); break;
        default: idit_pass<7>(a, n, m, tw, tid, nthreads); break;
        }
        m *= r;
        __syncthreads();
    }
}

#endif // !HPFW_SIMT_EMU

} // namespace hpfw
