"""What does v_mfma_f32_32x32x16_bf16 compute, bit for bit?  (Preparation for a projection on the bf16 matrix pipe with
three-way split operands: the oracle must restate the instruction exactly.)  Generates operand sets, runs
tools/mfma_bf16_probe.bin on them and scores candidate models.  python3 tools/mfma_bf16_probe.py [blocks]"""
import os
import subprocess
import sys
from fractions import Fraction

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(11)


def bf16_bits(x):
    """round-to-nearest-even float32 -> bf16 bit patterns"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def bf16_val(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def gen(kind):
    if kind == "wide":        # exponents spread over 2^-12 .. 2^12
        a = rng.standard_normal((32, 16)) * np.exp2(rng.integers(-6, 7, (32, 16)))
        b = rng.standard_normal((16, 32)) * np.exp2(rng.integers(-6, 7, (16, 32)))
        c = rng.standard_normal((32, 32)) * np.exp2(rng.integers(-8, 9, (32, 32)))
    elif kind == "narrow":    # like dB values times filter coefficients
        a = rng.uniform(-0.1, 0.1, (32, 16))
        b = rng.uniform(-80, 0, (16, 32))
        c = rng.uniform(-50, 50, (32, 32))
    elif kind == "mid":       # exponents spread over 2^-20 .. 2^20 in the products, accumulator anywhere
        a = rng.standard_normal((32, 16)) * np.exp2(rng.integers(-10, 11, (32, 16)))
        b = rng.standard_normal((16, 32)) * np.exp2(rng.integers(-10, 11, (16, 32)))
        c = rng.standard_normal((32, 32)) * np.exp2(rng.integers(-30, 31, (32, 32)))
    else:                     # "sticky": one big term and many tiny ones around the rounding threshold
        a = np.full((32, 16), 1.0) * np.exp2(-rng.integers(10, 16, (32, 16)).astype(np.float64))
        b = np.full((16, 32), 1.0) * np.exp2(-rng.integers(10, 16, (16, 32)).astype(np.float64)) * rng.choice([1.0, -1.0, 1.5, 1.25], (16, 32))
        c = rng.choice([1.0, -1.0, 1.0 + 2.0 ** -23, 3.0], (32, 32))
    return bf16_bits(a), bf16_bits(b), c.astype(np.float32)


kinds = ["wide", "narrow", "sticky", "mid"]
A = np.zeros((nb, 32, 16), np.uint16); B = np.zeros((nb, 16, 32), np.uint16); C = np.zeros((nb, 32, 32), np.float32)
for i in range(nb):
    A[i], B[i], C[i] = gen(kinds[i % 4])
inp, out = "/tmp/mfma_probe_in.bin", "/tmp/mfma_probe_out.bin"
with open(inp, "wb") as f:
    f.write(A.tobytes()); f.write(B.tobytes()); f.write(C.tobytes())
exe = os.path.join(here, "mfma_bf16_probe.bin")
if not os.path.exists(exe):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value",
                           os.path.join(here, "mfma_bf16_probe.hip"), "-o", exe])
subprocess.check_call([exe, inp, out, str(nb)])
D = np.fromfile(out, np.float32).reshape(nb, 32, 32)
if os.environ.get("MFMA_PROBE_DUMP"):
    np.savez_compressed(os.environ["MFMA_PROBE_DUMP"], A=A, B=B, C=C, D=D)


def f32_rne(fr):
    """Fraction -> nearest float32 (ties to even), via exact arithmetic"""
    if fr == 0:
        return np.float32(0.0)
    x = np.float32(float(fr))                      # float() rounds correctly to double; double -> f32 may double-round:
    lo, hi = np.nextafter(x, np.float32(-np.inf)), np.nextafter(x, np.float32(np.inf))
    best = min((lo, x, hi), key=lambda v: (abs(Fraction(float(v)) - fr), int(np.float32(v).view(np.uint32)) & 1))
    return np.float32(best)


def models(a, b, c):
    p = [Fraction(float(x)) * Fraction(float(y)) for x, y in zip(a, b)]
    out = {}
    out["exact, one rounding"] = f32_rne(sum(p) + Fraction(float(c)))
    acc = np.float32(c)
    for t in p:
        acc = f32_rne(Fraction(float(acc)) + t)
    out["fma chain k ascending"] = acc
    for blk in (2, 4, 8):
        acc = np.float32(c)
        for s in range(0, 16, blk):
            acc = f32_rne(Fraction(float(acc)) + sum(p[s:s + blk]))
        out[f"blocks of {blk} exact, rounded between"] = acc
    acc = np.float32(c)                            # blocks of 4 interleaved over the two lane halves (k, k + 8)
    for s in range(0, 8, 4):
        acc = f32_rne(Fraction(float(acc)) + sum(p[s:s + 4]) + sum(p[8 + s:12 + s]))
    out["blocks (k..k+3, k+8..k+11) exact, rounded between"] = acc
    prods = f32_rne(sum(p))
    out["products summed exactly and rounded, then + c rounded"] = f32_rne(Fraction(float(prods)) + Fraction(float(c)))
    return out


score, total = {}, {k: 0 for k in kinds}
per_kind = {}
sample = rng.integers(0, 32, (nb, 24, 2))
for i in range(nb):
    av, bv = bf16_val(A[i]), bf16_val(B[i])
    for r, cidx in sample[i]:
        m = models(av[r], bv[:, cidx], C[i, r, cidx])
        got = D[i, r, cidx]
        total[kinds[i % 4]] += 1
        for name, v in m.items():
            key = (kinds[i % 4], name)
            score[key] = score.get(key, 0) + int(np.float32(v).view(np.uint32) == np.float32(got).view(np.uint32))
for (kind, name), s in sorted(score.items()):
    print(f"{kind:7s} {name:55s} {s:6d} / {total[kind]}")
