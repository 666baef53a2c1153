"""The projection in fixed point (k_project_q.hip, exact integer sums on the int8 matrix pipe) against
the oracle's integer restatement, and beside the f32 fma chain (S9) it replaces as the default."""
import numpy as np
import pytest

import hpfw_amd
from hpfw_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def test_fixed_point_projection_is_exact(oracle):
    """hashprints in both projection modes against the oracle in the same mode, ragged clip lengths (tiles of 256
    hashprints with tails, widths below one tile), from PCM and from given dB spectrograms; the int64 sums themselves"""
    import torch
    filt = synth.make_filters()
    g = hpfw_amd.Gpu(0)
    g.set_filters(filt)
    assert g.get_projection() in (0, 1)
    before = oracle.get_projection()
    try:
        for sec, nclips in ((30.0, 3), (5.0, 4), (2.3, 2), (7.77, 1)):
            clips = np.stack([synth.gen_clip(900 + i, sec) for i in range(nclips)])
            plan = oracle.Plan(clips.shape[1])
            db = np.stack([oracle.db(plan.cqmag(plan.spectrum(c))) for c in clips])
            d_db = torch.from_numpy(db).cuda()
            for mode in (1, 0):
                g.set_projection(mode)
                oracle.set_projection(mode)
                want = np.stack([plan.extract(filt, c) for c in clips])
                assert np.array_equal(g.extract(clips), want), (sec, mode)
                hp = torch.zeros((nclips, plan.n_hp), dtype=torch.int64, device="cuda")
                g.hashprints_from_db_dev(d_db.data_ptr(), nclips, plan.c, hp.data_ptr())
                torch.cuda.synchronize()
                assert np.array_equal(hp.cpu().numpy().view(np.uint64), want), (sec, mode)
                if mode == 1:                                         # the exact integer sums behind the bits
                    dq = torch.zeros((nclips, 64, plan.n_hp), dtype=torch.int64, device="cuda")
                    g.stage_delta_q_dev(d_db.data_ptr(), nclips, plan.c, dq.data_ptr())
                    torch.cuda.synchronize()
                    for i in range(nclips):
                        assert np.array_equal(dq[i].cpu().numpy(), oracle.delta_q(filt, db[i])), (sec, i)
    finally:
        oracle.set_projection(before)
        g.close()


def test_the_two_projections_agree_within_rounding(oracle):
    """the fixed-point projection, scaled back, lies within 10^-3 of the f32 chain (|P| up to ~140: that is the f32
    chain's own rounding error); the hashprints of the two differ in at most 10^-4 of their bits"""
    filt = synth.make_filters()
    clip = synth.gen_clip(31, 20.0)
    plan = oracle.Plan(clip.size)
    db = oracle.db(plan.cqmag(plan.spectrum(clip)))
    pf, pq = oracle.project(filt, db), oracle.project_q(filt, db)
    fq = oracle.quantise_filters(filt)
    m = np.abs(filt.reshape(2420, 64)).max(axis=0)
    e = 21 - np.floor(np.log2(m)).astype(int)
    back = pq / (2.0 ** e)[:, None] / 98304.0
    assert np.abs(back - pf).max() < 1e-3
    x = oracle.pack(pf) ^ oracle.pack_q(pq)
    assert sum(bin(int(v)).count("1") for v in x) <= 1e-4 * x.size * 64


def test_fixed_point_projection_edge_values(oracle):
    """the digit splits at their limits: spectrogram values at the clamp (-80), at the reference level (0) and in
    between to the last bit; filter rows that are all zero, a single one, +/- powers of two (the row scale's
    boundary), denormal-small and alternating extremes; ragged widths down to one hashprint"""
    import torch
    rng = np.random.default_rng(77)
    f = rng.standard_normal((2420, 64)).astype(np.float32) * 0.03          # [k][r] = column-major [64][2420]
    f[:, 0] = 0.0                                                            # an all-zero row: every delta is 0 -> bit set
    f[:, 1] = 0.0; f[17, 1] = 1.0                                            # a single tap
    f[:, 2] = np.where(rng.random(2420) < 0.5, 0.25, -0.25)                  # +/- a power of two: fq = +/- 2^21 exactly
    f[:, 3] = np.float32(2.0) ** rng.integers(-20, 2, 2420) * rng.choice([-1, 1], 2420)
    f[:, 4] = np.nextafter(np.float32(0.5), np.float32(1.0))                 # just above a power of two
    f[:, 5] = np.nextafter(np.float32(0.5), np.float32(0.0))                 # just below: the row scale changes
    f[:, 6] = 1e-30
    filt = np.ascontiguousarray(f).ravel()
    g = hpfw_amd.Gpu(0)
    g.set_filters(filt)
    g.set_projection(1)
    before = oracle.get_projection()
    oracle.set_projection(1)
    try:
        for c in (100, 101, 355, 356, 1000):
            s = rng.uniform(-80, 0, (3, 121, c)).astype(np.float32)
            s[0, :, ::2] = -80.0
            s[0, :, 1::2] = 0.0
            s[1] = np.round(s[1] * 98304.0) / 98304.0                       # steps of the fixed-point grid
            s[1, 5] = np.nextafter(np.float32(-80.0), np.float32(0.0))
            s[2, :, : c // 2] = -80.0
            s[2, 7, 1::3] = 0.0                                             # |Du| at its largest: -80 against 0, 80 columns apart
            d_s = torch.from_numpy(s).cuda()
            hp = torch.zeros((3, c - 99), dtype=torch.int64, device="cuda")
            g.hashprints_from_db_dev(d_s.data_ptr(), 3, c, hp.data_ptr())
            torch.cuda.synchronize()
            got = hp.cpu().numpy().view(np.uint64)
            dq = torch.zeros((3, 64, c - 99), dtype=torch.int64, device="cuda")
            g.stage_delta_q_dev(d_s.data_ptr(), 3, c, dq.data_ptr())
            torch.cuda.synchronize()
            for i in range(3):
                assert np.array_equal(got[i], oracle.hashprints_from_db(filt, s[i])), (c, i)
                assert np.array_equal(dq[i].cpu().numpy(), oracle.delta_q(filt, s[i])), (c, i)
            assert ((got >> np.uint64(63)) == 1).all()                       # row 0: zero filters, delta 0 >= 0
    finally:
        oracle.set_projection(before)
        g.close()
