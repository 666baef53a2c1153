"""How much do the four unverifiable essentia conventions matter?  (include/hpfw_gpu.h
HPFW_CONV_*.)  For each switch, and all of them together, the oracle extracts the hashprints of synthetic clips
and counts the bits that differ from the default setting; geometry changes (M, C, hashprints per clip) are
reported as such.  CPU only (the GPU path is bit-identical to the oracle under every setting:
tests/test_gpu_parity.py::test_switchable_essentia_conventions).

    python tools/convention_sensitivity.py [out.json]       # default: tests/golden/convention_sensitivity.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpfw_amd import synth  # noqa: E402
from oracle import nsgt_f64, oracle  # noqa: E402

NAMES = {1: "hann_periodic", 2: "lg_round_half_even", 4: "float_geometry", 8: "no_ifft_scale", 15: "all_four"}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "convention_sensitivity.json")
    filt = synth.make_filters()
    report = {"filters": "hpfw_amd.synth.make_filters()", "clips": "hpfw_amd.synth.gen_clip(0..3, seconds)", "lengths": {}}
    for seconds, n_extra in ((30.0, 0), (5.0, 0), (30.0, 1), (7.3, 0)):
        clips = [synth.gen_clip(i, seconds + 0.01)[: int(round(seconds * 44100)) + n_extra] for i in range(4)]
        n = clips[0].size
        base_plan = oracle.Plan(n)
        base = [base_plan.extract(filt, c) for c in clips]
        base_mag = nsgt_f64.cq_magnitudes(clips[0])
        entry = {"M": base_plan.m, "C": base_plan.c, "hashprints_per_clip": base_plan.n_hp, "variants": {}}
        for conv, name in NAMES.items():
            plan = oracle.Plan(n, conventions=conv)
            v = {"M": plan.m, "C": plan.c, "hashprints_per_clip": plan.n_hp,
                 "bands_with_other_Lg": int((plan.lg != base_plan.lg).sum()),
                 "bands_with_other_start": int((plan.start != base_plan.start).sum())}
            if plan.n_hp == base_plan.n_hp:
                hp = [plan.extract(filt, c) for c in clips]
                diff = sum(bin(int(x)).count("1") for a, b in zip(hp, base) for x in (a ^ b))
                v["hashprint_bits_changed"] = diff
                v["fraction_of_bits"] = diff / (64.0 * plan.n_hp * len(clips))
            else:
                v["hashprint_bits_changed"] = None      # another number of columns: hashprints are not comparable 1:1
            mag = nsgt_f64.cq_magnitudes(clips[0], conv)
            if mag.shape == base_mag.shape:
                scale = (plan.m if conv & 8 else 1.0)
                v["max_rel_change_of_cq_magnitude"] = float((np.abs(mag / scale - base_mag).max(axis=1) / base_mag.max(axis=1)).max())
            entry["variants"][name] = v
        report["lengths"][str(n)] = entry
        print(n, json.dumps(entry["variants"]), flush=True)
    with open(out_path, "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
