"""Host-side SIMT emulation of the in-LDS transform kernel bodies (tests/emu): the same source the
HIP kernels compile, run thread by thread with bounds-checked LDS and compared with the oracle.
Checks index arithmetic where no GPU is available; the GPU parity tests remain the proof."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu_cq(tmp_path_factory):
    exe = tmp_path_factory.mktemp("emu") / "emu_cq"
    cmd = ["g++", "-O2", "-std=c++17", "-DHPFW_SIMT_EMU", "-ffp-contract=off", "-mfma", "-mavx2", "-o", str(exe),
           os.path.join(ROOT, "tests", "emu", "emu_cq.cpp"), os.path.join(ROOT, "hpfw_amd", "csrc", "plan.cpp"),
           os.path.join(ROOT, "oracle", "hpfw_oracle.c"), "-lm", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(exe)


@pytest.mark.parametrize("n", [88200, 132300, 220500, 1323000])
def test_chirpz_body_matches_oracle(emu_cq, n):
    r = subprocess.run([emu_cq, str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout


@pytest.fixture(scope="module")
def emu_rows(tmp_path_factory):
    exe = tmp_path_factory.mktemp("emu") / "emu_rows"
    cmd = ["g++", "-O2", "-std=c++17", "-DHPFW_SIMT_EMU", "-ffp-contract=off", "-mfma", "-mavx2", "-o", str(exe),
           os.path.join(ROOT, "tests", "emu", "emu_rows.cpp"), os.path.join(ROOT, "hpfw_amd", "csrc", "plan.cpp"),
           os.path.join(ROOT, "oracle", "hpfw_oracle.c"), "-lm", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(exe)


@pytest.mark.parametrize("n,threads", [(132300, 512), (220500, 448), (154350, 512), (1323000, 512)])
def test_row_transform_body_matches_oracle(emu_rows, n, threads):
    """forward row FFT + Hermitian split (fft_rows.h): the compile-time group sequence with its
    transposed last two groups on even residue pairs, the run-time sequence on the others"""
    r = subprocess.run([emu_rows, str(n), str(threads)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout
