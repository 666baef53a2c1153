"""CPU tests of the product library's host side: it loads, exports every symbol the header
declares, builds the same constant tables as the oracle, and its host-only helpers behave.
No compute call is made here (there is no GPU in this environment and no CPU fallback to call)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import hpfw_amd
from hpfw_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hpfw_gpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b((?:hpfw_gpu|par_collector)_\w+|prepare_result_free|calc_hashprint_result_free)\s*\(",
                              header))
    assert len(declared) >= 42
    L = hpfw_amd.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)
    assert b"gfx950" in L.hpfw_gpu_version()


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", _lib.LIB_PATH], capture_output=True, text=True)
    assert ".hip_fatbin" in out.stdout
    strings = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in strings and "hamming_scan_kernel" in strings and "project_kernel" in strings


@pytest.mark.parametrize("n", [1323000, 220500, 441000, 132300, 110250, 88200])
def test_plan_tables_identical_to_oracle(oracle, n):
    """twiddles, digit reversal, band geometry, window*chirp and chirp spectra: the product's host
    code (hpfw_amd/csrc/plan.cpp) and the oracle build them independently; FNV-1a checksums agree"""
    want = np.zeros(8, np.uint64)
    plan = oracle.Plan(n)
    oracle.lib().hpfw_oracle_plan_checksum.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    oracle.lib().hpfw_oracle_plan_checksum(plan._h, want.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(hpfw_amd.plan_checksum(n), want)


@pytest.mark.parametrize("n", [1323001, 352799, 99991, -220500])
def test_chirpz_plan_tables_identical_to_oracle(oracle, n):
    """lengths with a prime factor above 7 (and, negative, a 7-smooth length forced down the same path): geometry,
    band tables and the tables of the constant-Q stage are built independently by plan.cpp and the oracle; FNV-1a
    checksums agree.  (The chirp-z transform's own tables are generated on the device: tests/test_gpu_parity.py
    test_chirpz_tables_generated_on_device.)"""
    want = np.zeros(8, np.uint64)
    plan = oracle.Plan(abs(n), force_bluestein=n < 0)
    oracle.lib().hpfw_oracle_plan_checksum.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    oracle.lib().hpfw_oracle_plan_checksum(plan._h, want.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(hpfw_amd.plan_checksum(n), want)


def test_unsupported_lengths_fail_loudly():
    for n in (4410, 1, 44100 * 3600):              # too short, too short, chirp-z length above 2^19
        with pytest.raises(hpfw_amd.HpfwError):
            hpfw_amd.plan_checksum(n)


def test_supported_length():
    """every length between the shortest clip that yields a hashprint and the longest the tables allow is
    supported as it is: nothing is padded (the reference transforms the exact length, cqt.h:54-55)"""
    for n in (1323000, 1323001, 220501, 352799, 99991):
        assert hpfw_amd.supported_length(n) == n
    assert hpfw_amd.supported_length(44100 * 3600) == -1
    assert hpfw_amd.supported_length(10) == 54254                 # the shortest clip that yields a hashprint (1.23 s)


def test_merge_topk_host():
    h = np.zeros((3, 2, 3), hpfw_amd.HIT_DTYPE)
    h["dist"] = 0xFFFFFFFF
    h["clip"] = 0xFFFFFFFF
    h[0, 0] = [(5, 0, 1, 0), (9, 3, 2, 0), (9, 4, 7, 0)]
    h[1, 0] = [(5, 10, 1, 0), (6, 11, 0, 0), (0xFFFFFFFF, 0xFFFFFFFF, 0, 0)]
    h[2, 0] = [(4, 20, 9, 0), (9, 21, 0, 0), (12, 22, 0, 0)]
    h[1, 1, 0] = (7, 12, 3, 0)
    m = hpfw_amd.merge_topk(h, 3)
    assert [tuple(int(v) for v in x)[:3] for x in m[0]] == [(4, 20, 9), (5, 0, 1), (5, 10, 1)]
    assert tuple(int(v) for v in m[1, 0])[:3] == (7, 12, 3) and m[1, 1]["clip"] == 0xFFFFFFFF


def test_host_eigen_solver_against_numpy():
    """calc_filters' eigen-solve (hashprint_handle.h:105-112) on the host: leading eigenpairs of a
    covariance-like matrix against numpy.linalg.eigh"""
    import ctypes
    rng = np.random.default_rng(4)
    n, m = 300, 24
    x = rng.standard_normal((n, 900)) * np.linspace(4, 0.2, n)[:, None]
    x = np.linalg.qr(rng.standard_normal((n, n)))[0] @ x
    a = (x @ x.T / 900).astype(np.float32)
    out = np.zeros((m, n), np.float32)
    ev = np.zeros(m)
    rc = hpfw_amd.lib().hpfw_gpu_host_top_eigenvectors(a.ctypes.data_as(ctypes.c_void_p), n, m,
                                                       out.ctypes.data_as(ctypes.c_void_p),
                                                       ev.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    w, v = np.linalg.eigh(a.astype(np.float64))
    w, v = w[::-1], v[:, ::-1]
    assert np.abs(ev - w[:m]).max() / w[0] < 1e-12
    assert np.abs(out @ out.T - np.eye(m)).max() < 1e-5
    assert np.abs(np.sum(out * v[:, :m].T, axis=1)).min() > 0.99999
    assert (out[np.arange(m), np.abs(out).argmax(axis=1)] > 0).all()


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhpfw_gpu.so")
    with pytest.raises(hpfw_amd.HpfwError):
        _lib.lib()


def test_no_product_file_touches_the_oracle():
    """the oracle is test infrastructure: nothing under hpfw_amd/ or include/ may name it"""
    bad = []
    for base in ("hpfw_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                    text = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"hpfw_oracle|from oracle|import oracle|oracle/", text):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_cpp_facade_compiles_and_links(tmp_path):
    """the drop-in C++ surface: LiveSongIdentification<GpuCollector, GpuStorage>"""
    exe = tmp_path / "live_id"
    cmd = ["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "live_id.cpp"), "-o", str(exe),
           "-L", os.path.dirname(_lib.LIB_PATH), "-lhpfw_gpu", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH),
           "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_committed_sha_of_synthetic_clip_zero():
    """bench.py and the tests draw clip 0 from the same generator: its SHA-256 is committed"""
    import hashlib
    from hpfw_amd import synth
    want = open(os.path.join(ROOT, "tests", "golden", "clip0.sha256")).read().split()[0]
    assert hashlib.sha256(synth.gen_clip(0, 30.0).tobytes()).hexdigest() == want


def test_multi_library_exports_every_declared_symbol():
    """libhpfw_gpu_multi.so (include/hpfw_gpu_multi.h): loads, links librccl, exports what the header declares;
    the shard arithmetic is the one hpfw_amd.dist uses for the one-process-per-GPU launch"""
    from hpfw_amd import dist as hdist, multi
    header = open(os.path.join(ROOT, "include", "hpfw_gpu_multi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(hpfw_gpu_(?:group|shard)_\w+)\s*\(", header))
    assert declared == set(multi.EXPORTS) and len(declared) == 19
    L = multi.lib()
    assert not [s for s in sorted(declared) if not hasattr(L, s)]
    out = subprocess.run(["ldd", multi.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl.so" in out and "libhpfw_gpu.so" in out
    for n in (0, 1, 7, 100000, 12345):
        for world in (1, 3, 8):
            for r in range(world):
                assert multi.shard_range(n, r, world) == hdist.shard_range(n, r, world)


def test_multi_gpu_host_path_fails_loudly_without_a_gpu():
    """no device, no fallback: creating a group (or a handle) is an error with a message, never a silent CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from hpfw_amd import multi
    with pytest.raises(hpfw_amd.HpfwError):
        multi.GpuGroup([0])
    with pytest.raises(hpfw_amd.HpfwError):
        hpfw_amd.Gpu(0)
