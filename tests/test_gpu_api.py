"""GPU tests of the drop-in surface: the reference's eight FFI symbols (through the pyhpfw twin),
the cereal filter file, the C++ LiveSongIdentification facade (config 0: a handful of WAVs indexed
and queried end to end) and the error behaviour of the C-ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import hpfw_amd  # noqa: E402
from hpfw_amd import _lib, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_filters(cache_dir, filt):
    os.makedirs(cache_dir, exist_ok=True)
    with open(os.path.join(cache_dir, "filters.cereal"), "wb") as f:      # utils.h:84-90
        f.write(np.array([64, 2420], np.int32).tobytes())
        f.write(np.ascontiguousarray(filt, np.float32).tobytes())


@pytest.fixture(scope="module")
def wav_set(tmp_path_factory, torch_cuda):
    d = tmp_path_factory.mktemp("wavs")
    clips = [synth.gen_clip(500 + i, 8.0) for i in range(4)]
    paths = []
    for i, c in enumerate(clips):
        p = str(d / f"track{i:02d}.wav")
        synth.write_wav(p, c)
        paths.append(p)
    qpaths = []
    for q in range(3):
        pcm, ci, start = synth.gen_query(clips, q, seconds=3.0)
        p = str(d / f"live_track{ci:02d}_take{q}.wav")
        synth.write_wav(p, pcm)
        qpaths.append((p, ci, start))
    return d, clips, paths, qpaths


def test_legacy_ffi_through_python_twin(wav_set, oracle, filters):
    d, clips, paths, qpaths = wav_set
    cache = str(d / "cache") + "/"
    _write_filters(cache, filters)
    pc = hpfw_amd.ParallelCollector()
    with pytest.raises(hpfw_amd.HpfwError):            # no filters yet: an error, never garbage (D-9)
        pc.calc_hashprint(paths[0])
    pc.load(cache)
    os.environ["HPFW_PREPARE_KEEP_FILTERS"] = "1"        # use the loaded fixture instead of re-learning
    try:
        res = pc.prepare(paths + [str(d / "missing.wav")])  # a bad file is skipped, as parallel_collector.h:101-103
    finally:
        del os.environ["HPFW_PREPARE_KEEP_FILTERS"]
    assert [name for _, name in res] == [f"track{i:02d}" for i in range(4)]
    plan = oracle.Plan(clips[0].size)
    for (hp, _), c in zip(res, clips):
        assert hp.dtype == np.uint64 and np.array_equal(hp, plan.extract(filters, c))
    many = pc.calc_hashprints([paths[2], str(d / "missing.wav"), qpaths[0][0], paths[0]])   # one batched call
    assert [n for _, n in many] == ["track02", "missing", os.path.splitext(os.path.basename(qpaths[0][0]))[0], "track00"]
    assert many[1][0] is None and np.array_equal(many[0][0], res[2][0]) and np.array_equal(many[3][0], res[0][0])
    q = pc.calc_hashprint(qpaths[0][0])
    assert np.array_equal(many[2][0], q)
    assert np.array_equal(q, oracle.Plan(3 * 44100).extract(filters, synth.gen_query(clips, 0, seconds=3.0)[0]))
    pc.save(str(d / "cache2"))
    raw = open(str(d / "cache2" / "filters.cereal"), "rb").read()
    assert raw[:8] == np.array([64, 2420], np.int32).tobytes() and raw[8:] == np.asarray(filters, np.float32).tobytes()


def test_calc_hashprints_a_new_length_with_every_file(wav_set, oracle, filters, tmp_path):
    """a directory of tracks brings a new length with every file (parallel_collector.h:82-137 reads whatever the files
    hold): one batched call over seven files of seven lengths -- 7-smooth and not -- returns the oracle's hashprints in
    input order; the tables of each length are prepared by the reader threads and generated on the device while the
    previous file's kernels run"""
    d, clips, _, _ = wav_set
    pc = hpfw_amd.ParallelCollector()
    pc.load(str(d / "cache") + "/")
    paths, pcms = [], []
    for i in range(7):
        pcm = synth.gen_clip(900 + i, 3.0)[: 3 * 44100 - 11 * i]   # 132300, 132289, ... samples
        p = str(tmp_path / f"odd{i}.wav")
        synth.write_wav(p, pcm)
        paths.append(p)
        pcms.append(pcm)
    for rounds in range(2):                                        # the second call finds the tables cached
        got = pc.calc_hashprints(paths)
        assert [n for _, n in got] == [f"odd{i}" for i in range(7)]
        for (hp, _), pcm in zip(got, pcms):
            assert np.array_equal(hp, oracle.Plan(pcm.size).extract(filters, pcm))


def test_more_lengths_than_the_table_cache_holds(wav_set, oracle, filters, tmp_path, monkeypatch):
    """a corpus of distinct lengths larger than the cache of per-length tables (HPFW_PLAN_CACHE_GB; here room for a handful
    of 2 s lengths), in more than one window of files: tables are evicted while the collector runs -- least recently used
    first; without a device-wide wait where the collector has already waited for the window that used them, with one
    otherwise -- and their memory goes to the next lengths.  Every 9th file against the oracle, twice (the second call
    meets every length again after its eviction)"""
    d, clips, _, _ = wav_set
    monkeypatch.setenv("HPFW_PLAN_CACHE_GB", "0.03")
    pc = hpfw_amd.ParallelCollector()
    pc.load(str(d / "cache") + "/")
    base = synth.gen_clip(1300, 2.0)
    paths, pcms = [], []
    for i in range(300):                                           # 256 files make a window
        pcm = np.roll(base, 131 * i)[: base.size - 3 * i]
        p = str(tmp_path / f"len{i:03d}.wav")
        synth.write_wav(p, pcm)
        paths.append(p)
        pcms.append(pcm)
    for rounds in range(2):
        got = pc.calc_hashprints(paths)
        assert [n for _, n in got] == [f"len{i:03d}" for i in range(300)]
        for i in range(rounds, 300, 9):
            assert np.array_equal(got[i][0], oracle.Plan(pcms[i].size).extract(filters, pcms[i])), (rounds, i)


def test_stereo_and_unsupported_wav(wav_set, filters, tmp_path):
    d, clips, _, _ = wav_set
    pc = hpfw_amd.ParallelCollector()
    pc.load(str(d / "cache") + "/")
    stereo = np.stack([clips[0], clips[0]], axis=1)      # identical channels: the "mix" downmix returns the clip
    p = str(tmp_path / "stereo.wav")
    synth.write_wav(p, stereo, channels=2)
    mono = str(tmp_path / "mono.wav")
    synth.write_wav(mono, clips[0])
    assert np.array_equal(pc.calc_hashprint(p), pc.calc_hashprint(mono))
    # WAVE_FORMAT_EXTENSIBLE header, a LIST chunk of odd size before the data, and a streamed file whose
    # data size field is 0xffffffff: the same samples, the same hashprints
    import struct
    data = np.ascontiguousarray(clips[0], np.int16).tobytes()
    ext = str(tmp_path / "extensible.wav")
    fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, 1, 44100, 88200, 2, 16, 22, 16, 4, 1) + bytes.fromhex("000000001000800000aa00389b71")
    with open(ext, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + 6 + 8 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"LIST" + struct.pack("<I", 5) + b"INFOx" + b"\0")
        f.write(b"data" + struct.pack("<I", 0xFFFFFFFF) + data)
    assert len(fmt) == 40
    assert np.array_equal(pc.calc_hashprint(ext), pc.calc_hashprint(mono))
    odd = str(tmp_path / "odd.wav")
    synth.write_wav(odd, clips[0][:44100 * 8 - 1])       # 352799 samples = 13 * 27138...: exactly that length goes in
    assert hpfw_amd.supported_length(44100 * 8 - 1) == 44100 * 8 - 1
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    assert np.array_equal(pc.calc_hashprint(odd), g.extract(clips[0][:44100 * 8 - 1])[0])
    g.close()
    with pytest.raises(hpfw_amd.HpfwError):
        pc.calc_hashprint(str(tmp_path / "nope.wav"))


def test_cpp_live_song_identification(wav_set, filters, tmp_path):
    """config 0 plumbing: index + search through hpfw::LiveSongIdentification<GpuCollector, GpuStorage>"""
    d, clips, paths, qpaths = wav_set
    exe = str(tmp_path / "live_id")
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "live_id.cpp"), "-o", exe, "-L", libdir, "-lhpfw_gpu",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    work = tmp_path / "run"
    _write_filters(str(work / "cache"), filters)
    r = subprocess.run([exe, "--index"] + paths + ["--search"] + [q[0] for q in qpaths], cwd=str(work),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("=> ")]
    assert lines[-1] == "=> 0 1"                         # live_song_id.h:53: wrong count, accuracy
    hop = 3.0 * (8 * 44100) / round((2 ** (1 / 24) - 2 ** (-1 / 24)) * 130.81 * 32 * 8)   # samples per column
    for (qp, ci, start), k in zip(qpaths, range(3)):
        assert lines[2 * k] == f"=> Finding {qp}"
        name, cnt, off = lines[2 * k + 1][3:].split()
        assert name == f"track{ci:02d}" and int(cnt) > 0 and abs(int(off) - start / hop) <= 2
    # index() learned its own filters and the destructor saved them (live_song_id.h:23-29)
    learned = np.frombuffer(open(str(work / "cache" / "filters.cereal"), "rb").read()[8:], np.float32)
    assert learned.size == 64 * 2420 and not np.array_equal(learned, np.asarray(filters, np.float32).ravel())
    # f2: the database in MemoryStorage's cereal format (storage.h:67-86), written and read back
    dump = str(work / "dump.cereal")
    r2 = subprocess.run([exe, "--index"] + paths + ["--dump", dump], cwd=str(work), capture_output=True, text=True,
                        timeout=300, env=dict(os.environ, HPFW_PREPARE_KEEP_FILTERS="1"))
    assert r2.returncode == 0, r2.stdout + r2.stderr
    raw = open(dump, "rb").read()
    pos, entries = 8, []
    assert int(np.frombuffer(raw[:8], np.uint64)[0]) == 4
    for _ in range(4):
        n = int(np.frombuffer(raw[pos:pos + 8], np.uint64)[0])
        name = raw[pos + 8:pos + 8 + n].decode()
        pos += 8 + n
        n = int(np.frombuffer(raw[pos:pos + 8], np.uint64)[0])
        entries.append((name, np.frombuffer(raw[pos + 8:pos + 8 + 8 * n], np.uint64)))
        pos += 8 + 8 * n
    assert pos == len(raw) and [e[0] for e in entries] == [f"track{i:02d}" for i in range(4)]
    g = hpfw_amd.Gpu(0)
    g.set_filters(learned)
    assert all(np.array_equal(e[1], hp) for e, hp in zip(entries, g.extract(np.stack(clips))))
    g.close()
    r3 = subprocess.run([exe, "--db", dump, "--search"] + [q[0] for q in qpaths] + ["--votes"], cwd=str(work),
                        capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    assert [ln for ln in r3.stdout.splitlines() if ln.startswith("=> ")] == lines
    # the voting search (AnnStorage semantics) names the same tracks; its offset is i - p, minus the scan's
    voted = [ln[3:].split() for ln in r3.stdout.splitlines() if ln.startswith("=# ")]
    assert len(voted) == 3
    for (qp, ci, start), (name, cnt, off) in zip(qpaths, voted):
        assert name == f"track{ci:02d}" and float(cnt) > 0 and abs(-int(off) - start / hop) <= 2


def test_baseline_config0_plumbing(torch_cuda, tmp_path):
    """BASELINE.json configs[0]: 10 x 30 s synthetic WAVs indexed and 10 x 5 s noisy slices queried through
    LiveSongIdentification<GpuCollector, GpuStorage> (filters learned from the ten tracks): 10 of 10
    right, offsets at the planted positions"""
    exe = str(tmp_path / "live_id")
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "live_id.cpp"), "-o", exe, "-L", libdir, "-lhpfw_gpu",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    clips = [synth.gen_clip(i, 30.0) for i in range(10)]
    tracks, queries = [], []
    for i, c in enumerate(clips):
        p = str(tmp_path / f"song{i:02d}.wav")
        synth.write_wav(p, c)
        tracks.append(p)
    for q in range(10):
        pcm, ci, start = synth.gen_query(clips, q, seconds=5.0)
        p = str(tmp_path / f"live_song{ci:02d}_{q}.wav")
        synth.write_wav(p, pcm)
        queries.append((p, ci, start))
    work = tmp_path / "run"
    os.makedirs(str(work / "cache"))
    r = subprocess.run([exe, "--index"] + tracks + ["--search"] + [q[0] for q in queries], cwd=str(work),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("=> ")]
    assert lines[-1] == "=> 0 1"                                           # wrong count 0, accuracy 1
    hop = 3.0 * 1323000 / 7255                                            # samples per spectrogram column (M = 7255)
    for k, (qp, ci, start) in enumerate(queries):
        name, cnt, off = lines[2 * k + 1][3:].split()
        assert name == f"song{ci:02d}" and abs(int(off) - start / hop) <= 2
    # the same queries as one batch against the same filters (read back from cache/): identical output
    r2 = subprocess.run([exe, "--batch", "--index"] + tracks + ["--search"] + [q[0] for q in queries] + [str(tmp_path / "absent.wav")],
                        cwd=str(work), capture_output=True, text=True, timeout=600,
                        env=dict(os.environ, HPFW_PREPARE_KEEP_FILTERS="1"))
    assert r2.returncode == 0, r2.stdout + r2.stderr
    lines2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("=> ")]
    assert lines2[:-2] == lines[:-1] and lines2[-2].startswith("=> Finding") and "absent.wav" in r2.stderr


def test_index_readback_and_cached_spectrogram(torch_cuda, oracle, filters):
    g = hpfw_amd.Gpu(0)
    db = synth.random_hashprints(5, 300)
    off = np.array([0, 300, 420, 420, 1000, 1500], np.int64)               # ragged, one empty clip
    g.index_add(db.ravel(), off)
    hp, off2 = g.index_get()
    assert np.array_equal(hp, db.ravel()) and np.array_equal(off2, off)
    # hashprints from a dB spectrogram as cache/spectros/<stem> holds it (column-major [121][C])
    g.set_filters(filters)
    rng = np.random.default_rng(8)
    for c in (100, 137, 404):
        s = rng.uniform(-80, 0, (121, c)).astype(np.float32)
        got = g.extract_db(np.ascontiguousarray(s.T))
        assert np.array_equal(got, oracle.hashprints_from_db(filters, s))
    assert g.extract_db(np.zeros((99, 121), np.float32)).size == 0          # too short: no hashprints
    with pytest.raises(hpfw_amd.HpfwError):
        g.extract_db(np.zeros((200, 120), np.float32))                         # not a 121-bin spectrogram
    g.close()


def test_c_abi_error_codes(torch_cuda):
    L = hpfw_amd.lib()
    g = hpfw_amd.Gpu(0)
    pcm = np.zeros((1, 132300), np.int16)
    with pytest.raises(hpfw_amd.HpfwError) as e:
        g.extract(pcm)
    assert "-3" in str(e.value) and "filters" in str(e.value)        # HPFW_E_NOFILTERS
    assert g.geometry(132301).n2 == 6300                              # 11 | 132301: the chirp-z forward transform
    with pytest.raises(hpfw_amd.HpfwError) as e:
        g.geometry(4410)                                              # 0.1 s: the bands leave the half spectrum
    assert "-2" in str(e.value)                                       # HPFW_E_UNSUPPORTED
    with pytest.raises(hpfw_amd.HpfwError) as e:
        g.set_conventions(64)
    assert "-1" in str(e.value)                                       # HPFW_E_INVALID
    import ctypes
    h = ctypes.c_void_p()
    assert L.hpfw_gpu_create(99, ctypes.byref(h)) == -1               # HPFW_E_INVALID: no such device
    assert b"device" in L.hpfw_gpu_last_error()
    g.set_filters(synth.make_filters())
    hp = g.extract(pcm)                                               # silence: every delta is 0 -> all ones
    assert (hp == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    g.close()


def test_python_liveid_and_notebook_example(wav_set, oracle, filters, tmp_path):
    """the Python LiveSongIdentification twin and examples/liveid.py (the reference's notebook): ten best
    tracks per query equal the oracle's scan; the dump round-trips through pickle"""
    d, clips, paths, qpaths = wav_set
    cache = str(tmp_path / "cache") + "/"
    _write_filters(cache, filters)
    os.environ["HPFW_PREPARE_KEEP_FILTERS"] = "1"
    try:
        lid = hpfw_amd.LiveSongIdentification(cache=cache)
        lid.index(paths)
        ans = lid.top([q[0] for q in qpaths] + [str(d / "missing.wav")], 10)
        wrong, acc = lid.search([q[0] for q in qpaths])
        lid.close()
        dump = str(tmp_path / "dump.pkl")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "liveid.py"), "--cache", cache, "--index"] + paths +
                           ["--dump", dump, "--search"] + [q[0] for q in qpaths], capture_output=True, text=True, timeout=300)
    finally:
        del os.environ["HPFW_PREPARE_KEEP_FILTERS"]
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1] == "accuracy 1.0"
    assert (wrong, acc) == (0, 1.0) and ans[-1][1] is None
    plan = oracle.Plan(clips[0].size)
    db = [plan.extract(filters, c) for c in clips]
    off = np.zeros(len(db) + 1, np.int64)
    np.cumsum([x.size for x in db], out=off[1:])
    for (label, top), (qp, ci, start) in zip(ans, qpaths):
        q = oracle.Plan(3 * 44100).extract(filters, synth.gen_query(clips, qpaths.index((qp, ci, start)), seconds=3.0)[0])
        want = oracle.search_topk(np.concatenate(db), off, q, np.array([0, q.size], np.int64), 10)[0]
        assert [(d_, f"track{c:02d}", o) for d_, c, o in
                [(int(h["dist"]), int(h["clip"]), int(h["offset"])) for h in want if h["clip"] != 0xFFFFFFFF]] == top
    import pickle
    got = pickle.load(open(dump, "rb"))
    assert [n for _, n in got] == [f"track{i:02d}" for i in range(4)] and all(np.array_equal(a, b) for (a, _), b in zip(got, db))


def test_constant_q_classes_side_by_side_or_in_turn(torch_cuda, oracle, filters, monkeypatch):
    """the size classes of the constant-Q stage are launched on forked streams that join the caller's before the next stage
    (api.hip run_front); HPFW_CQ_SERIAL=1 launches them one after the other: the same hashprints either way, on the null
    stream and on a side stream, with another call enqueued right behind on a third stream"""
    torch = torch_cuda
    clips = np.stack([synth.gen_clip(640 + i, 12.0) for i in range(5)])   # (fewer than four clips are never forked)
    plan = oracle.Plan(clips.shape[1])
    want = plan.extract_batch(filters, clips, n_threads=5)
    d = torch.from_numpy(clips).cuda()
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("HPFW_CQ_SERIAL", "1")
        g = hpfw_amd.Gpu(0)                                   # the switch is read when a handle is created
        g.set_filters(filters)
        side, other = torch.cuda.Stream(), torch.cuda.Stream()
        for stream in (0, side.cuda_stream):
            hp = torch.zeros((5, plan.n_hp), dtype=torch.int64, device="cuda")
            hp2 = torch.zeros_like(hp)
            torch.cuda.synchronize()
            g.extract_dev(d.data_ptr(), clips.shape[1], 5, hp.data_ptr(), stream)
            g.extract_dev(d.data_ptr(), clips.shape[1], 5, hp2.data_ptr(), other.cuda_stream)   # shares the workspaces: ordered by the handle
            torch.cuda.synchronize()
            assert np.array_equal(hp.cpu().numpy().view(np.uint64), want), (serial, stream)
            assert np.array_equal(hp2.cpu().numpy().view(np.uint64), want), (serial, stream)
        g.close()


def test_forward_transform_in_chunks_on_streams_in_turn(torch_cuda, oracle, filters, monkeypatch):
    """a large batch goes through the forward transform in chunks of HPFW_FWD_CHUNK clips (default 16) that HPFW_FWD_STREAMS
    streams (default 2: the caller's and a side stream) take in turn, column stage and row stage of a chunk back to back
    (api.hip run_forward: the column stage's output stays in the Infinity Cache); HPFW_FWD_CHUNK=0 runs one launch per
    stage over the whole batch.  100 clips = six chunks of 16 and a ragged one of 4: the same hashprints whatever the
    chunking, on the null stream and on a side stream, with another call enqueued right behind on a third stream"""
    torch = torch_cuda
    base = np.stack([synth.gen_clip(900 + i, 2.0) for i in range(10)])
    clips = np.concatenate([np.roll(base, 37 * r, axis=1) for r in range(10)])        # 100 different clips
    clips[5::10] = -clips[5::10]
    n = clips.shape[1]
    plan = oracle.Plan(n)
    want = plan.extract_batch(filters, clips, n_threads=8)
    d = torch.from_numpy(clips).cuda()
    for chunk, streams in ((None, None), ("7", "3"), ("16", "1"), ("0", None), ("9", "5")):
        for key, val in (("HPFW_FWD_CHUNK", chunk), ("HPFW_FWD_STREAMS", streams)):
            if val is None:
                monkeypatch.delenv(key, raising=False)
            else:
                monkeypatch.setenv(key, val)
        g = hpfw_amd.Gpu(0)                                   # the switches are read when a handle is created
        g.set_filters(filters)
        side, other = torch.cuda.Stream(), torch.cuda.Stream()
        for stream in (0, side.cuda_stream):
            hp = torch.zeros((len(clips), plan.n_hp), dtype=torch.int64, device="cuda")
            hp2 = torch.zeros_like(hp)
            torch.cuda.synchronize()
            g.extract_dev(d.data_ptr(), n, len(clips), hp.data_ptr(), stream)
            g.extract_dev(d.data_ptr(), n, len(clips), hp2.data_ptr(), other.cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(hp.cpu().numpy().view(np.uint64), want), (chunk, streams, stream)
            assert np.array_equal(hp2.cpu().numpy().view(np.uint64), want), (chunk, streams, stream)
        g.close()
    # the chirp-z forward transform (a length with a prime factor above 7) has the same switch, HPFW_BZ_CHUNK (default 32)
    odd = np.ascontiguousarray(clips[:20, :n - 1199])
    plan = oracle.Plan(odd.shape[1])
    want = plan.extract_batch(filters, odd, n_threads=8)
    d = torch.from_numpy(odd).cuda()
    for chunk, streams in (("3", "2"), ("3", "4"), ("0", "2")):
        monkeypatch.setenv("HPFW_BZ_CHUNK", chunk)
        monkeypatch.setenv("HPFW_FWD_STREAMS", streams)
        g = hpfw_amd.Gpu(0)
        g.set_filters(filters)
        hp = torch.zeros((len(odd), plan.n_hp), dtype=torch.int64, device="cuda")
        g.extract_dev(d.data_ptr(), odd.shape[1], len(odd), hp.data_ptr(), 0)
        torch.cuda.synchronize()
        assert np.array_equal(hp.cpu().numpy().view(np.uint64), want), (chunk, streams)
        g.close()


def test_streams_mixed_without_sync(torch_cuda, oracle, filters):
    """One handle, three streams, no synchronisation by the caller: a device call on a non-blocking side
    stream, the host entry point (its own private streams), a device call on the null stream and an index
    that grows while appends are in flight.  The handle orders the calls itself (they share its workspaces):
    every result equals the oracle's."""
    torch = torch_cuda
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    a = np.stack([synth.gen_clip(610 + i, 5.0) for i in range(6)])
    b = np.stack([synth.gen_clip(620 + i, 3.0) for i in range(5)])
    plan_a, plan_b = oracle.Plan(a.shape[1]), oracle.Plan(b.shape[1])
    want_a = plan_a.extract_batch(filters, a, n_threads=6)
    want_b = plan_b.extract_batch(filters, b, n_threads=5)
    g.extract(a[:1]), g.extract(b[:1])                       # plans and workspaces exist before the mixing starts
    side = torch.cuda.Stream()                               # torch side streams do not sync with the null stream
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    hp_a = torch.zeros((6, plan_a.n_hp), dtype=torch.int64, device="cuda")
    hp_b = torch.zeros((5, plan_b.n_hp), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        g.extract_dev(d_a.data_ptr(), a.shape[1], 6, hp_a.data_ptr(), side.cuda_stream)   # enqueued, not waited for
        host_b = g.extract(b)                                                              # private streams
        g.extract_dev(d_b.data_ptr(), b.shape[1], 5, hp_b.data_ptr(), 0)                   # null stream
        host_a = g.extract(a)
        torch.cuda.synchronize()
        assert np.array_equal(host_a, want_a) and np.array_equal(host_b, want_b)
        assert np.array_equal(hp_a.cpu().numpy().view(np.uint64), want_a)
        assert np.array_equal(hp_b.cpu().numpy().view(np.uint64), want_b)
        hp_a.zero_(), hp_b.zero_()
        torch.cuda.synchronize()
    # index appends on the side stream while the capacity doubles (the grow-copy must see them), then a
    # search on the null stream
    rng = np.random.default_rng(3)
    n, per = 40, 3000                                        # 40 appends of 3000 hashprints: several doublings of 65536
    clips = torch.from_numpy(rng.integers(-2 ** 63, 2 ** 63 - 1, size=(n, per), dtype=np.int64)).cuda()
    torch.cuda.synchronize()
    g.index_clear()
    for i in range(n):
        g.index_add_dev(clips[i].data_ptr(), np.array([0, per], np.int64), side.cuda_stream)
    q = clips[31, 100:400].cpu().numpy().view(np.uint64)
    hits = g.search_topk(q, np.array([0, q.size], np.int64), 2)
    assert hits[0, 0]["clip"] == 31 and hits[0, 0]["offset"] == 100 and hits[0, 0]["dist"] == 0
    hp, off = g.index_get()
    assert np.array_equal(hp.view(np.int64).reshape(n, per), clips.cpu().numpy())
    g.close()


def test_tables_prepared_by_other_threads(torch_cuda, oracle, filters):
    """hpfw_gpu_prepare_length: host threads build the host half of the tables of the lengths they meet while the
    calling thread extracts -- lengths prepared ahead, a length in preparation at the moment it is needed, a length
    never prepared, a length prepared twice and an unsupported one; every hashprint equals the oracle's"""
    import threading
    g = hpfw_amd.Gpu(0)
    g.set_filters(filters)
    lengths = [132300, 132301, 99991, 88211, 220501, 176400]     # 7-smooth and not
    clips = {n: np.stack([synth.gen_clip(700 + i, n / 44100.0 + 0.1)[:n] for i in range(2)]) for n in lengths}
    errors = []

    def prepare(ns):
        try:
            for n in ns:
                g.prepare_length(n)
        except Exception as e:                                    # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=prepare, args=(lengths[i::3] + lengths[:2],)) for i in range(3)]   # overlaps: twice
    for t in th:
        t.start()
    got = {n: g.extract(clips[n]) for n in lengths[::-1]}         # may meet lengths still in preparation
    for t in th:
        t.join()
    assert not errors, errors
    extra = 352799                                                # never prepared: built by the extracting thread
    c = np.stack([synth.gen_clip(777, 8.1)[:extra]])
    assert np.array_equal(g.extract(c), np.stack([oracle.Plan(extra).extract(filters, c[0])]))
    for n in lengths:
        plan = oracle.Plan(n)
        assert np.array_equal(got[n], np.stack([plan.extract(filters, x) for x in clips[n]])), n
    for _ in range(2):
        with pytest.raises(hpfw_amd.HpfwError):
            g.prepare_length(1000)                                # too short, every time it is asked for
    g.prepare_length(132300)                                      # known already: nothing happens
    g.close()
