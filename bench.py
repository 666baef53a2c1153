#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): "hashprints/sec (index) + Hamming matches/sec (search), 30 s@44.1 kHz clips".
One step = one pass of the extraction hot path (int16 PCM -> CQT -> dB -> projection -> 64-bit
hashprints) over the batch of configs[1]: 1 000 x 30 s synthetic 44.1 kHz clips per GPU, inputs
already resident in HBM.  `value` = hashprints/s over all ranks (weak scaling: every rank extracts
its own 1 000 clips, no collective on this path).  The search half of the metric (configs[2]:
10 000-clip index per GPU, 1 000 x 5 s queries, top-10, one RCCL all-gather of the per-shard
top-k) is timed in the same run and reported under "search".

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 through
python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROJECT_FLOP_PER_CLIP = 2.0 * 64 * 2420 * 2400       # filters [64x2420] . frames [2420x2400] (the f32 chain forms all 2400 frames)
PROJECT_Q_MACS_PER_CLIP = 64 * 2420 * 2320           # fixed point: the lag-80 difference is taken first, 2320 columns remain
MFMA_F32_PEAK_TFLOPS = 157.3                         # MI355X_MICROARCH.md: dense f32 MFMA peak
# v_mfma_i32_32x32x32_i8 takes 32 cycles per instruction and SIMD (MI355X_MICROARCH.md "I8: 2x BF16 per clock"; measured
# by tools/mfma_i8_probe.hip, profiles/r03_mfma_i8_probe.jsonl: 32.1 cycles on 1, 2 or 4 accumulators): 65536 operations
# x 1024 SIMDs x 2.4 GHz / 32 = 5033 dense int8 TOP/s at the nominal clock.  The same probe, back to back on random
# operands with nothing else in the loop, holds 1.63-1.79 GHz: 2.9-3.5 POP/s is what the chip sustains on this instruction.
MFMA_I8_PEAK_TOPS = 65536 * 1024 * 2.4e9 / 32 / 1e12
MFMA_I8_PROBE_TOPS = 3535.0                          # best case of the probe on random operands (4 accumulators, 2 waves per SIMD)
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md
VALU_PAIR_PEAK = 256 * 4 * 32 * 2.4e9 / 4            # xor/popcount kernel: 4 VALU lane-ops per 64-bit pair
MFMA_FP4_PEAK_PFLOPS = 10.07                         # MI355X_MICROARCH.md: dense FP4 MFMA (32x32x64 in 32 cycles/SIMD)
FP4_PAIR_PEAK = MFMA_FP4_PEAK_PFLOPS * 1e15 / 2 / 64 # one pair = 64 multiply-adds of +-1
SPEC_CLIPS = 8                                       # clips per rank taken from hpfw_amd.synth.gen_clip


def synth_clips_gpu(torch, n_clips, n_samples, seed, device, chunk=25):
    """synthetic 'songs' generated on the device: 0.25 s notes of 6 partials in the CQ range with
    10 ms fades plus -30 dBFS noise, int16.  (Same recipe as hpfw_amd.synth.gen_clip, torch RNG.)"""
    sr, note = 44100, 11025
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_clips, n_samples), dtype=torch.int16, device=device)
    n_notes = (n_samples + note - 1) // note
    t = torch.arange(note, device=device, dtype=torch.float32) / sr
    ramp = torch.arange(note, device=device, dtype=torch.float32)
    fade = torch.clamp(torch.minimum(ramp, note - 1 - ramp) / (0.010 * sr), max=1.0)
    for c0 in range(0, n_clips, chunk):
        nc = min(chunk, n_clips - c0)
        f = 130.81 * (4186.01 / 130.81) ** torch.rand((nc, n_notes, 6, 1), generator=g, device=device)
        a = 0.05 + 0.15 * torch.rand((nc, n_notes, 6, 1), generator=g, device=device)
        ph = 6.2831853 * torch.rand((nc, n_notes, 6, 1), generator=g, device=device)
        x = (a * torch.sin(6.2831853 * f * t + ph)).sum(2) * fade          # [nc][notes][note]
        x = x.reshape(nc, n_notes * note)[:, :n_samples]
        x = x + 0.0316 * torch.randn((nc, n_samples), generator=g, device=device)
        x = x / torch.clamp(x.abs().amax(dim=1, keepdim=True), min=1.0)
        out[c0:c0 + nc] = torch.round(x * 32767).to(torch.int16)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=1000, help="30 s clips per GPU (configs[1]: 1000)")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--batch", type=int, default=0, help="clips per internal pass (0 = library default)")
    ap.add_argument("--index-clips", type=int, default=10000, help="search: indexed clips per GPU")
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the streaming-latency section (configs[4])")
    ap.add_argument("--no-learn", action="store_true", help="skip the filter-learning section (index() only)")
    ap.add_argument("--learn-clips", type=int, default=256)
    ap.add_argument("--stream-clips", type=int, default=125000, help="stream: indexed clips per GPU (1 M / 8)")
    ap.add_argument("--stream-queries", type=int, default=192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the oracle altogether (profiling runs: nothing but the product in the process)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) section")
    ap.add_argument("--no-ffi", action="store_true",
                    help="skip the file-level section (the reference's FFI entry points over WAV files on tmpfs)")
    ap.add_argument("--ffi-files", type=int, default=1000)
    ap.add_argument("--no-f32-chain", action="store_true",
                    help="skip the pass with the f32-chain projection and its bit comparison with the fixed-point default")
    ap.add_argument("--no-any-length", action="store_true",
                    help="skip the section on clips whose length is not 7-smooth (chirp-z forward transform)")
    ap.add_argument("--cpu-clips-per-core", type=int, default=64,
                    help="CPU baseline sample: clips per usable host CPU (about 10 s of CPU work on the 16 CPUs of a GPU box: all 1000 clips)")
    args = ap.parse_args()

    import torch
    import hpfw_amd
    from hpfw_amd import dist as hdist, synth

    rank, local_rank, world = hdist.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # HPFW_BENCH_REHEARSE_ONE_GPU=1: every rank on cuda:0 with gloo and CPU-side collectives -- only to
    # rehearse the N > 1 code path on a one-GPU box; the driver's runs use one GPU per rank and RCCL.
    rehearse = bool(os.environ.get("HPFW_BENCH_REHEARSE_ONE_GPU")) and world > 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as tdist
        if rehearse:
            tdist.init_process_group("gloo")
        else:
            tdist.init_process_group("nccl", device_id=device)

    def barrier():
        if world > 1:
            tdist.barrier()

    ranks_seen, devices_seen = [rank], [dev_index]
    if world > 1:                                            # who is here: one all-gather of (rank, local device)
        me = torch.tensor([rank, dev_index], dtype=torch.int64, device="cpu" if rehearse else device)
        got = [torch.empty_like(me) for _ in range(world)]
        tdist.all_gather(got, me)
        ranks_seen = sorted(int(g[0].item()) for g in got)
        devices_seen = [int(g[1].item()) for g in got]

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else device)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        return float(t.item())

    gpu = hpfw_amd.Gpu(dev_index)
    filt = synth.make_filters()
    gpu.set_filters(filt)
    if args.batch:
        gpu.set_batch(args.batch)
    n_samples = int(round(args.seconds * 44100))
    geo = gpu.geometry(n_samples)
    n_clips = args.clips
    stream = torch.cuda.current_stream().cuda_stream

    pcm = synth_clips_gpu(torch, n_clips, n_samples, 0x68706677 + rank, device)
    # the first clips come from the generator the tests use (hpfw_amd.synth.gen_clip);
    # clip 0's SHA-256 is committed in tests/golden/clip0.sha256 and checked below
    n_spec = min(SPEC_CLIPS, n_clips)
    spec = np.stack([synth.gen_clip(rank * SPEC_CLIPS + i, args.seconds) for i in range(n_spec)])
    pcm[:n_spec] = torch.from_numpy(spec).to(device)
    clip0_sha = hashlib.sha256(spec[0].tobytes()).hexdigest() if rank == 0 else None
    if rank == 0 and n_samples == 1323000:
        committed = open(os.path.join(ROOT, "tests", "golden", "clip0.sha256")).read().split()[0]
        assert clip0_sha == committed, "bench clip 0 is not the committed synthetic clip"
    hp = torch.empty((n_clips, geo.n_hp), dtype=torch.int64, device=device)
    torch.cuda.synchronize()

    def step():
        gpu.extract_dev(pcm.data_ptr(), n_samples, n_clips, hp.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # the two largest kernels (the row transforms and the projection): one HIP event pair per launch, on the launch stream
    # (the forward transform's stages run in chunks of 16 clips: an event pair around each of their 126 launches per pass
    # would cost the step 0.2 ms; they are timed as one span here and launch by launch in an extra pass further down)
    gpu.set_kernel_timing((1 << hpfw_amd.KERNEL_KINDS.index("project_mfma")) | (1 << hpfw_amd.KERNEL_KINDS.index("fwd_span")))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu.timer_start(stream)
    for _ in range(args.steps):
        step()
    ev_ms = gpu.timer_stop(stream)
    torch.cuda.synchronize()
    barrier()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local)
    per_rank_ms = [dt_local * 1e3 / args.steps]
    if world > 1:                                            # every rank's own time for the same per-GPU work as the N = 1 run
        mine = torch.tensor([dt_local * 1e3 / args.steps], dtype=torch.float64, device="cpu" if rehearse else device)
        allt = [torch.empty_like(mine) for _ in range(world)]
        tdist.all_gather(allt, mine)
        per_rank_ms = [float(t.item()) for t in allt]
    kt = gpu.kernel_timing()
    gpu.set_kernel_timing(0)
    # one extra (untimed) pass with an event pair around every launch: the per-kernel split, and the per-launch figures
    # of the forward transform's two kernels
    gpu.set_kernel_timing(-1)
    step()
    torch.cuda.synchronize()
    kt_extra = gpu.kernel_timing()
    split = {k: round(v[0], 3) for k, v in kt_extra.items() if v[1]}
    gpu.set_kernel_timing(0)
    ms_per_step = dt * 1e3 / args.steps
    hashprints = float(n_clips) * geo.n_hp * world * args.steps
    value = hashprints / dt

    pj_ms, pj_launches = kt["project_mfma"]
    clips_per_launch = n_clips * args.steps / max(pj_launches, 1)
    fixed_point = gpu.get_projection() == 1
    traffic_json = {}
    tr_path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_path):
        try:
            traffic_json = json.load(open(tr_path))
        except Exception:
            traffic_json = {}

    def pmc_traffic(key):                              # PMC bytes per clip (profiles/), per launch here
        per_clip = traffic_json.get(key)
        return per_clip * clips_per_launch if per_clip is not None else None

    if fixed_point:
        # work the kernel's instructions perform: 64 x 2420 x n_hp multiply-adds (the reference forms n_frames columns and
        # subtracts afterwards; here the difference comes first), each as nine int8 digit products (S9q), 2 operations each
        pj_s = pj_ms / max(pj_launches, 1) * 1e-3
        macs_per_clip = PROJECT_Q_MACS_PER_CLIP * (geo.n_hp / 2320.0)
        ops_per_clip = 2.0 * 9 * macs_per_clip
        achieved = ops_per_clip * clips_per_launch / pj_s / 1e12 if pj_launches else 0.0
        algorithmic = 2.0 * macs_per_clip * clips_per_launch / pj_s / 1e12 if pj_launches else 0.0
        roof_pj = {"kernel": "hashprint_q_kernel (v_mfma_i32_16x16x64_i8, nine digit products of 24-bit fixed-point factors)",
                   "bound": "mfma", "achieved": round(achieved, 1), "peak": round(MFMA_I8_PEAK_TOPS, 1), "unit": "TFLOP/s",
                   "unit_note": "integer work: tera int8 operations (2 per multiply-add) per second, TOP/s; 'achieved' counts the "
                                "nine digit products the fixed-point decomposition performs per reference multiply-add",
                   "frac": round(achieved / MFMA_I8_PEAK_TOPS, 4),
                   "frac_of_probe_ceiling": round(achieved / MFMA_I8_PROBE_TOPS, 4),
                   "probe_ceiling_tops": MFMA_I8_PROBE_TOPS,
                   "frac_algorithmic": round(algorithmic / MFMA_I8_PEAK_TOPS, 4),
                   "algorithmic_tops": round(algorithmic, 1),
                   "traffic": pmc_traffic("project_q_hbm_bytes_per_clip"),
                   "avg_launch_ms": round(pj_ms / max(pj_launches, 1), 4), "launches": pj_launches,
                   "ops_per_clip": ops_per_clip, "clips_per_launch": clips_per_launch,
                   "note": "frac_algorithmic counts one multiply-add per reference multiply-add (2 x 64 x 2420 x n_hp operations per "
                           "clip); probe_ceiling = tools/mfma_i8_probe.hip back to back on random operands (the clock the chip holds "
                           "under this instruction, profiles/r03_mfma_i8_probe.jsonl)"}
    else:
        achieved = PROJECT_FLOP_PER_CLIP * (geo.n_frames / 2400.0) * clips_per_launch / (pj_ms / max(pj_launches, 1) * 1e-3) / 1e12 \
            if pj_launches else 0.0
        roof_pj = {"kernel": "project_kernel (v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                   "achieved": round(achieved, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic("project_mfma_hbm_bytes_per_clip"),
                   "avg_launch_ms": round(pj_ms / max(pj_launches, 1), 4), "launches": pj_launches,
                   "flop_per_clip": PROJECT_FLOP_PER_CLIP * (geo.n_frames / 2400.0),
                   "clips_per_launch": clips_per_launch}
    # the forward transform's two kernels (7-smooth lengths), both bound by HBM on paper:
    #  column stage: int16 samples in, the rounded integer sums z out (planar f32, hq rows of n2 complex);
    #  row stage: z in (+ every fourth inter-stage twiddle), the consumed bins out
    hq, q2w = geo.n1 // 2 + 1, (geo.kmax - 1) // geo.n1 - geo.kmin // geo.n1 + 1

    def hbm_roof(kind, name, bytes_per_clip, key, note):
        ms, launches = kt_extra[kind]            # the extra pass: one pass over the rank's clips
        clips = n_clips / max(launches, 1)
        gbs = bytes_per_clip * clips / (ms / max(launches, 1) * 1e-3) / 1e9 if launches else 0.0
        return {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": (traffic_json.get(key) * clips if traffic_json.get(key) is not None else None),
                "avg_launch_ms": round(ms / max(launches, 1), 4), "launches": launches,
                "bytes_per_clip": bytes_per_clip, "clips_per_launch": clips, "note": note}

    cols_bytes = geo.n1 * geo.n2 * 2 + hq * geo.n2 * 8
    rows_bytes = hq * geo.n2 * 8 + hq * ((geo.n2 + 3) // 4) * 8 + geo.n1 * q2w * 8
    roof_cols = hbm_roof("fwd_cols", "fwd_cols_q3_kernel (column DFT of the sample matrix as six int8 digit products on "
                         "v_mfma_i32_32x32x32_i8, exact; one rounding; sample digits in registers, twiddle digits by "
                         "global_load_lds)", cols_bytes, "fwd_cols_hbm_bytes_per_clip",
                         f"also {6 * 2 * 2 * hq * geo.n1 * geo.n2 / 1e9:.2f} G int8 operations per clip on the matrix pipe "
                         "(34 % of its cycles busy: profiles/r04_sq.json); the twiddle digits (147 KB per 128 columns) come "
                         "from L2; the kernels of a step share one power budget -- a slower, cooler variant of this kernel "
                         "(2.9 ms) left the step time unchanged within 0.6 % (DESIGN.md section 9)")
    roof_rows = hbm_roof("fwd_rows", "fwd_rows2_kernel (per row: inter-stage twiddles, FFT_n2 in LDS, pruned stores)", rows_bytes,
                         "fwd_rows_hbm_bytes_per_clip",
                         "not HBM-bound: three fused LDS passes per row with a barrier each, 63 % of wave-cycles parked "
                         "(profiles/r04_sq.json); three workgroups per CU; twiddles of every fused group fetched before the barrier in front of it")
    # the two stages of the forward transform run in chunks of a few clips on two streams (the
    # column stage's output stays in the Infinity Cache), so their launches overlap and the stage is timed as ONE span on
    # the caller's stream, like the chirp-z classes; the per-launch figures above are durations of launches that share the chip
    span_ms, span_n = kt.get("fwd_span", (0.0, 0))
    chunked = kt_extra["fwd_cols"][1] > kt_extra["fwd_span"][1] > 0
    roof_fwd = None
    if span_n:
        clips = n_clips * args.steps / span_n
        span_s = span_ms / span_n * 1e-3
        # what has to cross HBM whatever the kernels do between themselves: the PCM in, the consumed bins out, and the
        # inter-stage twiddle seeds the row stage reads (the same for every clip, so mostly served by the caches -- counted
        # all the same); the column stage's output z (5.3 MB per clip written and read back) is the stage's OWN traffic
        compulsory = geo.n1 * geo.n2 * 2 + hq * ((geo.n2 + 3) // 4) * 8 + geo.n1 * q2w * 8
        gbs = compulsory * clips / span_s / 1e9
        own_gbs = (cols_bytes + rows_bytes) * clips / span_s / 1e9
        tr = [traffic_json.get("fwd_cols_hbm_bytes_per_clip"), traffic_json.get("fwd_rows_hbm_bytes_per_clip")]
        roof_fwd = {"kernel": "forward transform (the stage with the most time per step) as a span: fwd_cols_q3_kernel + fwd_rows2_kernel" +
                              (f", {kt_extra['fwd_cols'][1]} chunks per {n_clips} clips in turn on two streams" if chunked else ""),
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": (sum(tr) * clips if None not in tr else None),
                    "avg_launch_ms": round(span_ms / span_n, 4), "launches": span_n,
                    "bytes_per_clip": compulsory, "clips_per_launch": clips,
                    "stage_own_bytes": {"bytes_per_clip": cols_bytes + rows_bytes, "achieved": round(own_gbs, 1),
                                        "frac": round(own_gbs / HBM_PEAK_GBS, 4),
                                        "note": "both kernels' own algorithmic bytes, i.e. with the 5.3 MB per clip of column-stage "
                                                "output written and read back (chunked, it is read back out of the Infinity Cache)"},
                    "note": "'launch' = one span (a HIP event pair on the caller's stream around fork and join of the chunked "
                            "stage); achieved = HBM-compulsory bytes (PCM in, twiddle seeds, consumed bins out) over the span; "
                            "traffic: the two kernels' PMC counters from whole-batch launches (HPFW_FWD_CHUNK=0), an upper "
                            "bound for the chunked run"}
        if chunked:
            for r in (roof_cols, roof_rows):
                r["note"] = "launches of one chunk in an extra pass after the timed region, overlapping with the other stream's: " + r["note"]
    # `roofline`: the stage that takes the most time of a step (the forward transform), on the bytes that must cross HBM;
    # then the largest single kernel (hashprints from dB terms, on the int8 matrix pipe) and the forward transform's two kernels
    if roof_fwd is not None:
        roofline, roofline_second, roofline_third, roofline_fourth = roof_fwd, roof_pj, roof_cols, roof_rows
    else:
        roofline, roofline_second, roofline_third = sorted([roof_pj, roof_cols, roof_rows], key=lambda r: -r["avg_launch_ms"])
        roofline_fourth = None


    # parity and the CPU baseline (checker / reported baseline only, outside any timing): the oracle extracts a
    # bounded sample of the same clips on the host cores and EVERY hashprint it produced is compared
    parity = None
    cpu_baseline = None
    if rank == 0 and not args.no_parity:
        from oracle import oracle
        plan = oracle.Plan(n_samples)
        got_all = hp.cpu().numpy().view(np.uint64)
        if world == 1 and not args.no_cpu_baseline:
            from hpfw_amd import hostinfo
            budget = hostinfo.cpu_budget()                   # affinity and cgroup quota, not the machine's core count
            ref_libs = hostinfo.reference_cpu_libraries()    # FFTW3 / Eigen3 / TBB / essentia / MKL: looked for, not assumed
            cores = budget["usable"]
            n_cpu = min(n_clips, max(1, args.cpu_clips_per_core * cores))
            sample = pcm[:n_cpu].cpu().numpy()
            t1 = time.perf_counter()
            want = plan.extract_batch(filt, sample, n_threads=cores)
            cdt = time.perf_counter() - t1
            idx = np.arange(n_cpu)
            # one thread, stage by stage (4 clips): where the CPU path spends its time
            st = {"spectrum": 0.0, "cq": 0.0, "db": 0.0, "project": 0.0, "pack": 0.0}
            n_one = min(4, n_cpu)
            t_one = time.perf_counter()
            for c in sample[:n_one]:
                ta = time.perf_counter(); x = plan.spectrum(c)
                tb = time.perf_counter(); m = plan.cqmag(x)
                tc = time.perf_counter(); d = oracle.db(m)
                td = time.perf_counter(); pr = oracle.project_q(filt, d) if oracle.get_projection() else oracle.project(filt, d)
                te = time.perf_counter(); oracle.pack_q(pr) if oracle.get_projection() else oracle.pack(pr)
                tf = time.perf_counter()
                for key, dtk in zip(st, (tb - ta, tc - tb, td - tc, te - td, tf - te)):
                    st[key] += dtk
            one_dt = time.perf_counter() - t_one
            scaling = (n_cpu / cdt) / (n_one / one_dt)
            cpu_baseline = {"value": round(n_cpu * geo.n_hp / cdt, 1), "unit": "hashprints/s", "cores": cores,
                            "kind": "port",
                            "sample": f"{n_cpu} of the same {args.seconds:g} s clips, oracle/hpfw_oracle.c "
                                      f"(own C restatement, -O3 -mfma), {cores} threads = the CPUs this process may use "
                                      f"(machine: {budget['os_cpu_count']}, cgroup quota: {budget['cgroup_cpus']}) in static "
                                      f"chunks as flow_builder.hpp:321-325, {cdt:.1f} s",
                            "clips_per_s": round(n_cpu / cdt, 2),
                            "host": budget,
                            "reference_cpu_libraries": ref_libs,

                            "threads_over_one_thread": round(scaling, 2),
                            "one_thread": {"clips_per_s": round(n_one / one_dt, 2), "clips": n_one,
                                           "ms_per_clip_by_stage": {k: round(v * 1e3 / n_one, 1) for k, v in st.items()}}}
        else:
            # (N > 1, or no CPU baseline asked for: 32 clips spread over the batch -- with several ranks on a box the kernels of
            # one run beside the other's, which is where round 3's packed arithmetic went wrong)
            idx = np.unique(np.linspace(0, n_clips - 1, min(32, n_clips)).astype(np.int64))
            from hpfw_amd import hostinfo
            threads = max(1, hostinfo.cpu_budget()["usable"] // max(world, 1))   # (the cgroup's CPUs, not the machine's)
            want = plan.extract_batch(filt, pcm[idx].cpu().numpy(), n_threads=threads)
        parity = {"clips_checked": int(len(idx)), "bit_identical": bool(np.array_equal(got_all[idx], want)),
                  "hashprints_differing": int((got_all[idx] != want).sum())}

    # the reference's arithmetic type for filters * frames is f32 (parallel_collector.h:57,127): the same pass with the f32
    # fma chain (S9) instead of the fixed-point default (S9q), and every hashprint of the batch compared between the two
    f32_chain = None
    if fixed_point and not args.no_f32_chain:
        hp32 = torch.empty_like(hp)
        gpu.set_projection(0)
        try:
            def step32():
                gpu.extract_dev(pcm.data_ptr(), n_samples, n_clips, hp32.data_ptr(), stream)
            step32()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                step32()
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / 2
        finally:
            gpu.set_projection(1)
        x = torch.bitwise_xor(hp, hp32)
        words = int((x != 0).sum().item())
        bits = 0
        for sh in range(0, 64, 16):                          # popcount by 16-bit table-free folding on the device
            v = (x >> sh) & 0xFFFF
            v = (v & 0x5555) + ((v >> 1) & 0x5555)
            v = (v & 0x3333) + ((v >> 2) & 0x3333)
            v = (v & 0x0F0F) + ((v >> 4) & 0x0F0F)
            v = (v & 0x00FF) + ((v >> 8) & 0x00FF)
            bits += int(v.sum().item())
        f32_chain = {"ms_per_step": round(dt32 * 1e3, 3), "clips_per_s": round(n_clips / dt32, 1),
                     "hashprints_per_s": round(n_clips * geo.n_hp / dt32, 1),
                     "hashprints_differing_vs_fixed_point": words, "bits_differing_vs_fixed_point": bits,
                     "hashprints_compared": int(hp.numel()), "bits_compared": int(hp.numel()) * 64,
                     "note": "hpfw_gpu_set_projection(h, 0): filters * frames as the f32 fma chain on v_mfma_f32_32x32x2_f32, "
                             "rank 0's batch, outside the timed region"}
        del hp32, x

    pcie = None
    if rank == 0 and world == 1 and not args.no_pcie:
        pcie = bench_pcie(torch, gpu, pcm, n_samples, geo, hp)
        if hasattr(torch._C, "_host_emptyCache"):
            torch._C._host_emptyCache()        # the section's 0.7 GB of pinned host memory go back (torch caches them)

    any_len = None
    if rank == 0 and world == 1 and not args.no_any_length:
        any_len = bench_any_length(torch, gpu, pcm, n_samples, None if args.no_parity else __import__("oracle.oracle").oracle, filt)

    ffi = None
    if rank == 0 and world == 1 and not args.no_ffi:
        ffi = bench_ffi(torch, pcm, n_samples, filt, hp, args.ffi_files)

    search = None
    if not args.no_search:
        search = bench_search(torch, tdist if world > 1 else None, gpu, hdist, synth, args, rank, world, device,
                              stream, barrier, max_over_ranks, rehearse)

    learn = None
    if not args.no_learn and rank == 0 and world == 1:
        learn = bench_learn(torch, gpu, args, pcm, n_samples, stream, filt)

    stream_res = None
    if not args.no_search and not args.no_stream:
        stream_res = bench_stream(torch, tdist if world > 1 else None, gpu, hdist, synth, args, rank, world, device,
                                  stream, barrier, rehearse)

    if rank == 0:
        comm = None
        if world > 1:
            comm = {"backend": tdist.get_backend(), "world_size": tdist.get_world_size(), "ranks_seen": ranks_seen,
                    "devices_seen": devices_seen,
                    "note": "backend 'nccl' is RCCL on ROCm; one process and one GPU per rank; ranks_seen / devices_seen come from "
                            "an all-gather of every rank's (rank, local device) at start-up"}
        line = {
            "metric": "hashprints/sec (index) + Hamming matches/sec (search), 30 s@44.1 kHz clips",
            "value": round(value, 1), "unit": "hashprints/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (transforms, dB) + int8 digits / int64 sums (projection)" if fixed_point else "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {n_clips} x {args.seconds:g} s synthetic 44.1 kHz PCM16 clips per GPU, "
                                   "hashprint extraction only (CQT + dB + projection + bit pack), inputs resident in HBM",
                       "clips_per_gpu": n_clips, "clip_seconds": args.seconds, "hashprints_per_clip": geo.n_hp,
                       "generator": f"clips 0..{n_spec - 1} of every rank: hpfw_amd.synth.gen_clip (the tests' generator); "
                                    "the rest: the same recipe with the device RNG",
                       "clip0_sha256": clip0_sha,
                       "parallelism": f"clips sharded over {world} GPU(s), no collective on this path"},
            "clips_per_s": round(value / geo.n_hp, 1),
            "event_ms_per_step_rank0": round(ev_ms / args.steps, 3),
            "per_rank_ms_per_step": [round(t, 3) for t in per_rank_ms],
            "per_gpu_work": {"clips": n_clips, "hashprints": n_clips * geo.n_hp,
                             "note": "weak scaling: what every rank does per step is what the N = 1 run does"},
            "kernel_ms_one_pass": split,
            "roofline": roofline, "roofline_second_kernel": roofline_second, "roofline_third_kernel": roofline_third,
            "roofline_fourth_kernel": roofline_fourth,
            "cpu_baseline": cpu_baseline, "parity": parity,
            "projection_f32_chain": f32_chain,
            "pcie_inclusive": pcie, "ffi": ffi, "any_length": any_len, "search": search,
            "stream": stream_res, "filter_learning": learn, "rccl": comm,
        }
        print(json.dumps(line), flush=True)
    gpu.close()
    if world > 1:
        tdist.destroy_process_group()


def exchange_hits(tdist, hits, gathered, world, cpu_collectives=False):
    """the search's one exchange step (storage.h:56-60's global arg-min made a top-k merge): every
    rank's per-shard top-k list, Q x k x 16 bytes, to every rank.  RCCL (backend "nccl") on device tensors; the gloo
    rehearsal and the CPU test pass host tensors."""
    if world <= 1:
        return
    if cpu_collectives:
        tdist.all_gather([gathered[i] for i in range(world)], hits.cpu())
    else:
        tdist.all_gather_into_tensor(gathered, hits)          # RCCL over xGMI: Q x k x 16 B per rank


def merged_hits(hits, gathered, world, nq, topk):
    """the per-shard lists -> the global top-k, the same deterministic merge by (dist, clip) on every rank"""
    import hpfw_amd
    res = hits.cpu().numpy().reshape(nq, topk * 4).view(hpfw_amd.HIT_DTYPE).reshape(nq, topk)
    if world > 1:
        per = gathered.cpu().numpy().reshape(world, nq, topk * 4).view(hpfw_amd.HIT_DTYPE).reshape(world, nq, topk)
        res = hpfw_amd.merge_topk(per, topk)
    return res


def planted_found(res, nq, n_local, n_hp, kq):
    """query i was cut from clip i mod n_local of shard 0 at offset 37 i mod (n_hp - kq + 1)"""
    return bool((res[:, 0]["clip"] == (np.arange(nq) % n_local)).all()
                and (res[:, 0]["offset"] == ((np.arange(nq) * 37) % (n_hp - kq + 1))).all())


def bench_search(torch, tdist, gpu, hdist, synth, args, rank, world, device, stream, barrier, max_over_ranks,
                 rehearse=False):
    """configs[2] per GPU (configs[3] layout when world > 1): index shard resident in HBM, replicated
    queries, scan + per-shard top-k on every rank, one all-gather of Q x k x 16 B, identical merge."""
    import hpfw_amd
    n_hp = 2320
    kq = int(gpu.geometry(5 * 44100).n_hp)            # what extraction yields for a 5 s query: 304 (C = ceil(M / 3))
    n_local = args.index_clips
    g = torch.Generator(device=device)
    g.manual_seed(0x1D8 + rank)
    db = torch.randint(-2 ** 63, 2 ** 63 - 1, (n_local, n_hp), dtype=torch.int64, generator=g, device=device)
    # queries are planted in shard 0's clips (every rank regenerates shard 0's first rows the same way)
    g0 = torch.Generator(device=device)
    g0.manual_seed(0x1D8)
    db0 = db if rank == 0 else torch.randint(-2 ** 63, 2 ** 63 - 1, (n_local, n_hp), dtype=torch.int64,
                                              generator=g0, device=device)
    nq = args.queries
    src = torch.arange(nq, device=device) % n_local
    offs = (torch.arange(nq, device=device) * 37) % (n_hp - kq + 1)
    cols = offs[:, None] + torch.arange(kq, device=device)[None, :]
    q = db0[src[:, None], cols].clone()
    gq = torch.Generator(device=device)
    gq.manual_seed(0x51)
    for _ in range(6):   # six random bit flips per hashprint
        q ^= torch.ones_like(q) << torch.randint(0, 63, q.shape, generator=gq, device=device)
    del db0
    lo = rank * n_local
    gpu.index_clear()
    gpu.index_set_clip_base(lo)
    gpu.index_add_dev(db.data_ptr(), np.arange(0, (n_local + 1) * n_hp, n_hp, dtype=np.int64), stream)
    q_off = np.arange(0, (nq + 1) * kq, kq, dtype=np.int64)
    hits = torch.empty((nq, args.topk, 4), dtype=torch.int32, device=device)
    gathered = torch.empty((world, nq, args.topk, 4), dtype=torch.int32,
                           device="cpu" if rehearse else device) if world > 1 else None

    def one():
        gpu.search_topk_dev(q.data_ptr(), q_off, args.topk, hits.data_ptr(), stream)
        exchange_hits(tdist, hits, gathered, world, rehearse)

    one()
    torch.cuda.synchronize()
    gpu.set_kernel_timing(1 << hpfw_amd.KERNEL_KINDS.index("hamming_scan"))
    reps = 2
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    kt = gpu.kernel_timing()
    gpu.set_kernel_timing(0)
    pairs = float(nq) * n_local * world * kq * (n_hp - kq + 1) * reps
    res = merged_hits(hits, gathered, world, nq, args.topk)
    ok = planted_found(res, nq, n_local, n_hp, kq)
    scan_ms, scan_l = kt["hamming_scan"]
    pairs_rank_launch = float(nq) * n_local * kq * (n_hp - kq + 1) * reps / max(scan_l, 1)
    scan_rate = pairs_rank_launch / (scan_ms / max(scan_l, 1) * 1e-3) if scan_l else 0.0
    return {"value": round(pairs / dt, 1), "unit": "hashprint-pair Hamming matches/s",
            "workload": f"configs[2] per GPU: {n_local}-clip index ({n_hp} hashprints each) resident in HBM, "
                        f"{nq} x {kq}-hashprint queries, exhaustive sliding scan, top-{args.topk}"
                        + (f", all-gather of per-shard top-k over {world} ranks (RCCL)" if world > 1 else ""),
            "ms_per_search": round(dt * 1e3 / reps, 3), "queries_per_s": round(nq * reps / dt, 1),
            "planted_queries_found": ok,
            "scan_kernel": _scan_roofline(scan_rate)}


def bench_any_length(torch, gpu, pcm, n_samples, oracle_mod, filt, n=256):
    """the same clips one sample longer: a length with a prime factor above 7, which takes the chirp-z forward transform
    (real recordings have whatever length they have).  Reports the first use of the length (its tables are generated on
    the device) and the steady state; two clips against the oracle.  Never `value`."""
    n = min(n, pcm.shape[0] - 1)
    m = n_samples + 1
    flat = pcm.reshape(-1)
    odd = flat[: n * m].reshape(n, m)           # n clips of m samples cut from the same signal
    geo = gpu.geometry(m)                       # sizes only (host): no table is built for the question
    hp = torch.empty((n, geo.n_hp), dtype=torch.int64, device=pcm.device)
    one = torch.empty((1, geo.n_hp), dtype=torch.int64, device=pcm.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu.extract_dev(odd.data_ptr(), m, 1, one.data_ptr())   # the length's first use: host tables of the constant-Q stage,
    torch.cuda.synchronize()                                # device tables of the chirp-z transform, one clip extracted
    first_ms = (time.perf_counter() - t0) * 1e3
    gpu.extract_dev(odd.data_ptr(), m, n, hp.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        gpu.extract_dev(odd.data_ptr(), m, n, hp.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out = {"clip_samples": m, "n1": geo.n1, "n2": geo.n2, "clips": n, "clips_per_s": round(n / dt, 1),
           "hashprints_per_s": round(n * geo.n_hp / dt, 1), "first_use_ms": round(first_ms, 2),
           "note": "clip length with a prime factor above 7: chirp-z forward transform on the exact length (no padding)"}
    if oracle_mod is not None:                  # every clip of the pass against the oracle (all host cores, a few seconds)
        plan = oracle_mod.Plan(m)
        want = plan.extract_batch(filt, odd.cpu().numpy(), n_threads=os.cpu_count() or 1)
        got = hp.cpu().numpy().view(np.uint64)
        out["clips_checked"] = n
        out["bit_identical"] = bool(np.array_equal(got, want))
        out["hashprints_differing"] = int((got != want).sum())
    return out


def bench_ffi(torch, pcm, n_samples, filt, hp_dev, n_files):
    """The reference's own boundary takes FILES (parallel_collector_wrapper.hpp:25-30, parallel_collector.h:48-59,82-137):
    par_collector_calc_hashprints and par_collector_prepare (filters kept: no learning; the spectrogram cache written as
    the reference writes it) over WAV files on tmpfs, once with equal lengths and once with a different length in every
    file -- the shape of a directory of real tracks, where every file brings a new clip length and its tables.  Never `value`."""
    import shutil
    import tempfile
    import hpfw_amd
    from hpfw_amd import synth
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="hpfw_ffi_", dir=base)
    out = {"files": n_files, "where": "tmpfs" if base else "tmp", "clip_samples": n_samples}
    try:
        n_src = min(64, pcm.shape[0])
        src = pcm[:n_src].cpu().numpy()
        want = hp_dev[:n_src].cpu().numpy().view(np.uint64)
        cache = os.path.join(d, "cache") + "/"
        os.makedirs(cache)
        with open(os.path.join(cache, "filters.cereal"), "wb") as f:   # cereal image of Filters (utils.h:84-90)
            f.write(np.array([64, 2420], np.int32).tobytes() + np.ascontiguousarray(filt, np.float32).tobytes())
        pc = hpfw_amd.ParallelCollector()
        pc.load(cache)
        os.environ["HPFW_PREPARE_KEEP_FILTERS"] = "1"
        for label, lengths in (("equal_lengths", [n_samples] * n_files),
                               ("distinct_lengths", [n_samples - 3 * i for i in range(n_files)])):
            sub = os.path.join(d, label)
            os.makedirs(sub)
            paths = []
            for i, n in enumerate(lengths):
                p = os.path.join(sub, f"t{i:05d}.wav")
                synth.write_wav(p, src[i % n_src][:n])
                paths.append(p)
            res = {}
            pc.calc_hashprints(paths[:256])                               # first touch: the pinned arena of a full window, the tables
            # twice: the first pass is the first read of the files since they were written (on this pool's VMs that read
            # runs at a tenth of the rate of any later one: 50-90 ms against 7 ms per 677 MB window), the second finds
            # them as a corpus that has been read before does
            for key in ("calc_hashprints_first_read", "calc_hashprints"):
                t0 = time.perf_counter()
                got = pc.calc_hashprints(paths)
                dt = time.perf_counter() - t0
                ok = sum(1 for a, _ in got if a is not None)
                res[key + "_files_per_s"] = round(ok / dt, 1)
                res[key + "_s"] = round(dt, 3)
            if label == "equal_lengths":
                res["hashprints_equal_to_the_device_path"] = bool(all(np.array_equal(got[i][0], want[i % n_src]) for i in range(ok)))
            t0 = time.perf_counter()
            got = pc.prepare(paths)
            dt = time.perf_counter() - t0
            res["prepare_files_per_s"] = round(len(got) / dt, 1)
            res["prepare_s"] = round(dt, 3)
            res["files_returned"] = len(got)
            out[label] = res
            shutil.rmtree(sub, ignore_errors=True)
            shutil.rmtree(os.path.join(cache, "spectros"), ignore_errors=True)
        out["note"] = ("par_collector_calc_hashprints / par_collector_prepare through ctypes (hpfw_amd.ParallelCollector, the twin of "
                       "pyhpfw.py); prepare with HPFW_PREPARE_KEEP_FILTERS=1 (no learning), spectrogram cache written to tmpfs; "
                       "distinct lengths: every file a different sample count (chirp-z forward transform, tables per length); "
                       "(until round 4 these rates came out 15-20 % lower behind the host-buffer section than alone: the tables of a "
                       "new length were copied synchronously on the default stream, which waited behind the collector's extraction "
                       "stream when the two shared a hardware queue; they now go on a stream of the handle's own from a pinned "
                       "ring -- tools/ffi_interaction.sh)")
    finally:
        os.environ.pop("HPFW_PREPARE_KEEP_FILTERS", None)
        shutil.rmtree(d, ignore_errors=True)
    return out


def bench_pcie(torch, gpu, pcm, n_samples, geo, hp_dev, n=256):
    """the same extraction handed HOST buffers (hpfw_gpu_extract_pcm16_host: pinned int16 in, hashprints out;
    uploads in chunks on a copy stream under the kernels of the previous chunk).  Never `value`."""
    n = min(n, pcm.shape[0])
    host = torch.empty((n, n_samples), dtype=torch.int16).pin_memory()
    host.copy_(pcm[:n])
    out = torch.empty((n, geo.n_hp), dtype=torch.int64).pin_memory()
    import ctypes
    L = __import__("hpfw_amd").lib()

    def run():
        rc = L.hpfw_gpu_extract_pcm16_host(gpu._h, ctypes.c_void_p(host.data_ptr()), n_samples, n,
                                           ctypes.c_void_p(out.data_ptr()))
        assert rc == 0, L.hpfw_gpu_last_error()

    run()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    same = bool(torch.equal(out, hp_dev[:n].cpu()))
    return {"clips_per_s": round(n / dt, 1), "hashprints_per_s": round(n * geo.n_hp / dt, 1),
            "pcm_gb_per_s": round(n * n_samples * 2 / dt / 1e9, 2), "clips": n, "same_hashprints": same,
            "note": "host pinned int16 PCM in, host hashprints out; bounded by the PCIe link (2.65 MB per 30 s clip)"}


def bench_learn(torch, gpu, args, pcm, n_samples, stream, filt):
    """index()-only work: frame covariance of the first clips of the
    batch accumulated on the GPU (front end included), then the eigen-solve on the host."""
    n = min(args.learn_clips, pcm.shape[0])
    gpu.cov_reset()
    gpu.cov_accumulate_dev(pcm.data_ptr(), n_samples, n, stream)
    torch.cuda.synchronize()
    gpu.cov_reset()
    t0 = time.perf_counter()
    gpu.cov_accumulate_dev(pcm.data_ptr(), n_samples, n, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    gpu.learn_filters()
    et = time.perf_counter() - t1
    gpu.set_filters(filt)                                    # back to the fixture for the sections that follow
    n_frames = gpu.geometry(n_samples).n_frames
    flop = 2.0 * 2420 * 2420 * n_frames                      # per clip, full matrix (28.1 GFLOP)
    return {"covariance_clips_per_s": round(n / dt, 1), "clips": n, "ms": round(dt * 1e3, 2),
            "equivalent_tflops_full_matrix": round(flop * n / dt / 1e12, 1),
            "eigen_solve_s": round(et, 2),
            "note": "front end + covariance kernels per clip; the eigen-solve runs once per index on the host"}


def bench_stream(torch, tdist, gpu, hdist, synth, args, rank, world, device, stream, barrier, rehearse=False):
    """configs[4] per GPU: an index shard of 1 M / 8 clips resident in HBM; 5 s query windows arrive one
    at a time (and in batches of 32, as 32 concurrent streams would deliver them); each goes
    PCM -> hashprints -> scan of the shard -> (N > 1: all-gather + merge) -> host.  Reported: end-to-end
    latency percentiles on rank 0 and the number of real-time 5 s streams that rate sustains."""
    import hpfw_amd
    n_hp, n_local, n_q = 2320, args.stream_clips, args.stream_queries
    q_samples = 5 * 44100
    geo = gpu.geometry(q_samples)
    g = torch.Generator(device=device)
    g.manual_seed(0x57E + rank)
    db = torch.randint(-2 ** 63, 2 ** 63 - 1, (n_local, n_hp), dtype=torch.int64, generator=g, device=device)
    # planted songs: 16 synthetic 30 s songs (the same on every rank) extracted here and stored at known global clip
    # ids, spread over the shards; every streamed window is a noisy 5 s slice of one of them, so each round's
    # top-1 (clip, offset) is known and checked
    n_songs, song_samples = 16, 30 * 44100
    songs = synth_clips_gpu(torch, n_songs, song_samples, 0x50f6, device)
    song_hp = torch.empty((n_songs, n_hp), dtype=torch.int64, device=device)
    gpu.extract_dev(songs.data_ptr(), song_samples, n_songs, song_hp.data_ptr(), stream)
    total = n_local * world
    song_clip = [(7 + i * (total // n_songs + 1)) % total for i in range(n_songs)]
    for i, gc in enumerate(song_clip):
        if gc // n_local == rank:
            db[gc % n_local] = song_hp[i]
    gpu.index_clear()
    gpu.index_set_clip_base(rank * n_local)
    gpu.index_add_dev(db.data_ptr(), np.arange(0, (n_local + 1) * n_hp, n_hp, dtype=np.int64), stream)
    torch.cuda.synchronize()
    del db
    gq = torch.Generator(device=device)
    gq.manual_seed(0x57F)                                                 # the same windows on every rank
    q_song = [i % n_songs for i in range(n_q)]
    q_start = [44100 * (1 + (i * 5) % 23) for i in range(n_q)]
    pcm = torch.empty((n_q, q_samples), dtype=torch.int16, device=device)
    for i in range(n_q):
        seg = 0.5 * songs[q_song[i], q_start[i]:q_start[i] + q_samples].to(torch.float32)
        noise = torch.randn(q_samples, generator=gq, device=device) * (seg.pow(2).mean().sqrt() * 10 ** (-10 / 20))
        pcm[i] = torch.clamp(torch.round(seg + noise), -32768, 32767).to(torch.int16)
    hop = 1323000 / 7255 * 3                                              # samples per spectrogram column
    want_clip = np.array([song_clip[s] for s in q_song])
    want_off = np.array(q_start) / hop
    out = {}
    for batch in (1, 32):
        nb = n_q // batch
        d_hp = torch.empty((batch, geo.n_hp), dtype=torch.int64, device=device)
        hits = torch.empty((batch, args.topk, 4), dtype=torch.int32, device=device)
        gathered = torch.empty((world, batch, args.topk, 4), dtype=torch.int32,
                               device="cpu" if rehearse else device) if world > 1 else None
        q_off = np.arange(0, (batch + 1) * geo.n_hp, geo.n_hp, dtype=np.int64)
        lat = []
        wrong = 0
        barrier()
        rounds = max(nb, 100)                  # a percentile wants a hundred rounds: the windows are taken again in turn
        for i in range(rounds + 2):
            first = i % nb * batch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gpu.extract_dev(pcm[first].data_ptr(), q_samples, batch, d_hp.data_ptr(), stream)
            gpu.search_topk_dev(d_hp.data_ptr(), q_off, args.topk, hits.data_ptr(), stream)
            if world > 1 and rehearse:
                tdist.all_gather([gathered[r] for r in range(world)], hits.cpu())
                res = gathered.numpy()
            elif world > 1:
                tdist.all_gather_into_tensor(gathered, hits)
                res = gathered.cpu().numpy()
            else:
                res = hits.cpu().numpy()
            res = res.reshape(-1, batch, args.topk * 4).view(hpfw_amd.HIT_DTYPE).reshape(-1, batch, args.topk)
            final = hpfw_amd.merge_topk(res, args.topk) if world > 1 else res[0]
            if i >= 2:                                                       # two warm-up rounds
                lat.append((time.perf_counter() - t0) * 1e3)
            top = final[:, 0]                                                 # outside the timed span: check the hits
            wrong += int(((top["clip"] != want_clip[first:first + batch]) |
                          (np.abs(top["offset"] - want_off[first:first + batch]) > 2)).sum())
        lat = np.sort(np.array(lat))
        qps = batch * 1e3 / float(lat.mean())
        out[f"batch_{batch}"] = {"latency_ms_p50": round(float(np.percentile(lat, 50)), 3),
                                 "latency_ms_p99": round(float(np.percentile(lat, 99)), 3),
                                 "queries_per_s": round(qps, 1), "realtime_5s_streams": int(qps * 5), "rounds": int(lat.size),
                                 "windows_checked": int((rounds + 2) * batch), "wrong_hits": wrong}
    out["planted_windows_found"] = all(v["wrong_hits"] == 0 for v in out.values() if isinstance(v, dict))
    out["workload"] = (f"configs[4] per GPU: {n_local}-clip index shard ({n_local * n_hp * 8 / 1e9:.2f} GB of hashprints) "
                       f"in HBM, 5 s PCM windows -> extraction -> top-{args.topk} scan"
                       + (f" -> all-gather over {world} ranks + merge" if world > 1 else "") + " -> host, measured on rank 0")
    return out


def _scan_roofline(scan_rate):
    if os.environ.get("HPFW_SEARCH_POPC"):
        return {"kernel": "hamming_scan_kernel (v_xor_b32 + v_bcnt_u32_b32)", "pairs_per_s": round(scan_rate, 1),
                "bound": "valu", "peak_pairs_per_s": VALU_PAIR_PEAK, "frac": round(scan_rate / VALU_PAIR_PEAK, 4),
                "note": "integer VALU-issue bound (2 v_xor + 2 v_bcnt per pair); HBM is not binding"}
    return {"kernel": "hamming_mfma_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, fp4 +-1, exact)",
            "pairs_per_s": round(scan_rate, 1), "bound": "mfma", "peak_pairs_per_s": round(FP4_PAIR_PEAK, 1),
            "frac": round(scan_rate / FP4_PAIR_PEAK, 4),
            "note": "one pair = a 64-term +-1 dot product; the xor/popcount formulation (HPFW_SEARCH_POPC=1) peaks at "
                    f"{VALU_PAIR_PEAK:.3g} pairs/s on the VALU; HBM is not binding"}


if __name__ == "__main__":
    main()
