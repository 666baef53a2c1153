#!/bin/bash
# Round profile (run on the GPU box): everything lands under gpurun_out/prof; condense afterwards with
#   python profiles/summarize.py <tag> gpurun_out/prof 1000
#  1. trace/   rocprofv3 --kernel-trace --stats of the default bench command (python3 bench.py)
#  2. fetch/, write/   FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel trace only)
#  3. sq1..sq5/   SQ counters (occupancy, VALU / MFMA / LDS activity, waits, bank conflicts) in separate passes
#  4. calib_fetch/, calib_write/   the counters on known byte counts at 2/4/8/16 B per lane (tools/fetch_calib.bin)
#  5. chirpz/   kernel trace of the chirp-z forward path (tools/chirpz_profile.py: 256 clips of 1323001 samples)
# The program after "--" is python3 / the binary itself: no env, no shell, nothing that re-execs after the
# profiler's library has initialised the GPU; the --pmc passes run with --no-parity so that the profiled process
# holds nothing but the product.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o p --output-format csv -- python3 bench.py > gpurun_out/prof/bench_trace.log 2> gpurun_out/prof/bench_trace.err || { echo "trace run failed"; tail -5 gpurun_out/prof/bench_trace.err; exit 1; }
echo "trace done"
LIGHT="--no-parity --no-cpu-baseline --no-pcie --no-any-length --no-learn --steps 2 --warmup 1 --batch 1000"
# the counter passes launch the two stages of the forward transform over the whole batch (per-clip figures = per launch /
# 1000 clips: one pass, --batch 1000); the trace above ran them as shipped, four passes, chunks of 16 clips on two streams
export HPFW_FWD_CHUNK=0
for c in fetch:FETCH_SIZE write:WRITE_SIZE \
         "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
         "sq2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "sq3:SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
         "sq4:SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
         "sq5:SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  d=${c%%:*}; n=${c#*:}
  rocprofv3 --kernel-trace --pmc $n -d gpurun_out/prof/$d -o p --output-format csv -- python3 bench.py $LIGHT > gpurun_out/prof/bench_$d.log 2> gpurun_out/prof/bench_$d.err || { echo "$d run failed"; tail -5 gpurun_out/prof/bench_$d.err; }
  echo "$d done"
done
unset HPFW_FWD_CHUNK
for c in calib_fetch:FETCH_SIZE calib_write:WRITE_SIZE; do
  d=${c%%:*}; n=${c#*:}
  rocprofv3 --kernel-trace --pmc $n -d gpurun_out/prof/$d -o p --output-format csv -- tools/fetch_calib.bin > gpurun_out/prof/$d.log 2>&1 || echo "$d failed"
  echo "$d done"
done
# the chirp-z forward path (clip lengths with a prime factor above 7): 256 clips of 1323001 samples, 5 passes
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/chirpz -o p --output-format csv -- python3 tools/chirpz_profile.py 1323001 256 5 > gpurun_out/prof/chirpz.log 2>&1 || echo "chirpz trace failed"
echo "chirpz done"
# condense on the box (the raw traces are far above what travels back), keep the summaries and the logs
TAG=${1:-r02}
python3 profiles/summarize.py $TAG gpurun_out/prof 1000 gpurun_out/profiles_$TAG > gpurun_out/profiles_$TAG.log 2>&1 || tail -5 gpurun_out/profiles_$TAG.log
cp gpurun_out/prof/trace/p_kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_rocprof_stats_raw.csv 2>/dev/null || find gpurun_out/prof/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/profiles_$TAG/${TAG}_rocprof_stats_raw.csv \;
grep -h "^{" gpurun_out/prof/bench_trace.log | tail -1 > gpurun_out/profiles_$TAG/${TAG}_bench_under_trace.json
find gpurun_out/prof/chirpz -name "*kernel_stats.csv" -exec cp {} gpurun_out/profiles_$TAG/${TAG}_chirpz_kernel_stats.csv \;
mkdir -p gpurun_out/prof_logs && cp gpurun_out/prof/*.log gpurun_out/prof/*.err gpurun_out/prof_logs/ 2>/dev/null
rm -rf gpurun_out/prof
ls -la gpurun_out/profiles_$TAG
