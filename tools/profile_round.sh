#!/bin/bash
# Round profile: kernel trace + stats of the default bench command, then FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (kernel trace only), all under gpurun_out/prof; summarise with
#   python profiles/summarize.py <tag> gpurun_out/prof 1000
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o p --output-format csv -- python3 bench.py > gpurun_out/prof/bench_trace.log 2> gpurun_out/prof/bench_trace.err || { echo "trace run failed"; tail -5 gpurun_out/prof/bench_trace.err; exit 1; }
echo "trace done"
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
  d=${c%%:*}; n=${c##*:}
  rocprofv3 --kernel-trace --pmc $n -d gpurun_out/prof/$d -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/prof/bench_$d.log 2> gpurun_out/prof/bench_$d.err || { echo "$d run failed"; tail -5 gpurun_out/prof/bench_$d.err; exit 1; }
  echo "$d done"
done
# keep only what the summariser reads (the raw traces are large)
find gpurun_out/prof/fetch gpurun_out/prof/write -name "*_kernel_trace.csv" -delete
ls -R gpurun_out/prof | head -40
