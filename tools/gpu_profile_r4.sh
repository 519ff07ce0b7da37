#!/usr/bin/env bash
# Round-4 profile collection in ONE gpurun call: kernel-trace stats of the default bench command (the legs the driver's line now carries:
# headline, Q=256, c2, c5, hamming, shard, small), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the headline and the
# hamming launch, the latency map with the round-4 flavours on / off, the host-path breakdown, the odd-width timings, the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
guard() { if [ "$1" -ge 124 ]; then echo "step killed rc=$1"; exit "$1"; fi; }
echo "=== kernel trace: bench ($(date +%T))"; rm -rf $O/prof_r4
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_r4 --output-format csv -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/prof_r4.log 2>&1; rc=$?; tail -1 $O/prof_r4.log | cut -c1-300; guard $rc
echo "=== hbm traffic ($(date +%T))"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r4_$c
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_r4_$c --output-format csv -- python3 bench.py --steps 8 --warmup 2 --batch-q 0 --no-cpu-baseline --extra hamming > $O/pmc_r4_$c.log 2>&1; rc=$?; guard $rc
done
echo "=== latency map ($(date +%T))"
timeout -k 10 400 python tools/sweep_latency_r4.py > $O/r4_latency_map.txt 2>&1; rc=$?; grep -c "pass" $O/r4_latency_map.txt; guard $rc
echo "=== host path ($(date +%T))"
timeout -k 10 300 python tools/time_host_path.py 1000 10000 20000 100000 1250000 10000000 > $O/r4_host_path.txt 2>&1; rc=$?; grep -v amdgpu $O/r4_host_path.txt | cut -c1-260; guard $rc
echo "=== odd widths ($(date +%T))"
timeout -k 10 300 python tools/time_anyd.py > $O/r4_odd_widths.txt 2>&1; rc=$?; grep -v amdgpu $O/r4_odd_widths.txt | cut -c1-200; guard $rc
echo "=== bench ($(date +%T))"
timeout -k 10 600 python bench.py --steps 200 --warmup 10 > $O/bench_r4.log 2>&1; rc=$?; tail -1 $O/bench_r4.log | cut -c1-400; guard $rc
exit 0
