#!/usr/bin/env bash
# Round-3 profile collection, second part (after the bit-metric / pearson / float32-euclidean single launches): kernel-trace stats
# of the bench command and of tools/time_all_metrics.py, HBM traffic of the hamming launch, the bit-metric timings, the bench line.
# (The 256-query counters and clocks of tools/gpu_profile_r3.sh are not repeated: hdb_mfma_kernel.h is unchanged.)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
guard() { if [ "$1" -ge 124 ]; then echo "step killed rc=$1"; exit "$1"; fi; }
echo "=== kernel trace: bench"; rm -rf $O/prof_r3
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_r3 --output-format csv -- python3 bench.py --steps 200 --warmup 10 --extra c2,c5,hamming --no-cpu-baseline > $O/prof_r3.log 2>&1; rc=$?; tail -1 $O/prof_r3.log | cut -c1-300; guard $rc
echo "=== kernel trace: all metrics"; rm -rf $O/prof_r3_all
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_r3_all --output-format csv -- python3 tools/time_all_metrics.py > $O/prof_r3_all.log 2>&1; rc=$?; grep -v amdgpu $O/prof_r3_all.log | tail -16; guard $rc
echo "=== hbm traffic (hamming)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r3_ham_$c
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_r3_ham_$c --output-format csv -- python3 bench.py --steps 4 --warmup 2 --batch-q 0 --no-cpu-baseline --extra hamming > $O/pmc_r3_ham_$c.log 2>&1; rc=$?; guard $rc
done
echo "=== bit metrics: single launch vs six"
timeout -k 10 300 python tools/time_bits.py > $O/time_bits.log 2>&1; rc=$?; grep -v amdgpu $O/time_bits.log | tail -20 | cut -c1-200; guard $rc
echo "=== api / call overhead"
timeout -k 10 300 python tools/time_call_overhead.py > $O/overhead_r3.log 2>&1; grep -v amdgpu $O/overhead_r3.log | tail -12
echo "=== bench"
timeout -k 10 600 python bench.py --steps 200 --warmup 10 --extra c2,c5,hamming > $O/bench_r3.log 2>&1; rc=$?; tail -1 $O/bench_r3.log | cut -c1-600; guard $rc
exit 0
