#!/usr/bin/env bash
# Round-3 profile collection (one gpurun call): kernel-trace stats of the bench command and of tools/time_all_metrics.py,
# HBM traffic counters of the headline and hamming kernels, matrix-pipe counters + in-kernel clock of the 256-query pass, the
# plain bench line.  rocprofv3 always gets the program itself after `--`.
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
echo "=== hbm traffic (headline)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r3_$c
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_r3_$c --output-format csv -- python3 bench.py --steps 8 --warmup 2 --batch-q 0 --no-cpu-baseline > $O/pmc_r3_$c.log 2>&1; rc=$?; guard $rc
done
echo "=== hbm traffic (hamming)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r3_ham_$c
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_r3_ham_$c --output-format csv -- python3 bench.py --steps 4 --warmup 2 --batch-q 0 --no-cpu-baseline --extra hamming > $O/pmc_r3_ham_$c.log 2>&1; rc=$?; guard $rc
done
echo "=== q256 pmc"
for pass in a b; do
  case $pass in
    a) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES";;
    b) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
  esac
  rm -rf $O/pmc_r3_q256_$pass
  timeout -k 10 300 rocprofv3 --pmc $C -d $O/pmc_r3_q256_$pass --output-format csv -- python3 tools/run_q256.py 20 > $O/pmc_r3_q256_$pass.log 2>&1; rc=$?
  grep "q256 kernel" $O/pmc_r3_q256_$pass.log; guard $rc
done
echo "=== q256 clock (shipped single launch, knock-outs, five-kernel filter pass, four-wave variant)"
timeout -k 10 900 python tools/clock_q256.py run $O/r3_q256_clock.json > $O/clock_r3.log 2>&1; rc=$?; tail -3 $O/clock_r3.log | cut -c1-300; guard $rc
echo "=== rehearsal of the 2-rank control flow on one GPU (timings meaningless)"
HDB_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --rows 2000000 --batch-q 64 --batch-steps 3 > $O/bench_r3_rehearsal.log 2>&1; rc=$?; tail -1 $O/bench_r3_rehearsal.log | cut -c1-900; guard $rc
echo "=== api / call overhead"
timeout -k 10 300 python tools/time_call_overhead.py > $O/overhead_r3.log 2>&1; grep -v amdgpu $O/overhead_r3.log | tail -12
echo "=== bench"
timeout -k 10 600 python bench.py --steps 200 --warmup 10 --extra c2,c5,hamming > $O/bench_r3.log 2>&1; rc=$?; tail -1 $O/bench_r3.log | cut -c1-600; guard $rc
exit 0
