#!/usr/bin/env bash
# Round-2 profile collection (one gpurun call): kernel-trace stats of the bench command, HBM traffic counters of the
# headline kernel, matrix-pipe counters + in-kernel clock of the 256-query pass, then the plain bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
guard() { if [ "$1" -ge 124 ]; then echo "step killed rc=$1"; exit "$1"; fi; }
echo "=== kernel trace"; rm -rf $O/prof_r2
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_r2 --output-format csv -- python3 bench.py --steps 200 --warmup 10 --extra c2,c5 --no-cpu-baseline > $O/prof_r2.log 2>&1; rc=$?; tail -1 $O/prof_r2.log | cut -c1-400; guard $rc
echo "=== hbm traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r2_$c
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_r2_$c --output-format csv -- python3 bench.py --steps 8 --warmup 2 --batch-q 0 --no-cpu-baseline > $O/pmc_r2_$c.log 2>&1; rc=$?; guard $rc
done
echo "=== q256 pmc"
for pass in a b c; do
  case $pass in
    a) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES";;
    b) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    c) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE";;
  esac
  rm -rf $O/pmc_q256_$pass
  timeout -k 10 300 rocprofv3 --pmc $C -d $O/pmc_q256_$pass --output-format csv -- python3 tools/run_q256.py 20 > $O/pmc_q256_$pass.log 2>&1; rc=$?
  grep "q256 kernel" $O/pmc_q256_$pass.log; guard $rc
done
echo "=== q256 clock"
timeout -k 10 600 python tools/clock_q256.py run $O/r2_q256_clock.json > $O/clock.log 2>&1; rc=$?; tail -2 $O/clock.log | cut -c1-300; guard $rc
echo "=== bench"
timeout -k 10 600 python bench.py --steps 200 --warmup 10 --extra c2,c5 > $O/bench_r2.log 2>&1; rc=$?; tail -1 $O/bench_r2.log; guard $rc
echo "=== overhead"
timeout -k 10 300 python tools/time_call_overhead.py > $O/overhead_r2.log 2>&1; cat $O/overhead_r2.log
exit 0
