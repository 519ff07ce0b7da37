#!/usr/bin/env bash
# One gpurun call; steps chosen by arguments (tests clock pmc overhead exp trace bench).  A test FAILURE does not stop the
# later steps; a timeout or kill (rc >= 124) does.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
step() { echo "=== $* ($(date +%T))"; }
guard() { if [ "$1" -ge 124 ]; then echo "step killed rc=$1"; exit "$1"; fi; }
for s in "$@"; do
case $s in
tests)
  step tests
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests_r2.log 2>&1; rc=$?
  tail -15 $O/gpu_tests_r2.log; guard $rc;;
tests_x)
  step tests_x "${TESTSEL:-}"
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "${TESTSEL}" > $O/gpu_tests_sel.log 2>&1; rc=$?
  tail -15 $O/gpu_tests_sel.log; guard $rc; if [ $rc -ne 0 ]; then echo "selected tests failed: stopping"; exit $rc; fi;;
clock)
  step clock
  timeout -k 10 600 python tools/clock_q256.py run $O/r2_q256_clock.json > $O/clock.log 2>&1; rc=$?
  tail -3 $O/clock.log; guard $rc;;
pmc)
  step pmc
  for pass in a b c; do
    case $pass in
      a) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES";;
      b) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
      c) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE";;
    esac
    rm -rf $O/pmc_q256_$pass
    timeout -k 10 300 rocprofv3 --pmc $C -d $O/pmc_q256_$pass --output-format csv -- python3 tools/run_q256.py 20 > $O/pmc_q256_$pass.log 2>&1; rc=$?
    grep "q256 kernel" $O/pmc_q256_$pass.log; guard $rc
  done;;
overhead)
  step overhead
  timeout -k 10 300 python tools/time_call_overhead.py > $O/overhead_r2.log 2>&1; rc=$?
  cat $O/overhead_r2.log; guard $rc;;
exp)
  step exp ${EXPS:-0 1 2 3}
  timeout -k 10 1000 python tools/exp_q256.py run $O/exp_q256.json ${EXPS:-0 1 2 3} > $O/exp.log 2>&1; rc=$?
  cat $O/exp.log; guard $rc;;
trace)
  step trace
  rm -rf $O/trace_overhead
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_overhead --output-format csv -- python3 tools/time_call_overhead.py short > $O/trace_overhead.log 2>&1; rc=$?
  tail -4 $O/trace_overhead.log; guard $rc;;
bench)
  step bench
  timeout -k 10 600 python bench.py --steps 200 --warmup 10 > $O/bench_r2.log 2>&1; rc=$?
  tail -2 $O/bench_r2.log; guard $rc;;
esac
done
exit 0
