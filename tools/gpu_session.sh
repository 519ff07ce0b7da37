#!/usr/bin/env bash
# One gpurun call: GPU test suite, then the Q=256 clock / PMC evidence, then per-call latency.  A test FAILURE does not
# stop the measurements; a timeout or kill (rc >= 124) does.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
step() { echo "=== $* ($(date +%T))"; }
step tests
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_r2.log 2>&1; rc=$?
tail -5 $O/gpu_tests_r2.log
if [ $rc -ge 124 ]; then echo "tests killed rc=$rc"; exit $rc; fi
step clock
timeout -k 10 600 python tools/clock_q256.py run $O/r2_q256_clock.json > $O/clock.log 2>&1; rc=$?
tail -3 $O/clock.log
if [ $rc -ge 124 ]; then exit $rc; fi
step pmc
for pass in a b c; do
  case $pass in
    a) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES";;
    b) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    c) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE";;
  esac
  rm -rf $O/pmc_q256_$pass
  timeout -k 10 300 rocprofv3 --pmc $C -d $O/pmc_q256_$pass --output-format csv -- python3 tools/run_q256.py 20 > $O/pmc_q256_$pass.log 2>&1; rc=$?
  grep "q256 kernel" $O/pmc_q256_$pass.log
  if [ $rc -ge 124 ]; then exit $rc; fi
done
step overhead
timeout -k 10 300 python tools/time_call_overhead.py > $O/overhead_r2.log 2>&1; rc=$?
cat $O/overhead_r2.log
exit 0
