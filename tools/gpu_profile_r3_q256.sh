set -u
cd "${GRAFT_REPO_ROOT}"
export TMPDIR=/tmp
O=gpurun_out
guard() { if [ "$1" -ge 124 ]; then echo "step killed rc=$1"; exit "$1"; fi; }
for pass in a b; do
  case $pass in
    a) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES";;
    b) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
  esac
  rm -rf $O/pmc_r3_q256_$pass
  timeout -k 10 300 rocprofv3 --pmc $C -d $O/pmc_r3_q256_$pass --output-format csv -- python3 tools/run_q256.py 20 > $O/pmc_r3_q256_$pass.log 2>&1; rc=$?
  grep "q256 kernel" $O/pmc_r3_q256_$pass.log; guard $rc
done
timeout -k 10 900 python tools/clock_q256.py run $O/r3_q256_clock.json > $O/clock_r3.log 2>&1; rc=$?; tail -3 $O/clock_r3.log | cut -c1-300; guard $rc
exit 0
