set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for cfg in "fp32 1536 1000000 16" "fp32 1536 1000000 64" "fp16 4096 1000000 16"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $O/prof_ks_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_ks_$tag --output-format csv -- python3 tools/prof_ksplit.py $cfg > $O/prof_ks_$tag.log 2>&1 || exit 1
  f=$(find $O/prof_ks_$tag -name "*kernel_stats.csv" | head -1); echo "== $cfg"; cut -d, -f1-4 $f | cut -c1-200 | sed -n 1,12p
done
echo "=== rehearsal"
HDB_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --rows 2000000 --batch-q 64 --batch-steps 3 > $O/bench_r3_rehearsal.log 2>&1; rc=$?; tail -1 $O/bench_r3_rehearsal.log | cut -c1-1500; exit $rc
