cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
  export EG_SUMS_DOWN=$v
  rm -rf /tmp/tr_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 bench.py --mode train --batch 32 --steps 10 --warmup 3 --no-other-configs --no-graph-replay > /dev/null 2>&1
  f=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r6e_train_stats_sumsdown$v.csv
done
