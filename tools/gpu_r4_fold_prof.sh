# round 4: per-kernel durations of a 2^20 query (single lane) with and without the pseudo-Mersenne folds (tools/ab/libhydia_prefold.so)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
export HYDIA_LANES=1
for v in before folds; do
  if [ $v = before ]; then export HYDIA_LIBPATH=$R/tools/ab/libhydia_prefold.so; else unset HYDIA_LIBPATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fold_$v -- python3 $R/tools/prof_query_ledger.py 20 3 indexScenario > $R/gpurun_out/prof_fold_$v.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_fold_$v -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_fold_$v.csv; rm -rf $R/gpurun_out/prof_fold_$v
  python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_fold_$v.csv $R/gpurun_out/ledger_q20.json > $R/gpurun_out/kernel_rooflines_fold_$v.txt
  head -14 $R/gpurun_out/kernel_rooflines_fold_$v.txt
done
