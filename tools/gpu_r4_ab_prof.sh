# round 4: single-lane kernel tables of one query at 2^L vectors for the tree's build and a variant build (tools/ab/<name>.so)
# Usage: gpu_r4_ab_prof.sh <variant .so name> <tag> <L>
R=$GRAFT_REPO_ROOT; V=$R/tools/ab/$1; TAG=$2; L=$3
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
export HYDIA_LANES=1
for v in variant tree; do
  if [ $v = variant ]; then export HYDIA_LIBPATH=$V; else unset HYDIA_LIBPATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab_$v -- python3 $R/tools/prof_query_ledger.py $L 3 indexScenario > $R/gpurun_out/prof_ab_$v.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_ab_$v -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_${TAG}_q${L}_$v.csv; rm -rf $R/gpurun_out/prof_ab_$v
  python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_${TAG}_q${L}_$v.csv $R/gpurun_out/ledger_q$L.json > $R/gpurun_out/kernel_rooflines_${TAG}_q${L}_$v.txt
  head -32 $R/gpurun_out/kernel_rooflines_${TAG}_q${L}_$v.txt | cut -c1-90
done
