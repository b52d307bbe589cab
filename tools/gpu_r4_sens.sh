# round 4: where a kernel's time goes — single-lane kernel durations of a 2^L query for throw-away builds (tools/ab/libhydia_<V>.so, wrong
# results) that leave out one part of the kernel under study.  Usage: gpu_r4_sens.sh L <kernel name pattern> V1 V2 ...
R=$GRAFT_REPO_ROOT; L=$1; PAT=$2; shift 2
mkdir -p $R/gpurun_out; : > $R/gpurun_out/sens.txt; cd /tmp && export TMPDIR=/tmp
export HYDIA_LANES=1
for v in tree "$@"; do
  if [ $v = tree ]; then unset HYDIA_LIBPATH; else export HYDIA_LIBPATH=$R/tools/ab/libhydia_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sens -- python3 $R/tools/prof_query_ledger.py $L 3 indexScenario > $R/gpurun_out/prof_sens.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_sens -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/kernel_rooflines.py $f $R/gpurun_out/ledger_q$L.json | grep "$PAT" | sed "s/\$/   [$v]/" | cut -c1-110 >> $R/gpurun_out/sens.txt
  rm -rf $R/gpurun_out/prof_sens
done
cat $R/gpurun_out/sens.txt
