# round 4: same-box A/B of the tree's libhydia.so against a variant build (tools/ab/<name>.so, built in the container from another
# revision): a parity subset on the tree's build, whole-query times at 2^20 / 2^17 / 2^14 alternating between the two, single-lane
# kernel tables of a 2^20 query for both.  Usage: gpu_r4_ab_lib.sh <variant .so name in tools/ab> <tag> [pytest -k expression]
R=$GRAFT_REPO_ROOT; V=$R/tools/ab/$1; TAG=$2; K=${3:-"fast_paths or auto_tiers or hoisted_rotations or three_block or custom or small_ring"}
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/ab_${TAG}_parity.log 2>&1 || { tail -30 gpurun_out/ab_${TAG}_parity.log; exit 1; }
tail -2 gpurun_out/ab_${TAG}_parity.log
: > gpurun_out/ab_$TAG.txt
for L in 20 17 14 10; do
  for rep in 1 2; do
    HYDIA_LIBPATH=$V timeout -k 10 300 python tools/ab_env.py $L 10 - 2>&1 | sed 's/$/   [variant]/' >> gpurun_out/ab_$TAG.txt || exit 1
    timeout -k 10 300 python tools/ab_env.py $L 10 - 2>&1 | sed 's/$/   [tree]/' >> gpurun_out/ab_$TAG.txt || exit 1
  done
done
cat gpurun_out/ab_$TAG.txt
cd /tmp && export TMPDIR=/tmp
export HYDIA_LANES=1
for v in variant tree; do
  if [ $v = variant ]; then export HYDIA_LIBPATH=$V; else unset HYDIA_LIBPATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab_$v -- python3 $R/tools/prof_query_ledger.py 20 3 indexScenario > $R/gpurun_out/prof_ab_$v.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_ab_$v -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_${TAG}_$v.csv; rm -rf $R/gpurun_out/prof_ab_$v
  python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_${TAG}_$v.csv $R/gpurun_out/ledger_q20.json > $R/gpurun_out/kernel_rooflines_${TAG}_$v.txt
  head -9 $R/gpurun_out/kernel_rooflines_${TAG}_$v.txt
done
