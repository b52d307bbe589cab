# round 5: same-box A/B of ENGINE SWITCHES on the tree's library: a parity subset, whole-query times at 2^20 / 2^17 / 2^14 / 2^10 for
# every configuration (alternating, twice), single-lane kernel tables of a 2^20 query per configuration.
# Usage: gpu_r5_ab_env.sh <tag> "<pytest -k expression or ->" cfg [cfg ...]     (cfg: VAR=value[,VAR=value] or "-" for the defaults)
R=$GRAFT_REPO_ROOT; TAG=$1; K=$2; shift 2
cd $R; mkdir -p gpurun_out
if [ "$K" != "-" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/ab_${TAG}_parity.log 2>&1 || { tail -40 gpurun_out/ab_${TAG}_parity.log; exit 1; }
  tail -2 gpurun_out/ab_${TAG}_parity.log
fi
: > gpurun_out/ab_$TAG.txt
for L in 20 17 14 10; do
  AB_REPEATS=2 timeout -k 10 600 python tools/ab_env.py $L 10 "$@" >> gpurun_out/ab_$TAG.txt 2>&1 || { tail -20 gpurun_out/ab_$TAG.txt; exit 1; }
done
cat gpurun_out/ab_$TAG.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  name=$(echo "$cfg" | tr -c 'A-Za-z0-9_\n' '_'); [ "$cfg" = "-" ] && name=default
  if [ "$cfg" != "-" ]; then for kv in $(echo $cfg | tr ',' ' '); do export $kv; done; fi
  HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/prof_query_ledger.py 20 3 indexScenario > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/prof_$TAG.log; exit 1; }
  if [ "$cfg" != "-" ]; then for kv in $(echo $cfg | tr ',' ' '); do unset ${kv%%=*}; done; fi
  f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_${TAG}_$name.csv; rm -rf $R/gpurun_out/prof_$TAG
  python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_${TAG}_$name.csv $R/gpurun_out/ledger_q20.json > $R/gpurun_out/kernel_rooflines_${TAG}_$name.txt
  echo "== $cfg"; head -14 $R/gpurun_out/kernel_rooflines_${TAG}_$name.txt | cut -c1-90
done
