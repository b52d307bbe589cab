# round 3: per-kernel table (ms, GB, TB/s) of one indexScenario query at 2^$1 vectors, for each env configuration in $2..
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
L=$1; shift
for cfg in "$@"; do
export $cfg
HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_led$L -- python3 $R/tools/prof_query_ledger.py $L 3 indexScenario > $R/gpurun_out/prof_led$L.log 2>&1 || { tail -5 $R/gpurun_out/prof_led$L.log; exit 1; }
unset ${cfg%%=*}
f=$(find $R/gpurun_out/prof_led$L -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_q${L}_${cfg%%=*}.csv; rm -rf $R/gpurun_out/prof_led$L
echo "== $cfg"
python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_q${L}_${cfg%%=*}.csv $R/gpurun_out/ledger_q$L.json > $R/gpurun_out/kernel_rooflines_q${L}_${cfg%%=*}.txt; head -30 $R/gpurun_out/kernel_rooflines_q${L}_${cfg%%=*}.txt
done
