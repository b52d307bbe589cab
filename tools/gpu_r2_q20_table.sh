cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
L=20
HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_led$L -- python3 $R/tools/prof_query_ledger.py $L 3 indexScenario > $R/gpurun_out/prof_led$L.log 2>&1
f=$(find $R/gpurun_out/prof_led$L -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_q$L.csv; rm -rf $R/gpurun_out/prof_led$L
python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_q$L.csv $R/gpurun_out/ledger_q$L.json > $R/gpurun_out/kernel_rooflines_q$L.txt; head -32 $R/gpurun_out/kernel_rooflines_q$L.txt
