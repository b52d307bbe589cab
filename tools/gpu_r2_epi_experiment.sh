cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for lib in "" "$R/image_matching_amd/libhydia_EPICHEAP.so"; do
export HYDIA_LIBPATH=$lib
HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_epi -- python3 $R/tools/prof_query_ledger.py 20 3 indexScenario > $R/gpurun_out/prof_epi.log 2>&1
f=$(find $R/gpurun_out/prof_epi -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_epi.csv; rm -rf $R/gpurun_out/prof_epi
echo "== lib=$lib"; python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_epi.csv $R/gpurun_out/ledger_q20.json | grep "p2<false, 2, 3>\|one index"
done
