set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_q20d
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_q20d -- python3 $R/tools/prof_similarity.py ${1:-20} 2 indexScenario > $R/gpurun_out/rocprof_trace.log 2>&1
T=$(ls -t $R/gpurun_out/trace_q20d/*/*kernel_trace.csv | head -1)
python3 $R/tools/trace_summarise.py $T $R/gpurun_out/trace_q${1:-20}.csv
rm -rf $R/gpurun_out/trace_q20d
wc -l $R/gpurun_out/trace_q${1:-20}.csv
