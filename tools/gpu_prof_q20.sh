set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q20d -- python3 $R/tools/prof_similarity.py 20 3 indexScenario > $R/gpurun_out/rocprof_q20d.log 2>&1
rm -f $R/gpurun_out/prof_q20d/*/*kernel_trace.csv
