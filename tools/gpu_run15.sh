set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q14 -- python3 $R/tools/prof_similarity.py 14 10 indexScenario > $R/gpurun_out/rocprof_q14.log 2>&1
grep indexScenario $R/gpurun_out/rocprof_q14.log | cut -c1-150
rm -f $R/gpurun_out/prof_q14/*/*kernel_trace.csv
