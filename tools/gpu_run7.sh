set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
HYDIA_TENSOR_NW=4 timeout -k 10 300 python tools/prof_similarity.py 20 3 indexScenario > gpurun_out/prof20_fp.log 2>&1; tail -2 gpurun_out/prof20_fp.log
HYDIA_TENSOR_NW=4 HYDIA_NTT_INT=1 timeout -k 10 300 python tools/prof_similarity.py 20 3 indexScenario > gpurun_out/prof20_int.log 2>&1; tail -2 gpurun_out/prof20_int.log
cd /tmp && export TMPDIR=/tmp
HYDIA_TENSOR_NW=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q20 -- python3 $R/tools/prof_similarity.py 20 3 indexScenario > $R/gpurun_out/rocprof_q20.log 2>&1
rm -f $R/gpurun_out/prof_q20/*/*kernel_trace.csv
