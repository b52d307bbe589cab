set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))"; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
timeout -k 10 1000 python -m pytest tests/test_gpu_client.py -m gpu -x -q --durations=10 > gpurun_out/pytest_client.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_client.log
tail -25 gpurun_out/pytest_client.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1_17 -- python3 $GRAFT_REPO_ROOT/tools/prof_similarity.py 17 3 indexScenario > $GRAFT_REPO_ROOT/gpurun_out/rocprof17.log 2>&1
tail -5 $GRAFT_REPO_ROOT/gpurun_out/rocprof17.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1_17 -name "*stats*" | head
