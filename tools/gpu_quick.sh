set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_quick.log; tail -12 gpurun_out/pytest_quick.log
