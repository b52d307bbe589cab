# round 3: the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/pytest_gpu_r3.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_r3.log; tail -22 gpurun_out/pytest_gpu_r3.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu_r3.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
