cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_sharding.py -m gpu -q -x --durations=12 > gpurun_out/pytest_gpu_shard.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_shard.log; tail -22 gpurun_out/pytest_gpu_shard.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu_shard.log
