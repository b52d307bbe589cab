set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharding.py -m gpu -q -x --durations=15 > gpurun_out/pytest_gpu_shard.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_shard.log; tail -40 gpurun_out/pytest_gpu_shard.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu_shard.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r2_a.json 2> gpurun_out/bench_r2_a.err; echo "bench exit $?"; cut -c1-600 gpurun_out/bench_r2_a.json; tail -3 gpurun_out/bench_r2_a.err
