# round 2: GPU test suite + smoke + bench with the CPU baseline (G=1, G=3, disk variant)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/pytest_gpu_r2.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_r2.log; tail -25 gpurun_out/pytest_gpu_r2.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu_r2.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r2_a.json 2> gpurun_out/bench_r2_a.err; echo "bench exit $?"; cut -c1-600 gpurun_out/bench_r2_a.json; tail -3 gpurun_out/bench_r2_a.err
