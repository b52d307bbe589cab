set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu5.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu5.log
tail -8 gpurun_out/pytest_gpu5.log
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_2p20.json 2> gpurun_out/bench_2p20.err; echo "bench exit $?"; cat gpurun_out/bench_2p20.json; tail -5 gpurun_out/bench_2p20.err
