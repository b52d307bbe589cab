# round 5 (same as round 4): the whole GPU suite (+ smoke), timings of the slowest tests; "$@" = extra pytest arguments (e.g. -k expr)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 "$@" > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
