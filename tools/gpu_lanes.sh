set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_quick.log; tail -6 gpurun_out/pytest_quick.log
grep -q "pytest exit 0" gpurun_out/pytest_quick.log || exit 1
for L in 1 2 3 4; do HYDIA_LANES=$L timeout -k 10 300 python tools/prof_similarity.py 20 4 indexScenario > gpurun_out/lanes$L.log 2>&1; echo "lanes=$L $(grep indexScenario gpurun_out/lanes$L.log | cut -c1-130)"; done
for L in 1 2; do HYDIA_LANES=$L timeout -k 10 300 python tools/prof_similarity.py 14 5 indexScenario > gpurun_out/lanes14_$L.log 2>&1; echo "lanes=$L $(grep indexScenario gpurun_out/lanes14_$L.log | cut -c1-100)"; done
