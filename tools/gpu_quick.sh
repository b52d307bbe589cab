set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_quick.log; tail -6 gpurun_out/pytest_quick.log
grep -q "pytest exit 0" gpurun_out/pytest_quick.log || exit 1
timeout -k 10 300 python tools/prof_similarity.py 20 3 > gpurun_out/quick20.log 2>&1; grep -E "computeSimilarity|indexScenario" gpurun_out/quick20.log | cut -c1-120
HYDIA_KEYS_UNPACKED=1 timeout -k 10 300 python tools/prof_similarity.py 20 3 > gpurun_out/quick20b.log 2>&1; grep -E "computeSimilarity|indexScenario" gpurun_out/quick20b.log | cut -c1-120
timeout -k 10 300 python tools/prof_similarity.py 14 5 > gpurun_out/quick14.log 2>&1; grep -E "computeSimilarity|indexScenario" gpurun_out/quick14.log | cut -c1-100
