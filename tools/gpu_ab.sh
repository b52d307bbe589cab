set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/prof_similarity.py 20 3 indexScenario > gpurun_out/ab_a.log 2>&1; grep indexScenario gpurun_out/ab_a.log | cut -c1-120
HYDIA_NTT_NP1=1 timeout -k 10 300 python tools/prof_similarity.py 20 3 indexScenario > gpurun_out/ab_b.log 2>&1; grep indexScenario gpurun_out/ab_b.log | cut -c1-120
