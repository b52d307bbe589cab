cd $GRAFT_REPO_ROOT; : > gpurun_out/ab_lanes_small.txt
for L in 17 18 19 14; do AB_REPEATS=2 python tools/ab_env.py $L 20 - HYDIA_LANES=1 >> gpurun_out/ab_lanes_small.txt 2>&1 || exit 1; done
cat gpurun_out/ab_lanes_small.txt
