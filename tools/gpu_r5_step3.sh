# round 5, step 3: new parity test + sk<8> path + lanes / small-launch crossover with the narrow column-fused kernel
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "custom_chain_with_46bit or group_sequential or bench_gpus_flag or test_reference_datasets or devmath or loop_b" > gpurun_out/r05_step3_parity.log 2>&1 || { tail -40 gpurun_out/r05_step3_parity.log; exit 1; }
tail -3 gpurun_out/r05_step3_parity.log
: > gpurun_out/ab_lanes_cfsmall.txt
AB_REPEATS=2 timeout -k 10 600 python tools/ab_env.py 20 10 - HYDIA_LANES=3 HYDIA_LANES=4 >> gpurun_out/ab_lanes_cfsmall.txt 2>&1 || exit 1
for L in 17 14 10; do
  AB_REPEATS=2 timeout -k 10 600 python tools/ab_env.py $L 20 - HYDIA_CF_SMALL=0 HYDIA_CF_SMALL=128 HYDIA_CF_SMALL=64 HYDIA_LANES=3 >> gpurun_out/ab_lanes_cfsmall.txt 2>&1 || exit 1
done
cat gpurun_out/ab_lanes_cfsmall.txt
