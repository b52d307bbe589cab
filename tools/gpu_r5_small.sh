# round 5: the small-launch conversions on 16-column tiles: parity, then one-block / eight-block queries against round 4's 32-column kernels
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "three_block or config3 or auto_tiers or reference_dataset or fast_paths or custom_chain" > gpurun_out/r05_small_parity.log 2>&1 || { tail -40 gpurun_out/r05_small_parity.log; exit 1; }
tail -2 gpurun_out/r05_small_parity.log
: > gpurun_out/ab_small_cf8.txt
for L in 14 10 17; do AB_REPEATS=4 timeout -k 10 600 python tools/ab_env.py $L 40 - HYDIA_COLFUSE_WIDE=1 >> gpurun_out/ab_small_cf8.txt 2>&1 || exit 1; done
cat gpurun_out/ab_small_cf8.txt
