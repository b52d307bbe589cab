# round 5: loop A's last pass, rotations fastest (default) against limbs fastest (HYDIA_LOOPA_LIMB_FASTEST=1): parity of the 512 rotations,
# rotateQuery alone (alternating), whole queries at 2^20, PMC bytes of the new order
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "hoisted_rotations or three_block or fast_paths" > gpurun_out/r05_loop_a_parity.log 2>&1 || { tail -40 gpurun_out/r05_loop_a_parity.log; exit 1; }
tail -2 gpurun_out/r05_loop_a_parity.log
: > gpurun_out/ab_loop_a.txt
for rep in 1 2 3; do
  timeout -k 10 120 python tools/prof_rotate.py 10 | sed 's/$/   [rotations fastest]/' >> gpurun_out/ab_loop_a.txt || exit 1
  HYDIA_LOOPA_LIMB_FASTEST=1 timeout -k 10 120 python tools/prof_rotate.py 10 | sed 's/$/   [limbs fastest (round 4)]/' >> gpurun_out/ab_loop_a.txt || exit 1
done
AB_REPEATS=2 timeout -k 10 600 python tools/ab_env.py 20 10 - HYDIA_LOOPA_LIMB_FASTEST=1 >> gpurun_out/ab_loop_a.txt 2>&1 || exit 1
cat gpurun_out/ab_loop_a.txt
bash tools/gpu_r5_loop_a_pmc.sh
