# round 5: parity of the final build's pass-2 variants, then two lanes with an uneven split of the blocks (lanes drift out of step)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ntt_bit_exact or fast_paths or hoisted_rotations" > gpurun_out/r05_p2_variant_parity.log 2>&1 || { tail -40 gpurun_out/r05_p2_variant_parity.log; exit 1; }
tail -2 gpurun_out/r05_p2_variant_parity.log
AB_REPEATS=2 timeout -k 10 900 python tools/ab_env.py 20 10 - HYDIA_LANE_SPLIT=0.55 HYDIA_LANE_SPLIT=0.6 HYDIA_LANE_SPLIT=0.65 HYDIA_LANE_SPLIT=0.53 > gpurun_out/ab_lane_split.txt 2>&1 || { tail gpurun_out/ab_lane_split.txt; exit 1; }
cat gpurun_out/ab_lane_split.txt
