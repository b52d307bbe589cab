# round 5, step 1: the exchange microbenchmark (tools/ubench/p2_swap) and the same-box A/B of the barrier-free pass 2 (tree) against
# round 4's library (tools/ab/r4.so).  Usage: gpurun -- 'bash tools/gpu_r5_step1.sh'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 120 ./tools/ubench/p2_swap > gpurun_out/r05_p2_swap.txt 2>&1 || { cat gpurun_out/r05_p2_swap.txt; exit 1; }
cat gpurun_out/r05_p2_swap.txt
bash tools/gpu_r4_ab_lib.sh r4.so wavesync "ntt_bit_exact or fast_paths or hoisted_rotations or three_block or custom or small_ring"
