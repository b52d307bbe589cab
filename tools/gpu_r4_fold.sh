# round 4: parity of the pseudo-Mersenne folds (conversion sums / key-switching inner products) + same-box A/B against the build before them
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fast_paths or auto_tiers or hoisted_rotations or three_block or custom or comparator_depths or small_ring" > gpurun_out/fold_parity.log 2>&1 || { tail -30 gpurun_out/fold_parity.log; exit 1; }
tail -3 gpurun_out/fold_parity.log
: > gpurun_out/ab_fold.txt
for L in 20 17 14; do
  for rep in 1 2; do
    HYDIA_LIBPATH=$GRAFT_REPO_ROOT/tools/ab/libhydia_prefold.so timeout -k 10 300 python tools/ab_env.py $L 10 - 2>&1 | sed 's/$/   [before]/' >> gpurun_out/ab_fold.txt || exit 1
    timeout -k 10 300 python tools/ab_env.py $L 10 - 2>&1 | sed 's/$/   [folds]/' >> gpurun_out/ab_fold.txt || exit 1
  done
done
cat gpurun_out/ab_fold.txt
