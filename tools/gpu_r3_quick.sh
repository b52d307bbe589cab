# round 3: quick A/B — parity + full-ring tests, loop A alone, one-block and 2^20 queries, default vs HYDIA_NO_COLFUSE
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_ring.py -m gpu -q -x > gpurun_out/pytest_gpu_quick.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_quick.log; tail -4 gpurun_out/pytest_gpu_quick.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu_quick.log || exit 1
for cfg in "X=1" "HYDIA_NO_COLFUSE=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python tools/prof_rotate.py 10 || exit 1
  for L in 14 20; do
    env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --log2n $L > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('2^$L', round(d['ms_per_step'],2), 'ms/step  similarity', d['config']['secondary']['computeSimilarity_ms_per_query'], d['config']['result_correct'])"
  done
done
