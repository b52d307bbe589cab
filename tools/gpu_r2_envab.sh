# A/B of environment switches on ONE box: usage gpu_r2_envab.sh "VAR=1" ["VAR2=1" ...]; the default runs first and last
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "X=1" "$@" "X=1"; do
  for L in 14 20; do
    env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --log2n $L > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$cfg 2^$L', round(d['ms_per_step'],2), 'ms/step  similarity', d['config']['secondary']['computeSimilarity_ms_per_query'], d['config']['result_correct'])"
  done
done
