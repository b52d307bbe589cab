# round 3: quick timing of the 2^17 and 2^20 queries only; $@ = env assignments
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "$@"; do export $cfg; done
for L in 17 20; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --log2n $L > gpurun_out/ab_$L.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_$L.json')); print('2^$L', round(d['ms_per_step'],2), 'ms/step  similarity', d['config']['secondary']['computeSimilarity_ms_per_query'], 'membership', d['config']['secondary']['membershipScenario_ms_per_query'], d['config']['result_correct'], 'loopB frac', round(d['roofline']['frac'],3), 'loopB ms', round(d['roofline']['avg_launch_ms'],3))"
done
