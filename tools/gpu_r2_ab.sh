# A/B of one 2^20 query: one-pass NTT thresholds vs two-pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "X=1" "HYDIA_NTT_1PASS=1 HYDIA_NTT_1PASS_MIN=1024" "HYDIA_NTT_1PASS=1 HYDIA_NTT_1PASS_MIN=256" "HYDIA_NTT_1PASS=1 HYDIA_NTT_1PASS_MIN=4096"; do
  env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$cfg', round(d['ms_per_step'],2), 'ms/step  similarity', d['config']['secondary']['computeSimilarity_ms_per_query'], d['config']['result_correct'])"
done
