cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "HYDIA_LANES=2" "HYDIA_LANES=3" "HYDIA_LANES=4" "HYDIA_LANES=2 HYDIA_NTT_NP1=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$cfg', round(d['ms_per_step'],2), 'ms/step', d['config']['result_correct'])"
done
