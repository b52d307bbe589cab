# A/B: slicing batched key switches / loop A so that intermediates stay in the Infinity Cache (result: profiles/r02/slicing_for_infinity_cache_ab.txt;
# HYDIA_SLICE_A_MIB — loop A — was removed from the library after this run, HYDIA_SLICE_MIB remains as an experiment switch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in 20 14; do
  for cfg in "HYDIA_SLICE_MIB=0" "HYDIA_SLICE_MIB=192" "HYDIA_SLICE_MIB=96" "HYDIA_SLICE_MIB=48" "HYDIA_SLICE_MIB=0 HYDIA_SLICE_A_MIB=256" "HYDIA_SLICE_MIB=0 HYDIA_SLICE_A_MIB=128" "HYDIA_SLICE_MIB=0 HYDIA_SLICE_A_MIB=64"; do
    env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --log2n $L > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('2^$L', '$cfg', round(d['ms_per_step'],2), 'ms/step  similarity', d['config']['secondary']['computeSimilarity_ms_per_query'], d['config']['result_correct'])"
  done
done
