# A/B of alternative builds of libhydia.so (image_matching_amd/libhydia_<TAG>.so, built by hand with -D switches): bench at 2^14 and 2^20
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in "" $GRAFT_REPO_ROOT/image_matching_amd/libhydia_*.so; do
  export HYDIA_LIBPATH=$lib
  for L in 14 20; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --log2n $L > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('lib=$(basename "$lib")', '2^$L', round(d['ms_per_step'],2), 'ms/step  similarity', d['config']['secondary']['computeSimilarity_ms_per_query'], d['config']['result_correct'])"
  done
done
