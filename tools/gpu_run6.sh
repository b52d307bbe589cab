set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu6.log
tail -8 gpurun_out/pytest_gpu6.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu6.log || exit 1
for nw in 0 4 1; do HYDIA_TENSOR_NW=$nw timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench6_nw$nw.json 2> gpurun_out/bench6_nw$nw.err; python -c "
import json; d=json.load(open('gpurun_out/bench6_nw$nw.json')); print('nw=$nw', round(d['value']), 'vec/s', round(d['ms_per_step'],2), 'ms/step tensor', round(d['roofline']['avg_launch_ms'],2), 'ms', round(d['roofline']['achieved']), 'GB/s', d['config']['result_correct'])"; done
HYDIA_NTT_GENERIC=1 timeout -k 10 300 python tools/prof_similarity.py 17 3 > gpurun_out/prof17_gen6.log 2>&1; tail -3 gpurun_out/prof17_gen6.log
timeout -k 10 300 python tools/prof_similarity.py 17 3 > gpurun_out/prof17_fp6.log 2>&1; tail -3 gpurun_out/prof17_fp6.log
