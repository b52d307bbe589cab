set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu11.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu11.log; tail -3 gpurun_out/pytest_gpu11.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu11.log || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench11.json 2> gpurun_out/bench11.err; python -c "
import json; d=json.load(open('gpurun_out/bench11.json')); print(round(d['value']), 'vec/s', round(d['ms_per_step'],2), 'ms/step tensor', round(d['roofline']['avg_launch_ms'],2), 'ms', round(d['roofline']['achieved']), 'GB/s', d['config']['result_correct'])"
HYDIA_DB_UNPACKED=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench11u.json 2> gpurun_out/bench11u.err; python -c "
import json; d=json.load(open('gpurun_out/bench11u.json')); print('unpacked', round(d['value']), 'vec/s', round(d['ms_per_step'],2), 'ms/step tensor', round(d['roofline']['avg_launch_ms'],2), 'ms', round(d['roofline']['achieved']), 'GB/s', d['config']['result_correct'])"
