set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu8.log
tail -8 gpurun_out/pytest_gpu8.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu8.log || exit 1
HYDIA_TENSOR_NW=4 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench8.json 2> gpurun_out/bench8.err; python -c "
import json; d=json.load(open('gpurun_out/bench8.json')); print(round(d['value']), 'vec/s', round(d['ms_per_step'],2), 'ms/step tensor', round(d['roofline']['avg_launch_ms'],2), 'ms', round(d['roofline']['achieved']), 'GB/s', d['config']['result_correct'])"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
HYDIA_TENSOR_NW=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q20b -- python3 $R/tools/prof_similarity.py 20 3 indexScenario > $R/gpurun_out/rocprof_q20b.log 2>&1
rm -f $R/gpurun_out/prof_q20b/*/*kernel_trace.csv
tail -3 $R/gpurun_out/rocprof_q20b.log | cut -c1-300
