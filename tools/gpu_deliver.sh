# Round deliverables on the GPU box: tests, smoke, CLI run of the reference's config 1, bench (+cpu baseline), rocprof evidence
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_final.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_final.log; tail -3 gpurun_out/pytest_gpu_final.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu_final.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python tools/npz_to_dat.py tests/golden/dataset_2_10.npz /tmp/2_10.dat && (cd gpurun_out && rm -f latency.csv && touch latency.csv && timeout -k 10 300 $R/image_matching_amd/ImageMatching /tmp/2_10.dat 5 > cli_2_10.log 2>&1; echo "cli exit $?" >> cli_2_10.log; tail -12 cli_2_10.log; cat latency.csv)
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench exit $?"; cat gpurun_out/bench_final.json | cut -c1-400
for l in 10 14 17; do timeout -k 10 300 python bench.py --steps 5 --warmup 2 --log2n $l --no-cpu-baseline > gpurun_out/bench_2p$l.json 2> gpurun_out/bench_2p$l.err; python -c "
import json; d=json.load(open('gpurun_out/bench_2p$l.json')); print('2^$l:', round(d['value']), 'vec/s', round(d['ms_per_step'],2), 'ms/step', d['config']['result_correct'])"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_stats2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_stats2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_bench_fetch2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_fetch2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_bench_write2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_write2.log 2>&1
HYDIA_TENSOR_NW=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q20c -- python3 $R/tools/prof_similarity.py 20 3 indexScenario > $R/gpurun_out/rocprof_q20c.log 2>&1
cd $R
for d in fetch write; do f=$(find gpurun_out/prof_bench_${d}2 -name "*counter_collection.csv" | head -1); head -1 $f > gpurun_out/pmc2_${d}_tensor.csv; grep k_hydia_tensor $f >> gpurun_out/pmc2_${d}_tensor.csv; rm -rf gpurun_out/prof_bench_${d}2; done
rm -f gpurun_out/prof_bench_stats2/*/*kernel_trace.csv gpurun_out/prof_q20c/*/*kernel_trace.csv
