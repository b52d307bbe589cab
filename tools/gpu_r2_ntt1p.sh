# one-pass NTT: parity first, then A/B timing of one 2^20 query with per-kernel stats
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ntt or fast_paths or reference_dataset" > gpurun_out/pytest_ntt1p.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_ntt1p.log; tail -15 gpurun_out/pytest_ntt1p.log
grep -q "pytest exit 0" gpurun_out/pytest_ntt1p.log || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_full_ring.py -m gpu -q -x > gpurun_out/pytest_ntt1p_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_ntt1p_full.log; tail -5 gpurun_out/pytest_ntt1p_full.log
grep -q "pytest exit 0" gpurun_out/pytest_ntt1p_full.log || exit 1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1p.json 2> gpurun_out/bench_1p.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('gpurun_out/bench_1p.json')); print('1p:', d['ms_per_step'], d['config']['secondary'], d['config']['result_correct'])"
HYDIA_NTT_2PASS=1 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_2p.json 2> gpurun_out/bench_2p.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('gpurun_out/bench_2p.json')); print('2p:', d['ms_per_step'], d['config']['secondary'], d['config']['result_correct'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q20_1p -- python3 $R/tools/prof_similarity.py 20 3 indexScenario > $R/gpurun_out/rocprof_q20_1p.log 2>&1
cd $R
f=$(find gpurun_out/prof_q20_1p -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/q20_1p_kernel_stats.csv; rm -rf gpurun_out/prof_q20_1p; head -30 gpurun_out/q20_1p_kernel_stats.csv | cut -c1-150
